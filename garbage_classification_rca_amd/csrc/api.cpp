// Error plumbing shared by every libmmrca entry point.
#include <stdarg.h>
#include <stdio.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "../../include/mmrca.h"

static thread_local char g_err[512] = "";

int mmrca_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* mmrca_last_error(void) { return g_err; }
extern "C" int mmrca_version(void) { return 1; }

// ---- mask epoch (common.h): one launch refreshes every translation unit's copy -----------------------------------------------
extern "C" {
void* mmrca_seed_epoch_addr_attention_cls();
void* mmrca_seed_epoch_addr_attention_cross();
void* mmrca_seed_epoch_addr_attention_f32();
void* mmrca_seed_epoch_addr_attention_mfma();
void* mmrca_seed_epoch_addr_attention_ref();
void* mmrca_seed_epoch_addr_head();
void* mmrca_seed_epoch_addr_rowops();
}
#define MMRCA_N_EPOCH_COPIES 7
struct EpochCopies { unsigned long long* p[MMRCA_N_EPOCH_COPIES]; };

__global__ void seed_epoch_set_k(EpochCopies c, const unsigned long long* __restrict__ src, unsigned long long value) {
  const unsigned long long v = src ? *src : value;
  if (threadIdx.x < MMRCA_N_EPOCH_COPIES) *c.p[threadIdx.x] = v;
}

// hipGetSymbolAddress is per DEVICE (each GPU has its own copy of a __constant__): the resolved addresses are cached per device ordinal
// under a mutex, so a process that drives several GPUs -- or two host threads making their first call together -- writes the epoch of
// the device the launch runs on.
#define MMRCA_MAX_DEVICES 64
extern "C" int mmrca_seed_epoch_set(const uint64_t* device_value, uint64_t value, void* stream) {
  static EpochCopies copies[MMRCA_MAX_DEVICES];
  static bool ready[MMRCA_MAX_DEVICES] = {};
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MMRCA_MAX_DEVICES) return mmrca_fail(-10, "seed_epoch_set: bad current device %d", dev);
  EpochCopies c;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!ready[dev]) {
      void* a[MMRCA_N_EPOCH_COPIES] = {mmrca_seed_epoch_addr_attention_cls(), mmrca_seed_epoch_addr_attention_cross(), mmrca_seed_epoch_addr_attention_f32(),
                                       mmrca_seed_epoch_addr_attention_mfma(), mmrca_seed_epoch_addr_attention_ref(), mmrca_seed_epoch_addr_head(),
                                       mmrca_seed_epoch_addr_rowops()};
      for (int i = 0; i < MMRCA_N_EPOCH_COPIES; ++i) {
        if (!a[i]) return mmrca_fail(-10, "seed_epoch_set: hipGetSymbolAddress failed for copy %d on device %d", i, dev);
        copies[dev].p[i] = (unsigned long long*)a[i];
      }
      ready[dev] = true;
    }
    c = copies[dev];
  }
  hipLaunchKernelGGL(seed_epoch_set_k, dim3(1), dim3(64), 0, (hipStream_t)stream, c, (const unsigned long long*)device_value,
                     (unsigned long long)value);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mmrca_fail(-10, "seed_epoch_set: launch failed: %s", hipGetErrorString(e));
  return 0;
}
