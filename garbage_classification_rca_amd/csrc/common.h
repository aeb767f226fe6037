// Shared helpers for the libmmrca HIP sources (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/mmrca.h"

typedef __bf16 bf16_t;

int mmrca_fail(int code, const char* fmt, ...);   // records the message, returns code (api.cpp)

#define MMRCA_CHECK_LAUNCH(name)                                                         \
  do {                                                                                   \
    hipError_t e_ = hipGetLastError();                                                   \
    if (e_ != hipSuccess) return mmrca_fail(-10, "%s: launch failed: %s", name, hipGetErrorString(e_)); \
  } while (0)

#define MMRCA_REQUIRE(cond, ...)                         \
  do {                                                   \
    if (!(cond)) return mmrca_fail(-1, __VA_ARGS__);     \
  } while (0)

__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return (bf16_t)x; }

// 4-element vector access (16 B for f32, 8 B for bf16); pointers must be aligned accordingly.
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  float v[4];
  __device__ __forceinline__ static Vec4 load(const float* p) {
    Vec4 r; float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r;
  }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec4<bf16_t> {
  float v[4];
  __device__ __forceinline__ static Vec4 load(const bf16_t* p) {
    typedef __attribute__((ext_vector_type(4))) __bf16 b4;
    b4 t = *reinterpret_cast<const b4*>(p);
    Vec4 r; r.v[0] = (float)t[0]; r.v[1] = (float)t[1]; r.v[2] = (float)t[2]; r.v[3] = (float)t[3]; return r;
  }
  __device__ __forceinline__ void store(bf16_t* p) const {
    typedef __attribute__((ext_vector_type(4))) __bf16 b4;
    b4 t; t[0] = (bf16_t)v[0]; t[1] = (bf16_t)v[1]; t[2] = (bf16_t)v[2]; t[3] = (bf16_t)v[3];
    *reinterpret_cast<b4*>(p) = t;
  }
};

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}

// Largest dynamic-LDS size a kernel may be launched with (hipFuncAttributeMaxDynamicSharedMemorySize).  The attribute is sticky per
// (kernel, device): it is raised when a launch needs more than any launch before it did, not re-issued in front of every launch
// (the step is ~500 launches; the driver call costs about as much as the launch itself).  One cache per expansion site = per kernel
// instantiation; a racing first call from two host threads sets the same value twice, which is harmless.
struct MmrcaLdsAttr { int v[64]; };
static inline void mmrca_max_lds(const void* fn, int bytes, MmrcaLdsAttr& c) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& have = c.v[dev & 63];
  if (bytes > have && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) have = bytes;
}
#define MMRCA_MAX_LDS(bytes, ...) /* (bytes, kernel): the kernel last, it may be a template-id with commas */ \
  do {                                                                           \
    static MmrcaLdsAttr mmrca_lds_attr_;                                         \
    mmrca_max_lds((const void*)(__VA_ARGS__), (int)(bytes), mmrca_lds_attr_);    \
  } while (0)

// Step epoch of the counter-based masks.  Every launch carries its mask seed BY VALUE, so a launch captured in a HIP graph
// (training.GraphedTrainStep) would redraw the captured step's masks at every replay.  The hash therefore runs on
// seed + epoch * MMRCA_SEED_EPOCH_STRIDE, where the epoch is a device-side word that one node at the head of the graph refreshes
// from a counter in HBM (mmrca_seed_epoch_set): replay r of a step captured at step s draws exactly the masks the eager step
// s + r draws (the engine's seeds advance by the same stride per step, engine._site_seed).  0 outside graph replays.  One copy per
// translation unit (no relocatable device code in this build); constant address space, so kernels read it once.
#define MMRCA_SEED_EPOCH_STRIDE 1000003ull
__attribute__((unused)) static __constant__ unsigned long long mmrca_seed_epoch_v = 0ull;
#define MMRCA_SEED_EPOCH_EXPORT(tag)                                                       \
  extern "C" void* mmrca_seed_epoch_addr_##tag() {                                          \
    void* p = nullptr;                                                                     \
    return hipGetSymbolAddress(&p, HIP_SYMBOL(mmrca_seed_epoch_v)) == hipSuccess ? p : nullptr; \
  }

// counter-based uniform in [0,1): splitmix64 of (seed, index)
__device__ __forceinline__ float mmrca_uniform(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + mmrca_seed_epoch_v * MMRCA_SEED_EPOCH_STRIDE + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// GELU of the bf16 MFMA epilogues: 0.5 erfc(|x|/sqrt 2) by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7 in erf, far below
// bf16 resolution) on the hardware reciprocal and exp2 -- ~14 VALU ops per element instead of erff()'s ~40, which made
// the FFN1 epilogue VALU-bound (64 elements per lane per tile against 6k cycles of MFMA).  fp32 parity mode keeps erff.
__device__ __forceinline__ float gelu_fast_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float q = 0.5f * p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  return x >= 0.f ? fmaf(-x, q, x) : x * q;
}

// Packed forms for the MFMA epilogues, two elements per VALU instruction (v_pk_fma_f32 / v_pk_mul_f32; the reciprocal and the
// exponential stay scalar): ~10.5 instructions per element instead of ~20; the 0.5 of 0.5 erfc is folded into the
// polynomial's coefficients.  Measured in situ: the FFN1 forward of the 256x256 kernel stays at 314 us / 758 TFLOP/s either
// way -- its epilogue is bound by the two 128-KiB stores per tile (gelu and gelu'), not by this arithmetic.
typedef float mm_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mm_f2 mm_fma2(mm_f2 a, mm_f2 b, mm_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ mm_f2 gelu_half_erfc2(mm_f2 x, mm_f2* e_out) {      // q = 0.5 erfc(|x| / sqrt 2), e = exp(-x^2 / 2)
  const mm_f2 z = __builtin_elementwise_abs(x) * 0.70710678118654752f;
  const mm_f2 d = mm_fma2((mm_f2)(0.3275911f), z, (mm_f2)(1.0f));
  mm_f2 t; t.x = __builtin_amdgcn_rcpf(d.x); t.y = __builtin_amdgcn_rcpf(d.y);
  mm_f2 p = mm_fma2(t, (mm_f2)(0.5f * 1.061405429f), (mm_f2)(0.5f * -1.453152027f));
  p = mm_fma2(p, t, (mm_f2)(0.5f * 1.421413741f));
  p = mm_fma2(p, t, (mm_f2)(0.5f * -0.284496736f));
  p = mm_fma2(p, t, (mm_f2)(0.5f * 0.254829592f));
  const mm_f2 a = (z * z) * -1.4426950408889634f;
  mm_f2 e; e.x = __builtin_amdgcn_exp2f(a.x); e.y = __builtin_amdgcn_exp2f(a.y);
  *e_out = e;
  return (p * t) * e;
}
// v[0..3] -> gelu(v), g[0..3] = gelu'(v)
__device__ __forceinline__ void gelu_and_grad_fast4(float* v, float* g) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const mm_f2 x = {v[2 * h], v[2 * h + 1]};
    mm_f2 e;
    const mm_f2 q = gelu_half_erfc2(x, &e);
    mm_f2 Phi; Phi.x = x.x >= 0.f ? 1.0f - q.x : q.x; Phi.y = x.y >= 0.f ? 1.0f - q.y : q.y;
    const mm_f2 gr = mm_fma2(x * 0.3989422804014327f, e, Phi), y = x * Phi;
    v[2 * h] = y.x; v[2 * h + 1] = y.y; g[2 * h] = gr.x; g[2 * h + 1] = gr.y;
  }
}
__device__ __forceinline__ void gelu_fast4(float* v) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const mm_f2 x = {v[2 * h], v[2 * h + 1]};
    mm_f2 e;
    const mm_f2 q = gelu_half_erfc2(x, &e);
    mm_f2 Phi; Phi.x = x.x >= 0.f ? 1.0f - q.x : q.x; Phi.y = x.y >= 0.f ? 1.0f - q.y : q.y;
    const mm_f2 y = x * Phi;
    v[2 * h] = y.x; v[2 * h + 1] = y.y;
  }
}

// the same, also returning gelu'(x) = Phi(x) + x phi(x) (shares the exponential)
__device__ __forceinline__ float gelu_and_grad_fast_f(float x, float* grad) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);     // exp(-x^2 / 2)
  const float q = 0.5f * p * t * e;
  const float Phi = x >= 0.f ? 1.0f - q : q;
  *grad = fmaf(x * 0.3989422804014327f, e, Phi);
  return x * Phi;
}

#define MMRCA_DISPATCH_DTYPE(dtype, NAME, ...)                               \
  if ((dtype) == MMRCA_F32) { typedef float T; __VA_ARGS__ }                 \
  else if ((dtype) == MMRCA_BF16) { typedef bf16_t T; __VA_ARGS__ }          \
  else return mmrca_fail(-2, "%s: bad dtype %d", NAME, (int)(dtype));
