// Probe for VERDICT r5 #4 ("form S and dP of the attention backward as 32x32x16 tiles: half the LDS fragment bytes per MAC"):
// the S^T / dP^T blocks of the dK | dV part of mha_bwd_p_mfma_v_k, stripped to what the claim is about -- fragment reads of the
// Q | dO images from LDS and the matrix instructions they feed -- in three tilings, at the shipped occupancy (one workgroup per CU):
//   A  two 16-key tiles per wave, 16x16x32 MFMAs (the shipped form, <14, 8, 2>): a Q / dO fragment read feeds TWO MFMAs
//   B  one 32-key tile per wave, 32x32x16 MFMAs (the VERDICT's proposal)
//   C  one 16-key tile per wave, 16x16x32 MFMAs, sixteen waves (the form the "half the bytes" figure compares against)
// Every variant covers the same 14 x 16 = 224 keys x 224 queries x d = 64 per "head" (S^T and dP^T: 2 x 2 x 224 x 224 x 64 flops) and
// runs HEADS heads per workgroup; the images are the kernel's ROW images (128-byte rows, 16-byte chunks XOR-swizzled by row & 7).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/attn_tile_probe.hip -o tools/probes/attn_tile_probe.out ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define ROWS 224
#define IMG_BYTES (ROWS * 128)

__device__ __forceinline__ bf16x8 lds_read16(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

// ---- A / C: 16x16x32.  Operand fragment of a 16-row tile at depth step ks (32 columns): lane (l16 = lane & 15, g = lane >> 4) holds
// row l16, columns 32 ks + 8 g .. + 7  ->  chunk (4 ks + g) ^ (l16 & 7) of the row.
template <int KT>          // key tiles per wave (2 = shipped, 1 = one tile per wave)
__global__ void __launch_bounds__(KT == 2 ? 512 : 1024)
probe16_k(const __bf16* __restrict__ kv, float* __restrict__ sink, int heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // Q image | dO image
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  for (int i = threadIdx.x; i < 2 * IMG_BYTES / 16; i += blockDim.x)
    reinterpret_cast<bf16x8*>(smem)[i] = reinterpret_cast<const bf16x8*>(kv)[(i * 7 + blockIdx.x) % 4096];
  __syncthreads();
  // the wave's K and V fragments (registers for the whole head, as in the kernel)
  bf16x8 kf[KT][2], vf[KT][2];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[t][ks] = reinterpret_cast<const bf16x8*>(kv)[(wave * 64 + lane + 97 * (2 * t + ks)) % 4096];
      vf[t][ks] = reinterpret_cast<const bf16x8*>(kv)[(wave * 64 + lane + 131 * (2 * t + ks) + 7) % 4096];
    }
  const int ntile_waves = 14 / KT;                                   // waves that carry key tiles (7 of 8, 14 of 16)
  f32x4 tot = {0.f, 0.f, 0.f, 0.f};
  if (wave < ntile_waves) {
    for (int h = 0; h < heads; ++h) {
#pragma unroll 1
      for (int qp = 0; qp < 7; ++qp) {                               // query PAIRS: 2 x 16 queries
        bf16x8 qf[2][2], df[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int row = (2 * qp + qt) * 16 + l16;
            const int off = row * 128 + (((4 * ks + g) ^ (l16 & 7)) << 4);
            qf[qt][ks] = lds_read16(smem + off);
            df[qt][ks] = lds_read16(smem + IMG_BYTES + off);
          }
        f32x4 s[KT][2], dp[KT][2];
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            s[t][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[t][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              s[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][ks], qf[qt][ks], s[t][qt], 0, 0, 0);
              dp[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[t][ks], df[qt][ks], dp[t][qt], 0, 0, 0);
            }
          }
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) tot += s[t][qt] * dp[t][qt];
      }
    }
  }
  if (tot[0] + tot[1] + tot[2] + tot[3] == 12345.678f) sink[blockIdx.x * blockDim.x + threadIdx.x] = tot[0];
}

// ---- B: 32x32x16.  Operand fragment of a 32-row tile at depth step ks (16 columns): lane (l32 = lane & 31, g = lane >> 5) holds row
// l32, columns 16 ks + 8 g .. + 7  ->  chunk (2 ks + g) ^ (l32 & 7).
__global__ void __launch_bounds__(512)
probe32_k(const __bf16* __restrict__ kv, float* __restrict__ sink, int heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = lane & 31, g = lane >> 5;
  for (int i = threadIdx.x; i < 2 * IMG_BYTES / 16; i += blockDim.x)
    reinterpret_cast<bf16x8*>(smem)[i] = reinterpret_cast<const bf16x8*>(kv)[(i * 7 + blockIdx.x) % 4096];
  __syncthreads();
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[ks] = reinterpret_cast<const bf16x8*>(kv)[(wave * 64 + lane + 97 * ks) % 4096];
    vf[ks] = reinterpret_cast<const bf16x8*>(kv)[(wave * 64 + lane + 131 * ks + 7) % 4096];
  }
  f32x16 tot;
#pragma unroll
  for (int i = 0; i < 16; ++i) tot[i] = 0.f;
  if (wave < 7) {                                                    // 7 waves x 32 keys = 224
    for (int h = 0; h < heads; ++h) {
#pragma unroll 1
      for (int qt = 0; qt < 7; ++qt) {                               // 32-query tiles
        bf16x8 qf[4], df[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int row = qt * 32 + l32;
          const int off = row * 128 + (((2 * ks + g) ^ (l32 & 7)) << 4);
          qf[ks] = lds_read16(smem + off);
          df[ks] = lds_read16(smem + IMG_BYTES + off);
        }
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks], df[ks], dp, 0, 0, 0);
        }
        tot += s * dp;
      }
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += tot[i];
  if (t == 12345.678f) sink[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int heads = argc > 1 ? atoi(argv[1]) : 400;
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  const int ncu = pr.multiProcessorCount;
  __bf16* kv; float* sink;
  std::vector<unsigned short> host(4096 * 8);
  for (size_t i = 0; i < host.size(); ++i) host[i] = (unsigned short)(0x3c00 + (i * 37) % 512);       // finite bf16 values near 0.01..0.03
  CK(hipMalloc(&kv, host.size() * 2));
  CK(hipMemcpy(kv, host.data(), host.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&sink, (size_t)ncu * 1024 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = 2 * IMG_BYTES;
  const double flops_per_head = 2.0 * 2.0 * 224.0 * 224.0 * 64.0;     // S^T and dP^T
  struct V { const char* name; int threads; int which; } vs[3] = {
    {"A  16x16x32, two key tiles per wave (shipped), 8 waves", 512, 0},
    {"B  32x32x16, one 32-key tile per wave, 8 waves        ", 512, 1},
    {"C  16x16x32, one key tile per wave, 16 waves          ", 1024, 2}};
  printf("%d CUs, one workgroup per CU, %d heads per workgroup; per head: S^T and dP^T of 224 keys x 224 queries x d = 64\n", ncu, heads);
  for (int rep = 0; rep < 2; ++rep)
    for (auto& v : vs) {
      auto launch = [&]() {
        if (v.which == 0) hipLaunchKernelGGL(probe16_k<2>, dim3(ncu), dim3(512), lds, 0, kv, sink, heads);
        else if (v.which == 1) hipLaunchKernelGGL(probe32_k, dim3(ncu), dim3(512), lds, 0, kv, sink, heads);
        else hipLaunchKernelGGL(probe16_k<1>, dim3(ncu), dim3(1024), lds, 0, kv, sink, heads);
      };
      launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us_head = ms * 1e3 / heads;
      // LDS fragment bytes per head and workgroup: reads x 64 lanes x 16 B
      const double reads = v.which == 0 ? 7.0 * 7 * 8 : (v.which == 1 ? 7.0 * 7 * 8 : 14.0 * 7 * 8);
      printf("%s  %8.3f us per head   %7.1f TFLOP/s chip-wide   LDS fragment reads %5.0f KiB per head\n", v.name, us_head,
             flops_per_head * heads * ncu / (ms * 1e-3) / 1e12, reads * 1024.0 / 1024.0);
    }
  return 0;
}
