// K3f: multi-head attention forward / backward in fp32 on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), head dim 64.
// This is the attention of the two modes that meet the reference's precision (--dtype bf16x3 and --dtype fp32; the reference's
// attention is fp32: transformers modeling_distilbert.py:122-203 / torchvision MultiheadAttention under
// CVPR_code/multimodal_model.py:651-659, no autocast).  Same contract as the kernels of attention_ref.hip (fused qkv buffer,
// key mask, packed layout, lse, counter-based dropout applied to the numerators after the row sum) at ~8x their speed: they
// were VALU kernels (9-12 TFLOP/s); the MFMA form is bound by the fp32 matrix rate (157 TFLOP/s).
//
// One workgroup per (batch, head).  Forward and dQ kernel: the head's K and V sit in LDS, a wave owns 16 queries.  dK/dV
// kernel: Q and dO sit in LDS, a wave owns 16 keys.  No atomics, fixed summation order (bitwise reproducible).
//
// MFMA operand maps (16x16x4 f32, lane = 16 g + c): A[row = c][k = g], B[k = g][col = c], D[row = 4 g + r][col = c] in register r.
//   * the contraction index a lane holds is permuted so that ONE 16-byte LDS read feeds four MFMAs: over head-dim channels, k-step
//     (j, r) pairs lane group g with channel 16 j + 4 g + r; over keys / queries, k-step r pairs g with row 4 g + r of the tile --
//     which is exactly the row a lane's accumulator register r holds, so score tiles are used as the next product's B operand
//     straight from the accumulators (no LDS round trip, no shuffles);
//   * scores are computed transposed (S^T = K Q^T: keys on rows, queries on lanes) in the forward and dQ kernels, so a query's
//     softmax row is reduced in registers plus two cross-lane steps;
//   * output tiles come out with channel 16 g + 4 r + t in register r of tile t: four tiles give 16 contiguous bytes per lane.
// LDS images are [row][64 floats] (256-byte rows) with the 16-byte slot index XOR-ed with (row & 15): row-pattern reads
// (16 lanes = 16 rows, same slot) are conflict-free, transposed-pattern reads (16 lanes = 16 slots of one row) at most 2-way;
// the LDS array is idle most of the time next to the fp32 matrix pipe (one 16-byte read per four 32-cycle MFMAs).
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f4;

#define F32A_DH 64
#define F32A_MAX_WAVES 16
#define LOG2E 1.4426950408889634f

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float xlane_max(float x) { x = fmaxf(x, __shfl_xor(x, 16, 64)); return fmaxf(x, __shfl_xor(x, 32, 64)); }
__device__ __forceinline__ float xlane_sum(float x) { x += __shfl_xor(x, 16, 64); return x + __shfl_xor(x, 32, 64); }

// swizzled image address (in floats) of 16-byte slot `slot` of row `row`
__device__ __forceinline__ int img(int row, int slot) { return row * 64 + ((slot ^ (row & 15)) << 2); }

// four consecutive fp32 values at element offset `off`: from an fp32 array (lo == nullptr), or rebuilt from the two bf16 planes
// hi + lo of a bf16x3 GEMM's two-plane output (`src` then points at the hi plane; same element offsets)
__device__ __forceinline__ f4 load4(const float* __restrict__ src, const bf16_t* __restrict__ lo, int64_t off) {
  if (lo) {
    typedef __attribute__((ext_vector_type(4))) __bf16 b4;
    const b4 h = *reinterpret_cast<const b4*>(reinterpret_cast<const bf16_t*>(src) + off), l = *reinterpret_cast<const b4*>(lo + off);
    return (f4){(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]};
  }
  return *reinterpret_cast<const f4*>(src + off);
}

// stage `rows` rows of 64 floats (global row stride ld, first element at offset off0) into a swizzled (or linear) image; rows
// [rows, rows_pad) are zero-filled
template <bool SWZ>
__device__ __forceinline__ void stage_rows(float* dst, const float* __restrict__ src, int64_t ld, int rows, int rows_pad,
                                           const bf16_t* __restrict__ lo = nullptr, int64_t off0 = 0) {
  for (int e = threadIdx.x; e < rows_pad * 16; e += blockDim.x) {
    const int r = e >> 4, s = e & 15;
    f4 v = (f4){0.f, 0.f, 0.f, 0.f};
    if (r < rows) v = load4(src, lo, off0 + (int64_t)r * ld + s * 4);
    *reinterpret_cast<f4*>(dst + (SWZ ? img(r, s) : r * 64 + s * 4)) = v;
  }
}

__device__ __forceinline__ float keep_factor(uint64_t seed, uint64_t bh, int Smax, int i, int j, float p, float sc) {
  return mmrca_uniform(seed, (bh * Smax + i) * Smax + j) >= p ? sc : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward.  NKT = key tiles (of 16) the score registers are sized for (S <= 16 NKT)
// ---------------------------------------------------------------------------------------------------------------------------
template <int NKT, bool DROP>
__global__ void __launch_bounds__(64 * (NKT < F32A_MAX_WAVES ? NKT : F32A_MAX_WAVES))
mha_fwd_f32m_k(const float* __restrict__ qkv, const int32_t* __restrict__ key_mask, float* __restrict__ out, float* __restrict__ lse,
               int H, int Smax, float scale, float drop_p, uint64_t drop_seed, const int32_t* __restrict__ cu,
               bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo, const bf16_t* __restrict__ qkv_lo) {
  // qkv_lo != nullptr: q|k|v arrive as the two bf16 planes a bf16x3 GEMM wrote (`qkv` = hi plane); the fp32 values are rebuilt on load
  extern __shared__ __attribute__((aligned(16))) float sm[];      // K image (swizzled) | V image (linear) | key bias
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int row0 = cu ? cu[b] : b * Smax, S = cu ? cu[b + 1] - cu[b] : Smax;
  if (S <= 0) return;
  const int nt = (S + 15) >> 4, Sp = nt * 16;
  float* Ks = sm; float* Vs = Ks + NKT * 16 * 64; float* bias = Vs + NKT * 16 * 64;
  const int64_t ld = 3LL * H * F32A_DH, ldo = (int64_t)H * F32A_DH;
  const int64_t q0 = (int64_t)row0 * ld + h * F32A_DH;            // element offset of this head's first query row
  stage_rows<true>(Ks, qkv, ld, S, Sp, qkv_lo, q0 + H * F32A_DH);
  stage_rows<false>(Vs, qkv, ld, S, Sp, qkv_lo, q0 + 2 * H * F32A_DH);
  for (int j = threadIdx.x; j < Sp; j += blockDim.x) bias[j] = (j < S && (!key_mask || key_mask[row0 + j] != 0)) ? 0.f : -INFINITY;
  __syncthreads();
  const float c1 = scale * LOG2E;
  const float drop_sc = DROP ? 1.f / (1.f - drop_p) : 1.f;
  for (int qt = wave; qt < nt; qt += nw) {
    const int qi = qt * 16 + c;                      // this lane's query
    f4 qf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      qf[j] = qi < S ? load4(qkv, qkv_lo, q0 + (int64_t)qi * ld + 16 * j + 4 * g) : (f4){0.f, 0.f, 0.f, 0.f};
    f4 s[NKT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nt) {
        f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f4 kf = *reinterpret_cast<const f4*>(Ks + img(kt * 16 + c, 4 * j + g));
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = mfma4(kf[r], qf[j][r], acc);
        }
        const f4 b4 = *reinterpret_cast<const f4*>(bias + kt * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[r] += b4[r]; m = fmaxf(m, acc[r]); }
        s[kt] = acc;
      }
    }
    m = xlane_max(m);
    const bool dead = !(m > -INFINITY);              // every key masked: output 0, lse = +inf (torch SDPA semantics)
    const float mc = dead ? 0.f : m * c1;
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float e = dead ? 0.f : __builtin_amdgcn_exp2f(fmaf(s[kt][r], c1, -mc)); s[kt][r] = e; l += e; }
      }
    }
    l = xlane_sum(l);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if constexpr (DROP) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        if (kt < nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kt][r] *= keep_factor(drop_seed, blockIdx.x, Smax, qi, kt * 16 + 4 * g + r, drop_p, drop_sc);
        }
      }
    }
    f4 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f4 vf = *reinterpret_cast<const f4*>(Vs + (kt * 16 + 4 * g + r) * 64 + 4 * c);
#pragma unroll
          for (int t = 0; t < 4; ++t) o[t] = mfma4(vf[t], s[kt][r], o[t]);
        }
      }
    }
    if (qi < S) {
      if (out) {           // (nullptr: only the planes are wanted -- bf16x3f, whose backward reads the hi plane)
        float* orow = out + ((int64_t)row0 + qi) * ldo + h * F32A_DH + 16 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<f4*>(orow + 4 * r) = (f4){o[0][r] * inv, o[1][r] * inv, o[2][r] * inv, o[3][r] * inv};
      }
      if (out_hi) {        // bf16x3 mode: the context also as two bf16 planes, the operand form of the out-projection GEMM
        typedef __attribute__((ext_vector_type(4))) __bf16 b4;
        const int64_t off = ((int64_t)row0 + qi) * ldo + h * F32A_DH + 16 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          b4 hi, lo;
#pragma unroll
          for (int t = 0; t < 4; ++t) { const float v = o[t][r] * inv; hi[t] = (bf16_t)v; lo[t] = (bf16_t)(v - (float)hi[t]); }
          *reinterpret_cast<b4*>(out_hi + off + 4 * r) = hi;
          *reinterpret_cast<b4*>(out_lo + off + 4 * r) = lo;
        }
      }
      if (g == 0) lse[((int64_t)b * H + h) * Smax + qi] = l > 0.f ? m * scale + __logf(l) : INFINITY;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward, dQ: K and V in LDS (both swizzled: K is read by rows for S and transposed for dQ, V by rows for dP)
// ---------------------------------------------------------------------------------------------------------------------------
template <bool DROP>
__global__ void __launch_bounds__(64 * F32A_MAX_WAVES)
mha_bwd_dq_f32m_k(const float* __restrict__ qkv, const int32_t* __restrict__ key_mask, const float* __restrict__ out,
                  const float* __restrict__ dout, const float* __restrict__ lse, float* __restrict__ dqkv,
                  int H, int Smax, int nkt_alloc, float scale, float drop_p, uint64_t drop_seed, const int32_t* __restrict__ cu) {
  extern __shared__ __attribute__((aligned(16))) float sm[];      // K image | V image | key bias
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int row0 = cu ? cu[b] : b * Smax, S = cu ? cu[b + 1] - cu[b] : Smax;
  if (S <= 0) return;
  const int nt = (S + 15) >> 4, Sp = nt * 16;
  float* Ks = sm; float* Vs = Ks + nkt_alloc * 16 * 64; float* bias = Vs + nkt_alloc * 16 * 64;
  const int64_t ld = 3LL * H * F32A_DH, ldo = (int64_t)H * F32A_DH;
  const float* Q = qkv + (int64_t)row0 * ld + h * F32A_DH;
  stage_rows<true>(Ks, Q + H * F32A_DH, ld, S, Sp);
  stage_rows<true>(Vs, Q + 2 * H * F32A_DH, ld, S, Sp);
  for (int j = threadIdx.x; j < Sp; j += blockDim.x) bias[j] = (j < S && (!key_mask || key_mask[row0 + j] != 0)) ? 0.f : -INFINITY;
  __syncthreads();
  const float c1 = scale * LOG2E;
  const float drop_sc = DROP ? 1.f / (1.f - drop_p) : 1.f;
  for (int qt = wave; qt < nt; qt += nw) {
    const int qi = qt * 16 + c;
    const bool live = qi < S;
    f4 qf[4], dof[4];
    float dsum = 0.f;
    const float* dorow = dout + ((int64_t)row0 + qi) * ldo + h * F32A_DH + 4 * g;
    const float* orow = out + ((int64_t)row0 + qi) * ldo + h * F32A_DH + 4 * g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f4 z = (f4){0.f, 0.f, 0.f, 0.f};
      qf[j] = live ? *reinterpret_cast<const f4*>(Q + (int64_t)qi * ld + 16 * j + 4 * g) : z;
      dof[j] = live ? *reinterpret_cast<const f4*>(dorow + 16 * j) : z;
      const f4 of = live ? *reinterpret_cast<const f4*>(orow + 16 * j) : z;
#pragma unroll
      for (int r = 0; r < 4; ++r) dsum = fmaf(dof[j][r], of[r], dsum);
    }
    dsum = xlane_sum(dsum);
    const float Lq = live ? lse[((int64_t)b * H + h) * Smax + qi] * LOG2E : INFINITY;
    f4 dq[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) dq[t] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nt; ++kt) {
      f4 s = (f4){0.f, 0.f, 0.f, 0.f}, dp = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f4 kf = *reinterpret_cast<const f4*>(Ks + img(kt * 16 + c, 4 * j + g));
        const f4 vf = *reinterpret_cast<const f4*>(Vs + img(kt * 16 + c, 4 * j + g));
#pragma unroll
        for (int r = 0; r < 4; ++r) { s = mfma4(kf[r], qf[j][r], s); dp = mfma4(vf[r], dof[j][r], dp); }
      }
      const f4 b4 = *reinterpret_cast<const f4*>(bias + kt * 16 + 4 * g);
      f4 ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c1, b4[r]) - Lq);       // masked key / dead row: exp2(-inf) = 0
        float dpr = dp[r];
        if constexpr (DROP) dpr *= keep_factor(drop_seed, blockIdx.x, Smax, qi, kt * 16 + 4 * g + r, drop_p, drop_sc);
        ds[r] = p * (dpr - dsum) * scale;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f4 k4 = *reinterpret_cast<const f4*>(Ks + img(kt * 16 + 4 * g + r, c));
#pragma unroll
        for (int t = 0; t < 4; ++t) dq[t] = mfma4(k4[t], ds[r], dq[t]);
      }
    }
    if (live) {
      float* drow = dqkv + ((int64_t)row0 + qi) * ld + h * F32A_DH + 16 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<f4*>(drow + 4 * r) = (f4){dq[0][r], dq[1][r], dq[2][r], dq[3][r]};
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward, dK / dV: Q and dO in LDS (swizzled: read by rows for S / dP, transposed for dK / dV), lse and rowsum(dO O) per query
// ---------------------------------------------------------------------------------------------------------------------------
template <bool DROP>
__global__ void __launch_bounds__(64 * F32A_MAX_WAVES)
mha_bwd_dkv_f32m_k(const float* __restrict__ qkv, const int32_t* __restrict__ key_mask, const float* __restrict__ out,
                   const float* __restrict__ dout, const float* __restrict__ lse, float* __restrict__ dqkv,
                   int H, int Smax, int nkt_alloc, float scale, float drop_p, uint64_t drop_seed, const int32_t* __restrict__ cu) {
  extern __shared__ __attribute__((aligned(16))) float sm[];      // Q image | dO image | L*log2e [Sp] | rowsum(dO O) [Sp]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int row0 = cu ? cu[b] : b * Smax, S = cu ? cu[b + 1] - cu[b] : Smax;
  if (S <= 0) return;
  const int nt = (S + 15) >> 4, Sp = nt * 16;
  float* Qs = sm; float* Ds = Qs + nkt_alloc * 16 * 64; float* Ls = Ds + nkt_alloc * 16 * 64; float* dsum_s = Ls + nkt_alloc * 16;
  const int64_t ld = 3LL * H * F32A_DH, ldo = (int64_t)H * F32A_DH;
  const float* Q = qkv + (int64_t)row0 * ld + h * F32A_DH;
  const float* Kp = Q + H * F32A_DH;
  const float* Vp = Kp + H * F32A_DH;
  const float* dO = dout + (int64_t)row0 * ldo + h * F32A_DH;
  const float* O = out + (int64_t)row0 * ldo + h * F32A_DH;
  stage_rows<true>(Qs, Q, ld, S, Sp);
  stage_rows<true>(Ds, dO, ldo, S, Sp);
  for (int i = threadIdx.x; i < Sp; i += blockDim.x) {
    float a = 0.f, L = INFINITY;                     // padding queries: p = exp2(.. - inf) = 0
    if (i < S) {
      for (int d = 0; d < F32A_DH; d += 4) {
        const f4 x = *reinterpret_cast<const f4*>(dO + (int64_t)i * ldo + d), y = *reinterpret_cast<const f4*>(O + (int64_t)i * ldo + d);
        a += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
      }
      L = lse[((int64_t)b * H + h) * Smax + i] * LOG2E;
    }
    dsum_s[i] = a; Ls[i] = L;
  }
  __syncthreads();
  const float c1 = scale * LOG2E;
  const float drop_sc = DROP ? 1.f / (1.f - drop_p) : 1.f;
  for (int kt = wave; kt < nt; kt += nw) {
    const int kj = kt * 16 + c;                      // this lane's key
    const bool live = kj < S;
    const float kb = (live && (!key_mask || key_mask[row0 + kj] != 0)) ? 0.f : -INFINITY;
    f4 kf[4], vf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f4 z = (f4){0.f, 0.f, 0.f, 0.f};
      kf[j] = live ? *reinterpret_cast<const f4*>(Kp + (int64_t)kj * ld + 16 * j + 4 * g) : z;
      vf[j] = live ? *reinterpret_cast<const f4*>(Vp + (int64_t)kj * ld + 16 * j + 4 * g) : z;
    }
    f4 dk[4], dv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { dk[t] = (f4){0.f, 0.f, 0.f, 0.f}; dv[t] = (f4){0.f, 0.f, 0.f, 0.f}; }
    for (int qt = 0; qt < nt; ++qt) {
      f4 s = (f4){0.f, 0.f, 0.f, 0.f}, dp = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f4 qr = *reinterpret_cast<const f4*>(Qs + img(qt * 16 + c, 4 * j + g));
        const f4 dr = *reinterpret_cast<const f4*>(Ds + img(qt * 16 + c, 4 * j + g));
#pragma unroll
        for (int r = 0; r < 4; ++r) { s = mfma4(qr[r], kf[j][r], s); dp = mfma4(dr[r], vf[j][r], dp); }      // D[query 4g+r][key c]
      }
      const f4 L4 = *reinterpret_cast<const f4*>(Ls + qt * 16 + 4 * g), d4 = *reinterpret_cast<const f4*>(dsum_s + qt * 16 + 4 * g);
      f4 pk, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c1, kb) - L4[r]);
        float keep = 1.f;
        if constexpr (DROP) keep = keep_factor(drop_seed, blockIdx.x, Smax, qt * 16 + 4 * g + r, kj, drop_p, drop_sc);
        pk[r] = p * keep;
        ds[r] = p * (dp[r] * keep - d4[r]) * scale;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f4 q4 = *reinterpret_cast<const f4*>(Qs + img(qt * 16 + 4 * g + r, c));
        const f4 o4 = *reinterpret_cast<const f4*>(Ds + img(qt * 16 + 4 * g + r, c));
#pragma unroll
        for (int t = 0; t < 4; ++t) { dk[t] = mfma4(q4[t], ds[r], dk[t]); dv[t] = mfma4(o4[t], pk[r], dv[t]); }
      }
    }
    if (live) {
      float* krow = dqkv + ((int64_t)row0 + kj) * ld + (int64_t)H * F32A_DH + h * F32A_DH + 16 * g;
      float* vrow = krow + (int64_t)H * F32A_DH;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        *reinterpret_cast<f4*>(krow + 4 * r) = (f4){dk[0][r], dk[1][r], dk[2][r], dk[3][r]};
        *reinterpret_cast<f4*>(vrow + 4 * r) = (f4){dv[0][r], dv[1][r], dv[2][r], dv[3][r]};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
static const bool g_f32m_on = !(getenv("MMRCA_ATTN_F32_MFMA") && atoi(getenv("MMRCA_ATTN_F32_MFMA")) == 0);

bool mmrca_mha_f32m_ok(int S, int dh, int dtype) {
  return g_f32m_on && dtype == MMRCA_F32 && dh == F32A_DH && S >= 1 && S <= 208;
}

static int nkt_for(int S) { return S <= 64 ? 4 : (S <= 128 ? 8 : 13); }

int mmrca_mha_fwd_f32m(const void* qkv, const int32_t* key_mask, void* out, float* lse, int B, int H, int S, int dh, float scale,
                       float drop_p, uint64_t drop_seed, const int32_t* cu, hipStream_t st, void* out_hi, void* out_lo, const void* qkv_lo) {
  MMRCA_REQUIRE(mmrca_mha_f32m_ok(S, dh, MMRCA_F32), "mha_fwd(f32 mfma): S=%d dh=%d unsupported", S, dh);
  MMRCA_REQUIRE((((uintptr_t)qkv | (uintptr_t)out) & 15) == 0, "mha_fwd(f32 mfma): operands must be 16-byte aligned");
  const int nkt = nkt_for(S);
  const size_t lds = ((size_t)2 * nkt * 16 * 64 + nkt * 16) * sizeof(float);
#define LF(NKT_, DROP_)                                                                                                            \
  do {                                                                                                                             \
    MMRCA_MAX_LDS((int)lds, mha_fwd_f32m_k<NKT_, DROP_>);     \
    hipLaunchKernelGGL((mha_fwd_f32m_k<NKT_, DROP_>), dim3(B * H), dim3(64 * (NKT_ < F32A_MAX_WAVES ? NKT_ : F32A_MAX_WAVES)), lds, st, \
                       (const float*)qkv, key_mask, (float*)out, lse, H, S, scale, drop_p, drop_seed, cu, (bf16_t*)out_hi,         \
                       (bf16_t*)out_lo, (const bf16_t*)qkv_lo);                                                                    \
  } while (0)
  const bool drop = drop_p > 0.f;
  if (nkt == 4) { if (drop) LF(4, true); else LF(4, false); }
  else if (nkt == 8) { if (drop) LF(8, true); else LF(8, false); }
  else { if (drop) LF(13, true); else LF(13, false); }
#undef LF
  MMRCA_CHECK_LAUNCH("mha_fwd(f32 mfma)");
  return 0;
}

int mmrca_mha_bwd_f32m(const void* qkv, const int32_t* key_mask, const void* out, const void* dout, const float* lse, void* dqkv,
                       int B, int H, int S, int dh, float scale, float drop_p, uint64_t drop_seed, const int32_t* cu, hipStream_t st) {
  MMRCA_REQUIRE(mmrca_mha_f32m_ok(S, dh, MMRCA_F32), "mha_bwd(f32 mfma): S=%d dh=%d unsupported", S, dh);
  MMRCA_REQUIRE((((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dqkv) & 15) == 0, "mha_bwd(f32 mfma): operands must be 16-byte aligned");
  const int nkt = (S + 15) / 16;
  const int nw = nkt < F32A_MAX_WAVES ? nkt : F32A_MAX_WAVES;
  const size_t lds1 = ((size_t)2 * nkt * 16 * 64 + nkt * 16) * sizeof(float), lds2 = ((size_t)2 * nkt * 16 * 64 + 2 * nkt * 16) * sizeof(float);
#define LB(DROP_)                                                                                                                  \
  do {                                                                                                                             \
    MMRCA_MAX_LDS((int)lds1, mha_bwd_dq_f32m_k<DROP_>);       \
    MMRCA_MAX_LDS((int)lds2, mha_bwd_dkv_f32m_k<DROP_>);      \
    hipLaunchKernelGGL((mha_bwd_dq_f32m_k<DROP_>), dim3(B * H), dim3(64 * nw), lds1, st, (const float*)qkv, key_mask, (const float*)out, \
                       (const float*)dout, lse, (float*)dqkv, H, S, nkt, scale, drop_p, drop_seed, cu);                            \
    hipLaunchKernelGGL((mha_bwd_dkv_f32m_k<DROP_>), dim3(B * H), dim3(64 * nw), lds2, st, (const float*)qkv, key_mask, (const float*)out, \
                       (const float*)dout, lse, (float*)dqkv, H, S, nkt, scale, drop_p, drop_seed, cu);                            \
  } while (0)
  if (drop_p > 0.f) LB(true); else LB(false);
#undef LB
  MMRCA_CHECK_LAUNCH("mha_bwd(f32 mfma)");
  return 0;
}

MMRCA_SEED_EPOCH_EXPORT(attention_f32)   // this translation unit's copy of the mask epoch (common.h)
