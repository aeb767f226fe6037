// K3x: multi-head attention FORWARD with separate query / key / value operands, any head dim that is a multiple of 8
// (<= 128) and S_q != S_kv: what the BLIP-2 path needs (q_former_training.py:289 runs
// Blip2ForConditionalGeneration, whose attention lives in transformers 5.15.0 modeling_blip_2.py):
//   * the ViT-g vision tower's self-attention:  S = 257, 16 heads of 88  (modeling_blip_2.py:282-354)
//   * the Q-Former's cross-attention:           32 queries x 257 image tokens, 12 heads of 64, attention-probability
//                                               dropout in train mode   (modeling_blip_2.py:536-606)
// The path is forward-only: the reference freezes everything under the 4-class classifier (see q_former.py).
//
// Two kernels behind mmrca_mha_cross_fwd:
//   * mha_cross_ref_k : any dtype, fp32 math, one wave per query row (the checker and the fp32 mode);
//   * mha_cross_mfma_k: bf16 v_mfma_f32_16x16x32_bf16, one workgroup per (batch, head): the whole K and V of the head are
//     staged in LDS once (row pitch = padded head dim + 16 B, zero-filled beyond dh and beyond S_kv), every wave owns
//     16-query tiles; like attention_mfma.hip the score tile is computed TRANSPOSED (S^T = K Q^T) so that a query's
//     softmax row lives in one lane column and the probabilities feed O^T = V^T P^T straight from the accumulator
//     registers (contraction order inside a 32-key step permuted: key = 32u + 16(j>>2) + 4g + (j&3)); V^T fragments come
//     from the row-major V image through ds_read_b64_tr_b16.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

#define AX_MAX_DH 128
#define AX_MAX_S 1024
#define AX_LOG2E 1.4426950408889634f

// ------------------------------------------------------------------------------------------------------
// checker / fp32 mode: grid (ceil(Sq/4), B*H), block 256 = 4 query rows
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
mha_cross_ref_k(const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk, const T* __restrict__ v, int64_t ldv,
                T* __restrict__ out, int64_t ldo, int H, int Sq, int Skv, int dh, float scale, float drop_p, uint64_t drop_seed) {
  const float drop_sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  extern __shared__ float smf[];     // per wave: q[AX_MAX_DH] | p[Skv]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y / H, h = blockIdx.y % H;
  const int i = blockIdx.x * 4 + wave;
  if (i >= Sq) return;               // wave-uniform; no block barrier below
  float* qv = smf + wave * (AX_MAX_DH + Skv);
  float* p = qv + AX_MAX_DH;
  const T* qrow = q + ((int64_t)b * Sq + i) * ldq + h * dh;
  const T* kp = k + (int64_t)b * Skv * ldk + h * dh;
  const T* vp = v + (int64_t)b * Skv * ldv + h * dh;
  for (int d = lane; d < dh; d += 64) qv[d] = to_f(qrow[d]);
  __builtin_amdgcn_wave_barrier();
  float m = -INFINITY;
  for (int j = lane; j < Skv; j += 64) {
    const T* kr = kp + (int64_t)j * ldk;
    float a = 0.f;
    for (int d = 0; d < dh; ++d) a += to_f(kr[d]) * qv[d];
    a *= scale;
    p[j] = a;
    m = fmaxf(m, a);
  }
  m = wave_max(m);
  float l = 0.f;
  for (int j = lane; j < Skv; j += 64) { const float e = __expf(p[j] - m); p[j] = e; l += e; }
  l = wave_sum(l);
  const float inv = 1.f / l;
  if (drop_p > 0.f)
    for (int j = lane; j < Skv; j += 64)
      p[j] *= mmrca_uniform(drop_seed, ((uint64_t)blockIdx.y * Sq + i) * Skv + j) >= drop_p ? drop_sc : 0.f;
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < dh; d += 64) {
    float o = 0.f;
    for (int j = 0; j < Skv; ++j) o += p[j] * to_f(vp[(int64_t)j * ldv + d]);
    out[((int64_t)b * Sq + i) * ldo + h * dh + d] = from_f<T>(o * inv);
  }
}

// ------------------------------------------------------------------------------------------------------
// bf16 MFMA kernel: grid B*H, block 64*NW
// ------------------------------------------------------------------------------------------------------
// rows [0, Spad) x DHP columns of a [S, ld] slice -> LDS image with PITCH-byte rows; zeros beyond (S, dh)
template <int DHP>
__device__ __forceinline__ void ax_stage(char* img, const bf16_t* __restrict__ src, int64_t ld, int S, int Spad, int dh) {
  constexpr int CH = DHP / 8, PITCH = DHP * 2 + 16;
  for (int e = threadIdx.x; e < Spad * CH; e += blockDim.x) {
    const int row = e / CH, c = e % CH;
    bf16x8 val = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < S && c * 8 < dh) val = *reinterpret_cast<const bf16x8*>(src + (int64_t)row * ld + c * 8);
    *reinterpret_cast<bf16x8*>(img + row * PITCH + c * 16) = val;
  }
}

template <int NKT, int DHS, bool DROP>      // NKT key tiles of 16 (even); head dim padded to DHP = 32*DHS; DROP: dropout compiled in
__global__ void __launch_bounds__(512)
mha_cross_mfma_k(const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v,
                 int64_t ldv, bf16_t* __restrict__ out, int64_t ldo, int H, int Sq, int Skv, int dh, float scale, float drop_p,
                 uint64_t drop_seed) {
  constexpr int DHP = 32 * DHS, PITCH = DHP * 2 + 16, Spad = NKT * 16;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* Kimg = sm; char* Vimg = sm + Spad * PITCH;
  float* kb = reinterpret_cast<float*>(sm + 2 * Spad * PITCH);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, l16 = lane & 15, nw = blockDim.x >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const bf16_t* Q = q + (int64_t)b * Sq * ldq + h * dh;
  ax_stage<DHP>(Kimg, k + (int64_t)b * Skv * ldk + h * dh, ldk, Skv, Spad, dh);
  ax_stage<DHP>(Vimg, v + (int64_t)b * Skv * ldv + h * dh, ldv, Skv, Spad, dh);
  for (int key = threadIdx.x; key < Spad; key += blockDim.x) kb[key] = key < Skv ? 0.f : -INFINITY;
  __syncthreads();
  const float c1 = scale * AX_LOG2E;               // scores in the exp2 domain
  const float drop_sc = DROP ? 1.f / (1.f - drop_p) : 1.f;
  const int nqt = (Sq + 15) / 16;
  for (int qt = wave; qt < nqt; qt += nw) {
    const int q0 = qt * 16;
    const int qr = min(q0 + l16, Sq - 1);
    bf16x8 qf[DHS];
#pragma unroll
    for (int ks = 0; ks < DHS; ++ks) {
      const int col = 32 * ks + 8 * g;
      bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      qf[ks] = col < dh ? *reinterpret_cast<const bf16x8*>(Q + (int64_t)qr * ldq + col) : z;
    }
    f32x4 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DHS; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kimg + (kt * 16 + l16) * PITCH + (32 * ks + 8 * g) * 2);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], a, 0, 0, 0);       // S^T[key][query]
      }
      s[kt] = a;
      // keep the scheduler from hoisting every K fragment of the row up front (it then runs out of VGPRs and spills)
      if (kt & 1) __builtin_amdgcn_sched_barrier(0);
    }
    float mm = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(kb + kt * 16 + 4 * g);     // keys 16kt + 4g + r
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float x = fmaf(s[kt][r], c1, bias[r]); s[kt][r] = x; mm = fmaxf(mm, x); }
    }
    mm = fmaxf(mm, __shfl_xor(mm, 16, 64));
    mm = fmaxf(mm, __shfl_xor(mm, 32, 64));
    float ll = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(s[kt][r] - mm); s[kt][r] = e; ll += e; }
    ll += __shfl_xor(ll, 16, 64);
    ll += __shfl_xor(ll, 32, 64);
    if (DROP) {        // dropout acts on the normalised probabilities: mask the numerators, keep the denominator
      const uint64_t idx0 = ((uint64_t)blockIdx.x * Sq + (q0 + l16)) * Skv + 4 * g;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kt][r] *= mmrca_uniform(drop_seed, idx0 + (kt * 16 + r)) >= drop_p ? drop_sc : 0.f;
        __builtin_amdgcn_sched_barrier(0);      // one tile's 64-bit hash temporaries at a time
      }
    }
    f32x4 o[2 * DHS];
#pragma unroll
    for (int dt = 0; dt < 2 * DHS; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NKT / 2; ++u) {
      bf16x8 pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) { pf[r] = (bf16_t)s[2 * u][r]; pf[4 + r] = (bf16_t)s[2 * u + 1][r]; }
      const int r0 = 32 * u + 4 * g + (l16 >> 2);
#pragma unroll
      for (int dt = 0; dt < 2 * DHS; ++dt) {
        const char* a0 = Vimg + r0 * PITCH + (16 * dt + 4 * (l16 & 3)) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0 + 16 * PITCH));
        bf16x8 vf;
        vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3]; vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);     // O^T[d][query]
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const int qi = q0 + l16;
    if (qi < Sq) {
      const float inv = 1.f / ll;
      bf16_t* orow = out + ((int64_t)b * Sq + qi) * ldo + h * dh;
#pragma unroll
      for (int dt = 0; dt < 2 * DHS; ++dt) {
        const int col = dt * 16 + 4 * g;
        if (col < dh) {
          bf16x4 w;
#pragma unroll
          for (int r = 0; r < 4; ++r) w[r] = (bf16_t)(o[dt][r] * inv);
          *reinterpret_cast<bf16x4*>(orow + col) = w;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// bf16x3 form (round 5): fp32 q / k / v, every product three-pass on the bf16 matrix cores -- the fp32 arithmetic of
// Blip2Attention / Blip2QFormerMultiHeadAttention (the reference runs them in fp32, q_former_training.py:279-304) to ~1e-6 at the
// bf16 MFMA rate.  Head dim 88 at S = 257 does not fit four whole-head images (K_hi, K_lo, V_hi, V_lo) in 160 KB, so the keys are
// walked in CHUNKS of NKTC tiles with running (max, sum) per query row ("online softmax"); a wave keeps the accumulators of its
// TPW query tiles in registers across the chunks.  x = hi + lo with hi = bf16(x), lo = bf16(x - hi) is formed while staging (K, V)
// or loading (Q); the probabilities are split in registers; the context leaves as two bf16 planes -- the operand form of the
// bf16x3 projection GEMM that consumes it (mmrca_gemm_x3).
//   S^T = K_hi Q_hi + K_hi Q_lo + K_lo Q_hi        O^T += V_hi P_hi + V_hi P_lo + V_lo P_hi
// ------------------------------------------------------------------------------------------------------
template <int DHP>
__device__ __forceinline__ void ax_stage_split(char* hi_img, char* lo_img, const float* __restrict__ src, int64_t ld, int row0, int S,
                                               int rows, int dh) {
  constexpr int CH = DHP / 8, PITCH = DHP * 2 + 16;
  for (int e = threadIdx.x; e < rows * CH; e += blockDim.x) {
    const int row = e / CH, c = e % CH;
    bf16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row0 + row < S && c * 8 < dh) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + (int64_t)(row0 + row) * ld + c * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(src + (int64_t)(row0 + row) * ld + c * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = (bf16_t)a[j]; l[j] = (bf16_t)(a[j] - (float)h[j]);
        h[4 + j] = (bf16_t)b[j]; l[4 + j] = (bf16_t)(b[j] - (float)h[4 + j]);
      }
    }
    *reinterpret_cast<bf16x8*>(hi_img + row * PITCH + c * 16) = h;
    *reinterpret_cast<bf16x8*>(lo_img + row * PITCH + c * 16) = l;
  }
}

template <int NKTC, int DHS, bool DROP, int TPW>   // NKTC key tiles per chunk (even); TPW query tiles per wave
__global__ void __launch_bounds__(512)
mha_cross_x3_k(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k, int64_t ldk, const float* __restrict__ v,
               int64_t ldv, bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo, int64_t ldo, int H, int Sq, int Skv, int dh,
               float scale, float drop_p, uint64_t drop_seed) {
  constexpr int DHP = 32 * DHS, PITCH = DHP * 2 + 16, CROWS = NKTC * 16, IMG = CROWS * PITCH;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* Kh = sm; char* Kl = sm + IMG; char* Vh = sm + 2 * IMG; char* Vl = sm + 3 * IMG;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, l16 = lane & 15, nw = blockDim.x >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const float* Q = q + (int64_t)b * Sq * ldq + h * dh;
  const float* Kp = k + (int64_t)b * Skv * ldk + h * dh;
  const float* Vp = v + (int64_t)b * Skv * ldv + h * dh;
  const float c1 = scale * AX_LOG2E;
  const float drop_sc = DROP ? 1.f / (1.f - drop_p) : 1.f;
  const int nqt = (Sq + 15) / 16;
  f32x4 o[TPW][2 * DHS];
  float mrun[TPW], lrun[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    mrun[t] = -INFINITY; lrun[t] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2 * DHS; ++dt) o[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int k0 = 0; k0 < Skv; k0 += CROWS) {
    __syncthreads();                                   // every wave is done with the previous chunk's images
    ax_stage_split<DHP>(Kh, Kl, Kp, ldk, k0, Skv, CROWS, dh);
    ax_stage_split<DHP>(Vh, Vl, Vp, ldv, k0, Skv, CROWS, dh);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int qt = wave + t * nw;
      if (qt >= nqt) continue;                         // wave-uniform
      const int q0 = qt * 16;
      const int qr = min(q0 + l16, Sq - 1);
      bf16x8 qh[DHS], ql[DHS];
#pragma unroll
      for (int ks = 0; ks < DHS; ++ks) {
        const int col = 32 * ks + 8 * g;
        bf16x8 zh = {0, 0, 0, 0, 0, 0, 0, 0}, zl = {0, 0, 0, 0, 0, 0, 0, 0};
        if (col < dh) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(Q + (int64_t)qr * ldq + col);
          const f32x4 c = *reinterpret_cast<const f32x4*>(Q + (int64_t)qr * ldq + col + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            zh[j] = (bf16_t)a[j]; zl[j] = (bf16_t)(a[j] - (float)zh[j]);
            zh[4 + j] = (bf16_t)c[j]; zl[4 + j] = (bf16_t)(c[j] - (float)zh[4 + j]);
          }
        }
        qh[ks] = zh; ql[ks] = zl;
      }
      f32x4 s[NKTC];
#pragma unroll
      for (int kt = 0; kt < NKTC; ++kt) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < DHS; ++ks) {
          const int off = (kt * 16 + l16) * PITCH + (32 * ks + 8 * g) * 2;
          const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Kh + off);
          const bf16x8 kl = *reinterpret_cast<const bf16x8*>(Kl + off);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[ks], a, 0, 0, 0);       // small terms first
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[ks], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[ks], a, 0, 0, 0);       // S^T[key][query]
        }
        s[kt] = a;
        if (kt & 1) __builtin_amdgcn_sched_barrier(0);
      }
      float mm = mrun[t];
#pragma unroll
      for (int kt = 0; kt < NKTC; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = k0 + kt * 16 + 4 * g + r < Skv ? s[kt][r] * c1 : -INFINITY;
          s[kt][r] = x; mm = fmaxf(mm, x);
        }
      mm = fmaxf(mm, __shfl_xor(mm, 16, 64));
      mm = fmaxf(mm, __shfl_xor(mm, 32, 64));
      const float alpha = __builtin_amdgcn_exp2f(mrun[t] - mm);      // (first chunk: exp2(-inf) = 0; every chunk has a live key or mm stays)
      mrun[t] = mm;
      float ll = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKTC; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(s[kt][r] - mm); s[kt][r] = e; ll += e; }
      ll += __shfl_xor(ll, 16, 64);
      ll += __shfl_xor(ll, 32, 64);
      lrun[t] = lrun[t] * alpha + ll;
#pragma unroll
      for (int dt = 0; dt < 2 * DHS; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[t][dt][r] *= alpha;
      if (DROP) {        // dropout acts on the normalised probabilities: mask the numerators, keep the denominator
        const uint64_t idx0 = ((uint64_t)blockIdx.x * Sq + (q0 + l16)) * Skv + k0 + 4 * g;
#pragma unroll
        for (int kt = 0; kt < NKTC; ++kt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kt][r] *= mmrca_uniform(drop_seed, idx0 + (kt * 16 + r)) >= drop_p ? drop_sc : 0.f;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int u = 0; u < NKTC / 2; ++u) {
        bf16x8 ph, pl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ph[r] = (bf16_t)s[2 * u][r]; pl[r] = (bf16_t)(s[2 * u][r] - (float)ph[r]);
          ph[4 + r] = (bf16_t)s[2 * u + 1][r]; pl[4 + r] = (bf16_t)(s[2 * u + 1][r] - (float)ph[4 + r]);
        }
        const int r0 = 32 * u + 4 * g + (l16 >> 2);
#pragma unroll
        for (int dt = 0; dt < 2 * DHS; ++dt) {
          const int off = r0 * PITCH + (16 * dt + 4 * (l16 & 3)) * 2;
          bf16x8 vh, vl;
          {
            const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Vh + off));
            const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Vh + off + 16 * PITCH));
            vh[0] = a0[0]; vh[1] = a0[1]; vh[2] = a0[2]; vh[3] = a0[3]; vh[4] = a1[0]; vh[5] = a1[1]; vh[6] = a1[2]; vh[7] = a1[3];
            const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Vl + off));
            const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Vl + off + 16 * PITCH));
            vl[0] = b0[0]; vl[1] = b0[1]; vl[2] = b0[2]; vl[3] = b0[3]; vl[4] = b1[0]; vl[5] = b1[1]; vl[6] = b1[2]; vl[7] = b1[3];
          }
          f32x4 acc = o[t][dt];
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, ph, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph, acc, 0, 0, 0);       // O^T[d][query]
          o[t][dt] = acc;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int qt = wave + t * nw;
    const int qi = qt * 16 + l16;
    if (qt < nqt && qi < Sq) {
      const float inv = 1.f / lrun[t];
      bf16_t* oh = out_hi + ((int64_t)b * Sq + qi) * ldo + h * dh;
      bf16_t* ol = out_lo + ((int64_t)b * Sq + qi) * ldo + h * dh;
#pragma unroll
      for (int dt = 0; dt < 2 * DHS; ++dt) {
        const int col = dt * 16 + 4 * g;
        if (col < dh) {
          bf16x4 wh, wl;
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float x = o[t][dt][r] * inv; wh[r] = (bf16_t)x; wl[r] = (bf16_t)(x - (float)wh[r]); }
          *reinterpret_cast<bf16x4*>(oh + col) = wh;
          *reinterpret_cast<bf16x4*>(ol + col) = wl;
        }
      }
    }
  }
}

static bool ax_aligned(const void* p, int64_t ld, int a) { return (((uintptr_t)p) % a) == 0 && (ld * 2) % a == 0; }

template <int NKT, int DHS, bool DROP>
static int ax_launch_d(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out, int64_t ldo,
                     int B, int H, int Sq, int Skv, int dh, float scale, float drop_p, uint64_t seed, hipStream_t st) {
  constexpr int PITCH = 32 * DHS * 2 + 16;
  const int lds = 2 * NKT * 16 * PITCH + NKT * 16 * 4;
  const int nw = Sq > 64 ? 8 : 4;
  MMRCA_MAX_LDS(lds, mha_cross_mfma_k<NKT, DHS, DROP>);
  hipLaunchKernelGGL((mha_cross_mfma_k<NKT, DHS, DROP>), dim3(B * H), dim3(64 * nw), lds, st, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk,
                     (const bf16_t*)v, ldv, (bf16_t*)out, ldo, H, Sq, Skv, dh, scale, drop_p, seed);
  MMRCA_CHECK_LAUNCH("mha_cross_fwd(mfma)");
  return 0;
}

template <int NKT, int DHS>
static int ax_launch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out, int64_t ldo,
                     int B, int H, int Sq, int Skv, int dh, float scale, float drop_p, uint64_t seed, hipStream_t st) {
  if (drop_p > 0.f) return ax_launch_d<NKT, DHS, true>(q, ldq, k, ldk, v, ldv, out, ldo, B, H, Sq, Skv, dh, scale, drop_p, seed, st);
  return ax_launch_d<NKT, DHS, false>(q, ldq, k, ldk, v, ldv, out, ldo, B, H, Sq, Skv, dh, scale, drop_p, seed, st);
}

// the MFMA kernel takes bf16, dh % 8 == 0 <= 128, S_kv <= 288 (both LDS images of a head fit 160 KB), 16-byte aligned rows
// (8-byte for the output)
static bool ax_mfma_ok(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* out, int64_t ldo,
                       int Skv, int dh, int dtype) {
  return dtype == MMRCA_BF16 && dh % 8 == 0 && dh <= 128 && Skv <= 288 && ax_aligned(q, ldq, 16) && ax_aligned(k, ldk, 16) &&
         ax_aligned(v, ldv, 16) && ax_aligned(out, ldo, 8);
}

extern "C" int mmrca_mha_cross_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out,
                                   int64_t ldo, int B, int H, int Sq, int Skv, int dh, float scale, float drop_p, uint64_t drop_seed,
                                   int dtype, int impl, void* stream) {
  MMRCA_REQUIRE(q && k && v && out, "mha_cross_fwd: null pointer");
  MMRCA_REQUIRE(B > 0 && H > 0 && Sq > 0 && Skv > 0 && dh > 0, "mha_cross_fwd: bad shape");
  MMRCA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "mha_cross_fwd: dropout p must be in [0,1)");
  MMRCA_REQUIRE(ldq >= (int64_t)H * dh && ldk >= (int64_t)H * dh && ldv >= (int64_t)H * dh && ldo >= (int64_t)H * dh,
                "mha_cross_fwd: row strides must cover H*dh = %d columns", H * dh);
  hipStream_t st = (hipStream_t)stream;
  const bool ok = ax_mfma_ok(q, ldq, k, ldk, v, ldv, out, ldo, Skv, dh, dtype);
  if (impl == MMRCA_GEMM_MFMA && !ok)
    return mmrca_fail(-3, "mha_cross_fwd: S_kv=%d dh=%d dtype=%d (or operand alignment) does not qualify for the MFMA kernel", Skv, dh, dtype);
  if (ok && impl != MMRCA_GEMM_REF) {
    const int dhs = (dh + 31) / 32, nkt = (Skv + 15) / 16;
#define AX_CASE(N_)                                                                                                              \
  if (nkt <= N_) {                                                                                                               \
    if (dhs <= 2) return ax_launch<N_, 2>(q, ldq, k, ldk, v, ldv, out, ldo, B, H, Sq, Skv, dh, scale, drop_p, drop_seed, st);    \
    if (dhs == 3) return ax_launch<N_, 3>(q, ldq, k, ldk, v, ldv, out, ldo, B, H, Sq, Skv, dh, scale, drop_p, drop_seed, st);    \
    return ax_launch<N_, 4>(q, ldq, k, ldk, v, ldv, out, ldo, B, H, Sq, Skv, dh, scale, drop_p, drop_seed, st);                  \
  }
    AX_CASE(4) AX_CASE(8) AX_CASE(14) AX_CASE(18)
#undef AX_CASE
  }
  MMRCA_REQUIRE(Skv <= AX_MAX_S && dh <= AX_MAX_DH, "mha_cross_fwd(ref): S_kv=%d dh=%d unsupported", Skv, dh);
  dim3 grid((Sq + 3) / 4, B * H);
  const size_t lds = 4 * (AX_MAX_DH + Skv) * sizeof(float);
  MMRCA_DISPATCH_DTYPE(dtype, "mha_cross_fwd",
    hipLaunchKernelGGL(mha_cross_ref_k<T>, grid, dim3(256), lds, st, (const T*)q, ldq, (const T*)k, ldk, (const T*)v, ldv, (T*)out, ldo,
                       H, Sq, Skv, dh, scale, drop_p, drop_seed);)
  MMRCA_CHECK_LAUNCH("mha_cross_fwd(ref)");
  return 0;
}

template <int NKTC, int DHS, bool DROP, int TPW>
static int ax3_launch_t(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* oh, void* ol, int64_t ldo,
                        int B, int H, int Sq, int Skv, int dh, float scale, float drop_p, uint64_t seed, int nw, hipStream_t st) {
  constexpr int PITCH = 32 * DHS * 2 + 16;
  const int lds = 4 * NKTC * 16 * PITCH;
  MMRCA_MAX_LDS(lds, mha_cross_x3_k<NKTC, DHS, DROP, TPW>);
  hipLaunchKernelGGL((mha_cross_x3_k<NKTC, DHS, DROP, TPW>), dim3(B * H), dim3(64 * nw), lds, st, (const float*)q, ldq, (const float*)k, ldk,
                     (const float*)v, ldv, (bf16_t*)oh, (bf16_t*)ol, ldo, H, Sq, Skv, dh, scale, drop_p, seed);
  MMRCA_CHECK_LAUNCH("mha_cross_fwd_x3");
  return 0;
}
template <int NKTC, int DHS>
static int ax3_launch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* oh, void* ol, int64_t ldo,
                      int B, int H, int Sq, int Skv, int dh, float scale, float drop_p, uint64_t seed, hipStream_t st) {
  const int nqt = (Sq + 15) / 16;
  const int nw = nqt > 4 ? 8 : 4;
  const int tpw = (nqt + nw - 1) / nw;
#define AX3_ARGS q, ldq, k, ldk, v, ldv, oh, ol, ldo, B, H, Sq, Skv, dh, scale, drop_p, seed, nw, st
  if (drop_p > 0.f) {
    if (tpw <= 1) return ax3_launch_t<NKTC, DHS, true, 1>(AX3_ARGS);
    if (tpw == 2) return ax3_launch_t<NKTC, DHS, true, 2>(AX3_ARGS);
    if (tpw == 3) return ax3_launch_t<NKTC, DHS, true, 3>(AX3_ARGS);
  } else {
    if (tpw <= 1) return ax3_launch_t<NKTC, DHS, false, 1>(AX3_ARGS);
    if (tpw == 2) return ax3_launch_t<NKTC, DHS, false, 2>(AX3_ARGS);
    if (tpw == 3) return ax3_launch_t<NKTC, DHS, false, 3>(AX3_ARGS);
  }
#undef AX3_ARGS
  return mmrca_fail(-3, "mha_cross_fwd_x3: S_q=%d needs more than three query tiles per wave (S_q <= 384)", Sq);
}

extern "C" int mmrca_mha_cross_fwd_x3(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, void* out_hi,
                                      void* out_lo, int64_t ldo, int B, int H, int Sq, int Skv, int dh, float scale, float drop_p,
                                      uint64_t drop_seed, void* stream) {
  MMRCA_REQUIRE(q && k && v && out_hi && out_lo, "mha_cross_fwd_x3: null pointer");
  MMRCA_REQUIRE(B > 0 && H > 0 && Sq > 0 && Skv > 0 && dh > 0, "mha_cross_fwd_x3: bad shape");
  MMRCA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "mha_cross_fwd_x3: dropout p must be in [0,1)");
  MMRCA_REQUIRE(dh % 8 == 0 && dh <= 128 && Sq <= 384, "mha_cross_fwd_x3: head dim %d / S_q %d unsupported (dh %% 8 == 0, <= 128; S_q <= 384)", dh, Sq);
  MMRCA_REQUIRE(ldq >= (int64_t)H * dh && ldk >= (int64_t)H * dh && ldv >= (int64_t)H * dh && ldo >= (int64_t)H * dh,
                "mha_cross_fwd_x3: row strides must cover H*dh = %d columns", H * dh);
  MMRCA_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 &&
                (((uintptr_t)out_hi | (uintptr_t)out_lo) & 7) == 0 && ldo % 4 == 0, "mha_cross_fwd_x3: operands must be 16-byte aligned rows (8 for the planes)");
  hipStream_t st = (hipStream_t)stream;
  const int dhs = (dh + 31) / 32, nkt = (Skv + 15) / 16;
  // chunk = the largest even tile count whose four images fit 160 KB: 10 tiles (160 keys) up to head dim 96, 6 beyond; short key
  // sets take one small chunk
#define AX3_CASE(N_, D_) return ax3_launch<N_, D_>(q, ldq, k, ldk, v, ldv, out_hi, out_lo, ldo, B, H, Sq, Skv, dh, scale, drop_p, drop_seed, st)
  if (dhs <= 2) { if (nkt <= 2) AX3_CASE(2, 2); if (nkt <= 4) AX3_CASE(4, 2); AX3_CASE(10, 2); }
  if (dhs == 3) { if (nkt <= 2) AX3_CASE(2, 3); if (nkt <= 4) AX3_CASE(4, 3); AX3_CASE(10, 3); }
  if (nkt <= 2) AX3_CASE(2, 4); if (nkt <= 4) AX3_CASE(4, 4); AX3_CASE(6, 4);
#undef AX3_CASE
}

MMRCA_SEED_EPOCH_EXPORT(attention_cross)   // this translation unit's copy of the mask epoch (common.h)
