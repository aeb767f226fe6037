// K3c: attention of the class-token query only (query row 0 of every sequence) -- the top encoder layer under
// MMRCA_CLS_TAIL, where no other row of the attention output is ever read.  One block per (batch, head):
//   forward : out[b, h*dh..] = softmax(q0 K^T * scale) V, lse[b,h]
//   backward: dQ row 0, dK / dV for ALL key rows (they receive the class-token gradient), dQ rows >= 1 := 0, so the
//             fused dqkv buffer is complete for the in-projection backward.
// Same masking / dropout semantics and the same counter indices ((b*H+h)*S + 0)*S + key as the full kernels, so the
// results equal row 0 of mmrca_mha_fwd and mmrca_mha_bwd with dout zero outside row 0.  HBM-bound (reads K, V once,
// writes dqkv once, 16-lane-per-row coalesced); any dtype, fp32 math.
#include "common.h"

#define CLS_MAX_S 1024
#define CLS_MAX_DH 128

template <typename T>
__device__ __forceinline__ float dot_lds(const T* __restrict__ row, const float* __restrict__ vec, int dh) {
  float a = 0.f;
  for (int d = 0; d < dh; d += 4) {
    Vec4<T> v = Vec4<T>::load(row + d);
    a += v.v[0] * vec[d] + v.v[1] * vec[d + 1] + v.v[2] * vec[d + 2] + v.v[3] * vec[d + 3];
  }
  return a;
}

__device__ __forceinline__ float block_reduce(float x, float* red, bool is_max) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  x = is_max ? wave_max(x) : wave_sum(x);
  __syncthreads();                       // red may still be read from a previous reduction
  if (lane == 0) red[wave] = x;
  __syncthreads();
  return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1] + red[2] + red[3]);
}

// grid (B*H), block 256
template <typename T>
__global__ void __launch_bounds__(256)
mha_cls_fwd_k(const T* __restrict__ qkv, const int32_t* __restrict__ key_mask, T* __restrict__ out, float* __restrict__ lse,
              int B, int H, int Smax, int dh, float scale, float drop_p, uint64_t drop_seed, const int32_t* __restrict__ cu) {
  __shared__ float qv[CLS_MAX_DH], p[CLS_MAX_S], red[4], part[4][CLS_MAX_DH];
  const float drop_sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int b = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x;
  const int row0 = cu ? cu[b] : b * Smax, S = cu ? cu[b + 1] - cu[b] : Smax;     // packed / padded layout (see mmrca.h)
  if (S <= 0) return;
  const int64_t ld = 3LL * H * dh;
  const T* Q = qkv + (int64_t)row0 * ld + h * dh;
  const T* Kp = Q + H * dh;
  const T* Vp = Kp + H * dh;
  for (int d = tid; d < dh; d += 256) qv[d] = to_f(Q[d]);
  __syncthreads();
  float m = -INFINITY;
  for (int j = tid; j < S; j += 256) {
    float s = -INFINITY;
    if (!key_mask || key_mask[row0 + j] != 0) s = dot_lds(Kp + (int64_t)j * ld, qv, dh) * scale;
    p[j] = s;
    m = fmaxf(m, s);
  }
  m = block_reduce(m, red, true);
  float l = 0.f;
  for (int j = tid; j < S; j += 256) {
    const float e = m > -INFINITY ? __expf(p[j] - m) : 0.f;
    l += e;
    p[j] = (drop_p > 0.f && mmrca_uniform(drop_seed, ((uint64_t)blockIdx.x * Smax + 0) * Smax + j) < drop_p) ? 0.f : e * drop_sc;
  }
  l = block_reduce(l, red, false);       // (its barriers also publish p[])
  const float inv = l > 0.f ? 1.f / l : 0.f;
  // out[d] = inv * sum_j p[j] V[j][d]: thread = (column group of 4, key slice); 16-lane-per-row coalesced reads of V
  const int ncg = dh / 4, slices = 256 / ncg;
  const int cg = tid % ncg, sl = tid / ncg;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  if (sl < slices) {
    for (int j = sl; j < S; j += slices) {
      const float pj = p[j];
      Vec4<T> v = Vec4<T>::load(Vp + (int64_t)j * ld + cg * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] += pj * v.v[r];
    }
  }
  // reduce the key slices: 64-lane groups hold 64/ncg slices each -> shuffle within the wave, then 4 waves through LDS
  for (int off = ncg; off < 64; off <<= 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] += __shfl_xor(o[r], off, 64);
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < ncg) {
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][lane * 4 + r] = o[r];
  }
  __syncthreads();
  for (int d = tid; d < dh; d += 256)
    out[(int64_t)b * (H * dh) + h * dh + d] = from_f<T>((part[0][d] + part[1][d] + part[2][d] + part[3][d]) * inv);
  if (tid == 0) lse[blockIdx.x] = l > 0.f ? m + __logf(l) : INFINITY;
}

// grid (B*H), block 256
template <typename T>
__global__ void __launch_bounds__(256)
mha_cls_bwd_k(const T* __restrict__ qkv, const int32_t* __restrict__ key_mask, const T* __restrict__ out,
              const T* __restrict__ dout, const float* __restrict__ lse, T* __restrict__ dqkv,
              int B, int H, int Smax, int dh, float scale, float drop_p, uint64_t drop_seed, const int32_t* __restrict__ cu) {
  __shared__ float qv[CLS_MAX_DH], dov[CLS_MAX_DH], pk[CLS_MAX_S], ds[CLS_MAX_S], red[4], part[4][CLS_MAX_DH];
  const float drop_sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int b = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x;
  const int row0 = cu ? cu[b] : b * Smax, S = cu ? cu[b + 1] - cu[b] : Smax;
  if (S <= 0) return;
  const int64_t ld = 3LL * H * dh;
  const T* Q = qkv + (int64_t)row0 * ld + h * dh;
  const T* Kp = Q + H * dh;
  const T* Vp = Kp + H * dh;
  T* dQ = dqkv + (int64_t)row0 * ld + h * dh;
  T* dK = dQ + H * dh;
  T* dV = dK + H * dh;
  float dsum = 0.f;
  for (int d = tid; d < dh; d += 256) {
    qv[d] = to_f(Q[d]);
    const float g = to_f(dout[(int64_t)b * (H * dh) + h * dh + d]);
    dov[d] = g;
    dsum += g * to_f(out[(int64_t)b * (H * dh) + h * dh + d]);
  }
  dsum = block_reduce(dsum, red, false);   // (its barriers also publish qv / dov)
  const float Lse = lse[blockIdx.x];
  for (int j = tid; j < S; j += 256) {
    float pkj = 0.f, dsj = 0.f;
    if (!key_mask || key_mask[row0 + j] != 0) {
      const float s = dot_lds(Kp + (int64_t)j * ld, qv, dh) * scale;
      const float pj = __expf(s - Lse);
      const float dp = dot_lds(Vp + (int64_t)j * ld, dov, dh);
      const float keep = drop_p > 0.f ? (mmrca_uniform(drop_seed, ((uint64_t)blockIdx.x * Smax + 0) * Smax + j) >= drop_p ? drop_sc : 0.f) : 1.f;
      dsj = pj * (dp * keep - dsum) * scale;
      pkj = pj * keep;
    }
    pk[j] = pkj; ds[j] = dsj;
  }
  __syncthreads();
  // row-coalesced writes: (dh/4) lanes per row; dV[j] = pk[j] * do, dK[j] = ds[j] * q, dQ[j >= 1] = 0; and the
  // class-token dQ = sum_j ds[j] K[j] accumulated per (column group, key slice) on the way
  const int ncg = dh / 4, slices = 256 / ncg;
  const int cg = tid % ncg, sl = tid / ncg;
  float qa[4] = {0.f, 0.f, 0.f, 0.f};
  if (sl < slices) {
    float dor[4], qr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { dor[r] = dov[cg * 4 + r]; qr[r] = qv[cg * 4 + r]; }
    for (int j = sl; j < S; j += slices) {
      const float pkj = pk[j], dsj = ds[j];
      Vec4<T> k = Vec4<T>::load(Kp + (int64_t)j * ld + cg * 4), ov, ok, oz;
#pragma unroll
      for (int r = 0; r < 4; ++r) { ov.v[r] = pkj * dor[r]; ok.v[r] = dsj * qr[r]; oz.v[r] = 0.f; qa[r] += dsj * k.v[r]; }
      ov.store(dV + (int64_t)j * ld + cg * 4);
      ok.store(dK + (int64_t)j * ld + cg * 4);
      if (j > 0) oz.store(dQ + (int64_t)j * ld + cg * 4);
    }
  }
  for (int off = ncg; off < 64; off <<= 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) qa[r] += __shfl_xor(qa[r], off, 64);
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < ncg) {
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][lane * 4 + r] = qa[r];
  }
  __syncthreads();
  for (int d = tid; d < dh; d += 256) dQ[d] = from_f<T>(part[0][d] + part[1][d] + part[2][d] + part[3][d]);
}

static bool cls_shape_ok(int S, int dh) {
  // dh/4 lanes per row must divide a wave (dh in {16, 32, 64, 128}) so the slice reduction stays inside the wave
  return S <= CLS_MAX_S && dh <= CLS_MAX_DH && (dh == 16 || dh == 32 || dh == 64 || dh == 128);
}

extern "C" int mmrca_mha_cls_fwd(const void* qkv, const int32_t* key_mask, void* out, float* lse, int B, int H, int S, int dh,
                                 float scale, float drop_p, uint64_t drop_seed, const int32_t* cu_seqlens, int dtype, void* stream) {
  MMRCA_REQUIRE(qkv && out && lse, "mha_cls_fwd: null pointer");
  MMRCA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "mha_cls_fwd: dropout p must be in [0,1)");
  MMRCA_REQUIRE(B > 0 && H > 0 && S > 0 && cls_shape_ok(S, dh), "mha_cls_fwd: B=%d H=%d S=%d dh=%d unsupported", B, H, S, dh);
  MMRCA_DISPATCH_DTYPE(dtype, "mha_cls_fwd",
    hipLaunchKernelGGL(mha_cls_fwd_k<T>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, (const T*)qkv, key_mask, (T*)out, lse,
                       B, H, S, dh, scale, drop_p, drop_seed, cu_seqlens);)
  MMRCA_CHECK_LAUNCH("mha_cls_fwd");
  return 0;
}

extern "C" int mmrca_mha_cls_bwd(const void* qkv, const int32_t* key_mask, const void* out, const void* dout, const float* lse,
                                 void* dqkv, int B, int H, int S, int dh, float scale, float drop_p, uint64_t drop_seed,
                                 const int32_t* cu_seqlens, int dtype, void* stream) {
  MMRCA_REQUIRE(qkv && out && dout && lse && dqkv, "mha_cls_bwd: null pointer");
  MMRCA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "mha_cls_bwd: dropout p must be in [0,1)");
  MMRCA_REQUIRE(B > 0 && H > 0 && S > 0 && cls_shape_ok(S, dh), "mha_cls_bwd: B=%d H=%d S=%d dh=%d unsupported", B, H, S, dh);
  MMRCA_DISPATCH_DTYPE(dtype, "mha_cls_bwd",
    hipLaunchKernelGGL(mha_cls_bwd_k<T>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, (const T*)qkv, key_mask, (const T*)out,
                       (const T*)dout, lse, (T*)dqkv, B, H, S, dh, scale, drop_p, drop_seed, cu_seqlens);)
  MMRCA_CHECK_LAUNCH("mha_cls_bwd");
  return 0;
}

MMRCA_SEED_EPOCH_EXPORT(attention_cls)   // this translation unit's copy of the mask epoch (common.h)
