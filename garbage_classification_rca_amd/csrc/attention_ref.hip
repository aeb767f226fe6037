// K3 (reference-grade): multi-head attention forward / backward on a fused QKV buffer, any dtype, fp32 math.
// One wave per query row (forward, dQ) or per key row (dK, dV): lanes own keys (resp. queries) for the score
// phase and head-dim channels for the value phase; probabilities cross over through LDS.  This is the fp32
// "parity mode" kernel and the checker for the MFMA kernel in attention_mfma.hip; it is not the fast path.
#include "common.h"
#include <stdlib.h>

#define ATT_MAX_S 1024
#define ATT_MAX_DH 128

template <typename T>
__device__ __forceinline__ float dot_row(const T* __restrict__ row, const float* __restrict__ vec_lds, int dh) {
  float a = 0.f;
  for (int d = 0; d < dh; d += 4) {
    Vec4<T> v = Vec4<T>::load(row + d);
    a += v.v[0] * vec_lds[d] + v.v[1] * vec_lds[d + 1] + v.v[2] * vec_lds[d + 2] + v.v[3] * vec_lds[d + 3];
  }
  return a;
}

// grid (ceil(S/4), B*H), block 256 (4 waves = 4 query rows)
template <typename T>
__global__ void __launch_bounds__(256)
mha_fwd_ref_k(const T* __restrict__ qkv, const int32_t* __restrict__ key_mask, T* __restrict__ out, float* __restrict__ lse,
              int B, int H, int Smax, int dh, float scale, float drop_p, uint64_t drop_seed, const int32_t* __restrict__ cu) {
  const float drop_sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  extern __shared__ float sm[];     // per wave: q[dh] | p[S]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y / H, h = blockIdx.y % H;
  const int row0 = cu ? cu[b] : b * Smax, S = cu ? cu[b + 1] - cu[b] : Smax;     // packed / padded layout (see mmrca.h)
  const int i = blockIdx.x * 4 + wave;
  float* qv = sm + wave * (ATT_MAX_DH + Smax);
  float* p = qv + ATT_MAX_DH;
  const int64_t ld = 3LL * H * dh;
  const T* Q = qkv + (int64_t)row0 * ld + h * dh;
  const T* Kp = Q + H * dh;
  const T* Vp = Kp + H * dh;
  if (i >= S) return;                      // whole wave exits together (i is wave-uniform); no block barrier below
  for (int d = lane; d < dh; d += 64) qv[d] = to_f(Q[(int64_t)i * ld + d]);
  __builtin_amdgcn_wave_barrier();
  float m = -INFINITY;
  for (int j = lane; j < S; j += 64) {
    float s = -INFINITY;
    if (!key_mask || key_mask[row0 + j] != 0) s = dot_row(Kp + (int64_t)j * ld, qv, dh) * scale;
    p[j] = s;
    m = fmaxf(m, s);
  }
  m = wave_max(m);
  float l = 0.f;
  if (m > -INFINITY) {
    for (int j = lane; j < S; j += 64) { const float e = __expf(p[j] - m); p[j] = e; l += e; }
  } else {
    for (int j = lane; j < S; j += 64) p[j] = 0.f;
  }
  l = wave_sum(l);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  if (drop_p > 0.f)
    for (int j = lane; j < S; j += 64)
      p[j] *= mmrca_uniform(drop_seed, ((uint64_t)blockIdx.y * Smax + i) * Smax + j) >= drop_p ? drop_sc : 0.f;
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < dh; d += 64) {
    float o = 0.f;
    for (int j = 0; j < S; ++j) o += p[j] * to_f(Vp[(int64_t)j * ld + d]);
    out[((int64_t)row0 + i) * (H * dh) + h * dh + d] = from_f<T>(o * inv);
  }
  if (lane == 0) lse[((int64_t)b * H + h) * Smax + i] = l > 0.f ? m + __logf(l) : INFINITY;
}

// dQ: one wave per query row
template <typename T>
__global__ void __launch_bounds__(256)
mha_bwd_dq_ref_k(const T* __restrict__ qkv, const int32_t* __restrict__ key_mask, const T* __restrict__ out,
                 const T* __restrict__ dout, const float* __restrict__ lse, T* __restrict__ dqkv,
                 int B, int H, int Smax, int dh, float scale, float drop_p, uint64_t drop_seed, const int32_t* __restrict__ cu) {
  const float drop_sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  extern __shared__ float sm[];     // per wave: q[dh] | do[dh] | ds[S]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y / H, h = blockIdx.y % H;
  const int row0 = cu ? cu[b] : b * Smax, S = cu ? cu[b + 1] - cu[b] : Smax;
  const int i = blockIdx.x * 4 + wave;
  float* qv = sm + wave * (2 * ATT_MAX_DH + Smax);
  float* dov = qv + ATT_MAX_DH;
  float* ds = dov + ATT_MAX_DH;
  const int64_t ld = 3LL * H * dh, ldo = (int64_t)H * dh;
  const T* Q = qkv + (int64_t)row0 * ld + h * dh;
  const T* Kp = Q + H * dh;
  const T* Vp = Kp + H * dh;
  if (i >= S) return;
  float dsum = 0.f;
  for (int d = lane; d < dh; d += 64) {
    qv[d] = to_f(Q[(int64_t)i * ld + d]);
    const float g = to_f(dout[((int64_t)row0 + i) * ldo + h * dh + d]);
    dov[d] = g;
    dsum += g * to_f(out[((int64_t)row0 + i) * ldo + h * dh + d]);
  }
  dsum = wave_sum(dsum);
  __builtin_amdgcn_wave_barrier();
  const float L = lse[((int64_t)b * H + h) * Smax + i];
  for (int j = lane; j < S; j += 64) {
    float v = 0.f;
    if (!key_mask || key_mask[row0 + j] != 0) {
      const float s = dot_row(Kp + (int64_t)j * ld, qv, dh) * scale;
      const float pj = __expf(s - L);
      float dp = dot_row(Vp + (int64_t)j * ld, dov, dh);
      if (drop_p > 0.f) dp *= mmrca_uniform(drop_seed, ((uint64_t)blockIdx.y * Smax + i) * Smax + j) >= drop_p ? drop_sc : 0.f;
      v = pj * (dp - dsum) * scale;
    }
    ds[j] = v;
  }
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < dh; d += 64) {
    float a = 0.f;
    for (int j = 0; j < S; ++j) a += ds[j] * to_f(Kp[(int64_t)j * ld + d]);
    dqkv[((int64_t)row0 + i) * ld + h * dh + d] = from_f<T>(a);
  }
}

// dK, dV: one wave per key row
template <typename T>
__global__ void __launch_bounds__(256)
mha_bwd_dkv_ref_k(const T* __restrict__ qkv, const int32_t* __restrict__ key_mask, const T* __restrict__ out,
                  const T* __restrict__ dout, const float* __restrict__ lse, T* __restrict__ dqkv,
                  int B, int H, int Smax, int dh, float scale, float drop_p, uint64_t drop_seed, const int32_t* __restrict__ cu) {
  const float drop_sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  extern __shared__ float sm[];     // per wave: k[dh] | v[dh] | p[S] | ds[S]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y / H, h = blockIdx.y % H;
  const int row0 = cu ? cu[b] : b * Smax, S = cu ? cu[b + 1] - cu[b] : Smax;
  const int j = blockIdx.x * 4 + wave;
  float* kv = sm + wave * (2 * ATT_MAX_DH + 2 * Smax);
  float* vv = kv + ATT_MAX_DH;
  float* p = vv + ATT_MAX_DH;
  float* ds = p + Smax;
  const int64_t ld = 3LL * H * dh, ldo = (int64_t)H * dh;
  const T* Q = qkv + (int64_t)row0 * ld + h * dh;
  const T* Kp = Q + H * dh;
  const T* Vp = Kp + H * dh;
  if (j >= S) return;
  const bool masked = key_mask && key_mask[row0 + j] == 0;
  for (int d = lane; d < dh; d += 64) { kv[d] = to_f(Kp[(int64_t)j * ld + d]); vv[d] = to_f(Vp[(int64_t)j * ld + d]); }
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < S; i += 64) {
    float pi = 0.f, dsi = 0.f;
    if (!masked) {
      const T* dorow = dout + ((int64_t)row0 + i) * ldo + h * dh;
      const T* orow = out + ((int64_t)row0 + i) * ldo + h * dh;
      const float s = dot_row(Q + (int64_t)i * ld, kv, dh) * scale;
      pi = __expf(s - lse[((int64_t)b * H + h) * Smax + i]);
      float dp = dot_row(dorow, vv, dh);
      float dsum = 0.f;
      for (int d = 0; d < dh; ++d) dsum += to_f(dorow[d]) * to_f(orow[d]);
      const float keep = drop_p > 0.f ? (mmrca_uniform(drop_seed, ((uint64_t)blockIdx.y * Smax + i) * Smax + j) >= drop_p ? drop_sc : 0.f) : 1.f;
      dsi = pi * (dp * keep - dsum) * scale;
      pi *= keep;
    }
    p[i] = pi; ds[i] = dsi;
  }
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < dh; d += 64) {
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < S; ++i) {
      ak += ds[i] * to_f(Q[(int64_t)i * ld + d]);
      av += p[i] * to_f(dout[((int64_t)row0 + i) * ldo + h * dh + d]);
    }
    dqkv[((int64_t)row0 + j) * ld + (int64_t)H * dh + h * dh + d] = from_f<T>(ak);
    dqkv[((int64_t)row0 + j) * ld + 2LL * H * dh + h * dh + d] = from_f<T>(av);
  }
}

// ------------------------------------------------------------------------------------------------------
// The same three kernels with the head's operands staged once in LDS (fp32, rows padded by one float: a lane that walks
// its own row hits a different bank than its neighbours).  One workgroup of 16 waves per (b, h) walks all rows instead of
// four rows per workgroup re-reading K and V (resp. Q, dO and O) from L2 for every row: this is what the fp32 mode
// (bench.py --dtype fp32) runs when the head fits (2 S (dh + 1) floats + per-wave scratch <= 160 KiB, i.e. S <= ~270 at
// dh = 64); same arithmetic, same summation order as the kernels above, so the results are bitwise the same.
// ------------------------------------------------------------------------------------------------------
#define ATL_WAVES 16

__device__ __forceinline__ float dot_lds(const float* __restrict__ row, const float* __restrict__ vec, int dh) {
  float a = 0.f;
  for (int d = 0; d < dh; d += 4) a += row[d] * vec[d] + row[d + 1] * vec[d + 1] + row[d + 2] * vec[d + 2] + row[d + 3] * vec[d + 3];
  return a;
}

template <typename T>
__device__ __forceinline__ void stage_f32(float* dst, const T* __restrict__ src, int64_t ld, int S, int dh) {
  const int pitch = dh + 1;
  for (int e = threadIdx.x; e < S * (dh >> 2); e += blockDim.x) {
    const int r = e / (dh >> 2), c = (e % (dh >> 2)) * 4;
    const Vec4<T> v = Vec4<T>::load(src + (int64_t)r * ld + c);
    float* o = dst + r * pitch + c;
    o[0] = v.v[0]; o[1] = v.v[1]; o[2] = v.v[2]; o[3] = v.v[3];
  }
}

template <typename T>
__global__ void __launch_bounds__(64 * ATL_WAVES)
mha_fwd_lds_k(const T* __restrict__ qkv, const int32_t* __restrict__ key_mask, T* __restrict__ out, float* __restrict__ lse,
              int B, int H, int Smax, int dh, float scale, float drop_p, uint64_t drop_seed, const int32_t* __restrict__ cu) {
  const float drop_sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  extern __shared__ float sm[];     // K[S][dh+1] | V[S][dh+1] | per wave: q[dh] | p[S]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, pitch = dh + 1;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int row0 = cu ? cu[b] : b * Smax, S = cu ? cu[b + 1] - cu[b] : Smax;
  if (S <= 0) return;
  float* Ks = sm; float* Vs = Ks + Smax * pitch;
  float* qv = Vs + Smax * pitch + wave * (ATT_MAX_DH + Smax);
  float* p = qv + ATT_MAX_DH;
  const int64_t ld = 3LL * H * dh;
  const T* Q = qkv + (int64_t)row0 * ld + h * dh;
  stage_f32(Ks, Q + H * dh, ld, S, dh);
  stage_f32(Vs, Q + 2 * H * dh, ld, S, dh);
  __syncthreads();
  for (int i = wave; i < S; i += ATL_WAVES) {
    for (int d = lane; d < dh; d += 64) qv[d] = to_f(Q[(int64_t)i * ld + d]);
    __builtin_amdgcn_wave_barrier();
    float m = -INFINITY;
    for (int j = lane; j < S; j += 64) {
      float sc = -INFINITY;
      if (!key_mask || key_mask[row0 + j] != 0) sc = dot_lds(Ks + j * pitch, qv, dh) * scale;
      p[j] = sc;
      m = fmaxf(m, sc);
    }
    m = wave_max(m);
    float l = 0.f;
    if (m > -INFINITY) {
      for (int j = lane; j < S; j += 64) { const float e = __expf(p[j] - m); p[j] = e; l += e; }
    } else {
      for (int j = lane; j < S; j += 64) p[j] = 0.f;
    }
    l = wave_sum(l);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if (drop_p > 0.f)
      for (int j = lane; j < S; j += 64)
        p[j] *= mmrca_uniform(drop_seed, ((uint64_t)blockIdx.x * Smax + i) * Smax + j) >= drop_p ? drop_sc : 0.f;
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < dh; d += 64) {
      float o = 0.f;
      for (int j = 0; j < S; ++j) o += p[j] * Vs[j * pitch + d];
      out[((int64_t)row0 + i) * (H * dh) + h * dh + d] = from_f<T>(o * inv);
    }
    if (lane == 0) lse[((int64_t)b * H + h) * Smax + i] = l > 0.f ? m + __logf(l) : INFINITY;
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename T>
__global__ void __launch_bounds__(64 * ATL_WAVES)
mha_bwd_dq_lds_k(const T* __restrict__ qkv, const int32_t* __restrict__ key_mask, const T* __restrict__ out,
                 const T* __restrict__ dout, const float* __restrict__ lse, T* __restrict__ dqkv,
                 int B, int H, int Smax, int dh, float scale, float drop_p, uint64_t drop_seed, const int32_t* __restrict__ cu) {
  const float drop_sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  extern __shared__ float sm[];     // K | V | per wave: q[dh] | do[dh] | ds[S]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, pitch = dh + 1;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int row0 = cu ? cu[b] : b * Smax, S = cu ? cu[b + 1] - cu[b] : Smax;
  if (S <= 0) return;
  float* Ks = sm; float* Vs = Ks + Smax * pitch;
  float* qv = Vs + Smax * pitch + wave * (2 * ATT_MAX_DH + Smax);
  float* dov = qv + ATT_MAX_DH;
  float* ds = dov + ATT_MAX_DH;
  const int64_t ld = 3LL * H * dh, ldo = (int64_t)H * dh;
  const T* Q = qkv + (int64_t)row0 * ld + h * dh;
  stage_f32(Ks, Q + H * dh, ld, S, dh);
  stage_f32(Vs, Q + 2 * H * dh, ld, S, dh);
  __syncthreads();
  for (int i = wave; i < S; i += ATL_WAVES) {
    float dsum = 0.f;
    for (int d = lane; d < dh; d += 64) {
      qv[d] = to_f(Q[(int64_t)i * ld + d]);
      const float g = to_f(dout[((int64_t)row0 + i) * ldo + h * dh + d]);
      dov[d] = g;
      dsum += g * to_f(out[((int64_t)row0 + i) * ldo + h * dh + d]);
    }
    dsum = wave_sum(dsum);
    __builtin_amdgcn_wave_barrier();
    const float L = lse[((int64_t)b * H + h) * Smax + i];
    for (int j = lane; j < S; j += 64) {
      float v = 0.f;
      if (!key_mask || key_mask[row0 + j] != 0) {
        const float sc = dot_lds(Ks + j * pitch, qv, dh) * scale;
        const float pj = __expf(sc - L);
        float dp = dot_lds(Vs + j * pitch, dov, dh);
        if (drop_p > 0.f) dp *= mmrca_uniform(drop_seed, ((uint64_t)blockIdx.x * Smax + i) * Smax + j) >= drop_p ? drop_sc : 0.f;
        v = pj * (dp - dsum) * scale;
      }
      ds[j] = v;
    }
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < dh; d += 64) {
      float a = 0.f;
      for (int j = 0; j < S; ++j) a += ds[j] * Ks[j * pitch + d];
      dqkv[((int64_t)row0 + i) * ld + h * dh + d] = from_f<T>(a);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename T>
__global__ void __launch_bounds__(64 * ATL_WAVES)
mha_bwd_dkv_lds_k(const T* __restrict__ qkv, const int32_t* __restrict__ key_mask, const T* __restrict__ out,
                  const T* __restrict__ dout, const float* __restrict__ lse, T* __restrict__ dqkv,
                  int B, int H, int Smax, int dh, float scale, float drop_p, uint64_t drop_seed, const int32_t* __restrict__ cu) {
  const float drop_sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  extern __shared__ float sm[];     // Q | dO | dsum[S] | per wave: k[dh] | v[dh] | p[S] | ds[S]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, pitch = dh + 1;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int row0 = cu ? cu[b] : b * Smax, S = cu ? cu[b + 1] - cu[b] : Smax;
  if (S <= 0) return;
  float* Qs = sm; float* Ds = Qs + Smax * pitch; float* dsum_s = Ds + Smax * pitch;
  float* kv = dsum_s + Smax + wave * (2 * ATT_MAX_DH + 2 * Smax);
  float* vv = kv + ATT_MAX_DH;
  float* p = vv + ATT_MAX_DH;
  float* ds = p + Smax;
  const int64_t ld = 3LL * H * dh, ldo = (int64_t)H * dh;
  const T* Q = qkv + (int64_t)row0 * ld + h * dh;
  const T* Kp = Q + H * dh;
  const T* Vp = Kp + H * dh;
  const T* dO = dout + (int64_t)row0 * ldo + h * dh;
  const T* O = out + (int64_t)row0 * ldo + h * dh;
  stage_f32(Qs, Q, ld, S, dh);
  stage_f32(Ds, dO, ldo, S, dh);
  for (int i = threadIdx.x; i < S; i += blockDim.x) {       // rowsum(dO * O), summed over d in order like the row-wise kernel
    float a = 0.f;
    for (int d = 0; d < dh; ++d) a += to_f(dO[(int64_t)i * ldo + d]) * to_f(O[(int64_t)i * ldo + d]);
    dsum_s[i] = a;
  }
  __syncthreads();
  for (int j = wave; j < S; j += ATL_WAVES) {
    const bool masked = key_mask && key_mask[row0 + j] == 0;
    for (int d = lane; d < dh; d += 64) { kv[d] = to_f(Kp[(int64_t)j * ld + d]); vv[d] = to_f(Vp[(int64_t)j * ld + d]); }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < S; i += 64) {
      float pi = 0.f, dsi = 0.f;
      if (!masked) {
        const float sc = dot_lds(Qs + i * pitch, kv, dh) * scale;
        pi = __expf(sc - lse[((int64_t)b * H + h) * Smax + i]);
        const float dp = dot_lds(Ds + i * pitch, vv, dh);
        const float keep = drop_p > 0.f ? (mmrca_uniform(drop_seed, ((uint64_t)blockIdx.x * Smax + i) * Smax + j) >= drop_p ? drop_sc : 0.f) : 1.f;
        dsi = pi * (dp * keep - dsum_s[i]) * scale;
        pi *= keep;
      }
      p[i] = pi; ds[i] = dsi;
    }
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < dh; d += 64) {
      float ak = 0.f, av = 0.f;
      for (int i = 0; i < S; ++i) {
        ak += ds[i] * Qs[i * pitch + d];
        av += p[i] * Ds[i * pitch + d];
      }
      dqkv[((int64_t)row0 + j) * ld + (int64_t)H * dh + h * dh + d] = from_f<T>(ak);
      dqkv[((int64_t)row0 + j) * ld + 2LL * H * dh + h * dh + d] = from_f<T>(av);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

static size_t atl_bytes(int S, int dh, int per_wave_floats, int extra_floats) {
  return ((size_t)2 * S * (dh + 1) + extra_floats + (size_t)ATL_WAVES * per_wave_floats) * sizeof(float);
}
static const bool g_att_lds = !(getenv("MMRCA_ATTN_REF_LDS") && atoi(getenv("MMRCA_ATTN_REF_LDS")) == 0);

int mmrca_mha_fwd_ref(const void* qkv, const int32_t* key_mask, void* out, float* lse, int B, int H, int S, int dh,
                      float scale, float drop_p, uint64_t drop_seed, const int32_t* cu, int dtype, hipStream_t st) {
  MMRCA_REQUIRE(S <= ATT_MAX_S && dh <= ATT_MAX_DH && dh % 4 == 0, "mha_fwd(ref): S=%d dh=%d unsupported", S, dh);
  const size_t ldsl = atl_bytes(S, dh, ATT_MAX_DH + S, 0);
  if (g_att_lds && ldsl <= 160 * 1024) {
    MMRCA_DISPATCH_DTYPE(dtype, "mha_fwd",
      MMRCA_MAX_LDS((int)ldsl, mha_fwd_lds_k<T>);
      hipLaunchKernelGGL(mha_fwd_lds_k<T>, dim3(B * H), dim3(64 * ATL_WAVES), ldsl, st, (const T*)qkv, key_mask, (T*)out, lse, B, H, S, dh, scale, drop_p, drop_seed, cu);)
    MMRCA_CHECK_LAUNCH("mha_fwd(lds)");
    return 0;
  }
  dim3 grid((S + 3) / 4, B * H);
  const size_t lds = 4 * (ATT_MAX_DH + S) * sizeof(float);
  MMRCA_DISPATCH_DTYPE(dtype, "mha_fwd",
    hipLaunchKernelGGL(mha_fwd_ref_k<T>, grid, dim3(256), lds, st, (const T*)qkv, key_mask, (T*)out, lse, B, H, S, dh, scale, drop_p, drop_seed, cu);)
  MMRCA_CHECK_LAUNCH("mha_fwd(ref)");
  return 0;
}

int mmrca_mha_bwd_ref(const void* qkv, const int32_t* key_mask, const void* out, const void* dout, const float* lse,
                      void* dqkv, int B, int H, int S, int dh, float scale, float drop_p, uint64_t drop_seed, const int32_t* cu,
                      int dtype, hipStream_t st) {
  MMRCA_REQUIRE(S <= ATT_MAX_S && dh <= ATT_MAX_DH && dh % 4 == 0, "mha_bwd(ref): S=%d dh=%d unsupported", S, dh);
  const size_t ll1 = atl_bytes(S, dh, 2 * ATT_MAX_DH + S, 0), ll2 = atl_bytes(S, dh, 2 * ATT_MAX_DH + 2 * S, S);
  if (g_att_lds && ll1 <= 160 * 1024 && ll2 <= 160 * 1024) {
    MMRCA_DISPATCH_DTYPE(dtype, "mha_bwd",
      MMRCA_MAX_LDS((int)ll1, mha_bwd_dq_lds_k<T>);
      MMRCA_MAX_LDS((int)ll2, mha_bwd_dkv_lds_k<T>);
      hipLaunchKernelGGL(mha_bwd_dq_lds_k<T>, dim3(B * H), dim3(64 * ATL_WAVES), ll1, st, (const T*)qkv, key_mask, (const T*)out, (const T*)dout, lse, (T*)dqkv, B, H, S, dh, scale, drop_p, drop_seed, cu);
      hipLaunchKernelGGL(mha_bwd_dkv_lds_k<T>, dim3(B * H), dim3(64 * ATL_WAVES), ll2, st, (const T*)qkv, key_mask, (const T*)out, (const T*)dout, lse, (T*)dqkv, B, H, S, dh, scale, drop_p, drop_seed, cu);)
    MMRCA_CHECK_LAUNCH("mha_bwd(lds)");
    return 0;
  }
  dim3 grid((S + 3) / 4, B * H);
  const size_t lds1 = 4 * (2 * ATT_MAX_DH + S) * sizeof(float), lds2 = 4 * (2 * ATT_MAX_DH + 2 * S) * sizeof(float);
  MMRCA_DISPATCH_DTYPE(dtype, "mha_bwd",
    hipLaunchKernelGGL(mha_bwd_dq_ref_k<T>, grid, dim3(256), lds1, st, (const T*)qkv, key_mask, (const T*)out, (const T*)dout, lse, (T*)dqkv, B, H, S, dh, scale, drop_p, drop_seed, cu);
    hipLaunchKernelGGL(mha_bwd_dkv_ref_k<T>, grid, dim3(256), lds2, st, (const T*)qkv, key_mask, (const T*)out, (const T*)dout, lse, (T*)dqkv, B, H, S, dh, scale, drop_p, drop_seed, cu);)
  MMRCA_CHECK_LAUNCH("mha_bwd(ref)");
  return 0;
}

MMRCA_SEED_EPOCH_EXPORT(attention_ref)   // this translation unit's copy of the mask epoch (common.h)
