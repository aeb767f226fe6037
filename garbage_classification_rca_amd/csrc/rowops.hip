// Row-wise and elementwise kernels of the MM-RCA path: LayerNorm (+residual) fwd/bwd, GELU backward, bias
// (column) gradient, text embeddings, ViT patchify/assemble, loss, optimizers.  All are HBM-bound: one wave
// per row, 8- or 16-byte accesses per lane, fp32 statistics, wave-shuffle reductions.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

#define LN_MAXV 8   // row kept in registers: D <= 256 * LN_MAXV

// --------------------------------------------------------------------------------------------------------
// s = x (+res); y = LN(s)*gamma+beta
// --------------------------------------------------------------------------------------------------------
template <typename T, int NV>
__global__ void __launch_bounds__(256)
add_ln_fwd_k(const T* __restrict__ x, const T* __restrict__ res, const T* __restrict__ gamma, const T* __restrict__ beta,
             T* __restrict__ sum_out, T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
             int64_t rows, int D, int64_t ld_x, int64_t ld_y, float eps, float in_p, uint64_t in_seed, float out_p,
             uint64_t out_seed, bf16_t* __restrict__ y_hi = nullptr, bf16_t* __restrict__ y_lo = nullptr) {
  // y_hi / y_lo (bf16x3 mode): the normalised output also (or, with y == nullptr, only) as two bf16 planes for the GEMMs that read it
  const float in_sc = in_p > 0.f ? 1.f / (1.f - in_p) : 1.f, out_sc = out_p > 0.f ? 1.f / (1.f - out_p) : 1.f;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    float v[NV][4];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int c = it * 256 + lane * 4;
      if (c < D) {
        Vec4<T> a = Vec4<T>::load(x + row * ld_x + c);
        if (in_p > 0.f) {      // dropout on the branch before the residual add (FFN / attention output dropout)
#pragma unroll
          for (int j = 0; j < 4; ++j) a.v[j] = mmrca_uniform(in_seed, (uint64_t)row * D + c + j) >= in_p ? a.v[j] * in_sc : 0.f;
        }
        if (res) {
          Vec4<T> r = Vec4<T>::load(res + row * ld_x + c);
#pragma unroll
          for (int j = 0; j < 4; ++j) a.v[j] += r.v[j];
        }
        if (sum_out) {
          // the stored sum is what the backward re-reads: round once, then normalise the rounded values
#pragma unroll
          for (int j = 0; j < 4; ++j) a.v[j] = to_f(from_f<T>(a.v[j]));
          a.store(sum_out + row * ld_x + c);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[it][j] = a.v[j]; s += a.v[j]; }
      }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int c = it * 256 + lane * 4;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[it][j] - mu; q += d * d; }
      }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int c = it * 256 + lane * 4;
      if (c < D) {
        Vec4<T> g = Vec4<T>::load(gamma + c), b = Vec4<T>::load(beta + c), o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o.v[j] = (v[it][j] - mu) * rs * g.v[j] + b.v[j];
        if (out_p > 0.f) {     // dropout on the normalised output (embedding dropout)
#pragma unroll
          for (int j = 0; j < 4; ++j) o.v[j] = mmrca_uniform(out_seed, (uint64_t)row * D + c + j) >= out_p ? o.v[j] * out_sc : 0.f;
        }
        if (y) o.store(y + row * ld_y + c);
        if (y_hi) {
          Vec4<bf16_t> hi, lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) { hi.v[j] = (float)(bf16_t)o.v[j]; lo.v[j] = o.v[j] - hi.v[j]; }
          hi.store(y_hi + row * ld_y + c);
          lo.store(y_lo + row * ld_y + c);
        }
      }
    }
  }
}

// bf16 fast path of the kernel above: a lane owns 8 contiguous columns (16-byte accesses, 512-column slabs) and the raw
// vectors of the NEXT row are fetched before the current row's statistics are reduced, so every wave keeps two rows of
// loads in flight (the plain kernel above reached 3.0 TB/s at 50432 x 768; HBM-bound work wants ~2x the bytes in flight)
typedef __attribute__((ext_vector_type(8))) __bf16 raw8;

// PACK (round 5): the row is kept as the bf16 values it is normalised from (4 registers per 8 columns) instead of as fp32 (8): with
// the fp32 copy the <2,2> form needed 193 registers = TWO waves per SIMD, 48 KB of loads in flight per CU, and ran at 3.4 TB/s
// (Little's law at ~3.5 us of loaded latency); packed it fits four waves per SIMD.  Exact whenever the normalised values ARE bf16
// numbers: a stored sum (the backward re-reads the rounded sum, so the statistics are taken from it anyway) or a plain x.
// gamma | beta sit in LDS (16 registers less than a per-lane copy); DROP = either dropout site active (its 64-bit counters cost
// registers too, and the ViT has none).
template <int NV2, int RP, bool PACK, bool DROP>
__global__ void __launch_bounds__(256, PACK ? 4 : 1)
add_ln_fwd_bf16_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res, const bf16_t* __restrict__ gamma,
                  const bf16_t* __restrict__ beta, bf16_t* __restrict__ sum_out, bf16_t* __restrict__ y,
                  float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int D, int64_t ld_x, int64_t ld_y,
                  float eps, float in_p, uint64_t in_seed, float out_p, uint64_t out_seed) {
  // RP rows per wave per iteration (rows row, row + stride, ...): their loads, reductions and stores interleave
  const float in_sc = in_p > 0.f ? 1.f / (1.f - in_p) : 1.f, out_sc = out_p > 0.f ? 1.f / (1.f - out_p) : 1.f;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + wave;
  raw8 nx[RP][NV2], nr[RP][NV2];
  __shared__ raw8 gb_s[2][NV2 * 64];
  for (int i = threadIdx.x; i < NV2 * 64; i += 256)
    if (i * 8 < D) { gb_s[0][i] = *reinterpret_cast<const raw8*>(gamma + i * 8); gb_s[1][i] = *reinterpret_cast<const raw8*>(beta + i * 8); }
  __syncthreads();
#define LNF_FETCH(r0_)                                                                           \
  _Pragma("unroll") for (int p = 0; p < RP; ++p) {                                               \
    const int64_t r_ = (r0_) + p * stride;                                                       \
    if (r_ < rows) {                                                                             \
      _Pragma("unroll") for (int it = 0; it < NV2; ++it) {                                       \
        const int c = it * 512 + lane * 8;                                                       \
        if (c < D) {                                                                             \
          nx[p][it] = *reinterpret_cast<const raw8*>(x + r_ * ld_x + c);                         \
          if (res) nr[p][it] = *reinterpret_cast<const raw8*>(res + r_ * ld_x + c);              \
        }                                                                                        \
      }                                                                                          \
    }                                                                                            \
  }
  LNF_FETCH(row)
  for (; row < rows; row += RP * stride) {
    float v[PACK ? 1 : RP][PACK ? 1 : NV2][8];
    raw8 sv[PACK ? RP : 1][PACK ? NV2 : 1];
    float s[RP], q[RP], mu[RP], rs[RP];
#define LNF_VAL(p_, it_, j_) (PACK ? (float)sv[PACK ? (p_) : 0][PACK ? (it_) : 0][j_] : v[PACK ? 0 : (p_)][PACK ? 0 : (it_)][j_])
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int64_t r = row + p * stride;
      s[p] = 0.f;
      if (r < rows) {
#pragma unroll
        for (int it = 0; it < NV2; ++it) {
          const int c = it * 512 + lane * 8;
          if (c < D) {
            raw8 so;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float a = (float)nx[p][it][j];
              if (DROP && in_p > 0.f) a = mmrca_uniform(in_seed, (uint64_t)r * D + c + j) >= in_p ? a * in_sc : 0.f;
              if (res) a += (float)nr[p][it][j];
              if (sum_out || PACK) { so[j] = (bf16_t)a; a = (float)so[j]; }   // the stored sum is what the backward re-reads
              if constexpr (!PACK) v[p][it][j] = a;
              s[p] += a;
            }
            if constexpr (PACK) sv[p][it] = so;
            if (sum_out) *reinterpret_cast<raw8*>(sum_out + r * ld_x + c) = so;
          }
        }
      }
    }
    LNF_FETCH(row + RP * stride)
#pragma unroll
    for (int p = 0; p < RP; ++p) mu[p] = wave_sum(s[p]) / (float)D;
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      q[p] = 0.f;
      if (row + p * stride < rows) {
#pragma unroll
        for (int it = 0; it < NV2; ++it) {
          const int c = it * 512 + lane * 8;
          if (c < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = LNF_VAL(p, it, j) - mu[p]; q[p] += d * d; }
          }
        }
      }
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) rs[p] = rsqrtf(wave_sum(q[p]) / (float)D + eps);
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int64_t r = row + p * stride;
      if (r < rows) {
        if (lane == 0) { if (mean) mean[r] = mu[p]; if (rstd) rstd[r] = rs[p]; }
#pragma unroll
        for (int it = 0; it < NV2; ++it) {
          const int c = it * 512 + lane * 8;
          if (c < D) {
            raw8 o;
            const raw8 gm = gb_s[0][it * 64 + lane], bt = gb_s[1][it * 64 + lane];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float t = (LNF_VAL(p, it, j) - mu[p]) * rs[p] * (float)gm[j] + (float)bt[j];
              if (DROP && out_p > 0.f) t = mmrca_uniform(out_seed, (uint64_t)r * D + c + j) >= out_p ? t * out_sc : 0.f;
              o[j] = (bf16_t)t;
            }
            *reinterpret_cast<raw8*>(y + r * ld_y + c) = o;
          }
        }
      }
    }
  }
#undef LNF_FETCH
#undef LNF_VAL
}

static inline bool aligned16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int add_layernorm_fwd_impl(const void* x, const void* res, const void* gamma, const void* beta,
                                  void* sum_out, void* y, float* mean, float* rstd, int64_t rows, int D,
                                  int64_t ld_x, int64_t ld_y, float eps, float in_drop_p, uint64_t in_drop_seed,
                                  float out_drop_p, uint64_t out_drop_seed, int dtype, void* stream, void* y_hi, void* y_lo) {
  MMRCA_REQUIRE(in_drop_p >= 0.f && in_drop_p < 1.f && out_drop_p >= 0.f && out_drop_p < 1.f, "add_layernorm_fwd: dropout p must be in [0,1)");
  MMRCA_REQUIRE(x && gamma && beta && (y || y_hi), "add_layernorm_fwd: null pointer");
  MMRCA_REQUIRE((y_hi == nullptr) == (y_lo == nullptr) && (!y_hi || dtype == MMRCA_F32), "add_layernorm_fwd: a two-plane output needs both planes and fp32 inputs");
  MMRCA_REQUIRE(D > 0 && D % 4 == 0 && D <= 256 * LN_MAXV, "add_layernorm_fwd: D=%d unsupported (multiple of 4, <= %d)", D, 256 * LN_MAXV);
  MMRCA_REQUIRE(ld_x >= D && ld_y >= D && ld_x % 4 == 0 && ld_y % 4 == 0, "add_layernorm_fwd: bad leading dims");
  if (rows <= 0) return 0;
  if (dtype == MMRCA_BF16 && D % 8 == 0 && D <= 1536 && ld_x % 8 == 0 && ld_y % 8 == 0 && aligned16p(x) && aligned16p(y) &&
      aligned16p(gamma) && aligned16p(beta) && (!res || aligned16p(res)) && (!sum_out || aligned16p(sum_out))) {
    const int g2 = (int)((rows + 7) / 8 < 1024 ? (rows + 7) / 8 : 1024);
#define LN_FWD16(NV2_, RP_, PACK_, DROP_)                                                                                    \
    hipLaunchKernelGGL((add_ln_fwd_bf16_k<NV2_, RP_, PACK_, DROP_>), dim3(g2), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, \
                       (const bf16_t*)res, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)sum_out, (bf16_t*)y, mean, rstd, \
                       rows, D, ld_x, ld_y, eps, in_drop_p, in_drop_seed, out_drop_p, out_drop_seed)
#define LN_FWD16_PD(NV2_, RP_)                                                                                               \
    do {                                                                                                                     \
      if (pack) { if (drop) LN_FWD16(NV2_, RP_, true, true); else LN_FWD16(NV2_, RP_, true, false); }                        \
      else { if (drop) LN_FWD16(NV2_, RP_, false, true); else LN_FWD16(NV2_, RP_, false, false); }                           \
    } while (0)
    const int nv2 = (D + 511) / 512;
    // packed rows whenever the normalised values are bf16 numbers anyway (see the kernel); MMRCA_LN_PACK=0: the fp32-row form
    static const bool pack_on = !(getenv("MMRCA_LN_PACK") && atoi(getenv("MMRCA_LN_PACK")) == 0);
    const bool pack = pack_on && (sum_out || (!res && in_drop_p == 0.f));
    const bool drop = in_drop_p > 0.f || out_drop_p > 0.f;
    if (nv2 <= 1) LN_FWD16_PD(1, 2); else if (nv2 == 2) LN_FWD16_PD(2, 2); else LN_FWD16_PD(3, 1);      // 3: BLIP-2's ViT-g rows (D = 1408)
#undef LN_FWD16_PD
#undef LN_FWD16
    MMRCA_CHECK_LAUNCH("add_layernorm_fwd(bf16)");
    return 0;
  }
  const int grid = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
#define LN_FWD_LAUNCH(NV_)                                                                                             \
  hipLaunchKernelGGL((add_ln_fwd_k<T, NV_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)res,  \
                     (const T*)gamma, (const T*)beta, (T*)sum_out, (T*)y, mean, rstd, rows, D, ld_x, ld_y, eps, in_drop_p,  \
                     in_drop_seed, out_drop_p, out_drop_seed, (bf16_t*)y_hi, (bf16_t*)y_lo)
  const int nv = (D + 255) / 256;        // register slabs of 256 columns actually needed (3 for D=768)
  MMRCA_DISPATCH_DTYPE(dtype, "add_layernorm_fwd",
    if (nv <= 1) LN_FWD_LAUNCH(1); else if (nv <= 2) LN_FWD_LAUNCH(2); else if (nv <= 3) LN_FWD_LAUNCH(3);
    else if (nv <= 4) LN_FWD_LAUNCH(4); else LN_FWD_LAUNCH(8);)
#undef LN_FWD_LAUNCH
  MMRCA_CHECK_LAUNCH("add_layernorm_fwd");
  return 0;
}

extern "C" int mmrca_add_layernorm_fwd(const void* x, const void* res, const void* gamma, const void* beta,
                                       void* sum_out, void* y, float* mean, float* rstd, int64_t rows, int D,
                                       int64_t ld_x, int64_t ld_y, float eps, float in_drop_p, uint64_t in_drop_seed,
                                       float out_drop_p, uint64_t out_drop_seed, int dtype, void* stream) {
  MMRCA_REQUIRE(y, "add_layernorm_fwd: null pointer");
  return add_layernorm_fwd_impl(x, res, gamma, beta, sum_out, y, mean, rstd, rows, D, ld_x, ld_y, eps, in_drop_p, in_drop_seed,
                                out_drop_p, out_drop_seed, dtype, stream, nullptr, nullptr);
}

// bf16x3 mode (fp32 operands): the normalised output as two bf16 planes y_hi + y_lo (what the GEMMs that consume it read; see
// mmrca_gemm_x3), and as fp32 in y as well when y != NULL (post-LN encoders: the output is also the next residual)
extern "C" int mmrca_add_layernorm_fwd_x3(const void* x, const void* res, const void* gamma, const void* beta,
                                          void* sum_out, void* y, void* y_hi, void* y_lo, float* mean, float* rstd, int64_t rows,
                                          int D, int64_t ld_x, int64_t ld_y, float eps, float in_drop_p, uint64_t in_drop_seed,
                                          float out_drop_p, uint64_t out_drop_seed, void* stream) {
  MMRCA_REQUIRE(y_hi && y_lo, "add_layernorm_fwd_x3: null plane");
  return add_layernorm_fwd_impl(x, res, gamma, beta, sum_out, y, mean, rstd, rows, D, ld_x, ld_y, eps, in_drop_p, in_drop_seed,
                                out_drop_p, out_drop_seed, MMRCA_F32, stream, y_hi, y_lo);
}

// --------------------------------------------------------------------------------------------------------
// LayerNorm backward: ds = rstd*(g - mean(g) - xhat*mean(g*xhat)) (+dres), g = dy*gamma;
// dgamma += sum_rows dy*xhat, dbeta += sum_rows dy (per-lane register partials -> LDS -> one atomic per column/block)
// --------------------------------------------------------------------------------------------------------
template <typename T, int NV>
__global__ void __launch_bounds__(256)
ln_bwd_k(const T* __restrict__ dy, const T* __restrict__ s, const T* __restrict__ gamma, const float* __restrict__ mean,
         const float* __restrict__ rstd, const T* __restrict__ dres, T* __restrict__ ds, float* __restrict__ dgamma,
         float* __restrict__ dbeta, int64_t rows, int D, int64_t ld_dy, int64_t ld_s, int64_t ld_ds, float dy_p,
         uint64_t dy_seed, float br_p, uint64_t br_seed, T* __restrict__ dbranch, float* __restrict__ dcol,
         float* __restrict__ dcol_branch) {
  const float dy_sc = dy_p > 0.f ? 1.f / (1.f - dy_p) : 1.f, br_sc = br_p > 0.f ? 1.f / (1.f - br_p) : 1.f;
  __shared__ float red[4][4][256];   // [dgamma|dbeta|colsum(ds)|colsum(dbranch)][wave][column of the slab]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float adg[NV][4], adb[NV][4], adc[NV][4], adcb[NV][4];
#pragma unroll
  for (int it = 0; it < NV; ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j) { adg[it][j] = 0.f; adb[it][j] = 0.f; adc[it][j] = 0.f; adcb[it][j] = 0.f; }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float g[NV][4], xh[NV][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int c = it * 256 + lane * 4;
      if (c < D) {
        Vec4<T> d = Vec4<T>::load(dy + row * ld_dy + c), sv = Vec4<T>::load(s + row * ld_s + c), gm = Vec4<T>::load(gamma + c);
        if (dy_p > 0.f) {      // the forward dropped the normalised output: mask the incoming gradient the same way
#pragma unroll
          for (int j = 0; j < 4; ++j) d.v[j] = mmrca_uniform(dy_seed, (uint64_t)row * D + c + j) >= dy_p ? d.v[j] * dy_sc : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xhat = (sv.v[j] - mu) * rs;
          const float gg = d.v[j] * gm.v[j];
          xh[it][j] = xhat; g[it][j] = gg;
          s1 += gg; s2 += gg * xhat;
          adg[it][j] += d.v[j] * xhat; adb[it][j] += d.v[j];
        }
      }
    }
    s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int c = it * 256 + lane * 4;
      if (c < D) {
        Vec4<T> o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o.v[j] = rs * (g[it][j] - s1 - xh[it][j] * s2);
        if (dbranch) {         // gradient of the dropped branch (before the residual-stream gradient is added)
          Vec4<T> ob;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            ob.v[j] = (br_p > 0.f && mmrca_uniform(br_seed, (uint64_t)row * D + c + j) < br_p) ? 0.f : o.v[j] * br_sc;
          ob.store(dbranch + row * ld_ds + c);
          if (dcol_branch) {
#pragma unroll
            for (int j = 0; j < 4; ++j) adcb[it][j] += ob.v[j];
          }
        }
        if (dres) {
          Vec4<T> r = Vec4<T>::load(dres + row * ld_ds + c);
#pragma unroll
          for (int j = 0; j < 4; ++j) o.v[j] += r.v[j];
        }
        if (dcol) {            // column sums of the OUTPUT = bias gradient of the linear layer that produced this stream
#pragma unroll
          for (int j = 0; j < 4; ++j) adc[it][j] += o.v[j];
        }
        o.store(ds + row * ld_ds + c);
      }
    }
  }
  // cross-wave reduction of the parameter gradients, one 256-column slab at a time
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    if (it * 256 >= D) break;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[0][wave][lane * 4 + j] = adg[it][j]; red[1][wave][lane * 4 + j] = adb[it][j];
      red[2][wave][lane * 4 + j] = adc[it][j]; red[3][wave][lane * 4 + j] = adcb[it][j];
    }
    __syncthreads();
    const int c = it * 256 + threadIdx.x;
    if (c < D) {
      float* dst[4] = {dgamma, dbeta, dcol, dcol_branch};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (dst[q]) atomicAdd(dst[q] + c, red[q][0][threadIdx.x] + red[q][1][threadIdx.x] + red[q][2][threadIdx.x] + red[q][3][threadIdx.x]);
    }
  }
}

// bf16 fast path of ln_bwd_k: 16-byte lanes over 512-column slabs, gamma resident in registers, the raw vectors (and
// row statistics) of the next row fetched before the current row's two wave reductions; accumulators that a call site
// does not ask for are compiled out (COLS = dcol, BRANCH = dbranch / dcol_branch).
// SF32 (mmrca_layernorm_bwd_mixed): the saved pre-normalisation sum `s` is fp32 -- the residual stream of a forward that keeps
// it in fp32 (the bf16x3f mode) -- while every gradient stays bf16.
// LNB_WAVES waves per block: the column sums (dgamma, dbeta, bias gradients) leave a block as one fp32 atomic per column and array --
// 2.4 M atomics on 2,304 addresses per launch with 1,024 blocks of four waves, which a sweep of the grid size showed to cost more than
// the rows in flight they bought (86 us at 1,024 blocks, 78 at 512, 98 at 256 where latency takes over); eight waves per block at one block per CU keep half the
// rows in flight of the 1,024-block launch with a quarter of the atomics: 69 us (tools/rowops_bench.py, operands rotated past the MALL).
#define LNB_WAVES 8
template <int NV2, bool COLS, bool BRANCH, bool SF32 = false>
__global__ void __launch_bounds__(64 * LNB_WAVES)
ln_bwd_bf16_k(const bf16_t* __restrict__ dy, const void* __restrict__ s_, const bf16_t* __restrict__ gamma,
              const float* __restrict__ mean, const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
              bf16_t* __restrict__ ds, float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int D,
              int64_t ld_dy, int64_t ld_s, int64_t ld_ds, float dy_p, uint64_t dy_seed, float br_p, uint64_t br_seed,
              bf16_t* __restrict__ dbranch, float* __restrict__ dcol, float* __restrict__ dcol_branch) {
  const float dy_sc = dy_p > 0.f ? 1.f / (1.f - dy_p) : 1.f, br_sc = br_p > 0.f ? 1.f / (1.f - br_p) : 1.f;
  __shared__ float red[LNB_WAVES][512];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NC = COLS ? NV2 : 1, NB = BRANCH ? NV2 : 1;
  float adg[NV2][8], adb[NV2][8], adc[NC][8], adcb[NB][8];
#pragma unroll
  for (int it = 0; it < NV2; ++it)
#pragma unroll
    for (int j = 0; j < 8; ++j) { adg[it][j] = 0.f; adb[it][j] = 0.f; }
#pragma unroll
  for (int it = 0; it < NC; ++it)
#pragma unroll
    for (int j = 0; j < 8; ++j) adc[it][j] = 0.f;
#pragma unroll
  for (int it = 0; it < NB; ++it)
#pragma unroll
    for (int j = 0; j < 8; ++j) adcb[it][j] = 0.f;
  typedef typename std::conditional<SF32, float, bf16_t>::type ST;
  typedef ST sraw8 __attribute__((ext_vector_type(8)));
  const ST* __restrict__ s = reinterpret_cast<const ST*>(s_);
  raw8 gm[NV2], ndy[NV2], nres[NV2];
  sraw8 ns[NV2];
  float nmu = 0.f, nrs = 0.f;
#pragma unroll
  for (int it = 0; it < NV2; ++it) {
    const int c = it * 512 + lane * 8;
    if (c < D) gm[it] = *reinterpret_cast<const raw8*>(gamma + c);
  }
#define LNB_FETCH(r_)                                                                            \
  _Pragma("unroll") for (int it = 0; it < NV2; ++it) {                                           \
    const int c = it * 512 + lane * 8;                                                           \
    if (c < D) {                                                                                 \
      ndy[it] = *reinterpret_cast<const raw8*>(dy + (r_) * ld_dy + c);                           \
      ns[it] = *reinterpret_cast<const sraw8*>(s + (r_) * ld_s + c);                             \
      if (dres) nres[it] = *reinterpret_cast<const raw8*>(dres + (r_) * ld_ds + c);              \
    }                                                                                            \
  }                                                                                              \
  nmu = mean[r_]; nrs = rstd[r_];
  const int64_t stride = (int64_t)gridDim.x * LNB_WAVES;
  int64_t row = (int64_t)blockIdx.x * LNB_WAVES + wave;
  if (row < rows) { LNB_FETCH(row) }
  for (; row < rows; row += stride) {
    const float mu = nmu, rs = nrs;
    float g[NV2][8], xh[NV2][8];
    raw8 cres[NV2];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NV2; ++it) {
      const int c = it * 512 + lane * 8;
      if (c < D) {
        cres[it] = nres[it];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float d = (float)ndy[it][j];
          if (dy_p > 0.f) d = mmrca_uniform(dy_seed, (uint64_t)row * D + c + j) >= dy_p ? d * dy_sc : 0.f;
          const float xhat = ((float)ns[it][j] - mu) * rs;
          const float gg = d * (float)gm[it][j];
          xh[it][j] = xhat; g[it][j] = gg;
          s1 += gg; s2 += gg * xhat;
          adg[it][j] += d * xhat; adb[it][j] += d;
        }
      }
    }
    const int64_t nxt = row + stride;
    if (nxt < rows) { LNB_FETCH(nxt) }
    s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int it = 0; it < NV2; ++it) {
      const int c = it * 512 + lane * 8;
      if (c < D) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs * (g[it][j] - s1 - xh[it][j] * s2);
        if (BRANCH && dbranch) {
          raw8 ob;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float t = (br_p > 0.f && mmrca_uniform(br_seed, (uint64_t)row * D + c + j) < br_p) ? 0.f : o[j] * br_sc;
            ob[j] = (bf16_t)t;
            if (dcol_branch) adcb[BRANCH ? it : 0][j] += t;
          }
          *reinterpret_cast<raw8*>(dbranch + row * ld_ds + c) = ob;
        }
        raw8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (dres) o[j] += (float)cres[it][j];
          if (COLS) adc[COLS ? it : 0][j] += o[j];
          ov[j] = (bf16_t)o[j];
        }
        *reinterpret_cast<raw8*>(ds + row * ld_ds + c) = ov;
      }
    }
  }
#undef LNB_FETCH
  // cross-wave reduction of the column sums, one 512-column slab and one output array at a time
#pragma unroll
  for (int it = 0; it < NV2; ++it) {
    if (it * 512 >= D) break;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float* dst = q == 0 ? dgamma : q == 1 ? dbeta : q == 2 ? (COLS ? dcol : nullptr) : (BRANCH ? dcol_branch : nullptr);
      if (!dst) continue;                       // uniform across the block
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j)
        red[wave][lane * 8 + j] = q == 0 ? adg[it][j] : q == 1 ? adb[it][j] : q == 2 ? adc[COLS ? it : 0][j] : adcb[BRANCH ? it : 0][j];
      __syncthreads();
      for (int cc = threadIdx.x; cc < 512; cc += 64 * LNB_WAVES) {
        const int c = it * 512 + cc;
        if (c < D) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < LNB_WAVES; ++w) t += red[w][cc];
          atomicAdd(dst + c, t);
        }
      }
    }
  }
}

extern "C" int mmrca_layernorm_bwd(const void* dy, const void* s, const void* gamma, const float* mean, const float* rstd,
                                   const void* dres, void* ds, float* dgamma, float* dbeta, int64_t rows, int D,
                                   int64_t ld_dy, int64_t ld_s, int64_t ld_ds, float dy_drop_p, uint64_t dy_drop_seed,
                                   float branch_drop_p, uint64_t branch_drop_seed, void* dbranch, float* dcol, float* dcol_branch,
                                   int dtype, void* stream) {
  MMRCA_REQUIRE(dy_drop_p >= 0.f && dy_drop_p < 1.f && branch_drop_p >= 0.f && branch_drop_p < 1.f, "layernorm_bwd: dropout p must be in [0,1)");
  MMRCA_REQUIRE(dy && s && gamma && mean && rstd && ds, "layernorm_bwd: null pointer");
  MMRCA_REQUIRE(D > 0 && D % 4 == 0 && D <= 256 * LN_MAXV, "layernorm_bwd: D=%d unsupported", D);
  MMRCA_REQUIRE(ld_dy >= D && ld_s >= D && ld_ds >= D && ld_dy % 4 == 0 && ld_s % 4 == 0 && ld_ds % 4 == 0, "layernorm_bwd: bad leading dims");
  if (rows <= 0) return 0;
  int64_t want = (rows + 3) / 4;
  static const int ln_bwd_grid = getenv("MMRCA_LN_BWD_GRID") ? atoi(getenv("MMRCA_LN_BWD_GRID")) : 256;       // blocks of the bf16 fast path (LNB_WAVES waves each): one per CU
  const int grid = (int)(want < 1024 ? want : 1024);
  const int64_t want16 = (rows + LNB_WAVES - 1) / LNB_WAVES;
  const int grid16 = (int)(want16 < ln_bwd_grid ? want16 : ln_bwd_grid);
  MMRCA_REQUIRE(!(dcol_branch && !dbranch), "layernorm_bwd: dcol_branch needs dbranch");
  if (dtype == MMRCA_BF16 && D % 8 == 0 && D <= 1024 && ld_dy % 8 == 0 && ld_s % 8 == 0 && ld_ds % 8 == 0 && dgamma && dbeta &&
      aligned16p(dy) && aligned16p(s) && aligned16p(gamma) && aligned16p(ds) && (!dres || aligned16p(dres)) &&
      (!dbranch || aligned16p(dbranch))) {
#define LN_BWD16(NV2_, C_, B_)                                                                                              \
    hipLaunchKernelGGL((ln_bwd_bf16_k<NV2_, C_, B_>), dim3(grid16), dim3(64 * LNB_WAVES), 0, (hipStream_t)stream, (const bf16_t*)dy,      \
                       (const void*)s, (const bf16_t*)gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)ds, dgamma, dbeta,    \
                       rows, D, ld_dy, ld_s, ld_ds, dy_drop_p, dy_drop_seed, branch_drop_p, branch_drop_seed,              \
                       (bf16_t*)dbranch, dcol, dcol_branch)
#define LN_BWD16_NV(NV2_)                                                                                                   \
    do { if (dbranch) { if (dcol) LN_BWD16(NV2_, true, true); else LN_BWD16(NV2_, false, true); }                           \
         else { if (dcol) LN_BWD16(NV2_, true, false); else LN_BWD16(NV2_, false, false); } } while (0)
    const int nv2 = (D + 511) / 512;
    if (nv2 <= 1) LN_BWD16_NV(1); else LN_BWD16_NV(2);
#undef LN_BWD16_NV
#undef LN_BWD16
    MMRCA_CHECK_LAUNCH("layernorm_bwd(bf16)");
    return 0;
  }
#define LN_BWD_LAUNCH(NV_)                                                                                                  \
  hipLaunchKernelGGL((ln_bwd_k<T, NV_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)s,           \
                     (const T*)gamma, mean, rstd, (const T*)dres, (T*)ds, dgamma, dbeta, rows, D, ld_dy, ld_s, ld_ds, dy_drop_p, \
                     dy_drop_seed, branch_drop_p, branch_drop_seed, (T*)dbranch, dcol, dcol_branch)
  const int nv = (D + 255) / 256;
  MMRCA_DISPATCH_DTYPE(dtype, "layernorm_bwd",
    if (nv <= 1) LN_BWD_LAUNCH(1); else if (nv <= 2) LN_BWD_LAUNCH(2); else if (nv <= 3) LN_BWD_LAUNCH(3);
    else if (nv <= 4) LN_BWD_LAUNCH(4); else LN_BWD_LAUNCH(8);)
#undef LN_BWD_LAUNCH
  MMRCA_CHECK_LAUNCH("layernorm_bwd");
  return 0;
}

// LayerNorm backward of the bf16x3f mode: bf16 gradients (dy, dres, ds, dbranch) and bf16 gamma against the fp32 sum `s` that the
// fp32-stream forward saved.  Same argument meaning as mmrca_layernorm_bwd; D % 8 == 0, D <= 1024, 16-byte aligned operands.
extern "C" int mmrca_layernorm_bwd_mixed(const void* dy, const float* s, const void* gamma, const float* mean, const float* rstd,
                                         const void* dres, void* ds, float* dgamma, float* dbeta, int64_t rows, int D,
                                         int64_t ld_dy, int64_t ld_s, int64_t ld_ds, float dy_drop_p, uint64_t dy_drop_seed,
                                         float branch_drop_p, uint64_t branch_drop_seed, void* dbranch, float* dcol, float* dcol_branch,
                                         void* stream) {
  MMRCA_REQUIRE(dy_drop_p >= 0.f && dy_drop_p < 1.f && branch_drop_p >= 0.f && branch_drop_p < 1.f, "layernorm_bwd_mixed: dropout p must be in [0,1)");
  MMRCA_REQUIRE(dy && s && gamma && mean && rstd && ds && dgamma && dbeta, "layernorm_bwd_mixed: null pointer");
  MMRCA_REQUIRE(D > 0 && D % 8 == 0 && D <= 1024, "layernorm_bwd_mixed: D=%d unsupported (multiple of 8, <= 1024)", D);
  MMRCA_REQUIRE(ld_dy >= D && ld_s >= D && ld_ds >= D && ld_dy % 8 == 0 && ld_s % 8 == 0 && ld_ds % 8 == 0, "layernorm_bwd_mixed: bad leading dims");
  MMRCA_REQUIRE(aligned16p(dy) && aligned16p(s) && aligned16p(gamma) && aligned16p(ds) && (!dres || aligned16p(dres)) && (!dbranch || aligned16p(dbranch)),
                "layernorm_bwd_mixed: operands must be 16-byte aligned");
  MMRCA_REQUIRE(!(dcol_branch && !dbranch), "layernorm_bwd_mixed: dcol_branch needs dbranch");
  if (rows <= 0) return 0;
  int64_t want = (rows + LNB_WAVES - 1) / LNB_WAVES;
  const int grid = (int)(want < 256 ? want : 256);
#define LN_BWDM(NV2_, C_, B_)                                                                                               \
  hipLaunchKernelGGL((ln_bwd_bf16_k<NV2_, C_, B_, true>), dim3(grid), dim3(64 * LNB_WAVES), 0, (hipStream_t)stream, (const bf16_t*)dy,  \
                     (const void*)s, (const bf16_t*)gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)ds, dgamma, dbeta,      \
                     rows, D, ld_dy, ld_s, ld_ds, dy_drop_p, dy_drop_seed, branch_drop_p, branch_drop_seed,                \
                     (bf16_t*)dbranch, dcol, dcol_branch)
#define LN_BWDM_NV(NV2_)                                                                                                    \
  do { if (dbranch) { if (dcol) LN_BWDM(NV2_, true, true); else LN_BWDM(NV2_, false, true); }                               \
       else { if (dcol) LN_BWDM(NV2_, true, false); else LN_BWDM(NV2_, false, false); } } while (0)
  if (D <= 512) LN_BWDM_NV(1); else LN_BWDM_NV(2);
#undef LN_BWDM_NV
#undef LN_BWDM
  MMRCA_CHECK_LAUNCH("layernorm_bwd_mixed");
  return 0;
}

// --------------------------------------------------------------------------------------------------------
// GELU backward, bias gradient
// --------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) gelu_bwd_k(const T* __restrict__ dg, const T* __restrict__ h, T* __restrict__ dh, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    Vec4<T> a = Vec4<T>::load(dg + i * 4), b = Vec4<T>::load(h + i * 4), o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = a.v[j] * gelu_grad_f(b.v[j]);
    o.store(dh + i * 4);
  }
}

extern "C" int mmrca_gelu_bwd(const void* dG, const void* H, void* dH, int64_t n, int dtype, void* stream) {
  MMRCA_REQUIRE(dG && H && dH, "gelu_bwd: null pointer");
  MMRCA_REQUIRE(n % 4 == 0, "gelu_bwd: n must be a multiple of 4");
  if (n <= 0) return 0;
  const int64_t n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  MMRCA_DISPATCH_DTYPE(dtype, "gelu_bwd",
    hipLaunchKernelGGL(gelu_bwd_k<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)dG, (const T*)H, (T*)dH, n4);)
  MMRCA_CHECK_LAUNCH("gelu_bwd");
  return 0;
}

// dH = dG * gelu'(H) and, in the same pass, db[c] += sum_m dH[m][c] (the FFN1 bias gradient): a thread owns 4 columns and
// walks a slab of rows, so the column sums stay in registers; one atomic per column per block.
template <typename T>
__global__ void __launch_bounds__(256)
gelu_bwd_colsum_k(const T* __restrict__ dg, const T* __restrict__ h, T* __restrict__ dh, float* __restrict__ db, int64_t M,
                  int64_t N, int64_t ld, int rows_per_block) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= N) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int64_t r = r0;
  for (; r + 3 < r1; r += 4) {                     // four rows in flight per thread
    Vec4<T> a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = Vec4<T>::load(dg + (r + u) * ld + c); b[u] = Vec4<T>::load(h + (r + u) * ld + c); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      Vec4<T> o;
#pragma unroll
      for (int j = 0; j < 4; ++j) { o.v[j] = a[u].v[j] * gelu_grad_f(b[u].v[j]); }
      o.store(dh + (r + u) * ld + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += to_f(from_f<T>(o.v[j]));      // sum what was stored (rounded), as colsum_k would
    }
  }
  for (; r < r1; ++r) {
    Vec4<T> a = Vec4<T>::load(dg + r * ld + c), b = Vec4<T>::load(h + r * ld + c), o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = a.v[j] * gelu_grad_f(b.v[j]);
    o.store(dh + r * ld + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += to_f(from_f<T>(o.v[j]));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) atomicAdd(db + c + j, acc[j]);
}

extern "C" int mmrca_gelu_bwd_colsum(const void* dG, const void* H, void* dH, float* db, int64_t M, int64_t N, int64_t ld,
                                     int dtype, void* stream) {
  MMRCA_REQUIRE(dG && H && dH && db, "gelu_bwd_colsum: null pointer");
  MMRCA_REQUIRE(N % 4 == 0 && ld % 4 == 0 && ld >= N, "gelu_bwd_colsum: N and ld must be multiples of 4");
  if (M <= 0 || N <= 0) return 0;
  const int gx = (int)((N / 4 + 255) / 256);
  int rows_per_block = 64;
  int64_t gy = (M + rows_per_block - 1) / rows_per_block;
  while (gy * gx > 8192) { rows_per_block *= 2; gy = (M + rows_per_block - 1) / rows_per_block; }
  MMRCA_DISPATCH_DTYPE(dtype, "gelu_bwd_colsum",
    hipLaunchKernelGGL(gelu_bwd_colsum_k<T>, dim3(gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, (const T*)dG, (const T*)H,
                       (T*)dH, db, M, N, ld, rows_per_block);)
  MMRCA_CHECK_LAUNCH("gelu_bwd_colsum");
  return 0;
}

// db[c] += sum_m dY[m][c].  Block = 4 waves x 64 lanes; a block owns 256 columns (4 per lane) and a slab of rows;
// waves stride the rows, partials meet in LDS, one atomic per column per block.
template <typename T>
__global__ void __launch_bounds__(256) colsum_k(const T* __restrict__ dY, float* __restrict__ db, int64_t M, int64_t N, int64_t ld, int rows_per_block) {
  __shared__ float red[4][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * 256 + lane * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < N) {
    int64_t r = r0 + wave;
    for (; r + 12 < r1; r += 16) {                      // four rows in flight per thread (one load per iteration: 2.1 TB/s)
      Vec4<T> v0 = Vec4<T>::load(dY + r * ld + c), v1 = Vec4<T>::load(dY + (r + 4) * ld + c);
      Vec4<T> v2 = Vec4<T>::load(dY + (r + 8) * ld + c), v3 = Vec4<T>::load(dY + (r + 12) * ld + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] += (v0.v[j] + v1.v[j]) + (v2.v[j] + v3.v[j]);
    }
    for (; r < r1; r += 4) {
      Vec4<T> v = Vec4<T>::load(dY + r * ld + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] += v.v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[wave][lane * 4 + j] = a[j];
  __syncthreads();
  const int64_t cc = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (cc < N) atomicAdd(db + cc, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

extern "C" int mmrca_colsum_accum(const void* dY, float* db, int64_t M, int64_t N, int64_t ld, int dtype, void* stream) {
  MMRCA_REQUIRE(dY && db, "colsum_accum: null pointer");
  MMRCA_REQUIRE(N % 4 == 0 && ld % 4 == 0 && ld >= N, "colsum_accum: N and ld must be multiples of 4");
  if (M <= 0 || N <= 0) return 0;
  const int gx = (int)((N + 255) / 256);
  int rows_per_block = 256;
  int64_t gy = (M + rows_per_block - 1) / rows_per_block;
  while (gy * gx > 4096) { rows_per_block *= 2; gy = (M + rows_per_block - 1) / rows_per_block; }
  MMRCA_DISPATCH_DTYPE(dtype, "colsum_accum",
    hipLaunchKernelGGL(colsum_k<T>, dim3(gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, (const T*)dY, db, M, N, ld, rows_per_block);)
  MMRCA_CHECK_LAUNCH("colsum_accum");
  return 0;
}

// --------------------------------------------------------------------------------------------------------
// text embeddings
// --------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
embed_fwd_k(const int32_t* __restrict__ ids, const int32_t* __restrict__ pos_ids, const T* __restrict__ word,
            const T* __restrict__ pos, const T* __restrict__ type_row, T* __restrict__ out, int64_t rows, int D) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const int64_t id = ids[row], p = pos_ids[row];
    for (int c = lane * 4; c < D; c += 256) {
      Vec4<T> a = Vec4<T>::load(word + id * D + c), b = Vec4<T>::load(pos + p * D + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) a.v[j] += b.v[j];
      if (type_row) {
        Vec4<T> t = Vec4<T>::load(type_row + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) a.v[j] += t.v[j];
      }
      a.store(out + row * D + c);
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
embed_bwd_k(const T* __restrict__ dout, const int32_t* __restrict__ ids, const int32_t* __restrict__ pos_ids,
            float* __restrict__ dword, float* __restrict__ dpos, float* __restrict__ dtype_row, int64_t rows, int D,
            int pad_id, int pos_pad_id) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const int64_t id = ids[row], p = pos_ids[row];
    for (int c = lane * 4; c < D; c += 256) {
      Vec4<T> d = Vec4<T>::load(dout + row * D + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (dword && id != pad_id) atomicAdd(dword + id * D + c + j, d.v[j]);
        if (dpos && p != pos_pad_id) atomicAdd(dpos + p * D + c + j, d.v[j]);
        if (dtype_row) atomicAdd(dtype_row + c + j, d.v[j]);
      }
    }
  }
}

// D <= 1024: the token-type row (one address per column for ALL rows) and the position rows (one per position for the
// whole batch) were 16k-way and 256-way contended atomics, 1 ms per step.  A wave now keeps both sums in registers
// across the rows it walks -- its row stride is a multiple of any power-of-two sequence length, so its position id
// normally never changes -- and flushes them with one atomic per column (on a position change, and at the end).
template <typename T>
__global__ void __launch_bounds__(256)
embed_bwd_acc_k(const T* __restrict__ dout, const int32_t* __restrict__ ids, const int32_t* __restrict__ pos_ids,
                float* __restrict__ dword, float* __restrict__ dpos, float* __restrict__ dtype_row, int64_t rows, int D,
                int pad_id, int pos_pad_id) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float at[4][4], ap[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) { at[k][j] = 0.f; ap[k][j] = 0.f; }
  int64_t cur_p = -1;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const int64_t id = ids[row];
    int64_t p = pos_ids[row];
    if (p == pos_pad_id) p = -1;               // padding_idx of the position table (RoBERTa): that row takes no gradient
    if (dpos && p != cur_p) {                  // wave-uniform
      if (cur_p >= 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = lane * 4 + 256 * k;
          if (c < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { atomicAdd(dpos + cur_p * D + c + j, ap[k][j]); ap[k][j] = 0.f; }
          }
        }
      }
      cur_p = p;
    }
    Vec4<T> d[4];
    bool nz = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane * 4 + 256 * k;
      if (c < D) {
        d[k] = Vec4<T>::load(dout + row * D + c);
        nz = nz || d[k].v[0] != 0.f || d[k].v[1] != 0.f || d[k].v[2] != 0.f || d[k].v[3] != 0.f;
      }
    }
    // a row whose gradient is all zeros (masked positions under class-token pooling: most of a padded batch) adds
    // nothing anywhere; skipping it also removes the worst same-address contention (thousands of [PAD] rows -> one row)
    if (!__any(nz)) continue;
    const bool word_on = dword && id != pad_id;  // nn.Embedding(padding_idx=pad_token_id): the pad row stays zero
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane * 4 + 256 * k;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (word_on) atomicAdd(dword + id * D + c + j, d[k].v[j]);
          if (cur_p >= 0) ap[k][j] += d[k].v[j];
          at[k][j] += d[k].v[j];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane * 4 + 256 * k;
    if (c < D) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (dpos && cur_p >= 0) atomicAdd(dpos + cur_p * D + c + j, ap[k][j]);
        if (dtype_row) atomicAdd(dtype_row + c + j, at[k][j]);
      }
    }
  }
}

extern "C" int mmrca_embed_fwd(const int32_t* ids, const int32_t* pos_ids, const void* word, const void* pos,
                               const void* type_row, void* out, int64_t rows, int D, int dtype, void* stream) {
  MMRCA_REQUIRE(ids && pos_ids && word && pos && out, "embed_fwd: null pointer");
  MMRCA_REQUIRE(D % 4 == 0, "embed_fwd: D must be a multiple of 4");
  if (rows <= 0) return 0;
  const int grid = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
  MMRCA_DISPATCH_DTYPE(dtype, "embed_fwd",
    hipLaunchKernelGGL(embed_fwd_k<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, pos_ids, (const T*)word,
                       (const T*)pos, (const T*)type_row, (T*)out, rows, D);)
  MMRCA_CHECK_LAUNCH("embed_fwd");
  return 0;
}

extern "C" int mmrca_embed_bwd(const void* dout, const int32_t* ids, const int32_t* pos_ids, float* dword, float* dpos,
                               float* dtype_row, int64_t rows, int D, int pad_id, int pos_pad_id, int dtype, void* stream) {
  MMRCA_REQUIRE(dout && ids && pos_ids, "embed_bwd: null pointer");
  MMRCA_REQUIRE(D % 4 == 0, "embed_bwd: D must be a multiple of 4");
  if (rows <= 0) return 0;
  if (D <= 1024) {
    const int g2 = (int)((rows + 3) / 4 < 256 ? (rows + 3) / 4 : 256);     // 1024 waves: row stride 1024
    MMRCA_DISPATCH_DTYPE(dtype, "embed_bwd",
      hipLaunchKernelGGL(embed_bwd_acc_k<T>, dim3(g2), dim3(256), 0, (hipStream_t)stream, (const T*)dout, ids, pos_ids,
                         dword, dpos, dtype_row, rows, D, pad_id, pos_pad_id);)
    MMRCA_CHECK_LAUNCH("embed_bwd");
    return 0;
  }
  const int grid = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
  MMRCA_DISPATCH_DTYPE(dtype, "embed_bwd",
    hipLaunchKernelGGL(embed_bwd_k<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)dout, ids, pos_ids,
                       dword, dpos, dtype_row, rows, D, pad_id, pos_pad_id);)
  MMRCA_CHECK_LAUNCH("embed_bwd");
  return 0;
}

// --------------------------------------------------------------------------------------------------------
// ViT patch embedding glue
// --------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
patchify_k(const float* __restrict__ img, T* __restrict__ out, int B, int C, int Himg, int Wimg, int P, int64_t total4) {
  const int nPw = Wimg / P, nPh = Himg / P, Kp = C * P * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i * 4;
    const int k = (int)(e % Kp);
    const int64_t row = e / Kp;
    const int pw = (int)(row % nPw), ph = (int)((row / nPw) % nPh), b = (int)(row / ((int64_t)nPw * nPh));
    const int px = k % P, py = (k / P) % P, c = k / (P * P);
    const float* src = img + (((int64_t)b * C + c) * Himg + (ph * P + py)) * Wimg + pw * P + px;
    Vec4<float> v = Vec4<float>::load(src);
    Vec4<T> o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = v.v[j];
    o.store(out + e);
  }
}

// any patch size (BLIP-2's ViT-g/14: P = 14): one element per thread
template <typename T>
__global__ void __launch_bounds__(256)
patchify_any_k(const float* __restrict__ img, T* __restrict__ out, int B, int C, int Himg, int Wimg, int P, int64_t total) {
  const int nPw = Wimg / P, nPh = Himg / P, Kp = C * P * P;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(e % Kp);
    const int64_t row = e / Kp;
    const int pw = (int)(row % nPw), ph = (int)((row / nPw) % nPh), b = (int)(row / ((int64_t)nPw * nPh));
    const int px = k % P, py = (k / P) % P, c = k / (P * P);
    out[e] = from_f<T>(img[(((int64_t)b * C + c) * Himg + (ph * P + py)) * Wimg + pw * P + px]);
  }
}

extern "C" int mmrca_patchify_fwd(const float* images, void* patches, int B, int C, int Himg, int Wimg, int P, int dtype, void* stream) {
  MMRCA_REQUIRE(images && patches, "patchify_fwd: null pointer");
  MMRCA_REQUIRE(P > 0 && Himg % P == 0 && Wimg % P == 0, "patchify_fwd: image %dx%d is not a whole number of %dx%d patches", Himg, Wimg, P, P);
  if (P % 4 != 0 || Wimg % 4 != 0) {
    const int64_t total = (int64_t)B * C * Himg * Wimg;
    if (total <= 0) return 0;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    MMRCA_DISPATCH_DTYPE(dtype, "patchify_fwd",
      hipLaunchKernelGGL(patchify_any_k<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, images, (T*)patches, B, C, Himg, Wimg, P, total);)
    MMRCA_CHECK_LAUNCH("patchify_fwd");
    return 0;
  }
  const int64_t total4 = (int64_t)B * C * Himg * Wimg / 4;
  if (total4 <= 0) return 0;
  const int grid = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
  MMRCA_DISPATCH_DTYPE(dtype, "patchify_fwd",
    hipLaunchKernelGGL(patchify_k<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, images, (T*)patches, B, C, Himg, Wimg, P, total4);)
  MMRCA_CHECK_LAUNCH("patchify_fwd");
  return 0;
}

template <typename T>
__global__ void __launch_bounds__(256)
vit_assemble_fwd_k(const T* __restrict__ proj, const T* __restrict__ cls, const T* __restrict__ pos, T* __restrict__ x, int B, int nP, int D, int64_t total4) {
  const int Tn = nP + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i * 4;
    const int d = (int)(e % D);
    const int64_t row = e / D;
    const int t = (int)(row % Tn);
    const int64_t b = row / Tn;
    Vec4<T> a = (t == 0) ? Vec4<T>::load(cls + d) : Vec4<T>::load(proj + (b * nP + (t - 1)) * (int64_t)D + d);
    Vec4<T> p = Vec4<T>::load(pos + (int64_t)t * D + d);
#pragma unroll
    for (int j = 0; j < 4; ++j) a.v[j] += p.v[j];
    a.store(x + e);
  }
}

// dproj = dx rows 1..; dpos[t,d] += sum_b dx[b,t,d]; dcls[d] += sum_b dx[b,0,d].  One thread per 4 columns of one token.
template <typename T>
__global__ void __launch_bounds__(256)
vit_assemble_bwd_k(const T* __restrict__ dx, T* __restrict__ dproj, float* __restrict__ dcls, float* __restrict__ dpos, int B, int nP, int D) {
  const int Tn = nP + 1;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)Tn * D / 4) return;
  const int64_t e = i * 4;
  const int d = (int)(e % D), t = (int)(e / D);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int b = 0;
  for (; b + 8 <= B; b += 8) {                    // eight samples in flight per thread (only Tn*D/4 threads exist)
    Vec4<T> v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = Vec4<T>::load(dx + ((int64_t)(b + u) * Tn + t) * D + d);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (t > 0) v[u].store(dproj + ((int64_t)(b + u) * nP + (t - 1)) * D + d);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += v[u].v[j];
    }
  }
  for (; b < B; ++b) {
    Vec4<T> v = Vec4<T>::load(dx + ((int64_t)b * Tn + t) * D + d);
    if (t > 0) v.store(dproj + ((int64_t)b * nP + (t - 1)) * D + d);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += v.v[j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (dpos) atomicAdd(dpos + (int64_t)t * D + d + j, acc[j]);
    if (t == 0 && dcls) atomicAdd(dcls + d + j, acc[j]);
  }
}

extern "C" int mmrca_vit_assemble_fwd(const void* proj, const void* cls, const void* pos, void* x, int B, int nP, int D, int dtype, void* stream) {
  MMRCA_REQUIRE(proj && cls && pos && x, "vit_assemble_fwd: null pointer");
  MMRCA_REQUIRE(D % 4 == 0, "vit_assemble_fwd: D must be a multiple of 4");
  const int64_t total4 = (int64_t)B * (nP + 1) * D / 4;
  if (total4 <= 0) return 0;
  const int grid = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
  MMRCA_DISPATCH_DTYPE(dtype, "vit_assemble_fwd",
    hipLaunchKernelGGL(vit_assemble_fwd_k<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)proj, (const T*)cls,
                       (const T*)pos, (T*)x, B, nP, D, total4);)
  MMRCA_CHECK_LAUNCH("vit_assemble_fwd");
  return 0;
}

extern "C" int mmrca_vit_assemble_bwd(const void* dx, void* dproj, float* dcls, float* dpos, int B, int nP, int D, int dtype, void* stream) {
  MMRCA_REQUIRE(dx && dproj, "vit_assemble_bwd: null pointer");
  MMRCA_REQUIRE(D % 4 == 0, "vit_assemble_bwd: D must be a multiple of 4");
  const int64_t n = (int64_t)(nP + 1) * D / 4;
  if (n <= 0 || B <= 0) return 0;
  MMRCA_DISPATCH_DTYPE(dtype, "vit_assemble_bwd",
    hipLaunchKernelGGL(vit_assemble_bwd_k<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dx,
                       (T*)dproj, dcls, dpos, B, nP, D);)
  MMRCA_CHECK_LAUNCH("vit_assemble_bwd");
  return 0;
}

// --------------------------------------------------------------------------------------------------------
// loss: CrossEntropyLoss(weight, label_smoothing), mean over the weighted denominator (main_both.py:87-93)
// single block; B*C is tiny (batch x 4)
// --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
xent_k(const float* __restrict__ logits, const int32_t* __restrict__ labels, const float* __restrict__ cw, float eps,
       float* __restrict__ loss, float* __restrict__ dlogits, int B, int C, float grad_scale) {
  __shared__ float s_num[256], s_den[256];
  float num = 0.f, den = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float* z = logits + (int64_t)i * C;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, z[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(z[c] - m);
    const float lse = m + logf(se);
    const int y = labels[i];
    const float wy = cw ? cw[y] : 1.f;
    float smooth = 0.f;
    for (int c = 0; c < C; ++c) smooth += (cw ? cw[c] : 1.f) * (lse - z[c]);
    num += (1.f - eps) * wy * (lse - z[y]) + (eps / (float)C) * smooth;
    den += wy;
  }
  s_num[threadIdx.x] = num; s_den[threadIdx.x] = den;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { s_num[threadIdx.x] += s_num[threadIdx.x + o]; s_den[threadIdx.x] += s_den[threadIdx.x + o]; }
    __syncthreads();
  }
  const float D = s_den[0];
  if (threadIdx.x == 0 && loss) loss[0] = s_num[0] / D;
  if (dlogits) {
    for (int i = threadIdx.x; i < B; i += 256) {
      const float* z = logits + (int64_t)i * C;
      float m = -INFINITY;
      for (int c = 0; c < C; ++c) m = fmaxf(m, z[c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(z[c] - m);
      const int y = labels[i];
      const float wy = cw ? cw[y] : 1.f;
      float wsum = 0.f;
      for (int c = 0; c < C; ++c) wsum += cw ? cw[c] : 1.f;
      for (int c = 0; c < C; ++c) {
        const float p = expf(z[c] - m) / se;
        const float wc = cw ? cw[c] : 1.f;
        // d/dz_c of (1-eps)*wy*(lse - z_y) + eps/C * sum_k w_k (lse - z_k)
        const float g = (1.f - eps) * wy * (p - (c == y ? 1.f : 0.f)) + (eps / (float)C) * (wsum * p - wc);
        dlogits[(int64_t)i * C + c] = g / D * grad_scale;
      }
    }
  }
}

extern "C" int mmrca_xent_fwd_bwd(const float* logits, const int32_t* labels, const float* class_w, float smoothing,
                                  float* loss, float* dlogits, int B, int C, float grad_scale, void* stream) {
  MMRCA_REQUIRE(logits && labels, "xent: null pointer");
  MMRCA_REQUIRE(B > 0 && C > 0 && C <= 1024, "xent: bad shape B=%d C=%d", B, C);
  hipLaunchKernelGGL(xent_k, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, class_w, smoothing, loss, dlogits, B, C, grad_scale);
  MMRCA_CHECK_LAUNCH("xent");
  return 0;
}

// --------------------------------------------------------------------------------------------------------
// optimizers on the flat arenas (+ refresh of the bf16 working copy)
// --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sgd_k(float* __restrict__ p, const float* __restrict__ g, bf16_t* __restrict__ lp, int64_t n4, float lr, float wd, float gs,
      bf16_t* __restrict__ lp_lo) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    Vec4<float> pv = Vec4<float>::load(p + i * 4), gv = Vec4<float>::load(g + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) pv.v[j] -= lr * (gv.v[j] * gs + wd * pv.v[j]);
    pv.store(p + i * 4);
    if (lp) { Vec4<bf16_t> o; for (int j = 0; j < 4; ++j) o.v[j] = pv.v[j]; o.store(lp + i * 4); }
    if (lp_lo) { Vec4<bf16_t> o; for (int j = 0; j < 4; ++j) o.v[j] = pv.v[j] - (float)(bf16_t)pv.v[j]; o.store(lp_lo + i * 4); }   // bf16x3 mode: p = hi + lo
  }
}

__global__ void __launch_bounds__(256)
adamw_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, bf16_t* __restrict__ lp,
        int64_t n4, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, float gs, bf16_t* __restrict__ lp_lo) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    Vec4<float> pv = Vec4<float>::load(p + i * 4), gv = Vec4<float>::load(g + i * 4), mv = Vec4<float>::load(m + i * 4), vv = Vec4<float>::load(v + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = gv.v[j] * gs;
      pv.v[j] *= (1.f - lr * wd);
      mv.v[j] = b1 * mv.v[j] + (1.f - b1) * gg;
      vv.v[j] = b2 * vv.v[j] + (1.f - b2) * gg * gg;
      const float denom = sqrtf(vv.v[j]) / sqrtf(bc2) + eps;
      pv.v[j] -= (lr / bc1) * mv.v[j] / denom;
    }
    pv.store(p + i * 4); mv.store(m + i * 4); vv.store(v + i * 4);
    if (lp) { Vec4<bf16_t> o; for (int j = 0; j < 4; ++j) o.v[j] = pv.v[j]; o.store(lp + i * 4); }
    if (lp_lo) { Vec4<bf16_t> o; for (int j = 0; j < 4; ++j) o.v[j] = pv.v[j] - (float)(bf16_t)pv.v[j]; o.store(lp_lo + i * 4); }
  }
}

__global__ void __launch_bounds__(256) cast_k(const float* __restrict__ s, bf16_t* __restrict__ d, int64_t n4, bf16_t* __restrict__ d_lo) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    Vec4<float> a = Vec4<float>::load(s + i * 4);
    Vec4<bf16_t> o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = a.v[j];
    o.store(d + i * 4);
    if (d_lo) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o.v[j] = a.v[j] - (float)(bf16_t)a.v[j];
      o.store(d_lo + i * 4);
    }
  }
}

static inline int ew_grid(int64_t n4) { int64_t g = (n4 + 255) / 256; return (int)(g < 8192 ? g : 8192); }

extern "C" int mmrca_sgd_step_x3(float* p, const float* g, void* lp_hi, void* lp_lo, int64_t n, float lr, float wd, float grad_scale,
                                 void* stream) {
  MMRCA_REQUIRE(p && g, "sgd_step: null pointer");
  MMRCA_REQUIRE(n % 4 == 0, "sgd_step: arena length must be a multiple of 4");
  MMRCA_REQUIRE(!lp_lo || lp_hi, "sgd_step: a lo plane needs its hi plane");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sgd_k, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, (bf16_t*)lp_hi, n / 4, lr, wd, grad_scale,
                     (bf16_t*)lp_lo);
  MMRCA_CHECK_LAUNCH("sgd_step");
  return 0;
}
extern "C" int mmrca_sgd_step(float* p, const float* g, void* lp, int64_t n, float lr, float wd, float grad_scale, void* stream) {
  return mmrca_sgd_step_x3(p, g, lp, nullptr, n, lr, wd, grad_scale, stream);
}

extern "C" int mmrca_adamw_step_x3(float* p, const float* g, float* m, float* v, void* lp_hi, void* lp_lo, int64_t n, float lr,
                                   float beta1, float beta2, float eps, float wd, int step, float grad_scale, void* stream) {
  MMRCA_REQUIRE(p && g && m && v, "adamw_step: null pointer");
  MMRCA_REQUIRE(n % 4 == 0 && step >= 1, "adamw_step: bad arguments");
  MMRCA_REQUIRE(!lp_lo || lp_hi, "adamw_step: a lo plane needs its hi plane");
  if (n <= 0) return 0;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_k, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)lp_hi, n / 4, lr, beta1,
                     beta2, eps, wd, bc1, bc2, grad_scale, (bf16_t*)lp_lo);
  MMRCA_CHECK_LAUNCH("adamw_step");
  return 0;
}
extern "C" int mmrca_adamw_step(float* p, const float* g, float* m, float* v, void* lp, int64_t n, float lr, float beta1,
                                float beta2, float eps, float wd, int step, float grad_scale, void* stream) {
  return mmrca_adamw_step_x3(p, g, m, v, lp, nullptr, n, lr, beta1, beta2, eps, wd, step, grad_scale, stream);
}

extern "C" int mmrca_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  MMRCA_REQUIRE(src && dst, "cast: null pointer");
  MMRCA_REQUIRE(n % 4 == 0, "cast: n must be a multiple of 4");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_k, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n / 4, (bf16_t*)nullptr);
  MMRCA_CHECK_LAUNCH("cast");
  return 0;
}

// bf16x3 mode: x = hi + lo with hi = bf16(x) (round to nearest even), lo = bf16(x - hi): 16 significant bits in two bf16 planes
extern "C" int mmrca_split_f32(const float* src, void* hi, void* lo, int64_t n, void* stream) {
  MMRCA_REQUIRE(src && hi && lo, "split_f32: null pointer");
  MMRCA_REQUIRE(n % 4 == 0, "split_f32: n must be a multiple of 4");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_k, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)hi, n / 4, (bf16_t*)lo);
  MMRCA_CHECK_LAUNCH("split_f32");
  return 0;
}

// Stochastic depth of the conv image encoders (torchvision StochasticDepth, mode "row", train only): out[i][b] = keep / (1 - p[i]) with
// keep = uniform(seed; i * B + b) >= p[i].  A counter-based draw like every other mask of the step, so a HIP-graph replay (mask epoch,
// common.h) and the eager step of the same index apply the SAME masks.
__global__ void sd_rowscale_k(const float* __restrict__ p, float* __restrict__ out, int n, int B, uint64_t seed) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * B) return;
  const float pi = p[t / B];
  out[t] = mmrca_uniform(seed, (uint64_t)t) >= pi ? 1.0f / (1.0f - pi) : 0.0f;
}
extern "C" int mmrca_sd_rowscale(const float* p, float* out, int n, int B, uint64_t seed, void* stream) {
  MMRCA_REQUIRE(p && out, "sd_rowscale: null pointer");
  MMRCA_REQUIRE(n >= 0 && B >= 0, "sd_rowscale: bad shape");
  if (n == 0 || B == 0) return 0;
  hipLaunchKernelGGL(sd_rowscale_k, dim3((n * B + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, out, n, B, seed);
  MMRCA_CHECK_LAUNCH("sd_rowscale");
  return 0;
}

MMRCA_SEED_EPOCH_EXPORT(rowops)   // this translation unit's copy of the mask epoch (common.h)
