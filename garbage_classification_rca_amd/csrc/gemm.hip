// K2: C[M,N] = act(A (.) B + bias) + addend  -- every nn.Linear forward, input-gradient and weight-gradient
// on the MM-RCA path (see include/mmrca.h).
//
// Implementations behind the one entry point (gemm_dispatch picks; MMRCA_GEMM_* variables and the impl argument override):
//   * gemm_ref_k    : fp32-accumulating VALU tile kernel, any dtype / layout / shape: the checker the other kernels are
//                     tested against (impl = REF) and the last-resort fallback.
//   * gemm_gen_k    : the general-shape kernel on the matrix cores (fp32 32x32x2 / bf16 32x32x16), any layout / shape:
//                     the fp32 "<= 1e-3" mode and the bf16 shapes the tiled kernels cannot take (conv channel counts,
//                     the 4-class logits).
//   * gemm_mfma_k   : bf16 v_mfma_f32_16x16x32_bf16 kernel, 128x128x64 tiles, 4 waves (2x2, 64x64 per wave),
//                     operands staged HBM->LDS with global_load_lds (16 B/lane, no VGPR round trip), double-buffered.
//                     Both operand layouts are handled in LDS, so no transposed copies of weights or activations
//                     ever exist in HBM:
//                       ROWK [rows][k]  : 128-B LDS rows, 16-B chunks XOR-swizzled by (row>>1)&7 -> conflict-free
//                                         ds_read_b128 fragment reads (swizzle applied on the global source address,
//                                         because global_load_lds writes lane-linear).
//                       KROW [k][rows]  : 256-B LDS rows, 32-B granules XOR-swizzled by (k&3)|((k>>3)&1)<<2, read
//                                         with ds_read_b64_tr_b16 (hardware transpose) -> conflict-free.
//                     XCD-aware block->tile map (8 XCDs, private L2s) with grouped-M rastering.
//                     Weight gradients (fp32, +=) split the long contraction over blockIdx.y and combine with
//                     fp32 atomics shaped as 64-B row segments.
//   * gemm_mfma_k32 / gemm_mfma_k1s : the same tile with a 32-deep K step / a single LDS stage (more blocks per CU);
//                     used for the shapes the dispatch table names.
//   * gemm256.hip   : the persistent 256x256 kernel that runs the encoder-sized GEMMs (the headline K2 number).
#include "common.h"
#include "lds_asm.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;

// ======================================================================================================
// reference kernel
// ======================================================================================================
template <typename T>
__global__ void __launch_bounds__(256)
gemm_ref_k(const T* __restrict__ A, const T* __restrict__ B, void* __restrict__ Cv, const T* __restrict__ bias,
           const T* __restrict__ addend, T* __restrict__ preact, int64_t M, int64_t N, int64_t K,
           int64_t sam, int64_t sak, int64_t sbn, int64_t sbk, int64_t ldc, int act, int accum) {
  __shared__ float As[16][65], Bs[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
  // Three-level summation (16 products -> 16 partials -> the rest): the rounding error of a length-K fp32 dot product
  // then grows like sqrt(K/256) + 8 ulps instead of sqrt(K).  This kernel is the fp32 PARITY mode: with K up to 3,072
  // (and 50 k for weight gradients) a single running sum put the encoder features ~1e-5 away from a float64 evaluation,
  // which the head's softmax-gradient cancellation amplified to 2e-3 on its attention projections.
  float acc[4][4], mid[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = mid[i][j] = 0.f;
  int blk = 0;
  for (int64_t k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = threadIdx.x + i * 256;
      int mm, kk;
      if (sak == 1) { mm = idx >> 4; kk = idx & 15; } else { kk = idx >> 6; mm = idx & 63; }
      const int64_t gm = m0 + mm, gk = k0 + kk;
      As[kk][mm] = (gm < M && gk < K) ? to_f(A[gm * sam + gk * sak]) : 0.f;
      int nn, k2;
      if (sbk == 1) { nn = idx >> 4; k2 = idx & 15; } else { k2 = idx >> 6; nn = idx & 63; }
      const int64_t gn = n0 + nn, gk2 = k0 + k2;
      Bs[k2][nn] = (gn < N && gk2 < K) ? to_f(B[gn * sbn + gk2 * sbk]) : 0.f;
    }
    __syncthreads();
    float part[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) part[i][j] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) part[i][j] = fmaf(a[i], b[j], part[i][j]);
    }
    const bool fold = (++blk & 15) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mid[i][j] += part[i][j];
        if (fold) { acc[i][j] += mid[i][j]; mid[i][j] = 0.f; }
      }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] += mid[i][j];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float v = acc[i][j];
      if (bias) v += to_f(bias[n]);
      if (act == MMRCA_ACT_GELU_SAVE_GRAD) { preact[m * ldc + n] = from_f<T>(gelu_grad_f(v)); v = gelu_f(v); }
      else if (act == MMRCA_ACT_MUL) v *= to_f(preact[m * ldc + n]);
      else if (act == MMRCA_ACT_GELU_BWD) v *= gelu_grad_f(to_f(preact[m * ldc + n]));
      else {
        if (preact) preact[m * ldc + n] = from_f<T>(v);
        if (act == MMRCA_ACT_GELU) v = gelu_f(v);
      }
      if (addend) v += to_f(addend[m * ldc + n]);
      if (accum) ((float*)Cv)[m * ldc + n] += v;
      else ((T*)Cv)[m * ldc + n] = from_f<T>(v);
    }
  }
}

// ======================================================================================================
// gemm_gen_k: the general-shape kernel on the matrix cores (fp32: v_mfma_f32_32x32x2_f32, bf16: v_mfma_f32_32x32x16_bf16), any layout / shape
// ======================================================================================================
// Same contract and epilogue as gemm_ref_k at about twice its speed (64-72 vs 31-35 TFLOP/s on the fp32 encoder shapes,
// 157 peak): it is what AUTO runs for fp32 (the "<= 1e-3" mode, bench.py --dtype fp32) and for bf16 shapes the bf16 MFMA
// kernels cannot take (the conv backbones' channel counts).  Operands are converted to fp32 while they are staged
// (scalar, predicated loads), products accumulate in fp32, and -- like gemm_ref_k -- the running sum is folded into a
// second accumulator every 64 k and a third every 1024 k so that the rounding error of a long dot product (K = 50 k in
// the weight gradients) does not grow with sqrt(K).  128x64 tile, four waves of 32x64 (two MFMA tiles), K step 16,
// double-buffered LDS with one barrier per step.  MFMA operand layout (32x32x2): lane l supplies A[l%32][l/32] and
// B[l/32][l%32]; a lane that holds four consecutive k of its row feeds four MFMAs (k slot of MFMA m = 4 (l/32) + m: a
// permutation of the 8 k of a chunk, the same for both operands).
// (Measured and dropped: 16-byte group loads for the staging -- more registers, 47-55 TFLOP/s on the fp32 shapes, and the
// small-K conv shapes stay latency-bound at 20-35 TFLOP/s either way; they need their own bf16 kernel.)
typedef float f32x16g __attribute__((ext_vector_type(16)));
typedef float f32x4g __attribute__((ext_vector_type(4)));
#define GEN_LD 20     // LDS row pitch in floats (16 k + 4 pad)

// AK1 / BK1: the operand's k index is its contiguous dimension (ROWK).  With the layout a compile-time fact every staging
// coordinate of a thread is fixed for the whole K loop: element i of a step sits at base + i * (uniform stride) in memory and
// at base + i * (compile-time constant) in LDS, the row predicate is a per-thread bit mask, and only the last, partial K step
// checks k -- the generic index arithmetic cost about as many VALU cycles per step as the MFMAs.
template <typename T, bool AK1, bool BK1>
__global__ void __launch_bounds__(256)
gemm_gen_k(const T* __restrict__ A, const T* __restrict__ B, void* __restrict__ Cv, const T* __restrict__ bias,
           const T* __restrict__ addend, T* __restrict__ preact, int64_t M, int64_t N, int64_t K,
           int64_t sam, int64_t sak, int64_t sbn, int64_t sbk, int64_t ldc, int act, int accum, int tiles_n, int64_t ksplit_len) {
  __shared__ __attribute__((aligned(16))) float As[2][128 * GEN_LD], Bs[2][64 * GEN_LD];
  // accumulate mode with few tiles and a long contraction (conv weight gradients: K = B*H*W rows): blockIdx.y owns a K range and
  // adds its partial product with fp32 atomics
  const int64_t kbeg = (int64_t)blockIdx.y * ksplit_len;
  const int64_t Kend = kbeg + ksplit_len < K ? kbeg + ksplit_len : K;
  const bool split = gridDim.y > 1;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * 128, n0 = (int64_t)(blockIdx.x % tiles_n) * 64;
  // fp32 operands: three-level summation, 64 k -> 1024 k -> the rest (see gemm_ref_k).  bf16 operands carry 8 bits of
  // mantissa, a single fp32 running sum is exact enough for them and 64 fewer registers double the resident workgroups
  constexpr bool LEVELS = sizeof(T) == 4;
  f32x16g acc[2], mid[LEVELS ? 2 : 1], tot[LEVELS ? 2 : 1];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; if (LEVELS) { mid[j][r] = 0.f; tot[j][r] = 0.f; } }
  // staging: A tile = 128 rows x 16 k (8 elements per thread), B tile = 64 rows x 16 k (4 per thread); consecutive lanes follow
  // the operand's contiguous dimension
  const int a_row0 = AK1 ? (t >> 4) : (t & 127), a_k0 = AK1 ? (t & 15) : (t >> 7);
  const int b_row0 = BK1 ? (t >> 4) : (t & 63), b_k0 = BK1 ? (t & 15) : (t >> 6);
  constexpr int A_DROW = AK1 ? 16 : 0, A_DK = AK1 ? 0 : 2, B_DROW = BK1 ? 16 : 0, B_DK = BK1 ? 0 : 4;
  const T* pa = A + (m0 + a_row0) * sam + (kbeg + a_k0) * sak;
  const T* pb = B + (n0 + b_row0) * sbn + (kbeg + b_k0) * sbk;
  const int64_t a_is = A_DROW * sam + A_DK * sak, b_is = B_DROW * sbn + B_DK * sbk;      // memory stride between a thread's elements
  unsigned a_ok = 0, b_ok = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) a_ok |= (unsigned)(m0 + a_row0 + i * A_DROW < M) << i;
#pragma unroll
  for (int i = 0; i < 4; ++i) b_ok |= (unsigned)(n0 + b_row0 + i * B_DROW < N) << i;
  // bf16 operands stay bf16 in LDS (row pitch 24 = 16 k + 8 pad: 16-byte aligned rows) and go through v_mfma_f32_32x32x16_bf16:
  // one MFMA per 32x32 tile and K step instead of eight fp32 ones at a sixteenth of the cycles -- the products of two bf16
  // values are exact in fp32 either way
  constexpr bool BF = sizeof(T) == 2;
  // (a KROW bf16 operand uses pitch 20: its tile is written TRANSPOSED, eight 2-byte stores per 16-byte global load, and with
  //  40-byte rows the four row groups of a wave land in four different bank quarters; its fragments are read as 2 x 8 bytes)
  constexpr int LDA = BF ? (AK1 ? 24 : 20) : GEN_LD, LDB = BF ? ((BK1 || AK1) ? 24 : 20) : GEN_LD;     // (B: only beside a KROW A, see vec_b)
  const int a_lds = a_row0 * LDA + a_k0, b_lds = b_row0 * LDB + b_k0;
  constexpr int A_LI = A_DROW * LDA + A_DK, B_LI = B_DROW * LDB + B_DK;
  typedef typename std::conditional<BF, T, float>::type S;      // staged element type
  S ra[8], rb[4];
  auto fetch_a = [&](int64_t k0, bool full) {        // full: the whole 16-wide step lies inside [kbeg, Kend)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      ra[i] = ((a_ok >> i) & 1) && (full || k0 + a_k0 + i * A_DK < Kend) ? (S)pa[i * a_is] : (S)0.f;
    pa += 16 * sak;
  };
  auto fetch_b = [&](int64_t k0, bool full) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      rb[i] = ((b_ok >> i) & 1) && (full || k0 + b_k0 + i * B_DK < Kend) ? (S)pb[i * b_is] : (S)0.f;
    pb += 16 * sbk;
  };
  auto stash_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { if constexpr (BF) reinterpret_cast<T*>(As[buf])[a_lds + i * A_LI] = ra[i]; else As[buf][a_lds + i * A_LI] = ra[i]; }
  };
  auto stash_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { if constexpr (BF) reinterpret_cast<T*>(Bs[buf])[b_lds + i * B_LI] = rb[i]; else Bs[buf][b_lds + i * B_LI] = rb[i]; }
  };
  // bf16 ROWK operands, aligned: a thread moves ONE group of 8 contiguous k of a row per step -- a 16-byte load and a 16-byte
  // LDS store instead of eight two-byte ones.  KROW operands (weight gradients): ONE 16-byte load of 8 rows of a k, written
  // transposed with eight 2-byte stores into a pitch-20 image (with pitch 24 the stores were 8-way bank conflicted and the
  // variant lost): conv weight gradients 542 -> 426 us on average (configs[2]).
  typedef __attribute__((ext_vector_type(8))) __bf16 g_b8;
  bool vec_a = false, vec_b = false;
  if constexpr (BF) {
    const bool kok = (Kend - kbeg) % 8 == 0 && kbeg % 8 == 0;
    vec_a = AK1 ? (kok && (((uintptr_t)A) & 15) == 0 && sam % 8 == 0) : (sam == 1 && (((uintptr_t)A) & 15) == 0 && sak % 8 == 0 && M % 8 == 0);
    // (a KROW B beside a ROWK A -- the input-gradient form -- keeps the element-wise staging: measured 5 % faster there, the
    //  transposing stores of only half the threads outweigh the saved loads; beside a KROW A -- weight gradients -- 21 % slower)
    vec_b = BK1 ? (kok && (((uintptr_t)B) & 15) == 0 && sbn % 8 == 0) : (!AK1 && sbn == 1 && (((uintptr_t)B) & 15) == 0 && sbk % 8 == 0 && N % 8 == 0);
  }
  // ROWK group = (row, first of 8 k); KROW group = (one k, first of 8 rows): 16 k x 16 (A) / 8 (B) row groups per step, lanes of a
  // wave = 16 k x 4 row groups
  const int va_r = AK1 ? t >> 1 : 8 * (t >> 4), va_k = AK1 ? 8 * (t & 1) : (t & 15);
  const int vb_r = BK1 ? (t & 127) >> 1 : 8 * ((t & 127) >> 4), vb_k = BK1 ? 8 * (t & 1) : (t & 15);
  g_b8 ga, gb;
  auto zero8 = [&]() { g_b8 z; for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f; return z; };
  auto fetch = [&](int64_t k0, bool full) {
    if constexpr (BF) {
      if (vec_a) {
        const bool ok = m0 + va_r < M && k0 + va_k < Kend;
        ga = ok ? *reinterpret_cast<const g_b8*>(AK1 ? A + (m0 + va_r) * sam + (k0 + va_k) : A + (k0 + va_k) * sak + (m0 + va_r)) : zero8();
      } else fetch_a(k0, full);
      if (vec_b) {
        if (t < 128) {
          const bool ok = n0 + vb_r < N && k0 + vb_k < Kend;
          gb = ok ? *reinterpret_cast<const g_b8*>(BK1 ? B + (n0 + vb_r) * sbn + (k0 + vb_k) : B + (k0 + vb_k) * sbk + (n0 + vb_r)) : zero8();
        }
      } else fetch_b(k0, full);
    } else { fetch_a(k0, full); fetch_b(k0, full); }
  };
  auto stash = [&](int buf) {
    if constexpr (BF) {
      if (vec_a) {
        T* d = reinterpret_cast<T*>(As[buf]) + va_r * LDA + va_k;
        if constexpr (AK1) *reinterpret_cast<g_b8*>(d) = ga;
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) d[j * LDA] = ga[j];
        }
      } else stash_a(buf);
      if (vec_b) {
        if (t < 128) {
          T* d = reinterpret_cast<T*>(Bs[buf]) + vb_r * LDB + vb_k;
          if constexpr (BK1) *reinterpret_cast<g_b8*>(d) = gb;
          else {
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j * LDB] = gb[j];
          }
        }
      } else stash_b(buf);
    } else { stash_a(buf); stash_b(buf); }
  };
  fetch(kbeg, kbeg + 16 <= Kend);
  stash(0);
  __syncthreads();
  const int l32 = lane & 31, g = lane >> 5;
  int step = 0;
  for (int64_t k0 = kbeg; k0 < Kend; k0 += 16, ++step) {
    const int buf = step & 1;
    const bool more = k0 + 16 < Kend;
    if (more) fetch(k0 + 16, k0 + 32 <= Kend);
    if constexpr (BF) {
      typedef __attribute__((ext_vector_type(8))) __bf16 b16x8;
      typedef __attribute__((ext_vector_type(4))) __bf16 b16x4;
      auto frag = [&](const T* p, auto aligned16) -> b16x8 {
        if constexpr (decltype(aligned16)::value) return *reinterpret_cast<const b16x8*>(p);
        else {
          const b16x4 lo = *reinterpret_cast<const b16x4*>(p), hi = *reinterpret_cast<const b16x4*>(p + 4);
          return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      };
      const b16x8 a = frag(&reinterpret_cast<const T*>(As[buf])[(32 * wave + l32) * LDA + 8 * g], std::integral_constant<bool, LDA % 8 == 0>{});
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const b16x8 bj = frag(&reinterpret_cast<const T*>(Bs[buf])[(32 * j + l32) * LDB + 8 * g], std::integral_constant<bool, LDB % 8 == 0>{});
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bj, acc[j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x4g a = *reinterpret_cast<const f32x4g*>(&As[buf][(32 * wave + l32) * GEN_LD + 8 * c + 4 * g]);
        f32x4g b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f32x4g*>(&Bs[buf][(32 * j + l32) * GEN_LD + 8 * c + 4 * g]);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[j][m], acc[j], 0, 0, 0);
      }
    }
    if (LEVELS && (step & 3) == 3) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        mid[LEVELS ? j : 0] += acc[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      }
      if ((step & 63) == 63) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          tot[LEVELS ? j : 0] += mid[LEVELS ? j : 0];
#pragma unroll
          for (int r = 0; r < 16; ++r) mid[LEVELS ? j : 0][r] = 0.f;
        }
      }
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const f32x16g sum = LEVELS ? tot[LEVELS ? j : 0] + (mid[LEVELS ? j : 0] + acc[j]) : acc[j];
    const int64_t n = n0 + 32 * j + l32;
    if (n >= N) continue;
    const float bn = bias ? to_f(bias[n]) : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t m = m0 + 32 * wave + 8 * (r >> 2) + 4 * g + (r & 3);
      if (m >= M) continue;
      float v = sum[r] + bn;
      if (act == MMRCA_ACT_GELU_SAVE_GRAD) { preact[m * ldc + n] = from_f<T>(gelu_grad_f(v)); v = gelu_f(v); }
      else if (act == MMRCA_ACT_MUL) v *= to_f(preact[m * ldc + n]);
      else if (act == MMRCA_ACT_GELU_BWD) v *= gelu_grad_f(to_f(preact[m * ldc + n]));
      else {
        if (preact) preact[m * ldc + n] = from_f<T>(v);
        if (act == MMRCA_ACT_GELU) v = gelu_f(v);
      }
      if (addend) v += to_f(addend[m * ldc + n]);
      if (accum) { if (split) atomicAdd((float*)Cv + m * ldc + n, v); else ((float*)Cv)[m * ldc + n] += v; }
      else ((T*)Cv)[m * ldc + n] = from_f<T>(v);
    }
  }
}

// ======================================================================================================
// MFMA kernel
// ======================================================================================================
#define GBM 128
#define GBN 128
#define GBK 64
#define TILE_BYTES (128 * 64 * 2)    // one operand tile: 16 KiB in either layout
#define LDS128_BYTES (128 * (128 * 4 + 16))   // max(2 x (A,B) tiles = 64 KiB, epilogue staging 66 KiB)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ int krow_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// 16 zero bytes in global memory: the source of a B chunk past the end of the contraction (the EDGE forms of stage_tile / stage_tile32)
__device__ __attribute__((aligned(16))) static const unsigned int mmrca_zero_chunk[4] = {0u, 0u, 0u, 0u};

// stage one 128x64 operand tile (16 x 1-KiB wave instructions; this wave issues 4 of them)
// EDGE (round 5; see stage_tile32): the last step of a contraction that is a multiple of 8 but not of 64 -- chunks past `kend` are fetched
// as zeros on the B side (2) and as a duplicate of a valid chunk on the A side (1)
template <bool KROW, int EDGE = 0>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t rows_total,
                                           int64_t k0, char* lds_tile, int wave, int lane, int64_t kend = 0) {
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    const int i = wave * 4 + ii;
    const bf16_t* src;
    if (!KROW) {
      const int r = 8 * i + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      int64_t gr = row0 + r;
      if (gr > rows_total - 1) gr = rows_total - 1;        // edge rows: read a valid row, results are never stored
      int64_t kc = k0 + c * 8;
      if (EDGE == 1 && kc >= kend) kc = kend - 8;
      src = base + gr * ld + kc;
      if (EDGE == 2 && kc >= kend) src = reinterpret_cast<const bf16_t*>(mmrca_zero_chunk);
    } else {
      const int kr = 4 * i + (lane >> 4);
      const int chp = lane & 15;
      const int c = ((((chp >> 1) ^ krow_f(kr))) << 1) | (chp & 1);
      int64_t col = row0 + c * 8;
      if (col > rows_total - 8) col = rows_total - 8;       // ragged edge (rows_total % 8 == 0): read a valid chunk, results are never stored
      int64_t kk = k0 + kr;
      if (EDGE == 1 && kk >= kend) kk = kend - 1;
      src = base + kk * ld + col;
      if (EDGE == 2 && kk >= kend) src = reinterpret_cast<const bf16_t*>(mmrca_zero_chunk);
    }
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(lds_tile + i * 1024), 16, 0, 0);
  }
}

// the same with NI instructions per wave (an 8-wave block stages 256 rows with NI = 4, 128 rows / 64 k-rows with NI = 2)
template <bool KROW, int NI>
__device__ __forceinline__ void stage_tile_n(const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t rows_total,
                                             int64_t k0, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int ii = 0; ii < NI; ++ii) {
    const int i = wave * NI + ii;
    const bf16_t* src;
    if (!KROW) {
      const int r = 8 * i + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      int64_t gr = row0 + r;
      if (gr > rows_total - 1) gr = rows_total - 1;
      src = base + gr * ld + k0 + c * 8;
    } else {
      const int kr = 4 * i + (lane >> 4);
      const int chp = lane & 15;
      const int c = ((((chp >> 1) ^ krow_f(kr))) << 1) | (chp & 1);
      int64_t col = row0 + c * 8;
      if (col > rows_total - 8) col = rows_total - 8;       // ragged edge (rows_total % 8 == 0): read a valid chunk, results are never stored
      src = base + (k0 + kr) * ld + col;
    }
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(lds_tile + i * 1024), 16, 0, 0);
  }
}

// fragment of the 16 rows starting at tile-row `rb` (multiple of 16), k-step ks (32 contraction elements)
template <bool KROW>
__device__ __forceinline__ bf16x8 load_frag(const char* lds_tile, int rb, int ks, int lane) {
  if (!KROW) {
    const int r = rb + (lane & 15);
    const int ch = 4 * ks + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds_tile + r * 128 + ((ch ^ ((r >> 1) & 7)) << 4));
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = 32 * ks + 8 * g + q;
    const int off0 = row * 256 + ((((rb >> 4) ^ krow_f(row))) << 5) + p * 8;
    const int row1 = row + 4;
    const int off1 = row1 * 256 + ((((rb >> 4) ^ krow_f(row1))) << 5) + p * 8;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds_tile + off0));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds_tile + off1));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

// the same fragment through an asm read (lds_asm.h): LDS byte addresses of the one or two reads
template <bool KROW>
__device__ __forceinline__ void frag_addr(const char* lds_tile, int rb, int ks, int lane, unsigned& a0, unsigned& a1) {
  const unsigned base = (unsigned)(uintptr_t)(lds_void*)lds_tile;
  if (!KROW) {
    const int r = rb + (lane & 15);
    const int ch = 4 * ks + (lane >> 4);
    a0 = a1 = base + r * 128 + ((ch ^ ((r >> 1) & 7)) << 4);
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = 32 * ks + 8 * g + q;
    a0 = base + row * 256 + ((((rb >> 4) ^ krow_f(row))) << 5) + p * 8;
    a1 = base + (row + 4) * 256 + ((((rb >> 4) ^ krow_f(row + 4))) << 5) + p * 8;
  }
}

template <bool A_KROW, bool B_KROW, bool ATOMIC_F32, bool FUSE_DB>
__global__ void __launch_bounds__(256)
gemm_mfma_k(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, void* __restrict__ Cv, const bf16_t* __restrict__ bias,
            const bf16_t* __restrict__ addend, bf16_t* __restrict__ preact, int64_t M, int64_t N, int64_t K,
            int64_t lda, int64_t ldb, int64_t ldc, int act, int tiles_m, int tiles_n, int64_t ksplit_len, float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][A tile | B tile]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware remap (blocks b and b+8 share an XCD under round-robin dispatch; speed only), then grouped-M raster
  const int nwg = tiles_m * tiles_n;
  const int orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int GROUP = 8;
  const int group = wgid / (GROUP * tiles_n);
  const int first_m = group * GROUP;
  const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int tm = first_m + (wgid % (GROUP * tiles_n)) % gsize;
  const int tn = (wgid % (GROUP * tiles_n)) / gsize;
  const int64_t m_blk = (int64_t)tm * GBM, n_blk = (int64_t)tn * GBN;

  const int64_t kbeg = (int64_t)blockIdx.y * ksplit_len;
  int64_t kend = kbeg + ksplit_len; if (kend > K) kend = K;
  const int nt = (int)((kend - kbeg + GBK - 1) / GBK);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // bias gradient fused into the weight-gradient GEMM: row sums of the A operand (= column sums of dY) come from one
  // extra MFMA per A fragment against an all-ones B fragment; only the first column of tiles and waves does it.
  constexpr bool do_bias = ATOMIC_F32 && FUSE_DB;   // work split: k-step (mod tiles_n) -> tile column, ks -> wave column
  const int kstep0 = (int)(kbeg / GBK);
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

  if (nt > 0) {
    stage_tile<A_KROW>(A, lda, m_blk, M, kbeg, smem, wave, lane);
    stage_tile<B_KROW>(B, ldb, n_blk, N, kbeg, smem + TILE_BYTES, wave, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * 2 * TILE_BYTES;
    char* nxt = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
    if (t + 1 < nt) {
      stage_tile<A_KROW>(A, lda, m_blk, M, kbeg + (int64_t)(t + 1) * GBK, nxt, wave, lane);
      stage_tile<B_KROW>(B, ldb, n_blk, N, kbeg + (int64_t)(t + 1) * GBK, nxt + TILE_BYTES, wave, lane);
    }
    // fragment reads are asm (lds_asm.h): the prefetch of K-tile t+1 issued above stays in flight under these reads and
    // the 32 MFMAs below; it is waited for (vmcnt(0)) only at the end of the step
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Frag<A_KROW> afr[4];
      Frag<B_KROW> bfrr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { unsigned a0, a1; frag_addr<A_KROW>(cur, wr * 64 + i * 16, ks, lane, a0, a1); read_frag_rt<A_KROW>(afr[i], a0, a1); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { unsigned a0, a1; frag_addr<B_KROW>(cur + TILE_BYTES, wc * 64 + j * 16, ks, lane, a0, a1); read_frag_rt<B_KROW>(bfrr[j], a0, a1); }
      lgkm0(afr); lgkm0(bfrr);
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { af[i] = frag_val(afr[i]); bfr[i] = frag_val(bfrr[i]); }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (ATOMIC_F32) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);  // D[m][n]
          else            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // D[n][m]
        }
      if (do_bias && ks == wc && ((kstep0 + t) % tiles_n) == tn) {
#pragma unroll
        for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);       // the MFMAs stay ABOVE the wait (they are register-only: asm ordering does not hold them)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }

  const int g = lane >> 4, l16 = lane & 15;
  if (ATOMIC_F32) {
    float* C = (float*)Cv;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t n = n_blk + wc * 64 + j * 16 + l16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t m = m_blk + wr * 64 + i * 16 + 4 * g + r;
          if (m < M && n < N) atomicAdd(C + m * ldc + n, acc[i][j][r]);
        }
      }
    if (do_bias && l16 == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t m = m_blk + wr * 64 + i * 16 + 4 * g + r;
          if (m < M) atomicAdd(dbias + m, accb[i][r]);
        }
    }
  } else {
    // bf16 output through LDS (see gemm256.hip): fp32 tile [128][128] with rows padded by 16 B, read back two whole
    // rows per wave instruction so that every global access is a contiguous 256-byte row segment
    bf16_t* C = (bf16_t*)Cv;
    constexpr int EP_STRIDE = 128 * 4 + 16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wr * 64 + i * 16 + l16, col = wc * 64 + j * 16 + 4 * g;
        *reinterpret_cast<f32x4*>(smem + row * EP_STRIDE + col * 4) = acc[i][j];
      }
    __syncthreads();
    const int half = lane >> 5, l32 = lane & 31;
    const int64_t ncol = n_blk + l32 * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
      bf16x4 b4 = *reinterpret_cast<const bf16x4*>(bias + (ncol < N ? ncol : N - 4));
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = (float)b4[r];
    }
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      const int row = wave * 32 + rr * 2 + half;
      const int64_t m = m_blk + row;
      if (m < M && ncol < N) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(smem + row * EP_STRIDE + l32 * 16);
        float v[4] = {c[0] + bv[0], c[1] + bv[1], c[2] + bv[2], c[3] + bv[3]};
        if (act == MMRCA_ACT_MUL) {
          bf16x4 h4 = *reinterpret_cast<const bf16x4*>(preact + m * ldc + ncol);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= (float)h4[r];
        } else if (act == MMRCA_ACT_GELU_SAVE_GRAD) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = erff(v[r] * 0.70710678118654752f);
            o[r] = (bf16_t)(0.5f * (1.0f + e) + v[r] * 0.3989422804014327f * __expf(-0.5f * v[r] * v[r]));
            v[r] = 0.5f * v[r] * (1.0f + e);
          }
          *reinterpret_cast<bf16x4*>(preact + m * ldc + ncol) = o;
        } else if (act == MMRCA_ACT_GELU_BWD) {
          bf16x4 h4 = *reinterpret_cast<const bf16x4*>(preact + m * ldc + ncol);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= gelu_grad_f((float)h4[r]);
        } else if (preact) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
          *reinterpret_cast<bf16x4*>(preact + m * ldc + ncol) = o;
        }
        if (act == MMRCA_ACT_GELU) {
          gelu_fast4(v);
        }
        if (addend) {
          bf16x4 a4 = *reinterpret_cast<const bf16x4*>(addend + m * ldc + ncol);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)a4[r];
        }
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
        *reinterpret_cast<bf16x4*>(C + m * ldc + ncol) = o;
      }
    }
  }
}

// ======================================================================================================
// BK = 32 variant: 128x128x32 tiles, 2 x 16 KiB of LDS per block (epilogue staged 32 rows at a time inside it), so FOUR
// blocks (16 waves) are resident per CU instead of two.  Calibration (tools/probes/mfma_peak.hip): one wave per SIMD can
// only drive the matrix pipe to 57 %, two reach 72-90 %; with loads, waits and barriers in the way, a SIMD that holds two
// waves is often down to one issuing wave.  Four waves per SIMD trade a barrier every 16 MFMAs for that thread-level cover.
//   ROWK image: 64-B rows, 16-B chunk index XOR ((4 - (row>>2)) & 3): the 16 lanes of every ds_read_b128 group land on 16
//   distinct slots of the 256-B bank row.   KROW image: the 256-B-row image of the BK=64 kernel with 32 k-rows.
// ======================================================================================================
#define TILE32_BYTES (128 * 32 * 2)

// BatchNorm moments of the stored output tile in the epilogue of the plain bf16 kernels (mmrca_gemm_bnstats: the 1x1 convolutions
// of the conv backbones): per output tile row block `tm` and column, s1 = sum (c - shift), s2 = sum (c - shift)^2 over the tile's
// valid rows of the ROUNDED outputs, written (not added) to s1 / s2 [tiles_m, N]; mmrca_bn_finish_sums merges the row blocks.
// shift = the layer's running mean keeps s2 - s1^2 / n free of cancellation.
struct BnStat { const float* shift; float* s1; float* s2; };

// EDGE (round 5): the last 32-deep step of a contraction that is a multiple of 8 but not of 32 (EfficientNetV2-M's channel counts:
// 80, 176, 304 -- the reference's default image model put 35 + 33 of its 1x1 convolutions on the general kernel at ~65 TFLOP/s for
// that).  LDS-DMA cannot write zeros, so the chunks past `kend` are FETCHED as zeros on the B side (one 16-byte zero chunk in
// global memory) and as a duplicate of a valid chunk on the A side: finite x 0 contributes nothing.  0: no edge, 1: A, 2: B.
template <bool KROW, int EDGE = 0>
__device__ __forceinline__ void stage_tile32(const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t rows_total,
                                             int64_t k0, char* lds_tile, int wave, int lane, int64_t kend = 0) {
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    const int i = wave * 2 + ii;                  // 8 wave-instructions of 1 KiB per operand tile
    const bf16_t* src;
    if (!KROW) {
      const int r = 16 * i + (lane >> 2);
      const int c = (lane & 3) ^ ((4 - (r >> 2)) & 3);
      int64_t gr = row0 + r;
      if (gr > rows_total - 1) gr = rows_total - 1;
      int64_t kc = k0 + c * 8;
      if (EDGE == 1 && kc >= kend) kc = kend - 8;
      src = base + gr * ld + kc;
      if (EDGE == 2 && kc >= kend) src = reinterpret_cast<const bf16_t*>(mmrca_zero_chunk);
    } else {
      const int kr = 4 * i + (lane >> 4);
      const int chp = lane & 15;
      const int c = ((((chp >> 1) ^ krow_f(kr))) << 1) | (chp & 1);
      int64_t col = row0 + c * 8;
      if (col > rows_total - 8) col = rows_total - 8;       // ragged edge (rows_total % 8 == 0): read a valid chunk, results are never stored
      int64_t kk = k0 + kr;
      if (EDGE == 1 && kk >= kend) kk = kend - 1;
      src = base + kk * ld + col;
      if (EDGE == 2 && kk >= kend) src = reinterpret_cast<const bf16_t*>(mmrca_zero_chunk);
    }
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(lds_tile + i * 1024), 16, 0, 0);
  }
}

template <bool KROW>
__device__ __forceinline__ bf16x8 load_frag32(const char* lds_tile, int rb, int lane) {
  if (!KROW) {
    const int r = rb + (lane & 15);
    const int ch = lane >> 4;
    return *reinterpret_cast<const bf16x8*>(lds_tile + r * 64 + ((ch ^ ((4 - (r >> 2)) & 3)) << 4));
  } else {
    return load_frag<true>(lds_tile, rb, 0, lane);
  }
}

template <bool KROW>
__device__ __forceinline__ void frag_addr32(const char* lds_tile, int rb, int lane, unsigned& a0, unsigned& a1) {
  if (!KROW) {
    const int r = rb + (lane & 15);
    const int ch = lane >> 4;
    a0 = a1 = (unsigned)(uintptr_t)(lds_void*)lds_tile + r * 64 + ((ch ^ ((4 - (r >> 2)) & 3)) << 4);
  } else {
    frag_addr<true>(lds_tile, rb, 0, lane, a0, a1);
  }
}

template <bool A_KROW, bool B_KROW, bool ATOMIC_F32, bool FUSE_DB, bool BNS = false>
__global__ void __launch_bounds__(256, 4)
gemm_mfma_k32(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, void* __restrict__ Cv, const bf16_t* __restrict__ bias,
              const bf16_t* __restrict__ addend, bf16_t* __restrict__ preact, int64_t M, int64_t N, int64_t K,
              int64_t lda, int64_t ldb, int64_t ldc, int act, int tiles_m, int tiles_n, int64_t ksplit_len, float* __restrict__ dbias,
              const BnStat bst = BnStat{nullptr, nullptr, nullptr}) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][A tile 8 KiB | B tile 8 KiB]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int nwg = tiles_m * tiles_n;
  const int orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int GROUP = 8;
  const int group = wgid / (GROUP * tiles_n);
  const int first_m = group * GROUP;
  const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int tm = first_m + (wgid % (GROUP * tiles_n)) % gsize;
  const int tn = (wgid % (GROUP * tiles_n)) / gsize;
  const int64_t m_blk = (int64_t)tm * GBM, n_blk = (int64_t)tn * GBN;
  const int64_t kbeg = (int64_t)blockIdx.y * ksplit_len;
  int64_t kend = kbeg + ksplit_len; if (kend > K) kend = K;
  const int nt = (int)((kend - kbeg + 31) / 32);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient riding on the weight gradient (FUSE_DB): row sums of the A operand (= column sums of dY) from one extra
  // MFMA per A fragment against an all-ones B fragment; the 32-deep steps are dealt round robin to the 2 * tiles_n
  // (tile column, wave column) pairs that hold the same A fragments, so every block does 1 / (2 tiles_n) of it
  constexpr bool do_bias = ATOMIC_F32 && FUSE_DB;
  const int kstep0 = (int)(kbeg / 32);
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
  // a contraction range that ends off a 32-deep step (K % 32 != 0, K % 8 == 0; forward / input gradient only): its last step is
  // staged with the EDGE forms (B chunks past kend read as zeros)
  const bool kedge = !ATOMIC_F32 && ((kend - kbeg) & 31) != 0;
  if (nt > 0) {
    if (kedge && nt == 1) {
      stage_tile32<A_KROW, 1>(A, lda, m_blk, M, kbeg, smem, wave, lane, kend);
      stage_tile32<B_KROW, 2>(B, ldb, n_blk, N, kbeg, smem + TILE32_BYTES, wave, lane, kend);
    } else {
      stage_tile32<A_KROW>(A, lda, m_blk, M, kbeg, smem, wave, lane);
      stage_tile32<B_KROW>(B, ldb, n_blk, N, kbeg, smem + TILE32_BYTES, wave, lane);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * 2 * TILE32_BYTES;
    char* nxt = smem + ((t + 1) & 1) * 2 * TILE32_BYTES;
    if (t + 1 < nt) {
      if (kedge && t + 2 == nt) {
        stage_tile32<A_KROW, 1>(A, lda, m_blk, M, kbeg + (int64_t)(t + 1) * 32, nxt, wave, lane, kend);
        stage_tile32<B_KROW, 2>(B, ldb, n_blk, N, kbeg + (int64_t)(t + 1) * 32, nxt + TILE32_BYTES, wave, lane, kend);
      } else {
        stage_tile32<A_KROW>(A, lda, m_blk, M, kbeg + (int64_t)(t + 1) * 32, nxt, wave, lane);
        stage_tile32<B_KROW>(B, ldb, n_blk, N, kbeg + (int64_t)(t + 1) * 32, nxt + TILE32_BYTES, wave, lane);
      }
    }
    // asm fragment reads (lds_asm.h): the prefetch issued above stays in flight under them and under the 16 MFMAs
    Frag<A_KROW> afr[4];
    Frag<B_KROW> bfrr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { unsigned a0, a1; frag_addr32<A_KROW>(cur, wr * 64 + i * 16, lane, a0, a1); read_frag_rt<A_KROW>(afr[i], a0, a1); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { unsigned a0, a1; frag_addr32<B_KROW>(cur + TILE32_BYTES, wc * 64 + j * 16, lane, a0, a1); read_frag_rt<B_KROW>(bfrr[j], a0, a1); }
    lgkm0(afr); lgkm0(bfrr);
    bf16x8 af[4], bfr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { af[i] = frag_val(afr[i]); bfr[i] = frag_val(bfrr[i]); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (ATOMIC_F32) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        else            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    if (do_bias && ((kstep0 + t) % (2 * tiles_n)) == 2 * tn + wc) {
#pragma unroll
      for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);       // the MFMAs stay ABOVE the wait
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  const int g = lane >> 4, l16 = lane & 15;
  if (ATOMIC_F32) {
    float* C = (float*)Cv;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t n = n_blk + wc * 64 + j * 16 + l16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t m = m_blk + wr * 64 + i * 16 + 4 * g + r;
          if (m < M && n < N) atomicAdd(C + m * ldc + n, acc[i][j][r]);
        }
      }
    if (do_bias && l16 == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t m = m_blk + wr * 64 + i * 16 + 4 * g + r;
          if (m < M) atomicAdd(dbias + m, accb[i][r]);
        }
    }
  } else {
    bf16_t* C = (bf16_t*)Cv;
    constexpr int EP_STRIDE = 128 * 4 + 16;
    const int half = lane >> 5, l32 = lane & 31;
    const int64_t ncol = n_blk + l32 * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
      bf16x4 b4 = *reinterpret_cast<const bf16x4*>(bias + (ncol < N ? ncol : N - 4));
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = (float)b4[r];
    }
    [[maybe_unused]] float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f}, stsh[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BNS) {
      if (bst.shift && ncol < N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) stsh[r] = bst.shift[ncol + r];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(smem + (wr * 16 + l16) * EP_STRIDE + (wc * 64 + j * 16 + 4 * g) * 4) = acc[i][j];
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int lrow = wave * 8 + rr * 2 + half;
        const int64_t m = m_blk + (lrow >> 4) * 64 + i * 16 + (lrow & 15);
        if (m < M && ncol < N) {
          const f32x4 c = *reinterpret_cast<const f32x4*>(smem + lrow * EP_STRIDE + l32 * 16);
          float v[4] = {c[0] + bv[0], c[1] + bv[1], c[2] + bv[2], c[3] + bv[3]};
          if (act == MMRCA_ACT_MUL) {
            bf16x4 h4 = *reinterpret_cast<const bf16x4*>(preact + m * ldc + ncol);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= (float)h4[r];
          } else if (act == MMRCA_ACT_GELU_SAVE_GRAD) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float e = erff(v[r] * 0.70710678118654752f);
              o[r] = (bf16_t)(0.5f * (1.0f + e) + v[r] * 0.3989422804014327f * __expf(-0.5f * v[r] * v[r]));
              v[r] = 0.5f * v[r] * (1.0f + e);
            }
            *reinterpret_cast<bf16x4*>(preact + m * ldc + ncol) = o;
          } else if (act == MMRCA_ACT_GELU_BWD) {
            bf16x4 h4 = *reinterpret_cast<const bf16x4*>(preact + m * ldc + ncol);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= gelu_grad_f((float)h4[r]);
          } else if (preact) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
            *reinterpret_cast<bf16x4*>(preact + m * ldc + ncol) = o;
          }
          if (act == MMRCA_ACT_GELU) {
            gelu_fast4(v);
          }
          if (addend) {
            bf16x4 a4 = *reinterpret_cast<const bf16x4*>(addend + m * ldc + ncol);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)a4[r];
          }
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
          *reinterpret_cast<bf16x4*>(C + m * ldc + ncol) = o;
          if constexpr (BNS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = (float)o[r] - stsh[r]; st1[r] += d; st2[r] = fmaf(d, d, st2[r]); }
          }
        }
      }
      __syncthreads();
    }
    if constexpr (BNS) {   // 8 row groups (4 waves x 2 half-waves) -> LDS -> one plain store per column and moment
      float* red = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int r = 0; r < 4; ++r) { red[(wave * 2 + half) * 128 + l32 * 4 + r] = st1[r]; red[1024 + (wave * 2 + half) * 128 + l32 * 4 + r] = st2[r]; }
      __syncthreads();
      const int col = threadIdx.x & 127, which = threadIdx.x >> 7;
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += red[which * 1024 + q * 128 + col];
      if (n_blk + col < N) (which ? bst.s2 : bst.s1)[(int64_t)tm * N + n_blk + col] = t;
    }
  }
}

template <bool AK, bool BK2, bool AT, bool DB = false, bool BNS = false>
static void launch_mfma32(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact,
                          int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int act, int tiles_m,
                          int tiles_n, int ksplits, int64_t ksplit_len, hipStream_t st, float* dbias = nullptr,
                          BnStat bst = BnStat{nullptr, nullptr, nullptr}) {
  hipLaunchKernelGGL((gemm_mfma_k32<AK, BK2, AT, DB, BNS>), dim3(tiles_m * tiles_n, ksplits), dim3(256), 4 * TILE32_BYTES, st,
                     (const bf16_t*)A, (const bf16_t*)B, C, (const bf16_t*)bias, (const bf16_t*)addend, (bf16_t*)preact,
                     M, N, K, lda, ldb, ldc, act, tiles_m, tiles_n, ksplit_len, dbias, bst);
}

// single-stage 128x128x64 variant (32 KiB LDS, four to five blocks per CU): full 128-byte lines for ROWK operands
// X3 ("bf16x3", the <= 1e-3 fast mode, see gemm_x3.hip): both operands are fp32 values stored as two bf16 planes, x = hi + lo
// (hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits).  The K loop runs three times over the same output tile -- (A_hi, B_hi),
// (A_lo, B_hi), (A_hi, B_lo); lo x lo is below fp32 resolution -- into the same fp32 accumulators, and the epilogue's
// bias / side operands / outputs are fp32 (biasv / addendv / preactv / Cv point at floats), or, with C_lo given, the output is
// written as two bf16 planes (Cv = hi plane) for a consumer that is another X3 GEMM.
// F4 (X3, all three plane pairs, round 6): the FUSED form, as in gemm_p256_k<..., F4>.  The plain X3 loop above walks the contraction three
// times -- (A_hi, B_hi), (A_lo, B_hi), (A_hi, B_lo) -- staging A_hi and B_hi twice: 6 x 16 KiB of LDS-DMA and three barrier pairs per 64
// columns of K.  Here a K step is 32 columns of ALL FOUR planes (4 x 8 KiB = the same 32 KiB, the 32-deep images of gemm_mfma_k32) and the
// three products are formed from them: 4 x 16 KiB staged and two barrier pairs per 64 columns for the same 96 MFMAs per wave.  This is the
// kernel of every bf16x3 product the persistent kernel does not take: the text encoder's ragged-M shapes, the BLIP-2 towers' N = 1,408 /
// 4,224 / 6,144 products (configs[4] in its compliant mode).
template <bool A_KROW, bool B_KROW, bool ATOMIC_F32, int WPE, bool X3 = false, bool BNS = false, bool F4 = false>
__global__ void __launch_bounds__(256, WPE)
gemm_mfma_k1s(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, void* __restrict__ Cv, const bf16_t* __restrict__ bias,
              const bf16_t* __restrict__ addend, bf16_t* __restrict__ preact, int64_t M, int64_t N, int64_t K,
              int64_t lda, int64_t ldb, int64_t ldc, int act, int tiles_m, int tiles_n, int64_t ksplit_len, float* __restrict__ colsum,
              const bf16_t* __restrict__ A_lo = nullptr, const bf16_t* __restrict__ B_lo = nullptr, bf16_t* __restrict__ C_lo = nullptr,
              int nseg = 3, const BnStat bst = BnStat{nullptr, nullptr, nullptr}, int pre16 = 0) {
  // pre16 (X3, MMRCA_ACT_GELU_SAVE_GRAD_BF16): gelu' goes to `preact` as bf16 (what a bf16 backward reads) instead of fp32
  extern __shared__ __attribute__((aligned(16))) char smem[];   // ONE buffer: [A tile 16 KiB | B tile 16 KiB]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int nwg = tiles_m * tiles_n;
  const int orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int GROUP = 8;
  const int group = wgid / (GROUP * tiles_n);
  const int first_m = group * GROUP;
  const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int tm = first_m + (wgid % (GROUP * tiles_n)) % gsize;
  const int tn = (wgid % (GROUP * tiles_n)) / gsize;
  const int64_t m_blk = (int64_t)tm * GBM, n_blk = (int64_t)tn * GBN;
  const int64_t kbeg = (int64_t)blockIdx.y * ksplit_len;
  int64_t kend = kbeg + ksplit_len; if (kend > K) kend = K;
  const int nt = (int)((kend - kbeg + GBK - 1) / GBK);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (F4) {
    static_assert(X3 && !ATOMIC_F32, "the fused four-plane form is a bf16x3 product with an epilogue");
    const int nt32 = (int)((kend - kbeg) / 32);
    for (int t = 0; t < nt32; ++t) {
      const int64_t k0 = kbeg + (int64_t)t * 32;
      stage_tile32<A_KROW>(A, lda, m_blk, M, k0, smem, wave, lane);
      stage_tile32<A_KROW>(A_lo, lda, m_blk, M, k0, smem + TILE32_BYTES, wave, lane);
      stage_tile32<B_KROW>(B, ldb, n_blk, N, k0, smem + 2 * TILE32_BYTES, wave, lane);
      stage_tile32<B_KROW>(B_lo, ldb, n_blk, N, k0, smem + 3 * TILE32_BYTES, wave, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      bf16x8 ah[4], al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ah[i] = load_frag32<A_KROW>(smem, wr * 64 + i * 16, lane);
        al[i] = load_frag32<A_KROW>(smem + TILE32_BYTES, wr * 64 + i * 16, lane);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 bh = load_frag32<B_KROW>(smem + 2 * TILE32_BYTES, wc * 64 + j * 16, lane);
        const bf16x8 bl = load_frag32<B_KROW>(smem + 3 * TILE32_BYTES, wc * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {       // the small terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[i], acc[i][j], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  } else
  for (int tt = 0; tt < (X3 ? nseg : 1) * nt; ++tt) {      // bf16x3: nseg of the plane pairs (A,B), (A_lo,B), (A,B_lo)
    // single LDS stage: load -> wait -> barrier -> 32 MFMAs -> barrier; the load latency of this block is covered by the
    // other three or four blocks resident on the CU (32 KiB of LDS each) instead of by software prefetch
    int t = tt;
    const bf16_t* Ap = A;
    const bf16_t* Bp = B;
    if constexpr (X3) {
      const int seg = tt >= 2 * nt ? 2 : (tt >= nt ? 1 : 0);
      t = tt - seg * nt;
      if (seg == 1) Ap = A_lo;
      if (seg == 2) Bp = B_lo;
    }
    if (!ATOMIC_F32 && !X3 && t == nt - 1 && ((kend - kbeg) & (GBK - 1)) != 0) {      // the contraction ends inside this step (wave-uniform)
      stage_tile<A_KROW, 1>(Ap, lda, m_blk, M, kbeg + (int64_t)t * GBK, smem, wave, lane, kend);
      stage_tile<B_KROW, 2>(Bp, ldb, n_blk, N, kbeg + (int64_t)t * GBK, smem + TILE_BYTES, wave, lane, kend);
    } else {
      stage_tile<A_KROW>(Ap, lda, m_blk, M, kbeg + (int64_t)t * GBK, smem, wave, lane);
      stage_tile<B_KROW>(Bp, ldb, n_blk, N, kbeg + (int64_t)t * GBK, smem + TILE_BYTES, wave, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = load_frag<A_KROW>(smem, wr * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = load_frag<B_KROW>(smem + TILE_BYTES, wc * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (ATOMIC_F32) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
          else            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  const int g = lane >> 4, l16 = lane & 15;
  if (ATOMIC_F32) {
    float* C = (float*)Cv;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t n = n_blk + wc * 64 + j * 16 + l16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t m = m_blk + wr * 64 + i * 16 + 4 * g + r;
          if (m < M && n < N) atomicAdd(C + m * ldc + n, acc[i][j][r]);
        }
      }
  } else if constexpr (X3) {
    // fp32 epilogue of the bf16x3 mode: same LDS staging, 16-byte fp32 accesses.  The side operand (fp32) of a pass is
    // requested before the pass's staging barrier, so its latency overlaps the LDS round trip (the other resident blocks
    // cover the rest).
    const float* biasf = reinterpret_cast<const float*>(bias);
    const float* addf = reinterpret_cast<const float*>(addend);
    float* pref = reinterpret_cast<float*>(preact);
    constexpr int EP_STRIDE = 128 * 4 + 16;
    const int half = lane >> 5, l32 = lane & 31;
    const int64_t ncol = n_blk + l32 * 4;
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (biasf) bv = *reinterpret_cast<const f32x4*>(biasf + ncol);
    const float* side = act == MMRCA_ACT_MUL ? pref : addf;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 sd[4];
      if (side) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int lrow = wave * 8 + rr * 2 + half;
          int64_t m = m_blk + (lrow >> 4) * 64 + i * 16 + (lrow & 15);
          if (m > M - 1) m = M - 1;
          sd[rr] = *reinterpret_cast<const f32x4*>(side + m * ldc + ncol);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(smem + (wr * 16 + l16) * EP_STRIDE + (wc * 64 + j * 16 + 4 * g) * 4) = acc[i][j];
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int lrow = wave * 8 + rr * 2 + half;
        const int64_t m = m_blk + (lrow >> 4) * 64 + i * 16 + (lrow & 15);
        if (m < M && ncol < N) {
          const f32x4 c = *reinterpret_cast<const f32x4*>(smem + lrow * EP_STRIDE + l32 * 16);
          float v[4] = {c[0] + bv[0], c[1] + bv[1], c[2] + bv[2], c[3] + bv[3]};
          if (act == MMRCA_ACT_MUL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= sd[rr][r];
          } else if (act == MMRCA_ACT_GELU_SAVE_GRAD) {
            float dg[4];
            gelu_and_grad_fast4(v, dg);
            if (pre16) {
              bf16x4 o16;
#pragma unroll
              for (int r = 0; r < 4; ++r) o16[r] = (bf16_t)dg[r];
              *reinterpret_cast<bf16x4*>(preact + m * ldc + ncol) = o16;
            } else {
              *reinterpret_cast<f32x4*>(pref + m * ldc + ncol) = (f32x4){dg[0], dg[1], dg[2], dg[3]};
            }
          } else if (act == MMRCA_ACT_GELU_BWD) {
            const f32x4 h4 = *reinterpret_cast<const f32x4*>(pref + m * ldc + ncol);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= gelu_grad_f(h4[r]);
          } else if (pref) {
            *reinterpret_cast<f32x4*>(pref + m * ldc + ncol) = (f32x4){v[0], v[1], v[2], v[3]};
          }
          if (act == MMRCA_ACT_GELU) gelu_fast4(v);
          if (addf) {
            if (act == MMRCA_ACT_MUL) {
              const f32x4 a4 = *reinterpret_cast<const f32x4*>(addf + m * ldc + ncol);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += a4[r];
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += sd[rr][r];
            }
          }
          if (C_lo) {
            bf16x4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) { hi[r] = (bf16_t)v[r]; lo[r] = (bf16_t)(v[r] - (float)hi[r]); cs[r] += (float)hi[r] + (float)lo[r]; }
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(Cv) + m * ldc + ncol) = hi;
            *reinterpret_cast<bf16x4*>(C_lo + m * ldc + ncol) = lo;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[r] += v[r];
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Cv) + m * ldc + ncol) = (f32x4){v[0], v[1], v[2], v[3]};
          }
        }
      }
      __syncthreads();
    }
    if (colsum) {
      float* red = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 2 + half) * 128 + l32 * 4 + r] = cs[r];
      __syncthreads();
      if (threadIdx.x < 128) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q * 128 + threadIdx.x];
        if (n_blk + threadIdx.x < N) atomicAdd(colsum + n_blk + threadIdx.x, t);
      }
    }
  } else {
    bf16_t* C = (bf16_t*)Cv;
    constexpr int EP_STRIDE = 128 * 4 + 16;
    const int half = lane >> 5, l32 = lane & 31;
    const int64_t ncol = n_blk + l32 * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
      bf16x4 b4 = *reinterpret_cast<const bf16x4*>(bias + (ncol < N ? ncol : N - 4));
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = (float)b4[r];
    }
    // the side operand of the epilogue -- the residual addend, or the saved gelu'(h) that ACT_MUL multiplies by -- is
    // fetched for the whole tile up front (16 x 8 B per lane, in the registers the operand fragments just vacated): in
    // the model it was written many kernels ago and is long gone from L2/MALL, and loading it row by row inside the
    // staged loop exposed one HBM round trip per pass
    bf16x4 add4[4][4];
    const bf16_t* side = act == MMRCA_ACT_MUL ? preact : addend;
    if (side) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int lrow = wave * 8 + rr * 2 + half;
          int64_t m = m_blk + (lrow >> 4) * 64 + i * 16 + (lrow & 15);
          if (m > M - 1) m = M - 1;
          add4[i][rr] = *reinterpret_cast<const bf16x4*>(side + m * ldc + ncol);
        }
    }
    float cs[4] = {0.f, 0.f, 0.f, 0.f};      // column sums of what this lane stores (colsum != nullptr)
    [[maybe_unused]] float st2[4] = {0.f, 0.f, 0.f, 0.f}, stsh[4] = {0.f, 0.f, 0.f, 0.f};     // BNS: cs = sum d, st2 = sum d^2, d = c - shift
    if constexpr (BNS) {
      if (bst.shift && ncol < N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) stsh[r] = bst.shift[ncol + r];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(smem + (wr * 16 + l16) * EP_STRIDE + (wc * 64 + j * 16 + 4 * g) * 4) = acc[i][j];
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int lrow = wave * 8 + rr * 2 + half;
        const int64_t m = m_blk + (lrow >> 4) * 64 + i * 16 + (lrow & 15);
        if (m < M && ncol < N) {
          const f32x4 c = *reinterpret_cast<const f32x4*>(smem + lrow * EP_STRIDE + l32 * 16);
          float v[4] = {c[0] + bv[0], c[1] + bv[1], c[2] + bv[2], c[3] + bv[3]};
          if (act == MMRCA_ACT_MUL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= (float)add4[i][rr][r];
          } else if (act == MMRCA_ACT_GELU_SAVE_GRAD) {
            bf16x4 o;
            float dg[4];
            gelu_and_grad_fast4(v, dg);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)dg[r];
            *reinterpret_cast<bf16x4*>(preact + m * ldc + ncol) = o;
          } else if (act == MMRCA_ACT_GELU_BWD) {
            bf16x4 h4 = *reinterpret_cast<const bf16x4*>(preact + m * ldc + ncol);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= gelu_grad_f((float)h4[r]);
          } else if (preact) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
            *reinterpret_cast<bf16x4*>(preact + m * ldc + ncol) = o;
          }
          if (act == MMRCA_ACT_GELU) {
            gelu_fast4(v);
          }
          if (addend) {
            if (act == MMRCA_ACT_MUL) {       // (both side operands at once: the addend is read in place)
              bf16x4 a4 = *reinterpret_cast<const bf16x4*>(addend + m * ldc + ncol);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += (float)a4[r];
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += (float)add4[i][rr][r];
            }
          }
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = (bf16_t)v[r];
            if constexpr (BNS) { const float d = (float)o[r] - stsh[r]; cs[r] += d; st2[r] = fmaf(d, d, st2[r]); }
            else cs[r] += (float)o[r];
          }
          *reinterpret_cast<bf16x4*>(C + m * ldc + ncol) = o;
        }
      }
      __syncthreads();
    }
    if constexpr (BNS) {   // 8 row groups (4 waves x 2 half-waves) -> LDS -> one plain store per column and moment
      float* red = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int r = 0; r < 4; ++r) { red[(wave * 2 + half) * 128 + l32 * 4 + r] = cs[r]; red[1024 + (wave * 2 + half) * 128 + l32 * 4 + r] = st2[r]; }
      __syncthreads();
      const int col = threadIdx.x & 127, which = threadIdx.x >> 7;
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += red[which * 1024 + q * 128 + col];
      if (n_blk + col < N) (which ? bst.s2 : bst.s1)[(int64_t)tm * N + n_blk + col] = t;
    } else if (colsum) {
      // column sums of the stored tile (the bias gradient of the layer below, when this GEMM is its input gradient):
      // 8 row groups (4 waves x 2 half-waves) -> LDS -> one atomic per column per block
      float* red = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 2 + half) * 128 + l32 * 4 + r] = cs[r];
      __syncthreads();
      if (threadIdx.x < 128) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q * 128 + threadIdx.x];
        if (n_blk + threadIdx.x < N) atomicAdd(colsum + n_blk + threadIdx.x, t);
      }
    }
  }
}

template <bool AK, bool BK2, bool AT, int WPE, bool X3 = false, bool BNS = false, bool F4 = false>
static void launch_mfma1s(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact,
                          int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int act, int tiles_m,
                          int tiles_n, int ksplits, int64_t ksplit_len, float* colsum, hipStream_t st,
                          const void* A_lo = nullptr, const void* B_lo = nullptr, void* C_lo = nullptr, int nseg = 3,
                          BnStat bst = BnStat{nullptr, nullptr, nullptr}, int pre16 = 0) {
  hipLaunchKernelGGL((gemm_mfma_k1s<AK, BK2, AT, WPE, X3, BNS, F4>), dim3(tiles_m * tiles_n, ksplits), dim3(256), 2 * TILE_BYTES, st,
                     (const bf16_t*)A, (const bf16_t*)B, C, (const bf16_t*)bias, (const bf16_t*)addend, (bf16_t*)preact,
                     M, N, K, lda, ldb, ldc, act, tiles_m, tiles_n, ksplit_len, colsum, (const bf16_t*)A_lo, (const bf16_t*)B_lo,
                     (bf16_t*)C_lo, nseg, bst, pre16);
}

// entry used by gemm_x3.hip: the single-stage 128x128 kernel in its bf16x3 form (any epilogue; accumulate mode = fp32 atomics)
int mmrca_gemm_k1s_x3(const void* A_hi, const void* A_lo, const void* B_hi, const void* B_lo, void* C, void* C_lo, const void* bias,
                      const void* addend, void* preact, float* colsum, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                      int64_t ldc, int a_layout, int b_layout, int act, int accum, hipStream_t st, int pre16) {
  const int tiles_m = (int)((M + GBM - 1) / GBM), tiles_n = (int)(N / GBN);
  int ksplits = 1;
  int64_t ksplit_len = K;
  if (accum) {
    const int64_t ksteps = K / GBK;
    int64_t want = 1024 / ((int64_t)tiles_m * tiles_n);
    if (want < 1) want = 1;
    if (want > ksteps / 4) want = ksteps / 4 > 0 ? ksteps / 4 : 1;
    const int64_t steps_per = (ksteps + want - 1) / want;
    ksplit_len = steps_per * GBK;
    ksplits = (int)((ksteps + steps_per - 1) / steps_per);
  }
  const bool ak = a_layout == MMRCA_KROW, bk = b_layout == MMRCA_KROW, at = accum != 0;
  // the fused four-plane form: all three plane pairs, an epilogue (no accumulate mode), K % 32 == 0 (MMRCA_X3_FUSED_K1S=0: the three-pass loop)
  static const int fused_k1s = getenv("MMRCA_X3_FUSED_K1S") ? atoi(getenv("MMRCA_X3_FUSED_K1S")) : 1;
  if (fused_k1s && A_lo && B_lo && !at && K % 32 == 0) {
#define L1SF(AK_, BK_) launch_mfma1s<AK_, BK_, false, 4, true, false, true>(A_hi, B_hi, C, bias, addend, preact, M, N, K, lda, ldb, ldc, act, tiles_m, tiles_n, 1, K, colsum, st, A_lo, B_lo, C_lo, 3, BnStat{nullptr, nullptr, nullptr}, pre16)
    if (!ak && !bk) L1SF(false, false);
    else if (!ak && bk) L1SF(false, true);
    else if (ak && !bk) L1SF(true, false);
    else L1SF(true, true);
#undef L1SF
    MMRCA_CHECK_LAUNCH("gemm_x3(k1s,fused)");
    return 0;
  }
#define L1SX(AK_, BK_, AT_) launch_mfma1s<AK_, BK_, AT_, 4, true>(A_hi, B_hi, C, bias, addend, preact, M, N, K, lda, ldb, ldc, act, tiles_m, tiles_n, ksplits, ksplit_len, colsum, st, A_lo, B_lo, C_lo, A_lo ? (B_lo ? 3 : 2) : 1, BnStat{nullptr, nullptr, nullptr}, pre16)
  if (!ak && !bk && !at) L1SX(false, false, false);
  else if (!ak && bk && !at) L1SX(false, true, false);
  else if (ak && !bk && !at) L1SX(true, false, false);
  else if (ak && bk && !at) L1SX(true, true, false);
  else if (!ak && !bk && at) L1SX(false, false, true);
  else if (!ak && bk && at) L1SX(false, true, true);
  else if (ak && !bk && at) L1SX(true, false, true);
  else L1SX(true, true, true);
#undef L1SX
  MMRCA_CHECK_LAUNCH("gemm_x3(k1s)");
  return 0;
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

bool mmrca_gemm256_ok(int64_t M, int64_t N, int64_t K, int a_layout, int act, bool has_addend, bool has_preact, bool has_colsum, bool has_bias);
int mmrca_gemm256(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact, int64_t M,
                  int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_layout, int b_layout, int act,
                  float* colsum, hipStream_t st);

int mmrca_gemm256_streamk_split(int64_t M, int64_t N, int64_t ksteps, void* stream, bool x3);
extern int g_mmrca_dbg;
static const int g_mmrca_auto256_side = getenv("MMRCA_AUTO256_SIDE") ? atoi(getenv("MMRCA_AUTO256_SIDE")) : 0;
static const bool g_mmrca_auto256_gelu = getenv("MMRCA_AUTO256_GELU") ? atoi(getenv("MMRCA_AUTO256_GELU")) != 0 : true;
static const int g_mmrca_auto256_tail = getenv("MMRCA_AUTO256_TAIL") ? atoi(getenv("MMRCA_AUTO256_TAIL")) : 2;
static const int g_mmrca_auto256_tail_pct = getenv("MMRCA_AUTO256_TAIL_PCT") ? atoi(getenv("MMRCA_AUTO256_TAIL_PCT")) : 60;

template <bool AK, bool BK2, bool AT, bool DB>
static void launch_mfma(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact,
                        int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int act, int tiles_m,
                        int tiles_n, int ksplits, int64_t ksplit_len, float* dbias, hipStream_t st) {
  MMRCA_MAX_LDS(LDS128_BYTES, gemm_mfma_k<AK, BK2, AT, DB>);
  hipLaunchKernelGGL((gemm_mfma_k<AK, BK2, AT, DB>), dim3(tiles_m * tiles_n, ksplits), dim3(256), LDS128_BYTES, st,
                     (const bf16_t*)A, (const bf16_t*)B, C, (const bf16_t*)bias, (const bf16_t*)addend, (bf16_t*)preact,
                     M, N, K, lda, ldb, ldc, act, tiles_m, tiles_n, ksplit_len, dbias);
}

// colsum_fused: honoured by the single-stage kernel only; *fused_done tells the caller whether it was
// rows_readable: the caller vouches (mmrca_gemm_rows) that A and the side operand have round_up(M, 256) readable rows
static int gemm_dispatch(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact,
                         int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                         int a_layout, int b_layout, int act, int out_f32_accum, int dtype, int impl, void* stream,
                         float* colsum_fused, bool* fused_done, bool rows_readable = false) {
  MMRCA_REQUIRE(A && B && C, "gemm: null operand");
  MMRCA_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  MMRCA_REQUIRE((a_layout == MMRCA_ROWK || a_layout == MMRCA_KROW) && (b_layout == MMRCA_ROWK || b_layout == MMRCA_KROW), "gemm: bad layout");
  MMRCA_REQUIRE(act >= MMRCA_ACT_NONE && act <= MMRCA_ACT_MUL, "gemm: bad activation");
  MMRCA_REQUIRE(act < MMRCA_ACT_GELU_BWD || preact, "gemm: this activation needs the `preact` buffer");
  MMRCA_REQUIRE(lda >= (a_layout == MMRCA_ROWK ? K : M) && ldb >= (b_layout == MMRCA_ROWK ? K : N) && ldc >= N, "gemm: leading dimension too small");
  MMRCA_REQUIRE(!(out_f32_accum && (addend || preact || act != MMRCA_ACT_NONE)), "gemm: accumulate mode takes no epilogue");
  MMRCA_REQUIRE(!(out_f32_accum && bias && a_layout != MMRCA_KROW), "gemm: the fused bias gradient needs A in KROW layout");
  hipStream_t st = (hipStream_t)stream;

  bool ok_mfma = (dtype == MMRCA_BF16) && (N % GBN == 0) && (K % GBK == 0) && (lda % 8 == 0) && (ldb % 8 == 0) &&
                 (ldc % 4 == 0) && aligned16(A) && aligned16(B) && aligned16(C) && (!bias || aligned16(bias)) &&
                 (!addend || aligned16(addend)) && (!preact || aligned16(preact)) &&
                 (a_layout == MMRCA_ROWK || M % GBM == 0);
  // Ragged shapes on the 128x128 MFMA kernels (the conv backbones' channel counts: 192, 224, 1344, ... -- multiples of 8, not of
  // 128; contraction a multiple of 32, not of 64): edge tiles stage a clamped (duplicate) chunk and never store it, the 32-deep
  // kernel takes the odd half K step.  Plain epilogue only (bias allowed): side operands are read per tile and would run
  // past the last row.  These shapes used to fall to the general kernel at a fifth of the rate.
  static const bool ragged_on = !(getenv("MMRCA_GEMM_RAGGED") && atoi(getenv("MMRCA_GEMM_RAGGED")) == 0);
  const bool strict_mfma = ok_mfma;
  // (round 5: a contraction that is a multiple of 8 but not of 32 -- EfficientNetV2-M's 80 / 176 / 304 channels -- is taken by the
  // 32-deep kernel's edge step; forward / input gradient only, MMRCA_GEMM_KEDGE=0 restores the general kernel for them)
  static const bool kedge_on = !(getenv("MMRCA_GEMM_KEDGE") && atoi(getenv("MMRCA_GEMM_KEDGE")) == 0);
  const bool k_ok = K % 32 == 0 || (kedge_on && !out_f32_accum && K % 8 == 0 && K >= 32);
  if (!ok_mfma && ragged_on && dtype == MMRCA_BF16 && N % 8 == 0 && N >= 8 && k_ok && (a_layout == MMRCA_ROWK || (M % 8 == 0 && M >= 8)) &&
      lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && aligned16(A) && aligned16(B) && aligned16(C) && (!bias || (aligned16(bias) && !out_f32_accum)) &&
      !addend && !preact && act == MMRCA_ACT_NONE && !colsum_fused && (out_f32_accum ? K % 64 == 0 : true) && M >= (out_f32_accum ? 16 : 64) &&
      (impl == MMRCA_GEMM_AUTO || impl == MMRCA_GEMM_MFMA_BK32 || impl == MMRCA_GEMM_MFMA_1STAGE))
    ok_mfma = true;
  const bool ragged = ok_mfma && !strict_mfma;
  const bool ok256 = strict_mfma && !out_f32_accum && mmrca_gemm256_ok(M, N, K, a_layout, act, addend != nullptr, preact != nullptr, colsum_fused != nullptr, bias != nullptr) &&
                     M * lda * 2 < (1ll << 32) && (b_layout == MMRCA_KROW ? K * ldb : N * ldb) * 2 < (1ll << 32) && M * ldc * 2 < (1ll << 32);
  if (impl == MMRCA_GEMM_MFMA256 && !ok256)
    return mmrca_fail(-3, "gemm: shape M=%lld N=%lld K=%lld does not qualify for the 256x256 MFMA kernel", (long long)M, (long long)N, (long long)K);
  // The persistent kernel streams whole 256-row tiles of A (and fetches whole tiles of a side operand): with a ragged M it reads
  // rows M .. round_up(M, 256) - 1.  An explicit request must therefore either come with whole tiles or through mmrca_gemm_rows,
  // where the caller states how many rows are readable (AUTO only picks the kernel when M % 256 == 0).
  if (impl == MMRCA_GEMM_MFMA256 && M % 256 != 0 && !rows_readable)
    return mmrca_fail(-4, "gemm: impl=MFMA256 with M=%lld (not a multiple of 256) reads rows up to %lld of A and of the side operand: "
                      "call mmrca_gemm_rows and state the readable row counts", (long long)M, (long long)((M + 255) / 256 * 256));
  // AUTO (tools/gemm_bench.py, interleaved rounds, round 2): the persistent 256x256 kernel (operand stream in flight across
  // barriers and tile boundaries, barrier-free epilogue) wins on every bias-only shape once there is a full round of tiles:
  // K = 768: 870-1,120 vs 830-910 TFLOP/s for the 128x128 single-stage kernel, K >= 2304: 1,010-1,070 vs 910-920.
  // Epilogues with a side operand (residual addend, gelu' factor) are not built there and stay on the 128x128 kernel.
  const bool side256 = addend != nullptr || act == MMRCA_ACT_MUL;     // epilogues with a side operand: MMRCA_AUTO256_SIDE bit 0 = addend at K >= 1536, bit 1 = MUL, bit 2 = addend at any K
  const bool side_ok = !side256 || (addend != nullptr && act == MMRCA_ACT_NONE && (((g_mmrca_auto256_side & 1) && K >= 1536) || (g_mmrca_auto256_side & 4))) ||
                       (act == MMRCA_ACT_MUL && (g_mmrca_auto256_side & 2));
  const bool auto256 = impl == MMRCA_GEMM_AUTO && ok256 && M % 256 == 0 && K >= 768 && (M / 256) * (N / 256) >= 256 && side_ok &&
                       (act != MMRCA_ACT_GELU_SAVE_GRAD || g_mmrca_auto256_gelu);
  if (ok256 && (impl == MMRCA_GEMM_MFMA256 || auto256)) {
    // Tile quantisation: the persistent kernel runs whole rounds of one 256x256 tile per CU.  N = 768 at M = 50,432 makes
    // 591 tiles = 2.31 rounds on 256 CUs, i.e. the chip idles for 0.69 of a round.  AUTO gives the persistent kernel the
    // leading row blocks that make whole rounds and hands the remaining rows to the 128x128 kernel (MMRCA_AUTO256_TAIL:
    // 0 = off, 1 = single-stage kernel, 2 = two-stage kernel for the tail (default); only when the partial round is 25-60 % of a
    // round).  Isolated: FFN2 forward 223 -> 209 us, QKV input gradient 179 -> 160, FFN1 input gradient 233 -> 218.
    // Round 6: with a stream-K workspace registered for the stream (mmrca_gemm_streamk_workspace) the partial round stays INSIDE
    // the persistent launch -- the leftover tiles' K loops are cut into ranges across all CUs (gemm256.hip) -- and the split below
    // is not taken.
    const bool sk_tail = mmrca_gemm256_streamk_split(M, N, K / 64, stream, false) >= 2;
    if (auto256 && !sk_tail && g_mmrca_auto256_tail > 0 && !colsum_fused && a_layout == MMRCA_ROWK) {
      int ncu = 256, devi = 0;
      (void)hipGetDevice(&devi);
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, devi) != hipSuccess || ncu < 8) ncu = 256;
      ncu &= ~7;
      const int64_t tm = M / 256, tn = N / 256, tiles = tm * tn;
      const int64_t rounds = tiles / ncu, rem = tiles - rounds * ncu;
      const int64_t m_split = rounds * ncu / tn;                  // row blocks of the whole rounds
      if (rounds >= 1 && rem * 100 >= 25 * (int64_t)ncu && rem * 100 < (int64_t)g_mmrca_auto256_tail_pct * ncu && m_split >= 1 && m_split < tm) {
        const int64_t M1 = m_split * 256;
        if (int rc = mmrca_gemm256(A, B, C, bias, addend, preact, M1, N, K, lda, ldb, ldc, a_layout, b_layout, act, nullptr, st)) return rc;
        const char* A2 = (const char*)A + M1 * lda * 2;
        char* C2 = (char*)C + M1 * ldc * 2;
        const char* add2 = addend ? (const char*)addend + M1 * ldc * 2 : nullptr;
        char* pre2 = preact ? (char*)preact + M1 * ldc * 2 : nullptr;
        return gemm_dispatch(A2, B, C2, bias, add2, pre2, M - M1, N, K, lda, ldb, ldc, a_layout, b_layout, act, 0, dtype,
                             g_mmrca_auto256_tail == 2 ? MMRCA_GEMM_MFMA : MMRCA_GEMM_MFMA_1STAGE, stream, nullptr, nullptr);
      }
    }
    if (colsum_fused && fused_done) *fused_done = true;
    return mmrca_gemm256(A, B, C, bias, addend, preact, M, N, K, lda, ldb, ldc, a_layout, b_layout, act, colsum_fused, st);
  }
  if (impl == MMRCA_GEMM_MFMA_PERSIST || impl == MMRCA_GEMM_MFMA_TALL || impl == MMRCA_GEMM_MFMA_256W || impl == MMRCA_GEMM_MFMA_256X4)
    return mmrca_fail(-3, "gemm: impl %d was an experimental kernel of round 1 (measured slower, DESIGN.md K2) and has been removed", impl);
  if ((impl == MMRCA_GEMM_MFMA || impl == MMRCA_GEMM_MFMA_BK32 || impl == MMRCA_GEMM_MFMA_1STAGE) && !ok_mfma)
    return mmrca_fail(-3, "gemm: shape M=%lld N=%lld K=%lld dtype=%d does not qualify for the MFMA kernel", (long long)M, (long long)N, (long long)K, dtype);
  const bool use_mfma = ok_mfma && impl != MMRCA_GEMM_REF;   // 128x128 kernel

  if (use_mfma) {
    const int tiles_m = (int)((M + GBM - 1) / GBM), tiles_n = (int)((N + GBN - 1) / GBN);
    int ksplits = 1;
    int64_t ksplit_len = K;
    if (out_f32_accum) {
      const int64_t ksteps = K / GBK;
      // resident slots = 256 CUs x 2 blocks: aim for one (few tiles) or two (many tiles) full rounds
      const int64_t target = (g_mmrca_dbg >> 8) > 0 ? (g_mmrca_dbg >> 8) : ((int64_t)tiles_m * tiles_n >= 64 ? 1024 : 512);   // see below: 1024 = one round of 4 blocks/CU (BK=32 kernel) or two rounds of 2 (BK=64)
      int64_t want = target / ((int64_t)tiles_m * tiles_n);
      if (want < 1) want = 1;
      if (want > ksteps / 4) want = ksteps / 4 > 0 ? ksteps / 4 : 1;
      const int64_t steps_per = (ksteps + want - 1) / want;
      ksplit_len = steps_per * GBK;
      ksplits = (int)((ksteps + steps_per - 1) / steps_per);
    }
    const bool ak = a_layout == MMRCA_KROW, bk = b_layout == MMRCA_KROW, at = out_f32_accum != 0;
    // AUTO (tools/gemm_bench.py, interleaved A/B on the encoder shapes): thread-level parallelism beats software prefetch on
    // this chip -- the kernels with 32 KiB of LDS per block (four to five resident blocks per CU) win everywhere:
    //   forward (ROWK,ROWK) and dgrad (ROWK,KROW): single-stage 128x128x64 (790-910 TFLOP/s vs 680-875 for the two-stage
    //     kernel at two blocks per CU); 64-deep steps keep the ROWK operands' HBM reads in full 128-byte lines;
    //   wgrad (KROW,KROW, fp32 atomics): two-stage 128x128x32 (800-870 vs 680-740), unless there are too few tiles.
    // ragged contraction (K % 64 != 0): the 32-deep kernel takes K % 32 == 0 exactly and anything else through its edge step.  The
    // single-stage 64-deep kernel has the same edge step (MMRCA_GEMM_KEDGE64=1 sends forward / input gradient AUTO launches with
    // K % 64 != 0 there: longer steps, half the barriers per flop) -- measured neutral on both conv workloads (EfficientNetV2-M 909.4 vs
    // 906.9 samples/s, configs[2] 627.1 vs 626.9: these products wait on HBM, not on barriers), so it is off by default.
    static const int kedge64 = getenv("MMRCA_GEMM_KEDGE64") ? atoi(getenv("MMRCA_GEMM_KEDGE64")) : 0;
    const bool edge64 = kedge_on && ((kedge64 && impl == MMRCA_GEMM_AUTO) || impl == MMRCA_GEMM_MFMA_1STAGE) && !out_f32_accum && K % 64 != 0 &&
                        K % 8 == 0 && K >= 64;      // (an explicit MFMA_1STAGE request always gets it: the tests do that)
    const bool need32 = K % 64 != 0 && !edge64;
    // (experiment, round 4: below this many 128x128 tiles -- an under-filled chip, where the single-stage kernel has no sibling blocks to
    // hide its load -> barrier -> compute sequence -- AUTO takes the two-stage kernel; 0 = never.  Result in DESIGN K2.)
    static const int g_auto_2stage_below = getenv("MMRCA_AUTO_2STAGE_BELOW") ? atoi(getenv("MMRCA_AUTO_2STAGE_BELOW")) : 512;
    const bool auto1s = impl == MMRCA_GEMM_AUTO && !at && !need32 &&
                        (edge64 || !((int64_t)tiles_m * tiles_n < g_auto_2stage_below && K >= 768 && !colsum_fused && !ragged));
    const bool auto32 = (impl == MMRCA_GEMM_AUTO && at && (int64_t)tiles_m * tiles_n >= 64) || need32;
    if ((impl == MMRCA_GEMM_MFMA_BK32 || auto32) && !(at && bias)) {
#define L32(AK_, BK_, AT_) launch_mfma32<AK_, BK_, AT_>(A, B, C, bias, addend, preact, M, N, K, lda, ldb, ldc, act, tiles_m, tiles_n, ksplits, ksplit_len, st)
      if (!ak && !bk && !at) L32(false, false, false);
      else if (!ak && bk && !at) L32(false, true, false);
      else if (ak && !bk && !at) L32(true, false, false);
      else if (ak && bk && !at) L32(true, true, false);
      else if (!ak && !bk && at) L32(false, false, true);
      else if (!ak && bk && at) L32(false, true, true);
      else if (ak && !bk && at) L32(true, false, true);
      else L32(true, true, true);
#undef L32
      MMRCA_CHECK_LAUNCH("gemm(mfma,bk32)");
      return 0;
    }
    // (compiled for five waves per SIMD, <= 96 VGPRs, this kernel spills and runs 2-5x slower: four is the sweet spot)
    if ((impl == MMRCA_GEMM_MFMA_1STAGE || auto1s) && !(at && bias)) {
      float* cs1 = at ? nullptr : colsum_fused;
      if (cs1 && fused_done) *fused_done = true;
#define L1S(AK_, BK_, AT_) launch_mfma1s<AK_, BK_, AT_, 4>(A, B, C, bias, addend, preact, M, N, K, lda, ldb, ldc, act, tiles_m, tiles_n, ksplits, ksplit_len, cs1, st)
      if (!ak && !bk && !at) L1S(false, false, false);
      else if (!ak && bk && !at) L1S(false, true, false);
      else if (ak && !bk && !at) L1S(true, false, false);
      else if (ak && bk && !at) L1S(true, true, false);
      else if (!ak && !bk && at) L1S(false, false, true);
      else if (!ak && bk && at) L1S(false, true, true);
      else if (ak && !bk && at) L1S(true, false, true);
      else L1S(true, true, true);
#undef L1S
      MMRCA_CHECK_LAUNCH("gemm(mfma,1stage)");
      return 0;
    }
    if (at && bias && (impl == MMRCA_GEMM_MFMA_BK32 || auto32)) {   // the same on the 128x128x32 kernel (four blocks per CU)
      if (bk) launch_mfma32<true, true, true, true>(A, B, C, nullptr, addend, preact, M, N, K, lda, ldb, ldc, act, tiles_m, tiles_n, ksplits, ksplit_len, st, (float*)bias);
      else launch_mfma32<true, false, true, true>(A, B, C, nullptr, addend, preact, M, N, K, lda, ldb, ldc, act, tiles_m, tiles_n, ksplits, ksplit_len, st, (float*)bias);
      MMRCA_CHECK_LAUNCH("gemm(mfma,bk32,wgrad+dbias)");
      return 0;
    }
    if (at && bias) {   // weight gradient with the bias gradient fused (A is KROW by contract)
      if (bk) launch_mfma<true, true, true, true>(A, B, C, nullptr, addend, preact, M, N, K, lda, ldb, ldc, act, tiles_m, tiles_n, ksplits, ksplit_len, (float*)bias, st);
      else launch_mfma<true, false, true, true>(A, B, C, nullptr, addend, preact, M, N, K, lda, ldb, ldc, act, tiles_m, tiles_n, ksplits, ksplit_len, (float*)bias, st);
      MMRCA_CHECK_LAUNCH("gemm(mfma,wgrad+dbias)");
      return 0;
    }
#define L(AK_, BK_, AT_) launch_mfma<AK_, BK_, AT_, false>(A, B, C, out_f32_accum ? nullptr : bias, addend, preact, M, N, K, lda, ldb, ldc, act, tiles_m, tiles_n, ksplits, ksplit_len, out_f32_accum ? (float*)bias : nullptr, st)
    if (!ak && !bk && !at) L(false, false, false);
    else if (!ak && bk && !at) L(false, true, false);
    else if (ak && !bk && !at) L(true, false, false);
    else if (ak && bk && !at) L(true, true, false);
    else if (!ak && !bk && at) L(false, false, true);
    else if (!ak && bk && at) L(false, true, true);
    else if (ak && !bk && at) L(true, false, true);
    else L(true, true, true);
#undef L
    MMRCA_CHECK_LAUNCH("gemm(mfma)");
    return 0;
  }

  if (out_f32_accum && bias) {     // fused bias gradient, reference path: column sums of A[K rows][M cols]
    if (int rc = mmrca_colsum_accum(A, (float*)bias, K, M, lda, dtype, stream)) return rc;
    bias = nullptr;
  }
  const int64_t sam = a_layout == MMRCA_ROWK ? lda : 1, sak = a_layout == MMRCA_ROWK ? 1 : lda;
  const int64_t sbn = b_layout == MMRCA_ROWK ? ldb : 1, sbk = b_layout == MMRCA_ROWK ? 1 : ldb;
  // AUTO: the fp32-matrix-core kernel; impl == REF (or MMRCA_GEMM_GEN=0): the VALU reference kernel
  static const bool gen_on = !(getenv("MMRCA_GEMM_GEN") && atoi(getenv("MMRCA_GEMM_GEN")) == 0);
  if (impl != MMRCA_GEMM_REF && gen_on) {
    const int64_t tm = (M + 127) / 128, tn = (N + 63) / 64;
    MMRCA_REQUIRE(tm * tn < (1ll << 31), "gemm(gen): too many tiles");
    int64_t splits = 1, ksplit_len = K;
    if (out_f32_accum && tm * tn < 128 && K >= 512) {        // fill ~512 workgroup slots, at least 128 k per range (a small-batch weight
      splits = (512 + tm * tn - 1) / (tm * tn);               // gradient -- configs[0]: 122 x 122 over K = 3,136 -- ran on TWO workgroups)
      if (splits > K / 128) splits = K / 128;
      ksplit_len = ((K + splits - 1) / splits + 15) / 16 * 16;
      splits = (K + ksplit_len - 1) / ksplit_len;
    }
#define LGEN(AK_, BK_)                                                                                                            \
    MMRCA_DISPATCH_DTYPE(dtype, "gemm",                                                                                             \
      hipLaunchKernelGGL((gemm_gen_k<T, AK_, BK_>), dim3((unsigned)(tm * tn), (unsigned)splits), dim3(256), 0, st, (const T*)A,     \
                         (const T*)B, C, (const T*)bias, (const T*)addend, (T*)preact, M, N, K, sam, sak, sbn, sbk, ldc, act,     \
                         out_f32_accum, (int)tn, ksplit_len);)
    if (sak == 1 && sbk == 1) { LGEN(true, true) }
    else if (sak == 1) { LGEN(true, false) }
    else if (sbk == 1) { LGEN(false, true) }
    else { LGEN(false, false) }
#undef LGEN
    MMRCA_CHECK_LAUNCH("gemm(gen)");
    return 0;
  }
  dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
  MMRCA_REQUIRE(grid.y <= 65535, "gemm(ref): M too large for the reference kernel grid");
  MMRCA_DISPATCH_DTYPE(dtype, "gemm",
    hipLaunchKernelGGL(gemm_ref_k<T>, grid, dim3(256), 0, st, (const T*)A, (const T*)B, C, (const T*)bias, (const T*)addend,
                       (T*)preact, M, N, K, sam, sak, sbn, sbk, ldc, act, out_f32_accum);)
  MMRCA_CHECK_LAUNCH("gemm(ref)");
  return 0;
}

extern "C" int mmrca_gemm(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact,
                          int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                          int a_layout, int b_layout, int act, int out_f32_accum, int dtype, int impl, void* stream) {
  return gemm_dispatch(A, B, C, bias, addend, preact, M, N, K, lda, ldb, ldc, a_layout, b_layout, act, out_f32_accum, dtype,
                       impl, stream, nullptr, nullptr);
}

extern "C" int mmrca_gemm_colsum(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact,
                                 float* colsum, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                 int a_layout, int b_layout, int act, int dtype, int impl, void* stream) {
  MMRCA_REQUIRE(colsum, "gemm_colsum: null colsum");
  bool done = false;
  if (int rc = gemm_dispatch(A, B, C, bias, addend, preact, M, N, K, lda, ldb, ldc, a_layout, b_layout, act, 0, dtype, impl,
                             stream, colsum, &done))
    return rc;
  if (!done) return mmrca_colsum_accum(C, colsum, M, N, ldc, dtype, stream);    // kernels without the fused sums: one more pass
  return 0;
}

// C[M,N] = A[M,K] . B[N,K]^T (bf16, both row-major in K, no epilogue) with the BatchNorm moments of the stored C in the epilogue:
// s1[tm, n] = sum over the valid rows of 128-row block tm of (c - shift[n]), s2 likewise of the squares; s1 / s2 are
// [mmrca_gemm_bnstats_slots(M), N] fp32 and are WRITTEN.  shift may be NULL (= 0).  Runs on the 128x128 single-stage kernel
// (K % 64 == 0) or the 32-deep one (K % 32 == 0); ragged M / N (N % 8 == 0) as in mmrca_gemm.  Returns -3 for shapes it does not take.
extern "C" int64_t mmrca_gemm_bnstats_slots(int64_t M) { return M > 0 ? (M + GBM - 1) / GBM : 0; }
extern "C" int mmrca_gemm_bnstats(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                  int dtype, const float* shift, float* s1, float* s2, void* stream) {
  MMRCA_REQUIRE(A && B && C && s1 && s2 && M > 0 && N > 0 && K > 0, "gemm_bnstats: bad arguments");
  MMRCA_REQUIRE(lda >= K && ldb >= K && ldc >= N, "gemm_bnstats: leading dimension too small");
  if (!(dtype == MMRCA_BF16 && N % 8 == 0 && N >= 8 && K % 32 == 0 && M >= 64 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 &&
        aligned16(A) && aligned16(B) && aligned16(C) && (!shift || aligned16(shift))))
    return mmrca_fail(-3, "gemm_bnstats: shape M=%lld N=%lld K=%lld / dtype %d is not taken by the 128x128 bf16 kernels", (long long)M,
                      (long long)N, (long long)K, dtype);
  const int tiles_m = (int)((M + GBM - 1) / GBM), tiles_n = (int)((N + GBN - 1) / GBN);
  hipStream_t st = (hipStream_t)stream;
  const BnStat bst{shift, s1, s2};
  if (K % 64 == 0)
    launch_mfma1s<false, false, false, 4, false, true>(A, B, C, nullptr, nullptr, nullptr, M, N, K, lda, ldb, ldc, MMRCA_ACT_NONE, tiles_m, tiles_n, 1, K,
                                          nullptr, st, nullptr, nullptr, nullptr, 3, bst);
  else
    launch_mfma32<false, false, false, false, true>(A, B, C, nullptr, nullptr, nullptr, M, N, K, lda, ldb, ldc, MMRCA_ACT_NONE, tiles_m, tiles_n, 1, K, st,
                                       nullptr, bst);
  MMRCA_CHECK_LAUNCH("gemm_bnstats");
  return 0;
}

// mmrca_gemm / mmrca_gemm_colsum with the row contract of the 256x256 kernel made explicit: a_rows_readable / side_rows_readable
// = rows of A / of the side operand (addend, or preact under MMRCA_ACT_MUL) that exist in memory.  impl = MMRCA_GEMM_MFMA256 with
// a ragged M is accepted only when both cover round_up(M, 256).
extern "C" int mmrca_gemm_rows(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact,
                               float* colsum, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                               int64_t a_rows_readable, int64_t side_rows_readable, int a_layout, int b_layout, int act, int dtype,
                               int impl, void* stream) {
  const int64_t need = (M + 255) / 256 * 256;
  const bool has_side = addend != nullptr || act == MMRCA_ACT_MUL;
  MMRCA_REQUIRE(a_rows_readable >= M && (!has_side || side_rows_readable >= M), "gemm_rows: readable rows below M");
  bool vouched = true;
  if (impl == MMRCA_GEMM_MFMA256 && a_layout == MMRCA_ROWK) {
    if (a_rows_readable < need)
      return mmrca_fail(-4, "gemm_rows: the 256x256 kernel reads %lld rows of A, only %lld are readable", (long long)need, (long long)a_rows_readable);
    if (has_side && side_rows_readable < need)
      return mmrca_fail(-4, "gemm_rows: the 256x256 kernel reads %lld rows of the side operand, only %lld are readable", (long long)need, (long long)side_rows_readable);
  }
  bool done = false;
  if (int rc = gemm_dispatch(A, B, C, bias, addend, preact, M, N, K, lda, ldb, ldc, a_layout, b_layout, act, 0, dtype, impl, stream,
                             colsum, colsum ? &done : nullptr, vouched))
    return rc;
  if (colsum && !done) return mmrca_colsum_accum(C, colsum, M, N, ldc, dtype, stream);
  return 0;
}
