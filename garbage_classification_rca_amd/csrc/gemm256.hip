// K2, large-tile variant: 256x256x64 block tile, 8 waves (2 x 4), 128 KiB LDS, for the forward / input-gradient GEMMs
// of the encoders (bf16 out).  Same LDS images and fragment reads as gemm.hip (ROWK: swizzled 128-B rows read with
// ds_read_b128; KROW: swizzled 256-B rows read with ds_read_b64_tr_b16); what changes is the schedule:
//
//   * a K-tile is four HALF-tiles (A rows 0-127 / 128-255, B rows 0-127 / 128-255), each 16 KiB = two
//     global_load_lds_dwordx4 per thread;
//   * a K-tile is computed in four PHASES, one 64x32 quadrant of the wave's 128x64 output per phase
//     (16 x v_mfma_f32_16x16x32_bf16), in the order (a0,b0) (a0,b1) (a1,b1) (a1,b0) so that consecutive phases share
//     one operand's fragments;
//   * each phase issues the load of ONE half-tile of the NEXT K-tile (order A0, B0, B1, A1 = the order of first use)
//     into the other LDS stage, and ends with `s_waitcnt vmcnt(4); s_barrier`: everything except the two most
//     recently issued half-tiles has landed, i.e. loads stay in flight across three barriers and the chip never drains
//     its memory pipeline inside the K loop (the wait counts shrink to 2 and 0 in the last K-tile only).
//     Hazards: a slot is refilled at least one barrier after its last read (WAR) and read at least one barrier after the
//     counted wait that retires its fill (RAW); both follow from the fixed phase order above.
//
// A wave's output rows are two 64-row pieces, one in each A half (rows 128*ah + 64*wr + [0,64)), its columns two 32-column
// pieces, one in each B half (cols 128*bh + 32*wc + [0,32)), so every wave touches every half-tile.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

#define HT_BYTES (128 * 64 * 2)         // one half-tile
#define STAGE_BYTES (4 * HT_BYTES)      // A0 A1 B0 B1
#define LDS256_BYTES (128 * (256 * 4 + 16))   // max(2 stages = 128 KiB, epilogue staging 130 KiB)
enum { SLOT_A0 = 0, SLOT_A1 = 1, SLOT_B0 = 2, SLOT_B1 = 3 };

__device__ __forceinline__ int krow_f2(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// stage one 128x64 half-tile: 16 wave-instructions of 1 KiB, two per wave
template <bool KROW>
__device__ __forceinline__ void stage_half(const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t rows_total,
                                           int64_t k0, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    const int i = wave * 2 + ii;
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
      const int c = ((((chp >> 1) ^ krow_f2(kr))) << 1) | (chp & 1);
      src = base + (k0 + kr) * ld + row0 + c * 8;
    }
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(lds_tile + i * 1024), 16, 0, 0);
  }
}

template <bool KROW>
__device__ __forceinline__ bf16x8 load_frag2(const char* lds_tile, int rb, int ks, int lane) {
  if (!KROW) {
    const int r = rb + (lane & 15);
    const int ch = 4 * ks + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds_tile + r * 128 + ((ch ^ ((r >> 1) & 7)) << 4));
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = 32 * ks + 8 * g + q, row1 = row + 4;
    const int off0 = row * 256 + ((((rb >> 4) ^ krow_f2(row))) << 5) + p * 8;
    const int off1 = row1 * 256 + ((((rb >> 4) ^ krow_f2(row1))) << 5) + p * 8;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds_tile + off0));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds_tile + off1));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

#define WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory")

template <bool A_KROW, bool B_KROW>
__global__ void __launch_bounds__(512, 2)
gemm_mfma256_k(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, const bf16_t* __restrict__ bias,
               const bf16_t* __restrict__ addend, bf16_t* __restrict__ preact, int64_t M, int64_t N, int64_t K,
               int64_t lda, int64_t ldb, int64_t ldc, int act, int tiles_m, int tiles_n, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [2 stages][A0 A1 B0 B1]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wave >> 2, wc = wave & 3;

  const int nwg = tiles_m * tiles_n;
  const int orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int GROUP = 4;
  const int group = wgid / (GROUP * tiles_n);
  const int first_m = group * GROUP;
  const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int tm = first_m + (wgid % (GROUP * tiles_n)) % gsize;
  const int tn = (wgid % (GROUP * tiles_n)) / gsize;
  const int64_t m_blk = (int64_t)tm * 256, n_blk = (int64_t)tn * 256;
  const int nt = (int)(K / 64);

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Half-tiles are numbered in order of first use: seq = 4*t + {0:A0, 1:B0, 2:B1, 3:A1}; seq s lives in LDS stage (s>>2)&1.
  // Phase P = 4*t + p needs: p0 -> A0,B0 ; p1 -> B1 ; p2 -> A1 ; p3 -> nothing new (B0 fragments stay in registers from p0).
  // Each phase issues seq P+6, i.e. five to six phases before its first use; the slot it overwrites held seq P-2, whose
  // only LDS read was >= 2 phases ago.  Up to five half-tiles (80 KiB) are in flight per CU.
  const int total_seq = 4 * nt;
  auto issue_seq = [&](int sq) {
    const int t = sq >> 2, which = sq & 3;
    char* st = smem + (t & 1) * STAGE_BYTES;
    const int64_t k0 = (int64_t)t * 64;
    if (which == 0) stage_half<A_KROW>(A, lda, m_blk, M, k0, st + SLOT_A0 * HT_BYTES, wave, lane);
    else if (which == 1) stage_half<B_KROW>(B, ldb, n_blk, N, k0, st + SLOT_B0 * HT_BYTES, wave, lane);
    else if (which == 2) stage_half<B_KROW>(B, ldb, n_blk + 128, N, k0, st + SLOT_B1 * HT_BYTES, wave, lane);
    else stage_half<A_KROW>(A, lda, m_blk + 128, M, k0, st + SLOT_A1 * HT_BYTES, wave, lane);
  };
  // wait until at most `halves` of the most recently issued half-tiles are still in flight, then barrier
  auto wait_barrier = [&](int halves) {
    if (halves >= 4) { WAIT_BARRIER(8); }
    else if (halves == 3) { WAIT_BARRIER(6); }
    else if (halves == 2) { WAIT_BARRIER(4); }
    else if (halves == 1) { WAIT_BARRIER(2); }
    else { WAIT_BARRIER(0); }
  };
  // largest seq that must have landed before phase P starts
  auto needed_before = [&](int P) { const int t = P >> 2, p = P & 3; return 4 * t + (p == 0 ? 1 : (p == 1 ? 2 : 3)); };

#pragma unroll
  for (int sq = 0; sq < 6; ++sq)
    if (sq < total_seq) issue_seq(sq);
  {
    const int last = (total_seq < 6 ? total_seq : 6) - 1;
    wait_barrier(last - needed_before(0));
  }

  // Two barriers per phase: a LOAD segment (issue one half-tile + this phase's fragment reads) and a COMPUTE segment
  // (16 MFMAs).  Waves with wr==1 run one barrier behind the waves with wr==0, so one group of four waves (one per SIMD)
  // reads LDS while the other feeds the matrix pipe.  The counted vmcnt that retires the data of phase P+1 must precede
  // the barrier that ends group 0's COMPUTE segment P = group 1's LOAD segment P; both groups have issued the same
  // loads by then, so the count is the same.
  bf16x8 af[4][2], b0f[2][2], b1f[2][2];
  if (dbg & 2) {   // timing-only build path: fragments never loaded
    const bf16x8 z = *reinterpret_cast<const bf16x8*>(smem + lane * 16);
    for (int i = 0; i < 4; ++i) for (int k2 = 0; k2 < 2; ++k2) af[i][k2] = z;
    for (int j = 0; j < 2; ++j) for (int k2 = 0; k2 < 2; ++k2) { b0f[j][k2] = z; b1f[j][k2] = z; }
  }
  if (wr) asm volatile("s_barrier" ::: "memory");
#define SEG_END(IS_LOAD, P)                                                                                 \
  do {                                                                                                      \
    if ((IS_LOAD) == (wr != 0)) {                                                                           \
      const int issued_ = ((P) + 6 < total_seq ? (P) + 6 : total_seq - 1);                                  \
      wait_barrier(issued_ - needed_before((P) + 1));                                                       \
    } else {                                                                                                \
      asm volatile("s_barrier" ::: "memory");                                                               \
    }                                                                                                       \
  } while (0)
#define MFMA_QUAD(AH, BH, BF)                                                                               \
  if (!(dbg & 4)) {                                                                                         \
  if (!(dbg & 8)) __builtin_amdgcn_s_setprio(1);                                                            \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                           \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
        acc[AH][BH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[j][ks], af[i][ks], acc[AH][BH][i][j], 0, 0, 0); \
  if (!(dbg & 8)) __builtin_amdgcn_s_setprio(0); }
#define LOAD_B(BF, SLOT)                                                                                    \
  if (!(dbg & 2))                                                                                           \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
      BF[j][ks] = load_frag2<B_KROW>(st + (SLOT) * HT_BYTES, 32 * wc + 16 * j, ks, lane);
#define LOAD_A(SLOT)                                                                                        \
  if (!(dbg & 2))                                                                                           \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
      af[i][ks] = load_frag2<A_KROW>(st + (SLOT) * HT_BYTES, 64 * wr + 16 * i, ks, lane);

  for (int t = 0; t < nt; ++t) {
    const char* st = smem + (t & 1) * STAGE_BYTES;
    const int P0 = 4 * t;
    // phase 0: quadrant (a0, b0)
    if (P0 + 6 < total_seq && !(dbg & 1)) issue_seq(P0 + 6);
    LOAD_B(b0f, SLOT_B0) LOAD_A(SLOT_A0)
    SEG_END(true, P0);
    MFMA_QUAD(0, 0, b0f)
    SEG_END(false, P0);
    // phase 1: quadrant (a0, b1)
    if (P0 + 7 < total_seq && !(dbg & 1)) issue_seq(P0 + 7);
    LOAD_B(b1f, SLOT_B1)
    SEG_END(true, P0 + 1);
    MFMA_QUAD(0, 1, b1f)
    SEG_END(false, P0 + 1);
    // phase 2: quadrant (a1, b1)
    if (P0 + 8 < total_seq && !(dbg & 1)) issue_seq(P0 + 8);
    LOAD_A(SLOT_A1)
    SEG_END(true, P0 + 2);
    MFMA_QUAD(1, 1, b1f)
    SEG_END(false, P0 + 2);
    // phase 3: quadrant (a1, b0) -- B0 fragments are still in registers
    if (P0 + 9 < total_seq && !(dbg & 1)) issue_seq(P0 + 9);
    SEG_END(true, P0 + 3);
    MFMA_QUAD(1, 0, b0f)
    SEG_END(false, P0 + 3);
  }
  if (!wr) asm volatile("s_barrier" ::: "memory");
#undef SEG_END
#undef MFMA_QUAD
#undef LOAD_A
#undef LOAD_B

  // Epilogue through LDS: the MFMA layout gives a lane 4 consecutive columns of ONE row, i.e. 32-byte pieces of 16
  // different rows per store instruction (measured: 2.1 TB/s, fully exposed at one block per CU).  Instead each 128-row
  // half of the tile goes to LDS as fp32 (rows padded by 16 B: conflict-free 16-byte writes), and is read back one
  // whole 256-column row per wave instruction, so bias / GELU / addend are applied in fp32 and every global access
  // (pre-activation store, addend load, output store) is a contiguous 512-byte row segment.
  constexpr int EP_STRIDE = 256 * 4 + 16;
  const int g = lane >> 4, l16 = lane & 15;
  const int64_t ncol = n_blk + lane * 4;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
    bf16x4 b4 = *reinterpret_cast<const bf16x4*>(bias + ncol);
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (float)b4[r];
  }
  asm volatile("s_barrier" ::: "memory");
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = 64 * wr + 16 * i + l16, col = 128 * b + 32 * wc + 16 * j + 4 * g;
          *reinterpret_cast<f32x4*>(smem + row * EP_STRIDE + col * 4) = acc[a][b][i][j];
        }
    __syncthreads();
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      const int row = wave * 16 + rr;
      const int64_t m = m_blk + 128 * a + row;
      if (m < M) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(smem + row * EP_STRIDE + lane * 16);
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
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_fast_f(v[r]);
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
    __syncthreads();
  }
}

int g_mmrca_dbg = 0;
extern "C" int mmrca_debug_set(int v) { g_mmrca_dbg = v; return 0; }

template <bool AK, bool BK2>
static void launch256(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact, int64_t M,
                      int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int act, hipStream_t st) {
  const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)(N / 256);
  (void)hipFuncSetAttribute((const void*)gemm_mfma256_k<AK, BK2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS256_BYTES);
  hipLaunchKernelGGL((gemm_mfma256_k<AK, BK2>), dim3(tiles_m * tiles_n), dim3(512), LDS256_BYTES, st, (const bf16_t*)A,
                     (const bf16_t*)B, (bf16_t*)C, (const bf16_t*)bias, (const bf16_t*)addend, (bf16_t*)preact, M, N, K, lda,
                     ldb, ldc, act, tiles_m, tiles_n, g_mmrca_dbg);
}

// called by mmrca_gemm (gemm.hip) for bf16-out GEMMs that qualify
bool mmrca_gemm256_ok(int64_t M, int64_t N, int64_t K, int a_layout) {
  return N % 256 == 0 && K % 64 == 0 && K >= 64 && (a_layout == MMRCA_ROWK || M % 256 == 0);
}

int mmrca_gemm256(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact, int64_t M,
                  int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_layout, int b_layout, int act,
                  hipStream_t st) {
  const bool ak = a_layout == MMRCA_KROW, bk = b_layout == MMRCA_KROW;
  if (!ak && !bk) launch256<false, false>(A, B, C, bias, addend, preact, M, N, K, lda, ldb, ldc, act, st);
  else if (!ak && bk) launch256<false, true>(A, B, C, bias, addend, preact, M, N, K, lda, ldb, ldc, act, st);
  else if (ak && !bk) launch256<true, false>(A, B, C, bias, addend, preact, M, N, K, lda, ldb, ldc, act, st);
  else launch256<true, true>(A, B, C, bias, addend, preact, M, N, K, lda, ldb, ldc, act, st);
  MMRCA_CHECK_LAUNCH("gemm(mfma256)");
  return 0;
}
