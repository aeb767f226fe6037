// K2, large-tile kernel: 256x256x64 block tile, 8 waves (2 x 4), 128 KiB LDS, one workgroup per CU, with the operand
// stream kept in flight ACROSS barriers (counted vmcnt).  Same LDS images as gemm.hip (ROWK: XOR-swizzled 128-B rows read
// with ds_read_b128; KROW: XOR-swizzled 256-B rows read with ds_read_b64_tr_b16).
//
// Schedule
//   * a K-tile is four HALF-tiles (A rows 0-127 / 128-255, B rows 0-127 / 128-255), 16 KiB each = two
//     global_load_lds_dwordx4 per thread; two LDS stages of four half-tiles;
//   * a K-tile is computed in four PHASES, one 64x32 quadrant of the wave's output per phase (16 x
//     v_mfma_f32_16x16x32_bf16), in the order (a0,b0) (a0,b1) (a1,b1) (a1,b0): a wave's rows are two 64-row pieces, one
//     in each A half (rows 128*ah + 64*wr + [0,64)), its columns two 32-column pieces, one in each B half (cols
//     128*bh + 32*wc + [0,32)), so that half-tile A0 and B0 are read in phase 0 only, B1 in phase 1, A1 in phase 2
//     (B0's fragments stay in registers for phase 3) and their LDS slots are free for the stream early;
//   * half-tiles are numbered in order of first use, seq = 4t + {0:A0, 1:B0, 2:B1, 3:A1}; phase P = 4t + p issues the
//     loads of seq P+6 (one and a half K-tiles ahead, 64 KiB in flight per CU in steady state) and every wait is a
//     COUNTED `s_waitcnt vmcnt(8)`: the memory pipeline never drains inside the K loop;
//   * every phase is a LOAD segment (issue one half-tile + this phase's fragment reads) and a COMPUTE segment (16 MFMAs)
//     separated by raw s_barriers; the waves with wr == 1 run one barrier behind those with wr == 0, so of the two
//     waves that share a SIMD one reads LDS while the other feeds the matrix pipe.
//
// Why the LDS fragment reads are inline asm: with `ds_read` emitted from C++ the compiler cannot tell the reads from the
// LDS-DMA writes in flight (one LDS array, runtime addresses) and inserts `s_waitcnt vmcnt(0)` in front of every group of
// reads -- the whole stream drained four times per K-tile (round 1's version of this kernel ran at 0.5-0.8 PFLOP/s for
// that reason; the .s showed the waits).  An asm read is invisible to that pass; ordering is by the hand-placed counted
// waits below.  Rules: a slot is read at least one barrier after the counted wait that retires its DMA (RAW) and is
// refilled at least two phases after its last read, whose lgkmcnt(0) precedes a barrier in between (WAR).
//
// Hazard bookkeeping (P = 4t + p; needed(P) = highest seq phase P reads: p0 -> 4t+1, p1 -> 4t+2, p2,p3 -> 4t+3):
//   before the barrier that lets a group start LOAD(P+1), every wave has issued through seq P+6 and waits until seq
//   <= needed(P+1) has landed: (P + 6 - needed(P+1)) half-tiles may stay in flight = 4, 4, 5, 4 for p = 0..3 -> vmcnt(8)
//   (two instructions per half-tile).  Group 0 places that wait at the end of COMPUTE(P), group 1 at the end of LOAD(P):
//   both then precede the same barrier.  The last two K-tiles issue nothing new and their counts shrink to 8,8,8,4 /
//   2,0,0,0.
#include "common.h"
#include "lds_asm.h"
#include <stdlib.h>
#include <type_traits>
#include <mutex>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

#define HT_BYTES (128 * 64 * 2)         // one half-tile
#define STAGE_BYTES (4 * HT_BYTES)      // A0 A1 B0 B1
#define LDS256_BYTES (128 * (256 * 4 + 16))   // max(2 stages = 128 KiB, epilogue staging 130 KiB)
#define SLOT_A0 0
#define SLOT_A1 1
#define SLOT_B0 2
#define SLOT_B1 3

__device__ __forceinline__ int krow_f2(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// stage one 128x64 half-tile: 16 wave-instructions of 1 KiB, two per wave.  src0/src1: this lane's source pointers of the
// wave's two pieces for the CURRENT K-tile of that half (advanced by the caller)
__device__ __forceinline__ void stage_half(const bf16_t* src0, const bf16_t* src1, char* lds_piece0) {
  __builtin_amdgcn_global_load_lds((gbl_void*)src0, (lds_void*)(lds_piece0), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gbl_void*)src1, (lds_void*)(lds_piece0 + 1024), 16, 0, 0);
}

// the same as a template: F4 = the fused four-plane form, whose second piece (the lo plane's 16 rows) lands 8 KiB behind the first
template <bool F4>
__device__ __forceinline__ void stage_half_t(const bf16_t* src0, const bf16_t* src1, char* lds_piece0) {
  __builtin_amdgcn_global_load_lds((gbl_void*)src0, (lds_void*)(lds_piece0), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gbl_void*)src1, (lds_void*)(lds_piece0 + (F4 ? 8192 : 1024)), 16, 0, 0);
}

// per-lane source pointer of piece `i` (0..15) of a half-tile at K offset 0.
// STRIPE (B operand of the persistent kernel): half-tile h holds the 32-row stripes {64w + 32h + [0,32) : w = 0..3} of the
// 256-row tile instead of rows 128h + [0,128), so that a wave's two 32-column pieces are ADJACENT in the output (columns
// 64wc + [0,64): whole 128-byte lines of bf16 per row in its epilogue).  row0 is then the tile origin, h the half.
template <bool KROW, bool STRIPE = false>
__device__ __forceinline__ const bf16_t* piece_src(const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t rows_total,
                                                   int i, int lane, int h = 0) {
  if (!KROW) {
    const int r = 8 * i + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int64_t gr = row0 + (STRIPE ? 64 * (r >> 5) + 32 * h + (r & 31) : r);
    if (gr > rows_total - 1) gr = rows_total - 1;
    return base + gr * ld + c * 8;
  } else {
    const int kr = 4 * i + (lane >> 4);
    const int chp = lane & 15;
    const int c = ((((chp >> 1) ^ krow_f2(kr))) << 1) | (chp & 1);
    // ragged tiles (the split-K weight gradients of the conv layers: 176 x 1056, 48 x 192, ...): an 8-column chunk past the
    // operand's last column is fetched from column 0 of the same row instead -- finite values that only reach output rows /
    // columns the reduction never stores -- so nothing outside the [K, rows_total] matrix is ever read (rows_total % 8 == 0)
    int64_t col = row0 + (STRIPE ? 64 * (c >> 2) + 32 * h + 8 * (c & 3) : c * 8);
    if (col + 8 > rows_total) col = 0;
    return base + (int64_t)kr * ld + col;
  }
}

#define VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define BARRIER() asm volatile("s_barrier" ::: "memory")

// SLAB: split-K form for the weight gradients.  blockIdx.y selects a K range of `ksplit_len`; the block's fp32 partial
// tile goes to `slab` with plain 16-byte stores in accumulator order ([split][tile][wave][fragment][lane][4]: every wave
// instruction writes 1 KiB contiguously, no LDS staging), and splitk_reduce256_k adds the partials of a tile into C.
// (fp32 atomics would put 256 KiB per block through the memory-side atomic units: ~50 us per launch at 1.3 TB/s.)
// MODE: the epilogue, fixed at compile time (a run-time `act` switch inside the unrolled epilogue made 18 k lines of code):
//   MODE_SLAB, or the activation MMRCA_ACT_NONE / MMRCA_ACT_GELU_SAVE_GRAD / MMRCA_ACT_MUL (the ones the engine uses)
#define MODE_SLAB (-1)
// X3 (bf16x3 mode, gemm_x3.hip): operands are (hi, lo) bf16 plane pairs and the contraction is the VIRTUAL one of length 3K --
// K-tiles [0, K/64) pair (A_hi, B_hi), [K/64, 2K/64) pair (A_lo, B_hi), [2K/64, 3K/64) pair (A_hi, B_lo) -- which the split-K
// ranges cut like any other contraction; the stream re-points its source pointers when it crosses a segment boundary.
// nseg < 3 runs only the first nseg plane pairs (2: the B operand's lo plane is dropped, 1: a plain bf16 product into fp32).
template <bool A_KROW, bool B_KROW, int MODE, bool X3 = false>
__global__ void __launch_bounds__(512, 2)
gemm_mfma256_k(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, const bf16_t* __restrict__ bias,
               const bf16_t* __restrict__ addend, bf16_t* __restrict__ preact, int64_t M, int64_t N, int64_t K,
               int64_t lda, int64_t ldb, int64_t ldc, int tiles_m, int tiles_n, int ksteps_base, int ksteps_rem,
               float* __restrict__ slab, float* __restrict__ colsum, int splits,
               const bf16_t* __restrict__ A_lo = nullptr, const bf16_t* __restrict__ B_lo = nullptr, int nseg = 3) {
  constexpr bool SLAB = MODE == MODE_SLAB;
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [2 stages][A0 A1 B0 B1]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wave >> 2, wc = wave & 3;

  const int nwg = tiles_m * tiles_n;
  // One-dimensional grid of tiles x K-ranges (rounded up to a multiple of 8).  All tiles of a K-range read the SAME rows of
  // both operands, so they are placed on ONE XCD (workgroup id % 8, observed round-robin placement: speed only): that
  // XCD's L2 fetches the rows once and the launch's HBM traffic falls from ~2x to ~1x the algorithmic bytes.
  const int total = nwg * splits;
  const int per_xcd = (total + 7) >> 3;
  const int idx = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (((int)blockIdx.x >> 3) >= per_xcd || idx >= total) return;
  const int yb = idx / nwg, wgid = idx - yb * nwg;
  const int tm = wgid / tiles_n, tn = wgid - tm * tiles_n;
  const int64_t m_blk = (int64_t)tm * 256, n_blk = (int64_t)tn * 256;
  // split-K: range y covers ksteps_base (+1 for the first ksteps_rem ranges) K-tiles
  // X3: kbeg counts virtual K (segment = kbeg / K)
  const int64_t kbeg_v = SLAB ? 64 * ((int64_t)yb * ksteps_base + (yb < ksteps_rem ? yb : ksteps_rem)) : 0;
  const int nt = SLAB ? ksteps_base + (yb < ksteps_rem ? 1 : 0) : (int)(K / 64);     // >= 2 (host-checked)
  const int ksteps_seg = (int)(K / 64);
  int issue_seg = X3 ? (int)(kbeg_v / K) : 0;
  const int64_t kbeg = X3 ? kbeg_v - (int64_t)issue_seg * K : kbeg_v;
  int issue_kt = (int)(kbeg / 64);               // K-tile (within its segment) the next closing issue slot completes

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- the stream: per-lane source pointers of this wave's two pieces of each half-tile, advanced as tiles are issued
  const int64_t a_step = A_KROW ? 64 * lda : 64, b_step = B_KROW ? 64 * ldb : 64;       // elements per K-tile
  const bf16_t* pa[2][2];     // [half][piece]
  const bf16_t* pb[2][2];
  auto repoint = [&](int seg, int64_t k0) {
    const bf16_t* Ab = (X3 && seg == 1) ? A_lo : A;
    const bf16_t* Bb = (X3 && seg == 2) ? B_lo : B;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        pa[h][ii] = piece_src<A_KROW>(Ab, lda, m_blk + 128 * h, M, wave * 2 + ii, lane) + (A_KROW ? k0 * lda : k0);
        pb[h][ii] = piece_src<B_KROW>(Bb, ldb, n_blk + 128 * h, N, wave * 2 + ii, lane) + (B_KROW ? k0 * ldb : k0);
      }
  };
  repoint(issue_seg, kbeg);
  char* const my_piece = smem + wave * 2048;              // this wave's first piece inside a half-tile
  int issue_stage = 0;                                     // LDS stage (byte offset) of the next issue group's K-tile
  // issue order within a K-tile: A0, B0, B1, A1 (order of first use); `which` is a compile-time constant at every call
#define ISSUE(which)                                                                                            \
  do {                                                                                                          \
    if ((which) == 0) { stage_half(pa[0][0], pa[0][1], my_piece + issue_stage + SLOT_A0 * HT_BYTES); pa[0][0] += a_step; pa[0][1] += a_step; } \
    else if ((which) == 1) { stage_half(pb[0][0], pb[0][1], my_piece + issue_stage + SLOT_B0 * HT_BYTES); pb[0][0] += b_step; pb[0][1] += b_step; } \
    else if ((which) == 2) { stage_half(pb[1][0], pb[1][1], my_piece + issue_stage + SLOT_B1 * HT_BYTES); pb[1][0] += b_step; pb[1][1] += b_step; } \
    else { stage_half(pa[1][0], pa[1][1], my_piece + issue_stage + SLOT_A1 * HT_BYTES); pa[1][0] += a_step; pa[1][1] += a_step; issue_stage ^= STAGE_BYTES; \
           if constexpr (X3) { if (++issue_kt == ksteps_seg) { issue_kt = 0; if (++issue_seg < nseg) repoint(issue_seg, 0); } } } \
  } while (0)

  // ---- fragment read addresses (LDS byte addresses of stage 0; the stage is toggled by XOR per K-tile)
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
  const int g = lane >> 4, l16 = lane & 15;
  unsigned abase[4], bbase[2];      // ROWK: [ks] (2 used); KROW: [fragment]
  if (!A_KROW) {
    const int r0 = 64 * wr + l16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) abase[ks] = lds0 + r0 * 128 + (((4 * ks + g) ^ ((r0 >> 1) & 7)) << 4);
    abase[2] = abase[3] = 0;
  } else {
    const int q = l16 >> 2, p = l16 & 3, row = 8 * g + q;
#pragma unroll
    for (int i = 0; i < 4; ++i) abase[i] = lds0 + row * 256 + ((((4 * wr + i) ^ krow_f2(row))) << 5) + p * 8;
  }
  if (!B_KROW) {
    const int r0 = 32 * wc + l16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) bbase[ks] = lds0 + r0 * 128 + (((4 * ks + g) ^ ((r0 >> 1) & 7)) << 4);
  } else {
    const int q = l16 >> 2, p = l16 & 3, row = 8 * g + q;
#pragma unroll
    for (int j = 0; j < 2; ++j) bbase[j] = lds0 + row * 256 + ((((2 * wc + j) ^ krow_f2(row))) << 5) + p * 8;
  }
  // fragment (f, ks) of a slot: ROWK -> base[ks] + f*2048; KROW -> base[f] + ks*8192
#define A_FRAG(SLOT, f, ks) read_frag<A_KROW, (SLOT) * HT_BYTES + (A_KROW ? (ks) * 8192 : (f) * 2048)>(af[f][ks], abase[A_KROW ? (f) : (ks)])
#define B_FRAG(BF, SLOT, f, ks) read_frag<B_KROW, (SLOT) * HT_BYTES + (B_KROW ? (ks) * 8192 : (f) * 2048)>(BF[f][ks], bbase[B_KROW ? (f) : (ks)])

  Frag<A_KROW> af[4][2];
  Frag<B_KROW> b0f[2][2], b1f[2][2];
  [[maybe_unused]] unsigned read_stage = 0;
#define LOAD_A(SLOT)                                                                                          \
  do {                                                                                                        \
    A_FRAG(SLOT, 0, 0); A_FRAG(SLOT, 0, 1); A_FRAG(SLOT, 1, 0); A_FRAG(SLOT, 1, 1);                           \
    A_FRAG(SLOT, 2, 0); A_FRAG(SLOT, 2, 1); A_FRAG(SLOT, 3, 0); A_FRAG(SLOT, 3, 1);                           \
  } while (0)
#define LOAD_B(BF, SLOT)                                                                                      \
  do {                                                                                                        \
    B_FRAG(BF, SLOT, 0, 0); B_FRAG(BF, SLOT, 0, 1); B_FRAG(BF, SLOT, 1, 0); B_FRAG(BF, SLOT, 1, 1);           \
  } while (0)
#define LGKM0_A() lgkm0(af)
#define LGKM0_B(BF) lgkm0(BF)
#define MFMA_QUAD(AH, BH, BF)                                                                                 \
  do {                                                                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                          \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
          acc[AH][BH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_val(BF[j][ks]), frag_val(af[i][ks]), acc[AH][BH][i][j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                            \
  } while (0)
  // segment ends: the group whose turn it is waits (counted) before the barrier; N = vmcnt argument (literal)
#define END_LOAD(N)    do { if (wr) VMCNT(N); BARRIER(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define END_COMPUTE(N) do { __builtin_amdgcn_sched_barrier(0); if (!wr) VMCNT(N); BARRIER(); } while (0)
  // one K-tile = four phases.  I0..I3: 1 = issue seq P+6 in phase p; W0..W3: vmcnt argument of the phase's counted wait
#define K_TILE_G(ISS, I0, I1, I2, I3, W0, W1, W2, W3)                                                         \
  do {                                                                                                        \
    if (I0) ISS(2);                                                                                           \
    LOAD_B(b0f, SLOT_B0); LOAD_A(SLOT_A0);                                                                    \
    END_LOAD(W0);                                                                                             \
    LGKM0_B(b0f); LGKM0_A();                                                                                  \
    MFMA_QUAD(0, 0, b0f);                                                                                     \
    END_COMPUTE(W0);                                                                                          \
    if (I1) ISS(3);                                                                                           \
    LOAD_B(b1f, SLOT_B1);                                                                                     \
    END_LOAD(W1);                                                                                             \
    LGKM0_B(b1f);                                                                                             \
    MFMA_QUAD(0, 1, b1f);                                                                                     \
    END_COMPUTE(W1);                                                                                          \
    if (I2) ISS(0);                                                                                           \
    LOAD_A(SLOT_A1);                                                                                          \
    END_LOAD(W2);                                                                                             \
    LGKM0_A();                                                                                                \
    MFMA_QUAD(1, 1, b1f);                                                                                     \
    END_COMPUTE(W2);                                                                                          \
    if (I3) ISS(1);                                                                                           \
    END_LOAD(W3);                                                                                             \
    MFMA_QUAD(1, 0, b0f);                                                                                     \
    END_COMPUTE(W3);                                                                                          \
    _Pragma("unroll") for (int x = 0; x < 4; ++x) abase[x] ^= STAGE_BYTES;                                    \
    bbase[0] ^= STAGE_BYTES; bbase[1] ^= STAGE_BYTES; read_stage ^= STAGE_BYTES;                              \
  } while (0)

  // The fused four-plane form of a K step (gemm_p256_k<..., F4>): fragment index [f][pl] = plane pl (0 hi, 1 lo) of 16-row piece f
  // (1 KiB apart in a 64-byte-row image); a quadrant is three products -- hi.hi, lo.hi, hi.lo -- of 8 MFMAs each.
#define A_FRAG4(SLOT, f, pl) read_frag<false, (SLOT) * HT_BYTES + (f) * 1024>(af[f][pl], abase[pl])
#define B_FRAG4(BF, SLOT, f, pl) read_frag<false, (SLOT) * HT_BYTES + (f) * 1024>(BF[f][pl], bbase[pl])
#define LOAD_A4(SLOT)                                                                                         \
  do {                                                                                                        \
    A_FRAG4(SLOT, 0, 0); A_FRAG4(SLOT, 0, 1); A_FRAG4(SLOT, 1, 0); A_FRAG4(SLOT, 1, 1);                       \
    A_FRAG4(SLOT, 2, 0); A_FRAG4(SLOT, 2, 1); A_FRAG4(SLOT, 3, 0); A_FRAG4(SLOT, 3, 1);                       \
  } while (0)
#define LOAD_B4(BF, SLOT)                                                                                     \
  do {                                                                                                        \
    B_FRAG4(BF, SLOT, 0, 0); B_FRAG4(BF, SLOT, 0, 1); B_FRAG4(BF, SLOT, 1, 0); B_FRAG4(BF, SLOT, 1, 1);       \
  } while (0)
#define MFMA_QUAD4(AH, BH, BF)                                                                                \
  do {                                                                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                            \
    _Pragma("unroll") for (int pr = 0; pr < 3; ++pr)      /* (B plane, A plane): (hi, hi), (hi, lo), (lo, hi) */ \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
          acc[AH][BH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_val(BF[j][pr == 2 ? 1 : 0]), frag_val(af[i][pr == 1 ? 1 : 0]), acc[AH][BH][i][j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                            \
  } while (0)
#define K_TILE_G4(ISS, I0, I1, I2, I3, W0, W1, W2, W3)                                                        \
  do {                                                                                                        \
    if (I0) ISS(2);                                                                                           \
    LOAD_B4(b0f, SLOT_B0); LOAD_A4(SLOT_A0);                                                                  \
    END_LOAD(W0);                                                                                             \
    LGKM0_B(b0f); LGKM0_A();                                                                                  \
    MFMA_QUAD4(0, 0, b0f);                                                                                    \
    END_COMPUTE(W0);                                                                                          \
    if (I1) ISS(3);                                                                                           \
    LOAD_B4(b1f, SLOT_B1);                                                                                    \
    END_LOAD(W1);                                                                                             \
    LGKM0_B(b1f);                                                                                             \
    MFMA_QUAD4(0, 1, b1f);                                                                                    \
    END_COMPUTE(W1);                                                                                          \
    if (I2) ISS(0);                                                                                           \
    LOAD_A4(SLOT_A1);                                                                                         \
    END_LOAD(W2);                                                                                             \
    LGKM0_A();                                                                                                \
    MFMA_QUAD4(1, 1, b1f);                                                                                    \
    END_COMPUTE(W2);                                                                                          \
    if (I3) ISS(1);                                                                                           \
    END_LOAD(W3);                                                                                             \
    MFMA_QUAD4(1, 0, b0f);                                                                                    \
    END_COMPUTE(W3);                                                                                          \
    _Pragma("unroll") for (int x = 0; x < 4; ++x) abase[x] ^= STAGE_BYTES;                                    \
    bbase[0] ^= STAGE_BYTES; bbase[1] ^= STAGE_BYTES; read_stage ^= STAGE_BYTES;                              \
  } while (0)

  // prologue: seq 0..5 (K-tile 0 and A0, B0 of K-tile 1); phase 0 needs seq 0, 1 -> four half-tiles may stay in flight
  ISSUE(0); ISSUE(1); ISSUE(2); ISSUE(3); ISSUE(0); ISSUE(1);
  VMCNT(8);
  BARRIER();
  if (wr) BARRIER();                    // group 1 runs one barrier behind
  __builtin_amdgcn_sched_barrier(0);

  for (int t = 0; t < nt - 2; ++t) K_TILE_G(ISSUE, 1, 1, 1, 1, 8, 8, 10, 8);
  K_TILE_G(ISSUE, 1, 1, 0, 0, 8, 8, 8, 4);       // K-tile nt-2: issues B1, A1 of the last K-tile
  K_TILE_G(ISSUE, 0, 0, 0, 0, 2, 0, 0, 0);       // K-tile nt-1
  if (!wr) BARRIER();                   // group 0 waits for group 1's last segment

  if constexpr (SLAB) {
    float* dst = slab + ((int64_t)yb * nwg + (int64_t)tm * tiles_n + tn) * 65536 + wave * 8192 + lane * 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            *reinterpret_cast<f32x4*>(dst + (((a * 2 + b) * 4 + i) * 2 + j) * 256) = acc[a][b][i][j];
    return;
  }

}

// =====================================================================================================================
// Persistent form for the forward / input-gradient GEMMs (bf16 out): ONE workgroup per CU walks its share of the output
// tiles, and the operand stream never stops at a tile boundary -- the issue slots of a tile's last K-tiles already load
// the first K-tiles of the workgroup's NEXT tile (same seq numbering, same counted waits), so a tile starts computing the
// moment the previous epilogue ends.  Measured before this form: ~10 us per tile outside the K loop at one workgroup
// per CU (workgroup launch, a cold prologue, an LDS-staged epilogue that nothing overlaps) against 1.3 us per K-tile,
// i.e. 40 % of the time of a K = 768 tile.
//
// Epilogue: the 32 KiB of LDS beyond the two stages (131,072 .. 163,839) are eight PRIVATE 4-KiB staging areas, one per
// wave, so the epilogue has no barrier at all (a block-wide staging with two barriers per 32-row pass cost 4.6 us per
// tile).  A wave owns output columns 64wc + [0,64) (the B half-tiles are 32-row stripes, see piece_src) and rows
// 128a + 64wr + [0,64): per pass (a, i) it writes its four 16x16 fragments (16 rows x 64 columns fp32, 16-byte chunks
// XOR-swizzled by the row) and reads four times four whole 256-byte rows back -- LDS operations of one wave execute in
// order, so neither the read-after-write nor the next pass's write-after-read needs a wait -- then bias / activation /
// addend in fp32 and 128-byte (whole cache line) bf16 row stores, which drain while the next pass or the next tile's K
// loop runs.  All LDS accesses of the kernel are inline asm, so that no compiler-inserted vmcnt(0) drains the stream.
// =====================================================================================================================
#define EP_BASE (2 * STAGE_BYTES)
#define DS_WRITE_B128(addr, val, off) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(val), "i"(off) : "memory")

// Epilogue variants are compile-time (run-time flags cost ~20 branches and twice the code per pass): ACT is
// MMRCA_ACT_NONE (with or without a residual addend, ADD), MMRCA_ACT_GELU, MMRCA_ACT_GELU_SAVE_GRAD (stores gelu' to `preact`) or
// MMRCA_ACT_MUL (multiplies by `preact`; column sums of the stored result to `colsum` when given).
// X3 (bf16x3 mode, gemm_x3.hip): A / B are the hi planes, A_lo / B_lo the lo planes of fp32 operands; every output tile runs
// the K loop three times -- (A_hi, B_hi), (A_lo, B_hi), (A_hi, B_lo) -- the stream switching planes where it used to switch
// tiles; bias / preact / C are fp32, or, with PLANES, C is written as two bf16 planes (C = hi, C_lo = lo).  No side operands.
// F4 (X3, row-major A and B, all three plane pairs): the FUSED form of the bf16x3 product.  The plain X3 form walks the K loop
// three times per output tile -- (A_hi, B_hi), (A_lo, B_hi), (A_hi, B_lo) -- i.e. it stages A_hi and B_hi twice: 3 x 64 KiB of
// LDS-DMA and 3 x 24 fragment reads per wave for every 64 columns of K, and this kernel is co-bound by LDS bandwidth and the
// matrix pipe.  Here a K step is 32 columns of ALL FOUR planes -- a 16-KiB slot holds the hi image [128 rows x 64 B] and the lo
// image of one half-tile -- and every staged plane is used by two of the three products: per 64 columns of K 2 x 64 KiB staged and
// 2 x 24 fragment reads for the same 192 MFMAs, a third less LDS traffic per flop.  The schedule (slots, issue order, two DMA
// instructions per wave and half-tile, counted waits, phases) is unchanged: "half-tile X of K-tile t" now reads "half-tile X, planes
// hi | lo, of K step t".  64-byte image rows: 16-byte chunk index XOR (row >> 2) & 3 (conflict-free for the 16 rows of a fragment).
template <bool A_KROW, bool B_KROW, int ACT, bool ADD, bool X3 = false, bool PLANES = false, bool F4 = false>
__global__ void __launch_bounds__(512, 2)
gemm_p256_k(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, const bf16_t* __restrict__ bias,
            const bf16_t* __restrict__ addend, bf16_t* __restrict__ preact, int64_t M, int64_t N, int64_t K,
            int64_t lda, int64_t ldb, int64_t ldc, int tiles_m, int tiles_n, float* __restrict__ colsum, int skew_ticks, int dbg,
            const bf16_t* __restrict__ A_lo = nullptr, const bf16_t* __restrict__ B_lo = nullptr, bf16_t* __restrict__ C_lo = nullptr,
            int nseg = 3, int pre16 = 0, float* __restrict__ sk_slab = nullptr, unsigned* __restrict__ sk_count = nullptr, int sk_split = 0) {
  // pre16 (X3 only; MMRCA_ACT_GELU_SAVE_GRAD_BF16): gelu' is stored as bf16 -- the form a bf16 backward reads (bf16x3f mode)
  static_assert(!X3 || (!ADD && ACT != MMRCA_ACT_MUL), "the bf16x3 form has no side-operand epilogue");
  static_assert(!F4 || (X3 && !A_KROW && !B_KROW), "the fused four-plane form is the row-major bf16x3 forward product");
  constexpr int act = ACT;
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [2 stages][A0 A1 B0 B1] + 32 KiB epilogue staging
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wave >> 2, wc = wave & 3;
  const int ntiles = tiles_m * tiles_n, G = (int)gridDim.x;                  // G % 8 == 0 (host)
  // workgroups that share an XCD (blockIdx % 8, observed round-robin placement: speed only) take consecutive tile ids
  const int slot = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
  // Stream-K tail (sk_split >= 2; host: launch_p256).  ntiles = R whole rounds of G tiles + L leftover tiles.  The whole rounds
  // are walked as ever (tile ids slot, slot + G, ...); each leftover tile's K loop is cut into sk_split ranges of >= 2 K steps and
  // workgroup `slot` < L * sk_split takes ONE (tile, range) unit as its last work item -- the partial round then costs
  // ~1 / sk_split of a tile on (nearly) every CU instead of a whole tile on L of them.  Every unit leaves its fp32 partial tile in
  // `sk_slab` (accumulator order, 1 KiB per wave instruction); the unit that arrives LAST at the tile's counter adds the sk_split
  // partials in the fixed order 0, 1, ... (bitwise reproducible whatever the arrival order), resets the counter and runs the
  // ordinary epilogue.  Nobody waits for anybody: no spinning, no residency assumption.
  const bool sk_on = sk_split > 1;
  const int dp_tiles = sk_on ? (ntiles / G) * G : ntiles;                     // tiles walked whole
  const int my_dp = dp_tiles > slot ? (dp_tiles - slot + G - 1) / G : 0;      // tile ids slot, slot + G, ...
  const int sk_left = ntiles - dp_tiles;                                      // L (0 when the tail is off)
  const bool has_unit = sk_on && slot < sk_left * sk_split;
  const int my_tiles = my_dp + (has_unit ? 1 : 0);
  if (my_tiles <= 0) return;
  // Desynchronise the workgroups: all tiles take the same time, so without this every CU reaches its epilogue at the same
  // moment and the chip alternates between "every CU stores 128 KiB" (32 MiB burst = the whole L2) and "nobody stores".
  // Four groups per XCD start skew_ticks * {0,1,2,3} x 10 ns apart and keep that phase for the whole launch.
  if (skew_ticks > 0) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t wait = (uint64_t)skew_ticks * (((unsigned)blockIdx.x >> 3) & 3);
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
  const int nt = (int)(K / (F4 ? 32 : 64));                                   // K steps per plane pair (F4: per tile); >= 2 (host-checked)
  int unit_tile = 0, unit_part = 0, unit_k0 = 0, unit_len = nt;               // this workgroup's (leftover tile, K range)
  if (has_unit) {
    unit_tile = slot % sk_left; unit_part = slot / sk_left;                   // consecutive slots (one XCD) -> consecutive tiles
    const int kbase = nt / sk_split, krem = nt % sk_split;                    // the first krem ranges take one step more
    unit_k0 = unit_part * kbase + (unit_part < krem ? unit_part : krem);
    unit_len = kbase + (unit_part < krem ? 1 : 0);                            // >= 2 (host: sk_split <= nt / 2)
  }
  // work item idx of this workgroup: (tile id, first K step, K steps)
  auto work_item = [&](int idx, int& id, int& k0, int& nk) {
    if (idx < my_dp) { id = slot + idx * G; k0 = 0; nk = nt; }
    else { id = dp_tiles + unit_tile; k0 = unit_k0; nk = unit_len; }
  };
  const int GROUP = 4;
  auto tile_origin = [&](int id, int64_t& m_blk, int64_t& n_blk) {
    const int group = id / (GROUP * tiles_n);
    const int first_m = group * GROUP;
    const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int tm = first_m + (id % (GROUP * tiles_n)) % gsize;
    const int tn = (id % (GROUP * tiles_n)) / gsize;
    m_blk = (int64_t)tm * 256; n_blk = (int64_t)tn * 256;
  };

  // ---- the stream
  // per-lane BYTE offsets from A / B (32 bits: the host checks that both operands are < 4 GiB) of this wave's two pieces of
  // half-tile 0, advanced per K-tile; half-tile 1 is a wave-uniform distance away (128 rows of A; 32 rows of the striped B).
  // Rows of A past M (last row tile) are read as they are -- the caller provides round_up(M, 256) readable rows -- and the
  // output rows they produce are never stored.
  const unsigned a_step = (unsigned)(F4 ? 64 : (A_KROW ? 128 * lda : 128)), b_step = (unsigned)(F4 ? 64 : (B_KROW ? 128 * ldb : 128));
  const int64_t a_half = A_KROW ? 256 : 256 * lda, b_half = B_KROW ? 64 : 64 * ldb;       // bytes
  unsigned oa[2], ob[2];
  auto stream_to = [&](int id, int k0) {
    int64_t mb, nb;
    tile_origin(id, mb, nb);
    if constexpr (F4) {
      // one DMA instruction = 16 image rows of 64 B: lane -> row 16 wave + (lane >> 2), chunk slot lane & 3; the hi and the lo
      // plane share the geometry (oa[0] == oa[1]); B in the striped order of the epilogue (image row r = tile row 64 (r >> 5) + (r & 31))
      const int r = 16 * wave + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
      oa[0] = oa[1] = (unsigned)(((mb + r) * lda) * 2 + c * 16);
      ob[0] = ob[1] = (unsigned)(((nb + 64 * (r >> 5) + (r & 31)) * ldb) * 2 + c * 16);
    } else {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        oa[ii] = (unsigned)(2 * (piece_src<A_KROW>(A, lda, mb, (int64_t)1 << 40, wave * 2 + ii, lane) - A));
        ob[ii] = (unsigned)(2 * (piece_src<B_KROW, true>(B, ldb, nb, (int64_t)1 << 40, wave * 2 + ii, lane, 0) - B));
      }
    }
    oa[0] += (unsigned)k0 * a_step; oa[1] += (unsigned)k0 * a_step;
    ob[0] += (unsigned)k0 * b_step; ob[1] += (unsigned)k0 * b_step;
  };
  int stream_nt;                               // K steps of the stream's current work item
  {
    int id0, k00;
    work_item(0, id0, k00, stream_nt);
    stream_to(id0, k00);
  }
  int stream_kt = 0, stream_tile = 0;          // K-tiles already issued of the stream's current tile; its index in my list
  [[maybe_unused]] int stream_seg = 0;         // X3: which plane pair the stream is in
  const bf16_t* Acur = A;
  const bf16_t* Bcur = B;
  char* const my_piece = smem + wave * (F4 ? 1024 : 2048);
  int issue_stage = 0;
  // the issue slot of half-tile A1 closes a K-tile of the stream: after the last K-tile of a tile, move on to the next tile
  // (F4: piece 0 comes from the hi plane, piece 1 from the lo plane, and lands 8 KiB behind piece 0)
#define SRC_A(h, ii) reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(F4 ? ((ii) ? A_lo : A) : Acur) + (h) * a_half + oa[ii])
#define SRC_B(h, ii) reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(F4 ? ((ii) ? B_lo : B) : Bcur) + (h) * b_half + ob[ii])
#define ISSUE_P(which)                                                                                          \
  do {                                                                                                          \
    if ((which) == 0) { stage_half_t<F4>(SRC_A(0, 0), SRC_A(0, 1), my_piece + issue_stage + SLOT_A0 * HT_BYTES); }    \
    else if ((which) == 1) { stage_half_t<F4>(SRC_B(0, 0), SRC_B(0, 1), my_piece + issue_stage + SLOT_B0 * HT_BYTES); } \
    else if ((which) == 2) { stage_half_t<F4>(SRC_B(1, 0), SRC_B(1, 1), my_piece + issue_stage + SLOT_B1 * HT_BYTES); ob[0] += b_step; ob[1] += b_step; } \
    else {                                                                                                      \
      stage_half_t<F4>(SRC_A(1, 0), SRC_A(1, 1), my_piece + issue_stage + SLOT_A1 * HT_BYTES); oa[0] += a_step; oa[1] += a_step;  \
      issue_stage ^= STAGE_BYTES;                                                                               \
      if (++stream_kt == stream_nt) {                                                                           \
        stream_kt = 0;                                                                                          \
        if constexpr (X3 && !F4) {                                                                              \
          if (++stream_seg == nseg) { stream_seg = 0; ++stream_tile; }                                             \
          Acur = stream_seg == 1 ? A_lo : A; Bcur = stream_seg == 2 ? B_lo : B;                                 \
        } else ++stream_tile;                                                                                   \
        if (stream_tile < my_tiles) { int id_, k0_; work_item(stream_tile, id_, k0_, stream_nt); stream_to(id_, k0_); } \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)

  // ---- fragment read addresses: recomputed at every tile start from an opaque lane id and the scalar stage parity, so
  // that they do not occupy registers during the epilogue (the side-operand registers need the room)
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
  unsigned abase[4], bbase[2];
  unsigned read_stage = 0;                 // LDS stage (byte offset) of the K-tile the fragment reads are at
  auto frag_bases = [&]() {
    int lane_f = lane;
    asm volatile("" : "+v"(lane_f));
    const int g = lane_f >> 4, l16 = lane_f & 15;
    if constexpr (F4) {           // index = plane (0: hi image, 1: lo image, 8 KiB behind); 64-byte rows
      const int ra = 64 * wr + l16, rb = 32 * wc + l16;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        abase[pl] = (lds0 + pl * 8192 + ra * 64 + ((g ^ ((ra >> 2) & 3)) << 4)) ^ read_stage;
        bbase[pl] = (lds0 + pl * 8192 + rb * 64 + ((g ^ ((rb >> 2) & 3)) << 4)) ^ read_stage;
      }
      abase[2] = abase[3] = 0;
      return;
    }
    if (!A_KROW) {
      const int r0 = 64 * wr + l16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) abase[ks] = (lds0 + r0 * 128 + (((4 * ks + g) ^ ((r0 >> 1) & 7)) << 4)) ^ read_stage;
      abase[2] = abase[3] = 0;
    } else {
      const int q = l16 >> 2, p = l16 & 3, row = 8 * g + q;
#pragma unroll
      for (int i = 0; i < 4; ++i) abase[i] = (lds0 + row * 256 + ((((4 * wr + i) ^ krow_f2(row))) << 5) + p * 8) ^ read_stage;
    }
    if (!B_KROW) {
      const int r0 = 32 * wc + l16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bbase[ks] = (lds0 + r0 * 128 + (((4 * ks + g) ^ ((r0 >> 1) & 7)) << 4)) ^ read_stage;
    } else {
      const int q = l16 >> 2, p = l16 & 3, row = 8 * g + q;
#pragma unroll
      for (int j = 0; j < 2; ++j) bbase[j] = (lds0 + row * 256 + ((((2 * wc + j) ^ krow_f2(row))) << 5) + p * 8) ^ read_stage;
    }
  };
  Frag<A_KROW> af[4][2];
  Frag<B_KROW> b0f[2][2], b1f[2][2];
  f32x4 acc[2][2][4][2];

  // prologue of the stream: seq 0..5 of the first tile
  ISSUE_P(0); ISSUE_P(1); ISSUE_P(2); ISSUE_P(3); ISSUE_P(0); ISSUE_P(1);
  VMCNT(8);
  BARRIER();

  for (int ti = 0; ti < my_tiles; ++ti) {
    int64_t m_blk, n_blk;
    int tile_id, tile_k0, tile_nk;
    work_item(ti, tile_id, tile_k0, tile_nk);
    tile_origin(tile_id, m_blk, n_blk);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    frag_bases();
    if (wr) BARRIER();                    // group 1 runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);
    const int ntv = (X3 && !F4) ? nseg * nt : tile_nk;  // K-tiles of this work item (bf16x3: nseg plane pairs; fused: K / 32 steps of all planes)
    if constexpr (F4) {
      if (ti + 1 < my_tiles) {
        for (int t = 0; t < ntv; ++t) K_TILE_G4(ISSUE_P, 1, 1, 1, 1, 8, 8, 10, 8);
      } else {
        for (int t = 0; t < ntv - 2; ++t) K_TILE_G4(ISSUE_P, 1, 1, 1, 1, 8, 8, 10, 8);
        K_TILE_G4(ISSUE_P, 1, 1, 0, 0, 8, 8, 8, 4);
        K_TILE_G4(ISSUE_P, 0, 0, 0, 0, 2, 0, 0, 0);
      }
    } else
    if (ti + 1 < my_tiles) {              // the stream runs on into the next tile: every K-tile is a steady-state one
      for (int t = 0; t < ntv; ++t) K_TILE_G(ISSUE_P, 1, 1, 1, 1, 8, 8, 10, 8);
    } else {
      for (int t = 0; t < ntv - 2; ++t) K_TILE_G(ISSUE_P, 1, 1, 1, 1, 8, 8, 10, 8);
      K_TILE_G(ISSUE_P, 1, 1, 0, 0, 8, 8, 8, 4);
      K_TILE_G(ISSUE_P, 0, 0, 0, 0, 2, 0, 0, 0);
    }
    if (!wr) BARRIER();                   // group 0 waits for group 1's last segment: all eight waves aligned
    __builtin_amdgcn_sched_barrier(0);

    // ---- stream-K unit: partial tile out, and the last arrival at the tile's counter sums the partials (see the top).
    // Coherence without cache flushes: the partial tiles are the ONLY data that cross workgroups, so they are written and read
    // with agent-scope accesses (sc1: coherent across the eight XCDs' L2s by themselves) and ordered by vmcnt(0) -> barrier ->
    // agent-scope atomic -> barrier -> loads.  A release / acquire FENCE pair instead (`__threadfence()`) writes back and
    // invalidates whole L2s -- full of this launch's C tiles -- once per unit: measured +100 us per launch.
    if constexpr (!X3 || F4) {
      if (ti >= my_dp) {                    // (uniform; a unit is always the workgroup's last work item: nothing is in flight)
        int lane_s = lane;                  // (opaque: otherwise the 32 slab addresses are computed at kernel entry and spilled)
        asm volatile("" : "+v"(lane_s));
        float* const mine = sk_slab + ((int64_t)(unit_tile * sk_split + unit_part) * 65536 + wave * 8192 + lane_s * 4);
#pragma unroll
        for (int f = 0; f < 32; ++f)
          asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(mine + f * 256), "v"(acc[f >> 4][(f >> 3) & 1][(f >> 1) & 3][f & 1]) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every partial of this wave has reached the coherence point
        __syncthreads();
        volatile unsigned* const flag = reinterpret_cast<volatile unsigned*>(smem + EP_BASE);
        if (threadIdx.x == 0) {
          const unsigned arrived = __hip_atomic_fetch_add(sk_count + unit_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (arrived == (unsigned)sk_split - 1u)            // all sk_split units have arrived: ready for the next launch
            __hip_atomic_store(sk_count + unit_tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          *flag = arrived;
        }
        __syncthreads();
        const bool last = *flag == (unsigned)sk_split - 1u;
        __syncthreads();                    // (wave 0's epilogue staging overwrites the flag)
        if (!last) break;
        asm volatile("" : "+v"(lane_s));
        const float* const all = sk_slab + ((int64_t)(unit_tile * sk_split) * 65536 + wave * 8192 + lane_s * 4);
        // part 0 lands in the accumulators themselves (dead since the store above); parts 1.. go through 16 temporaries at a time.
        // An asm statement = 16 loads + the wait for them: the compiler never sees a register with a load in flight.
#define SK_LD16(R, SRC)                                                                                                       \
        asm volatile("global_load_dwordx4 %0, %16, off sc1\n\tglobal_load_dwordx4 %1, %16, off offset:1024 sc1\n\t"         \
                     "global_load_dwordx4 %2, %16, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %16, off offset:3072 sc1\n\t" \
                     "global_load_dwordx4 %4, %17, off sc1\n\tglobal_load_dwordx4 %5, %17, off offset:1024 sc1\n\t"         \
                     "global_load_dwordx4 %6, %17, off offset:2048 sc1\n\tglobal_load_dwordx4 %7, %17, off offset:3072 sc1\n\t" \
                     "global_load_dwordx4 %8, %18, off sc1\n\tglobal_load_dwordx4 %9, %18, off offset:1024 sc1\n\t"         \
                     "global_load_dwordx4 %10, %18, off offset:2048 sc1\n\tglobal_load_dwordx4 %11, %18, off offset:3072 sc1\n\t" \
                     "global_load_dwordx4 %12, %19, off sc1\n\tglobal_load_dwordx4 %13, %19, off offset:1024 sc1\n\t"       \
                     "global_load_dwordx4 %14, %19, off offset:2048 sc1\n\tglobal_load_dwordx4 %15, %19, off offset:3072 sc1\n\t" \
                     "s_waitcnt vmcnt(0)"                                                                                     \
                     : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3]), "=&v"(R[4]), "=&v"(R[5]), "=&v"(R[6]), "=&v"(R[7]),     \
                       "=&v"(R[8]), "=&v"(R[9]), "=&v"(R[10]), "=&v"(R[11]), "=&v"(R[12]), "=&v"(R[13]), "=&v"(R[14]), "=&v"(R[15]) \
                     : "v"((SRC)), "v"((SRC) + 1024), "v"((SRC) + 2048), "v"((SRC) + 3072) : "memory")
        for (int part = 0; part < sk_split; ++part) {
          const float* const src = all + (int64_t)part * 65536;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x4 t[16];
            SK_LD16(t, src + h * 4096);
#pragma unroll
            for (int f = 0; f < 16; ++f) {
              f32x4& c = acc[h][(f >> 3) & 1][(f >> 1) & 3][f & 1];
              c = part == 0 ? t[f] : c + t[f];
            }
          }
        }
#undef SK_LD16
      }
    }

    // ---- epilogue (see the header of this kernel).  Its per-lane addresses are derived from an opaque copy of the lane id,
    // so that none of them is kept in a register (or spilled) across the K loop.
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int e16 = lane_e & 15, eg = lane_e >> 4;
    // write: row e16, chunk (8b + 4j + eg) ^ (e16 & 7)  (b -> address bit 7, j -> bit 6 of the swizzled chunk);
    // read pass `it`: row 4it + eg, chunk e16 ^ (row & 7)
    const unsigned wbase0 = lds0 + EP_BASE + wave * 4096 + e16 * 256 + ((eg ^ (e16 & 7)) << 4);
    const unsigned rbase0 = lds0 + EP_BASE + wave * 4096 + eg * 256;
    const int64_t ncol = n_blk + 64 * wc + 4 * e16;
    const unsigned lane_off = (unsigned)(eg * (int)ldc + 4 * e16) * 2u;       // row eg, columns 4 e16 .. +3 (bytes)
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (act != MMRCA_ACT_MUL) {       // (the input-gradient form has no bias; the dispatcher checks)
      if (bias) {
        if constexpr (X3) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(bias) + ncol);
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[r] = b4[r];
        } else {
          bf16x4 b4 = *reinterpret_cast<const bf16x4*>(bias + ncol);
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[r] = (float)b4[r];
        }
      }
    }
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool full = m_blk + 256 <= M;
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // the last MFMAs' results are read by asm stores below
    // (dbg & 2: timing-only ablation, no epilogue at all)
    // Side operand (residual addend, or the saved gelu' of ACT_MUL): all 32 row pieces of the tile are requested here, before
    // the tile's first store, by ORDINARY loads.  The compiler waits for an ordinary load that is in flight beside LDS-DMA
    // with vmcnt(0), and vmcnt retires in order: issued later, between the stores, every pass would wait for all earlier
    // stores and for the stream (51 such waits in the .s, 591 vs 870 TFLOP/s); issued up front there is ONE such wait per
    // tile, at the first use in pass 0, when only loads are outstanding.  (Asm loads with hand-counted waits were tried: at
    // this register pressure the compiler moved / re-used their destination registers before the data landed -- memory
    // fault.)  Like A, the side operand must have round_up(M, 256) readable rows; rows past M are not stored.
    constexpr bool SIDE = ADD || act == MMRCA_ACT_MUL;
    bf16x4 side[SIDE ? 32 : 1];
    if constexpr (SIDE) {
      const char* sp = reinterpret_cast<const char*>(ADD ? (const bf16_t*)addend : (const bf16_t*)preact);
      if (!(dbg & 2)) {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
          const int64_t urow = m_blk + 128 * (q >> 4) + 64 * wr + 4 * (q & 15);    // wave-uniform; this lane's row: urow + eg
          const int64_t ub = (urow * ldc + n_blk + 64 * wc) * 2;
          side[q] = *reinterpret_cast<const bf16x4*>(sp + ub + lane_off);
        }
      }
    }
    auto passes = [&](auto ragged_tag) {
      constexpr bool RAGGED = decltype(ragged_tag)::value;
  #pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int a = pass >> 2, i = pass & 3;
        {
          const unsigned wbase1 = wbase0 ^ 64u;         // fragment j = 1: chunk bit 2
          DS_WRITE_B128(wbase0, acc[a][0][i][0], 0);
          DS_WRITE_B128(wbase1, acc[a][0][i][1], 0);
          DS_WRITE_B128(wbase0, acc[a][1][i][0], 128);   // b = 1: chunk bit 3 (not touched by the 3-bit swizzle)
          DS_WRITE_B128(wbase1, acc[a][1][i][1], 128);
        }
        // global addresses = wave-uniform row base (scalar registers) + one per-lane byte offset for the whole tile
        // (per-row 64-bit multiplies made this epilogue VALU-bound: 4 us per tile)
        const int64_t urow0 = m_blk + 128 * a + 64 * wr + 16 * i;               // uniform; this lane's rows: urow0 + eg + 4it
        // rows are read back two at a time when a side operand occupies registers (four at a time otherwise)
        constexpr int RB = SIDE ? 2 : 4;
  #pragma unroll
        for (int it0 = 0; it0 < 4; it0 += RB) {
          f32x4 cr[RB];
  #pragma unroll
          for (int u = 0; u < RB; ++u) {
            const int it = it0 + u;
            // row 4it + eg: (row & 7) = (4it + eg) & 7
            const unsigned rb = rbase0 + it * 1024 + ((unsigned)(e16 ^ ((4 * it + eg) & 7)) << 4);
            DS_READ_B128(cr[u], rb, 0);
          }
          if constexpr (RB == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cr[0]), "+v"(cr[1]), "+v"(cr[2]), "+v"(cr[3]) :: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cr[0]), "+v"(cr[1]) :: "memory");
  #pragma unroll
          for (int u = 0; u < RB; ++u) {
            const int it = it0 + u;
            float v[4] = {cr[u][0], cr[u][1], cr[u][2], cr[u][3]};
            if constexpr (act != MMRCA_ACT_MUL) {
  #pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bv[r];
            }
            bf16x4 po;
            [[maybe_unused]] float pf[4] = {0.f, 0.f, 0.f, 0.f};
            constexpr bool store_pre = act == MMRCA_ACT_GELU_SAVE_GRAD;
            if constexpr (act == MMRCA_ACT_MUL) {
  #pragma unroll
              for (int r = 0; r < 4; ++r) v[r] *= (float)side[4 * pass + it][r];
            } else if constexpr (act == MMRCA_ACT_GELU_SAVE_GRAD) {
              float gr[4];
              gelu_and_grad_fast4(v, gr);
  #pragma unroll
              for (int r = 0; r < 4; ++r) { po[r] = (bf16_t)gr[r]; pf[r] = gr[r]; }
            } else if constexpr (act == MMRCA_ACT_GELU) {      // forward-only callers (the frozen BLIP-2 towers)
              gelu_fast4(v);
            }
            if constexpr (ADD) {
  #pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += (float)side[4 * pass + it][r];
            }
            bf16x4 o;
  #pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
            const bool ok = !RAGGED || urow0 + 4 * it + eg < M;
            if (ok) {
              if constexpr (act == MMRCA_ACT_MUL) {
  #pragma unroll
                for (int r = 0; r < 4; ++r) csum[r] += (float)o[r];
              }
              const int64_t ub = ((urow0 + 4 * it) * ldc + n_blk + 64 * wc) * 2;
              if constexpr (X3) {
                // fp32 outputs: 16 bytes per lane (a wave instruction writes four whole 256-byte row segments); the byte
                // offsets above are those of 2-byte elements
                if (store_pre) {
                  if (pre16) *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(preact) + ub + lane_off) = po;
                  else *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(preact) + 2 * (ub + lane_off)) = (f32x4){pf[0], pf[1], pf[2], pf[3]};
                }
                if constexpr (PLANES) {
                  bf16x4 lo;
#pragma unroll
                  for (int r = 0; r < 4; ++r) lo[r] = (bf16_t)(v[r] - (float)o[r]);
                  *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(C) + ub + lane_off) = o;
                  *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(C_lo) + ub + lane_off) = lo;
                } else {
                  *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(C) + 2 * (ub + lane_off)) = (f32x4){v[0], v[1], v[2], v[3]};
                }
              } else {
              if (store_pre) *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(preact) + ub + lane_off) = po;
              *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(C) + ub + lane_off) = o;
              }
            }
          }
        }
      }
    };
    if (!(dbg & 2)) {
      if (full) passes(std::false_type{}); else passes(std::true_type{});   // whole tiles: no per-row predicates at all
    }
    if constexpr (act == MMRCA_ACT_MUL) {
      if (colsum) {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(colsum + ncol + r, csum[r]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

extern int g_mmrca_dbg;
static int g_num_cus = 0;
static int g_p256_skew = -1;      // x 10 ns between the four start groups (MMRCA_P256_SKEW; see gemm_p256_k)

// ---- stream-K tail workspaces: caller-owned, one per stream that launches GEMMs (mmrca_gemm_streamk_workspace, include/mmrca.h)
#define SK_COUNTER_BYTES 4096          // 1,024 tile counters (at most 255 are used), zero between launches
#define SK_MAX_UNITS 256               // one unit per CU at most
struct SkWorkspace { int device; void* stream; char* base; int64_t bytes; };      // keyed on (device, stream): stream 0 exists on every device
static SkWorkspace g_sk_ws[32];
static int g_sk_ws_n = 0;
static std::mutex g_sk_mutex;
extern "C" int64_t mmrca_gemm_streamk_workspace_bytes(void) { return SK_COUNTER_BYTES + (int64_t)SK_MAX_UNITS * 65536 * 4; }
extern "C" int mmrca_gemm_streamk_workspace(void* workspace, int64_t bytes, void* stream) {
  MMRCA_REQUIRE(workspace == nullptr || (bytes >= mmrca_gemm_streamk_workspace_bytes() && (((uintptr_t)workspace) & 15) == 0),
                "gemm_streamk_workspace: needs %lld bytes, 16-byte aligned, zero-filled", (long long)mmrca_gemm_streamk_workspace_bytes());
  int dev = 0;
  (void)hipGetDevice(&dev);            // the workspace belongs to the CURRENT device (the caller's, as for every launch)
  std::lock_guard<std::mutex> lock(g_sk_mutex);
  for (int i = 0; i < g_sk_ws_n; ++i)
    if (g_sk_ws[i].stream == stream && g_sk_ws[i].device == dev) {
      if (workspace) { g_sk_ws[i].base = (char*)workspace; g_sk_ws[i].bytes = bytes; }
      else g_sk_ws[i] = g_sk_ws[--g_sk_ws_n];
      return 0;
    }
  if (!workspace) return 0;
  MMRCA_REQUIRE(g_sk_ws_n < 32, "gemm_streamk_workspace: more than 32 (device, stream) pairs registered");
  g_sk_ws[g_sk_ws_n++] = SkWorkspace{dev, stream, (char*)workspace, bytes};
  return 0;
}
static char* sk_workspace_of(void* stream) {
  if (g_sk_ws_n == 0) return nullptr;   // (unlocked fast path: nothing was ever registered)
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_sk_mutex);
  for (int i = 0; i < g_sk_ws_n; ++i)
    if (g_sk_ws[i].stream == stream && g_sk_ws[i].device == dev) return g_sk_ws[i].base;
  return nullptr;
}
static int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    g_num_cus = n & ~7;
  }
  return g_num_cus;
}
// how many K ranges the leftover tiles of a launch are cut into (0 / 1 = no stream-K tail): as many as fill the chip once,
// each at least two K steps long (the stream's pipeline), at most MMRCA_SK_MAX (default 4: the unit that finishes a tile reads
// all its partials, 256 KiB each, on ONE CU).  No workspace registered for the stream = no tail.
static int g_sk_max = -1, g_sk_min_ksteps = -1, g_sk_bf16 = -1;
static void sk_config_init() {
  // plain bf16 products: measured neutral at K >= 1536 and slower below (profiles/r06_streamk_ab.txt) -> off unless asked for;
  // the fused bf16x3 form does three times the matrix work per tile for the same partial-tile traffic: +1.3 % on the compliant step
  if (g_sk_bf16 < 0) g_sk_bf16 = getenv("MMRCA_SK_BF16") ? atoi(getenv("MMRCA_SK_BF16")) : 0;
  if (g_sk_max < 0) g_sk_max = getenv("MMRCA_SK_MAX") ? atoi(getenv("MMRCA_SK_MAX")) : 4;
  // (the partial tiles are 2 x 256 KiB of HBM traffic per unit whatever K is: below this many K steps the round-5 split -- whole
  // rounds here, the rest on the 128x128 kernel -- is faster; tools/streamk_bench.py, profiles/r06_streamk_ab.txt)
  if (g_sk_min_ksteps < 0) g_sk_min_ksteps = getenv("MMRCA_SK_MIN_KSTEPS") ? atoi(getenv("MMRCA_SK_MIN_KSTEPS")) : 24;
}
extern "C" int mmrca_gemm_streamk_config(int max_split, int min_ksteps, int bf16_products) {
  sk_config_init();
  if (max_split >= 0) g_sk_max = max_split;
  if (min_ksteps >= 0) g_sk_min_ksteps = min_ksteps;
  if (bf16_products >= 0) g_sk_bf16 = bf16_products;
  return 0;
}
int mmrca_gemm256_streamk_split(int64_t M, int64_t N, int64_t ksteps, void* stream, bool x3) {
  sk_config_init();
  const int sk_max = g_sk_max;
  if ((!x3 && !g_sk_bf16) || ksteps < g_sk_min_ksteps || !sk_workspace_of(stream)) return 0;
  const int64_t tiles = ((M + 255) / 256) * (N / 256);
  const int ncu = num_cus();
  if (tiles < ncu) return 0;             // (fewer tiles than CUs, ranges over the whole chip: measured slower than the 128x128 kernels,
                                         //  profiles/r06_text_gemm_shapes.txt)
  const int64_t left = tiles % ncu;
  if (left == 0) return 0;
  int64_t split = ncu / left;
  if (split > ksteps / 2) split = ksteps / 2;
  if (split > sk_max) split = sk_max;
  return split >= 2 ? (int)split : 0;
}
template <bool AK, bool BK2, int ACT, bool ADD, bool X3 = false, bool PLANES = false, bool F4 = false>
static void launch_p256(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact, int64_t M,
                        int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float* colsum, hipStream_t st,
                        const void* A_lo = nullptr, const void* B_lo = nullptr, void* C_lo = nullptr, int nseg = 3, int pre16 = 0) {
  const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)(N / 256);
  (void)num_cus();
  if (g_p256_skew < 0) {
    const char* e = getenv("MMRCA_P256_SKEW");
    g_p256_skew = e ? atoi(e) : 0;
  }
  int grid = tiles_m * tiles_n < g_num_cus ? ((tiles_m * tiles_n) & ~7) : g_num_cus;
  if (grid < 8) grid = 8;
  constexpr int LDS_P = 2 * STAGE_BYTES + 32768;       // all 160 KiB of the CU
  // stream-K tail (bf16 and the fused four-plane form; the three-pass bf16x3 form walks its K loop per plane pair)
  int sk_split = 0;
  char* sk_ws = nullptr;
  if constexpr (!X3 || F4) {
    sk_split = mmrca_gemm256_streamk_split(M, N, K / (F4 ? 32 : 64), (void*)st, X3);
    if (sk_split >= 2) sk_ws = sk_workspace_of((void*)st);
    if (!sk_ws) sk_split = 0;
  }
  MMRCA_MAX_LDS(LDS_P, gemm_p256_k<AK, BK2, ACT, ADD, X3, PLANES, F4>);
  hipLaunchKernelGGL((gemm_p256_k<AK, BK2, ACT, ADD, X3, PLANES, F4>), dim3(grid), dim3(512), LDS_P, st, (const bf16_t*)A, (const bf16_t*)B,
                     (bf16_t*)C, (const bf16_t*)bias, (const bf16_t*)addend, (bf16_t*)preact, M, N, K, lda, ldb, ldc, tiles_m,
                     tiles_n, colsum, tiles_m * tiles_n >= 2 * g_num_cus ? g_p256_skew : 0, g_mmrca_dbg, (const bf16_t*)A_lo,
                     (const bf16_t*)B_lo, (bf16_t*)C_lo, nseg, pre16, (float*)(sk_ws ? sk_ws + SK_COUNTER_BYTES : nullptr),
                     (unsigned*)sk_ws, sk_split);
}

int g_mmrca_dbg = 0;
extern "C" int mmrca_debug_set(int v) { g_mmrca_dbg = v; return 0; }

// called by mmrca_gemm (gemm.hip) for bf16-out GEMMs that qualify: A row-major (forward / input gradient), the epilogues
// the engine uses (see gemm_p256_k), operands below 4 GiB (32-bit stream offsets)
bool mmrca_gemm256_ok(int64_t M, int64_t N, int64_t K, int a_layout, int act, bool has_addend, bool has_preact, bool has_colsum, bool has_bias) {
  if (!(N % 256 == 0 && K % 64 == 0 && K >= 128 && a_layout == MMRCA_ROWK)) return false;
  if (act == MMRCA_ACT_NONE) return !has_preact && !has_colsum;
  if (act == MMRCA_ACT_GELU_SAVE_GRAD) return has_preact && !has_addend && !has_colsum;
  if (act == MMRCA_ACT_GELU) return !has_preact && !has_addend && !has_colsum;
  if (act == MMRCA_ACT_MUL) return has_preact && !has_addend && !has_bias;
  return false;
}

int mmrca_gemm256(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact, int64_t M,
                  int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_layout, int b_layout, int act,
                  float* colsum, hipStream_t st) {
  const bool bk = b_layout == MMRCA_KROW;
  MMRCA_REQUIRE(mmrca_gemm256_ok(M, N, K, a_layout, act, addend != nullptr, preact != nullptr, colsum != nullptr, bias != nullptr),
                "gemm(mfma256): shape / epilogue combination not built");
  MMRCA_REQUIRE(M * lda * 2 < (1ll << 32) && (bk ? K * ldb : N * ldb) * 2 < (1ll << 32) && M * ldc * 2 < (1ll << 32),
                "gemm(mfma256): operands must be smaller than 4 GiB");
#define L256(BK_, MODE_, ADD_) launch_p256<false, BK_, MODE_, ADD_>(A, B, C, bias, addend, preact, M, N, K, lda, ldb, ldc, colsum, st)
  if (act == MMRCA_ACT_GELU_SAVE_GRAD) {
    if (bk) L256(true, MMRCA_ACT_GELU_SAVE_GRAD, false); else L256(false, MMRCA_ACT_GELU_SAVE_GRAD, false);
  } else if (act == MMRCA_ACT_GELU) {
    if (bk) L256(true, MMRCA_ACT_GELU, false); else L256(false, MMRCA_ACT_GELU, false);
  } else if (act == MMRCA_ACT_MUL) {
    if (bk) L256(true, MMRCA_ACT_MUL, false); else L256(false, MMRCA_ACT_MUL, false);
  } else if (addend) {
    if (bk) L256(true, MMRCA_ACT_NONE, true); else L256(false, MMRCA_ACT_NONE, true);
  } else {
    if (bk) L256(true, MMRCA_ACT_NONE, false); else L256(false, MMRCA_ACT_NONE, false);
  }
#undef L256
  MMRCA_CHECK_LAUNCH("gemm(mfma256)");
  return 0;
}


// bf16x3 forms (called by gemm_x3.hip): persistent forward / input-gradient kernel with fp32 or two-plane output
bool mmrca_gemm256_x3_ok(int64_t M, int64_t N, int64_t K, int a_layout, int act, bool has_addend, bool has_colsum) {
  if (!(N % 256 == 0 && K % 64 == 0 && K >= 128 && a_layout == MMRCA_ROWK) || has_addend || has_colsum) return false;
  return act == MMRCA_ACT_NONE || act == MMRCA_ACT_GELU_SAVE_GRAD;
}

static bool x3_fused_on() {
  static const int x3_fused = getenv("MMRCA_X3_FUSED") ? atoi(getenv("MMRCA_X3_FUSED")) : 1;
  return x3_fused != 0;
}
// stream-K ranges per leftover tile of a bf16x3 launch (0 = none): only the fused four-plane form walks one K loop per tile
int mmrca_gemm256_x3_streamk_split(int64_t M, int64_t N, int64_t K, int b_layout, bool has_a_lo, bool has_b_lo, void* stream) {
  if (!(x3_fused_on() && b_layout != MMRCA_KROW && has_a_lo && has_b_lo && K % 32 == 0 && K >= 64)) return 0;
  return mmrca_gemm256_streamk_split(M, N, K / 32, stream, true);
}

int mmrca_gemm256_x3(const void* A_hi, const void* A_lo, const void* B_hi, const void* B_lo, void* C, void* C_lo, const void* bias,
                     void* preact, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int b_layout, int act,
                     hipStream_t st, int pre16) {
  const bool bk = b_layout == MMRCA_KROW;
  MMRCA_REQUIRE(M * lda * 2 < (1ll << 32) && (bk ? K * ldb : N * ldb) * 2 < (1ll << 32) && M * ldc * 4 < (1ll << 32),
                "gemm_x3(mfma256): operands must be smaller than 4 GiB");
  // the fused four-plane form (gemm_p256_k<..., F4>): row-major B, all three plane pairs, K % 32 == 0 (MMRCA_X3_FUSED=0: the three-pass form)
  if (x3_fused_on() && !bk && A_lo && B_lo && K % 32 == 0 && K >= 64) {
#define L256F(ACT_, PL_) launch_p256<false, false, ACT_, false, true, PL_, true>(A_hi, B_hi, C, bias, nullptr, preact, M, N, K, lda, ldb, ldc, nullptr, st, A_lo, B_lo, C_lo, 3, pre16)
    if (act == MMRCA_ACT_GELU_SAVE_GRAD) { if (C_lo) L256F(MMRCA_ACT_GELU_SAVE_GRAD, true); else L256F(MMRCA_ACT_GELU_SAVE_GRAD, false); }
    else { if (C_lo) L256F(MMRCA_ACT_NONE, true); else L256F(MMRCA_ACT_NONE, false); }
#undef L256F
    MMRCA_CHECK_LAUNCH("gemm_x3(mfma256,fused)");
    return 0;
  }
#define L256X(BK_, ACT_, PL_) launch_p256<false, BK_, ACT_, false, true, PL_>(A_hi, B_hi, C, bias, nullptr, preact, M, N, K, lda, ldb, ldc, nullptr, st, A_lo, B_lo, C_lo, A_lo ? (B_lo ? 3 : 2) : 1, pre16)
  if (act == MMRCA_ACT_GELU_SAVE_GRAD) {
    if (C_lo) { if (bk) L256X(true, MMRCA_ACT_GELU_SAVE_GRAD, true); else L256X(false, MMRCA_ACT_GELU_SAVE_GRAD, true); }
    else { if (bk) L256X(true, MMRCA_ACT_GELU_SAVE_GRAD, false); else L256X(false, MMRCA_ACT_GELU_SAVE_GRAD, false); }
  } else {
    if (C_lo) { if (bk) L256X(true, MMRCA_ACT_NONE, true); else L256X(false, MMRCA_ACT_NONE, true); }
    else { if (bk) L256X(true, MMRCA_ACT_NONE, false); else L256X(false, MMRCA_ACT_NONE, false); }
  }
#undef L256X
  MMRCA_CHECK_LAUNCH("gemm_x3(mfma256)");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-K weight gradient on the 256x256 tile: C[M,N] (fp32) += A^T-contracted product over K, partial tiles through a
// caller-owned workspace.  One block per CU: (M/256)*(N/256) tiles x `splits` K ranges <= 256 blocks.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
splitk_reduce256_k(const float* __restrict__ slab, float* __restrict__ C, int64_t ldc, int ntiles, int tiles_n, int splits, int64_t M, int64_t N) {
  const int tile = blockIdx.x >> 6;
  const int idx = ((blockIdx.x & 63) << 8) + threadIdx.x;          // which f32x4 of the tile, accumulator order
  const float* src = slab + (int64_t)tile * 65536 + (int64_t)idx * 4;
  f32x4 s = *reinterpret_cast<const f32x4*>(src);
  for (int k = 1; k < splits; ++k) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)k * ntiles * 65536);
    s += v;
  }
  const int w = idx >> 11, f = (idx >> 6) & 31, lane = idx & 63;
  const int a = f >> 4, b = (f >> 3) & 1, i = (f >> 1) & 3, j = f & 1;
  const int wr = w >> 2, wc = w & 3, l16 = lane & 15, g = lane >> 4;
  const int row = 128 * a + 64 * wr + 16 * i + l16, col = 128 * b + 32 * wc + 16 * j + 4 * g;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  if ((int64_t)tm * 256 + row >= M || (int64_t)tn * 256 + col >= N) return;     // ragged edge tiles (N % 4 == 0: whole float4s)
  float* dst = C + ((int64_t)tm * 256 + row) * ldc + (int64_t)tn * 256 + col;
  f32x4 c = *reinterpret_cast<f32x4*>(dst);
  c += s;
  *reinterpret_cast<f32x4*>(dst) = c;
}

extern "C" int64_t mmrca_gemm_splitk_workspace_bytes(int64_t M, int64_t N) {
  if (M <= 0 || N <= 0 || M % 8 || N % 8) return 0;
  const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
  if (tiles > 256) return 0;
  return (256 / tiles) * tiles * 65536 * 4;
}

static int gemm_splitk_impl(const void* A, const void* A_lo, const void* B, const void* B_lo, float* C, void* workspace,
                           int64_t workspace_bytes, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                           int a_layout, int b_layout, void* stream) {
  const bool x3 = A_lo != nullptr;
  const int nseg = x3 ? (B_lo ? 3 : 2) : 1;
  MMRCA_REQUIRE(A && B && C && workspace, "gemm_splitk: null operand");
  // M, N: multiples of 256, or -- both operands K-major (the weight-gradient layout), plain bf16 -- any multiples of 8: edge tiles
  // then fetch the chunks past the last column from column 0 (piece_src) and the reduction skips what lies outside C
  const bool ragged = M % 256 != 0 || N % 256 != 0;
  MMRCA_REQUIRE(M > 0 && N > 0 && M % 8 == 0 && N % 8 == 0 && K >= 128 && K % 64 == 0 &&
                (!ragged || (a_layout == MMRCA_KROW && b_layout == MMRCA_KROW && !A_lo)),
                "gemm_splitk: needs K %% 64 == 0, K >= 128 and M, N multiples of 256 (K-major bf16 operands: of 8) (got M=%lld N=%lld K=%lld)",
                (long long)M, (long long)N, (long long)K);
  MMRCA_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)workspace) & 15) == 0,
                "gemm_splitk: operands must be 16-byte aligned");
  MMRCA_REQUIRE(lda >= (a_layout == MMRCA_ROWK ? K : M) && ldb >= (b_layout == MMRCA_ROWK ? K : N) && ldc >= N, "gemm_splitk: leading dimension too small");
  const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + 255) / 256), tiles = tiles_m * tiles_n;
  MMRCA_REQUIRE(tiles <= 256, "gemm_splitk: more than 256 output tiles (use mmrca_gemm)");
  const int64_t ksteps = (int64_t)nseg * (K / 64);   // bf16x3: the virtual contraction [A_hi|A_lo|A_hi] . [B_hi|B_hi|B_lo]
  int64_t splits64 = 256 / tiles;                   // one workgroup per CU
  if (splits64 > ksteps / 2) splits64 = ksteps / 2; // the kernel's pipeline needs two K-tiles per range
  const int splits = (int)splits64;
  const int ksteps_base = (int)(ksteps / splits), ksteps_rem = (int)(ksteps % splits);   // the first `rem` ranges take one more
  MMRCA_REQUIRE(workspace_bytes >= (int64_t)splits * tiles * 65536 * 4, "gemm_splitk: workspace too small (%lld < %lld bytes)",
                (long long)workspace_bytes, (long long)splits * tiles * 65536 * 4);
  hipStream_t st = (hipStream_t)stream;
  const bool ak = a_layout == MMRCA_KROW, bk = b_layout == MMRCA_KROW;
#define LSLAB(AK_, BK_, X3_)                                                                                                     \
  do {                                                                                                                           \
    MMRCA_MAX_LDS(LDS256_BYTES, gemm_mfma256_k<AK_, BK_, MODE_SLAB, X3_>); \
    hipLaunchKernelGGL((gemm_mfma256_k<AK_, BK_, MODE_SLAB, X3_>), dim3((tiles * splits + 7) / 8 * 8), dim3(512), LDS256_BYTES, st, (const bf16_t*)A,      \
                       (const bf16_t*)B, (bf16_t*)nullptr, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (bf16_t*)nullptr, M, N, K, \
                       lda, ldb, ldc, tiles_m, tiles_n, ksteps_base, ksteps_rem, (float*)workspace, (float*)nullptr, splits,    \
                       (const bf16_t*)A_lo, (const bf16_t*)B_lo, nseg);                                                         \
  } while (0)
  if (x3) {
    if (!ak && !bk) LSLAB(false, false, true);
    else if (!ak && bk) LSLAB(false, true, true);
    else if (ak && !bk) LSLAB(true, false, true);
    else LSLAB(true, true, true);
  } else {
    if (!ak && !bk) LSLAB(false, false, false);
    else if (!ak && bk) LSLAB(false, true, false);
    else if (ak && !bk) LSLAB(true, false, false);
    else LSLAB(true, true, false);
  }
#undef LSLAB
  MMRCA_CHECK_LAUNCH("gemm_splitk(mfma256)");
  hipLaunchKernelGGL(splitk_reduce256_k, dim3(tiles * 64), dim3(256), 0, st, (const float*)workspace, C, ldc, tiles, tiles_n, splits, M, N);
  MMRCA_CHECK_LAUNCH("gemm_splitk(reduce)");
  return 0;
}

extern "C" int mmrca_gemm_splitk(const void* A, const void* B, float* C, void* workspace, int64_t workspace_bytes, int64_t M,
                                 int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_layout, int b_layout,
                                 void* stream) {
  return gemm_splitk_impl(A, nullptr, B, nullptr, C, workspace, workspace_bytes, M, N, K, lda, ldb, ldc, a_layout, b_layout, stream);
}

extern "C" int mmrca_gemm_splitk_x3(const void* A_hi, const void* A_lo, const void* B_hi, const void* B_lo, float* C, void* workspace,
                                    int64_t workspace_bytes, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                    int a_layout, int b_layout, void* stream) {
  // B_lo == NULL: two passes (A_hi B_hi + A_lo B_hi); A_lo == NULL as well: one pass = mmrca_gemm_splitk on the hi planes
  MMRCA_REQUIRE(A_lo || !B_lo, "gemm_x3(splitk): a B lo plane without an A lo plane is not a supported pass set");
  return gemm_splitk_impl(A_hi, A_lo, B_hi, B_lo, C, workspace, workspace_bytes, M, N, K, lda, ldb, ldc, a_layout, b_layout, stream);
}
