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

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
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

// ---- LDS fragment reads (inline asm, see header) ------------------------------------------------------------------
#define DS_READ_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define DS_READ_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))

// One 16(rows) x 32(k) bf16 operand fragment in registers.  ROWK: one 16-byte read from base[ks]; KROW: two transposed
// 8-byte reads from base[frag] (k rows +0..3 and +4..7) kept as two register pairs until the lgkmcnt wait has passed --
// they are only joined into the MFMA operand afterwards, so no compiler-made copy can read them before the data lands.
template <bool KROW> struct Frag;
template <> struct Frag<false> { bf16x8 v; };
template <> struct Frag<true> { bf16x4 lo, hi; };

template <bool KROW, int OFF>
__device__ __forceinline__ void read_frag(Frag<KROW>& f, unsigned base) {
  if constexpr (!KROW) {
    DS_READ_B128(f.v, base, OFF);
  } else {
    DS_READ_TR(f.lo, base, OFF);
    DS_READ_TR(f.hi, base, OFF + 1024);
  }
}
__device__ __forceinline__ bf16x8 frag_val(const Frag<false>& f) { return f.v; }
__device__ __forceinline__ bf16x8 frag_val(const Frag<true>& f) { return __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7); }

// s_waitcnt lgkmcnt(0) with the fragment registers as read-write operands: nothing that uses them is scheduled above it
__device__ __forceinline__ void lgkm0(Frag<false> (&f)[4][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0].v), "+v"(f[0][1].v), "+v"(f[1][0].v), "+v"(f[1][1].v), "+v"(f[2][0].v),
               "+v"(f[2][1].v), "+v"(f[3][0].v), "+v"(f[3][1].v) :: "memory");
}
__device__ __forceinline__ void lgkm0(Frag<true> (&f)[4][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0].lo), "+v"(f[0][0].hi), "+v"(f[0][1].lo), "+v"(f[0][1].hi), "+v"(f[1][0].lo),
               "+v"(f[1][0].hi), "+v"(f[1][1].lo), "+v"(f[1][1].hi), "+v"(f[2][0].lo), "+v"(f[2][0].hi), "+v"(f[2][1].lo),
               "+v"(f[2][1].hi), "+v"(f[3][0].lo), "+v"(f[3][0].hi), "+v"(f[3][1].lo), "+v"(f[3][1].hi) :: "memory");
}
__device__ __forceinline__ void lgkm0(Frag<false> (&f)[2][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0].v), "+v"(f[0][1].v), "+v"(f[1][0].v), "+v"(f[1][1].v) :: "memory");
}
__device__ __forceinline__ void lgkm0(Frag<true> (&f)[2][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0].lo), "+v"(f[0][0].hi), "+v"(f[0][1].lo), "+v"(f[0][1].hi), "+v"(f[1][0].lo),
               "+v"(f[1][0].hi), "+v"(f[1][1].lo), "+v"(f[1][1].hi) :: "memory");
}

// stage one 128x64 half-tile: 16 wave-instructions of 1 KiB, two per wave.  src0/src1: this lane's source pointers of the
// wave's two pieces for the CURRENT K-tile of that half (advanced by the caller)
__device__ __forceinline__ void stage_half(const bf16_t* src0, const bf16_t* src1, char* lds_piece0) {
  __builtin_amdgcn_global_load_lds((gbl_void*)src0, (lds_void*)(lds_piece0), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gbl_void*)src1, (lds_void*)(lds_piece0 + 1024), 16, 0, 0);
}

// per-lane source pointer of piece `i` (0..15) of a half-tile at K offset 0
template <bool KROW>
__device__ __forceinline__ const bf16_t* piece_src(const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t rows_total,
                                                   int i, int lane) {
  if (!KROW) {
    const int r = 8 * i + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int64_t gr = row0 + r;
    if (gr > rows_total - 1) gr = rows_total - 1;
    return base + gr * ld + c * 8;
  } else {
    const int kr = 4 * i + (lane >> 4);
    const int chp = lane & 15;
    const int c = ((((chp >> 1) ^ krow_f2(kr))) << 1) | (chp & 1);
    return base + (int64_t)kr * ld + row0 + c * 8;
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
template <bool A_KROW, bool B_KROW, int MODE>
__global__ void __launch_bounds__(512, 2)
gemm_mfma256_k(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, const bf16_t* __restrict__ bias,
               const bf16_t* __restrict__ addend, bf16_t* __restrict__ preact, int64_t M, int64_t N, int64_t K,
               int64_t lda, int64_t ldb, int64_t ldc, int tiles_m, int tiles_n, int ksteps_base, int ksteps_rem,
               float* __restrict__ slab, float* __restrict__ colsum) {
  constexpr bool SLAB = MODE == MODE_SLAB;
  constexpr int act = SLAB ? 0 : MODE;
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
  // split-K: range y covers ksteps_base (+1 for the first ksteps_rem ranges) K-tiles
  const int yb = (int)blockIdx.y;
  const int64_t kbeg = SLAB ? 64 * ((int64_t)yb * ksteps_base + (yb < ksteps_rem ? yb : ksteps_rem)) : 0;
  const int nt = SLAB ? ksteps_base + (yb < ksteps_rem ? 1 : 0) : (int)(K / 64);     // >= 2 (host-checked)

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
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      pa[h][ii] = piece_src<A_KROW>(A, lda, m_blk + 128 * h, M, wave * 2 + ii, lane) + (A_KROW ? kbeg * lda : kbeg);
      pb[h][ii] = piece_src<B_KROW>(B, ldb, n_blk + 128 * h, N, wave * 2 + ii, lane) + (B_KROW ? kbeg * ldb : kbeg);
    }
  char* const my_piece = smem + wave * 2048;              // this wave's first piece inside a half-tile
  int issue_stage = 0;                                     // LDS stage (byte offset) of the next issue group's K-tile
  // issue order within a K-tile: A0, B0, B1, A1 (order of first use); `which` is a compile-time constant at every call
#define ISSUE(which)                                                                                            \
  do {                                                                                                          \
    if ((which) == 0) { stage_half(pa[0][0], pa[0][1], my_piece + issue_stage + SLOT_A0 * HT_BYTES); pa[0][0] += a_step; pa[0][1] += a_step; } \
    else if ((which) == 1) { stage_half(pb[0][0], pb[0][1], my_piece + issue_stage + SLOT_B0 * HT_BYTES); pb[0][0] += b_step; pb[0][1] += b_step; } \
    else if ((which) == 2) { stage_half(pb[1][0], pb[1][1], my_piece + issue_stage + SLOT_B1 * HT_BYTES); pb[1][0] += b_step; pb[1][1] += b_step; } \
    else { stage_half(pa[1][0], pa[1][1], my_piece + issue_stage + SLOT_A1 * HT_BYTES); pa[1][0] += a_step; pa[1][1] += a_step; issue_stage ^= STAGE_BYTES; } \
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
#define K_TILE(I0, I1, I2, I3, W0, W1, W2, W3)                                                                \
  do {                                                                                                        \
    if (I0) ISSUE(2);                                                                                         \
    LOAD_B(b0f, SLOT_B0); LOAD_A(SLOT_A0);                                                                    \
    END_LOAD(W0);                                                                                             \
    LGKM0_B(b0f); LGKM0_A();                                                                                  \
    MFMA_QUAD(0, 0, b0f);                                                                                     \
    END_COMPUTE(W0);                                                                                          \
    if (I1) ISSUE(3);                                                                                         \
    LOAD_B(b1f, SLOT_B1);                                                                                     \
    END_LOAD(W1);                                                                                             \
    LGKM0_B(b1f);                                                                                             \
    MFMA_QUAD(0, 1, b1f);                                                                                     \
    END_COMPUTE(W1);                                                                                          \
    if (I2) ISSUE(0);                                                                                         \
    LOAD_A(SLOT_A1);                                                                                          \
    END_LOAD(W2);                                                                                             \
    LGKM0_A();                                                                                                \
    MFMA_QUAD(1, 1, b1f);                                                                                     \
    END_COMPUTE(W2);                                                                                          \
    if (I3) ISSUE(1);                                                                                         \
    END_LOAD(W3);                                                                                             \
    MFMA_QUAD(1, 0, b0f);                                                                                     \
    END_COMPUTE(W3);                                                                                          \
    _Pragma("unroll") for (int x = 0; x < 4; ++x) abase[x] ^= STAGE_BYTES;                                    \
    bbase[0] ^= STAGE_BYTES; bbase[1] ^= STAGE_BYTES;                                                         \
  } while (0)

  // prologue: seq 0..5 (K-tile 0 and A0, B0 of K-tile 1); phase 0 needs seq 0, 1 -> four half-tiles may stay in flight
  ISSUE(0); ISSUE(1); ISSUE(2); ISSUE(3); ISSUE(0); ISSUE(1);
  VMCNT(8);
  BARRIER();
  if (wr) BARRIER();                    // group 1 runs one barrier behind
  __builtin_amdgcn_sched_barrier(0);

  for (int t = 0; t < nt - 2; ++t) K_TILE(1, 1, 1, 1, 8, 8, 10, 8);
  K_TILE(1, 1, 0, 0, 8, 8, 8, 4);       // K-tile nt-2: issues B1, A1 of the last K-tile
  K_TILE(0, 0, 0, 0, 2, 0, 0, 0);       // K-tile nt-1
  if (!wr) BARRIER();                   // group 0 waits for group 1's last segment
#undef K_TILE
#undef END_LOAD
#undef END_COMPUTE
#undef MFMA_QUAD
#undef LOAD_A
#undef LOAD_B
#undef ISSUE

  if constexpr (SLAB) {
    float* dst = slab + ((int64_t)blockIdx.y * nwg + (int64_t)tm * tiles_n + tn) * 65536 + wave * 8192 + lane * 4;
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

  // Epilogue through LDS: the MFMA layout gives a lane 4 consecutive columns of ONE row, i.e. 32-byte pieces of 16
  // different rows per store instruction (measured: 2.1 TB/s, fully exposed at one block per CU).  Instead each 128-row
  // half of the tile goes to LDS as fp32 (rows padded by 16 B: conflict-free 16-byte writes), and is read back one
  // whole 256-column row per wave instruction, so bias / GELU / addend are applied in fp32 and every global access
  // (pre-activation store, addend load, output store) is a contiguous 512-byte row segment.
  constexpr int EP_STRIDE = 256 * 4 + 16;
  const int64_t ncol = n_blk + lane * 4;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
    bf16x4 b4 = *reinterpret_cast<const bf16x4*>(bias + ncol);
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (float)b4[r];
  }
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
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
    // the wave's 16 rows of this half: all LDS reads and all side-operand loads are issued before the first use, so the
    // pass has 16 independent row streams in flight (as a rolled loop it was one latency chain per row: ~10 us per tile)
    const bool full = m_blk + 256 <= M;
    f32x4 cr[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) cr[rr] = *reinterpret_cast<const f32x4*>(smem + (wave * 16 + rr) * EP_STRIDE + lane * 16);
    const int64_t m0 = m_blk + 128 * a + wave * 16;
    constexpr bool need_h = act == MMRCA_ACT_MUL;
    bf16x4 hr[16], ar[16];
    if constexpr (need_h) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int64_t m = (full || m0 + rr < M) ? m0 + rr : M - 1;
        hr[rr] = *reinterpret_cast<const bf16x4*>(preact + m * ldc + ncol);
      }
    }
    if (addend) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int64_t m = (full || m0 + rr < M) ? m0 + rr : M - 1;
        ar[rr] = *reinterpret_cast<const bf16x4*>(addend + m * ldc + ncol);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int64_t m = m0 + rr;
      float v[4] = {cr[rr][0] + bv[0], cr[rr][1] + bv[1], cr[rr][2] + bv[2], cr[rr][3] + bv[3]};
      bf16x4 po;
      bool store_pre = false;
      if constexpr (act == MMRCA_ACT_MUL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= (float)hr[rr][r];
      } else if constexpr (act == MMRCA_ACT_GELU_SAVE_GRAD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float gr;
          v[r] = gelu_and_grad_fast_f(v[r], &gr);
          po[r] = (bf16_t)gr;
        }
        store_pre = true;
      } else if (preact) {
#pragma unroll
        for (int r = 0; r < 4; ++r) po[r] = (bf16_t)v[r];
        store_pre = true;
      }
      if (addend) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)ar[rr][r];
      }
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
      if (full || m < M) {
#pragma unroll
        for (int r = 0; r < 4; ++r) csum[r] += (float)o[r];       // column sums of what is STORED (the rounded values)
        if (store_pre) *reinterpret_cast<bf16x4*>(preact + m * ldc + ncol) = po;
        *reinterpret_cast<bf16x4*>(C + m * ldc + ncol) = o;
      }
    }
    __syncthreads();
  }
  if (colsum) {      // lane owns columns ncol..ncol+3 over this wave's 32 rows: 8 partial sums per column per block
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(colsum + ncol + r, csum[r]);
  }
}

int g_mmrca_dbg = 0;
extern "C" int mmrca_debug_set(int v) { g_mmrca_dbg = v; return 0; }

template <bool AK, bool BK2, int MODE>
static void launch256(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact, int64_t M,
                      int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float* colsum, hipStream_t st) {
  const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)(N / 256);
  (void)hipFuncSetAttribute((const void*)gemm_mfma256_k<AK, BK2, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS256_BYTES);
  hipLaunchKernelGGL((gemm_mfma256_k<AK, BK2, MODE>), dim3(tiles_m * tiles_n), dim3(512), LDS256_BYTES, st, (const bf16_t*)A,
                     (const bf16_t*)B, (bf16_t*)C, (const bf16_t*)bias, (const bf16_t*)addend, (bf16_t*)preact, M, N, K, lda,
                     ldb, ldc, tiles_m, tiles_n, 0, 0, (float*)nullptr, colsum);
}

// called by mmrca_gemm (gemm.hip) for bf16-out GEMMs that qualify.  Epilogues built: no activation for every layout pair;
// GELU_SAVE_GRAD and MUL for a ROWK A operand (forward / input gradient).
bool mmrca_gemm256_ok(int64_t M, int64_t N, int64_t K, int a_layout, int act) {
  return N % 256 == 0 && K % 64 == 0 && K >= 128 && (a_layout == MMRCA_ROWK || M % 256 == 0) &&
         (act == MMRCA_ACT_NONE || (a_layout == MMRCA_ROWK && (act == MMRCA_ACT_GELU_SAVE_GRAD || act == MMRCA_ACT_MUL)));
}

int mmrca_gemm256(const void* A, const void* B, void* C, const void* bias, const void* addend, void* preact, int64_t M,
                  int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_layout, int b_layout, int act,
                  float* colsum, hipStream_t st) {
  const bool ak = a_layout == MMRCA_KROW, bk = b_layout == MMRCA_KROW;
#define L256(AK_, BK_, MODE_) launch256<AK_, BK_, MODE_>(A, B, C, bias, addend, preact, M, N, K, lda, ldb, ldc, colsum, st)
  if (ak) {
    if (bk) L256(true, true, MMRCA_ACT_NONE); else L256(true, false, MMRCA_ACT_NONE);
  } else if (act == MMRCA_ACT_GELU_SAVE_GRAD) {
    if (bk) L256(false, true, MMRCA_ACT_GELU_SAVE_GRAD); else L256(false, false, MMRCA_ACT_GELU_SAVE_GRAD);
  } else if (act == MMRCA_ACT_MUL) {
    if (bk) L256(false, true, MMRCA_ACT_MUL); else L256(false, false, MMRCA_ACT_MUL);
  } else {
    if (bk) L256(false, true, MMRCA_ACT_NONE); else L256(false, false, MMRCA_ACT_NONE);
  }
#undef L256
  MMRCA_CHECK_LAUNCH("gemm(mfma256)");
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// Split-K weight gradient on the 256x256 tile: C[M,N] (fp32) += A^T-contracted product over K, partial tiles through a
// caller-owned workspace.  One block per CU: (M/256)*(N/256) tiles x `splits` K ranges <= 256 blocks.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
splitk_reduce256_k(const float* __restrict__ slab, float* __restrict__ C, int64_t ldc, int ntiles, int tiles_n, int splits) {
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
  float* dst = C + ((int64_t)tm * 256 + row) * ldc + (int64_t)tn * 256 + col;
  f32x4 c = *reinterpret_cast<f32x4*>(dst);
  c += s;
  *reinterpret_cast<f32x4*>(dst) = c;
}

extern "C" int64_t mmrca_gemm_splitk_workspace_bytes(int64_t M, int64_t N) {
  if (M <= 0 || N <= 0 || M % 256 || N % 256) return 0;
  const int64_t tiles = (M / 256) * (N / 256);
  if (tiles > 256) return 0;
  return (256 / tiles) * tiles * 65536 * 4;
}

extern "C" int mmrca_gemm_splitk(const void* A, const void* B, float* C, void* workspace, int64_t workspace_bytes, int64_t M,
                                 int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_layout, int b_layout,
                                 void* stream) {
  MMRCA_REQUIRE(A && B && C && workspace, "gemm_splitk: null operand");
  MMRCA_REQUIRE(M > 0 && N > 0 && M % 256 == 0 && N % 256 == 0 && K >= 128 && K % 64 == 0,
                "gemm_splitk: needs M %% 256 == 0, N %% 256 == 0, K %% 64 == 0, K >= 128 (got M=%lld N=%lld K=%lld)", (long long)M, (long long)N, (long long)K);
  MMRCA_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)workspace) & 15) == 0,
                "gemm_splitk: operands must be 16-byte aligned");
  MMRCA_REQUIRE(lda >= (a_layout == MMRCA_ROWK ? K : M) && ldb >= (b_layout == MMRCA_ROWK ? K : N) && ldc >= N, "gemm_splitk: leading dimension too small");
  const int tiles_m = (int)(M / 256), tiles_n = (int)(N / 256), tiles = tiles_m * tiles_n;
  MMRCA_REQUIRE(tiles <= 256, "gemm_splitk: more than 256 output tiles (use mmrca_gemm)");
  const int64_t ksteps = K / 64;
  int64_t splits64 = 256 / tiles;                   // one workgroup per CU
  if (splits64 > ksteps / 2) splits64 = ksteps / 2; // the kernel's pipeline needs two K-tiles per range
  const int splits = (int)splits64;
  const int ksteps_base = (int)(ksteps / splits), ksteps_rem = (int)(ksteps % splits);   // the first `rem` ranges take one more
  MMRCA_REQUIRE(workspace_bytes >= (int64_t)splits * tiles * 65536 * 4, "gemm_splitk: workspace too small (%lld < %lld bytes)",
                (long long)workspace_bytes, (long long)splits * tiles * 65536 * 4);
  hipStream_t st = (hipStream_t)stream;
  const bool ak = a_layout == MMRCA_KROW, bk = b_layout == MMRCA_KROW;
#define LSLAB(AK_, BK_)                                                                                                          \
  do {                                                                                                                           \
    (void)hipFuncSetAttribute((const void*)gemm_mfma256_k<AK_, BK_, MODE_SLAB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS256_BYTES); \
    hipLaunchKernelGGL((gemm_mfma256_k<AK_, BK_, MODE_SLAB>), dim3(tiles, splits), dim3(512), LDS256_BYTES, st, (const bf16_t*)A,      \
                       (const bf16_t*)B, (bf16_t*)nullptr, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (bf16_t*)nullptr, M, N, K, \
                       lda, ldb, ldc, tiles_m, tiles_n, ksteps_base, ksteps_rem, (float*)workspace, (float*)nullptr);           \
  } while (0)
  if (!ak && !bk) LSLAB(false, false);
  else if (!ak && bk) LSLAB(false, true);
  else if (ak && !bk) LSLAB(true, false);
  else LSLAB(true, true);
#undef LSLAB
  MMRCA_CHECK_LAUNCH("gemm_splitk(mfma256)");
  hipLaunchKernelGGL(splitk_reduce256_k, dim3(tiles * 64), dim3(256), 0, st, (const float*)workspace, C, ldc, tiles, tiles_n, splits);
  MMRCA_CHECK_LAUNCH("gemm_splitk(reduce)");
  return 0;
}
