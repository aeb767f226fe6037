// LDS fragment reads as inline asm (shared by gemm.hip and gemm256.hip).
//
// Why: with `ds_read` emitted from C++ the compiler cannot tell the reads from global_load_lds (LDS-DMA) writes that are
// still in flight -- one LDS array, runtime addresses -- and inserts `s_waitcnt vmcnt(0)` in front of every group of
// fragment reads, i.e. a software prefetch issued just before them is waited for before the K-step even starts.  An asm
// read is invisible to that pass; ordering is by hand-placed waits: `s_waitcnt lgkmcnt(0)` with the fragment registers as
// read-write operands (nothing that uses them can be scheduled above it), and counted `s_waitcnt vmcnt(N)` + barriers for
// the DMA (RAW: read a slot at least one barrier after the wait that retires its DMA; WAR: refill a slot only after a
// barrier that follows the lgkmcnt(0) of its last reads).
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

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


// the same with a run-time LDS address (no immediate offset)
template <bool KROW>
__device__ __forceinline__ void read_frag_rt(Frag<KROW>& f, unsigned addr_lo, unsigned addr_hi) {
  if constexpr (!KROW) {
    DS_READ_B128(f.v, addr_lo, 0);
  } else {
    DS_READ_TR(f.lo, addr_lo, 0);
    DS_READ_TR(f.hi, addr_hi, 0);
  }
}
__device__ __forceinline__ void lgkm0(Frag<false> (&f)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].v), "+v"(f[1].v), "+v"(f[2].v), "+v"(f[3].v) :: "memory");
}
__device__ __forceinline__ void lgkm0(Frag<true> (&f)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi),
               "+v"(f[3].lo), "+v"(f[3].hi) :: "memory");
}
