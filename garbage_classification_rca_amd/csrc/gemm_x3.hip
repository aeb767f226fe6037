// K2x: the "bf16x3" GEMM -- fp32-accurate products on the bf16 matrix cores (the <= 1e-3 fast mode, --dtype bf16x3).
//
// The reference computes every nn.Linear in fp32 (CVPR_code/multimodal_model.py:651-726, no autocast anywhere).  gfx950 has no
// TF32-like path: its fp32 MFMA runs at 1/16 of the bf16 rate.  Here an fp32 operand x is carried as TWO bf16 planes,
//     hi = bf16(x)          (round to nearest even)
//     lo = bf16(x - hi)     (the next 8 mantissa bits; x - hi is exact in fp32)
// so that x = hi + lo to 2^-17 relative, and a product is formed as
//     A . B  ~=  A_hi . B_hi  +  A_lo . B_hi  +  A_hi . B_lo            (A_lo . B_lo < 2^-16 of the result: dropped)
// i.e. ONE bf16 GEMM over the contraction [A_hi | A_lo | A_hi] . [B_hi | B_hi | B_lo] of length 3K with fp32 accumulation,
// run by the tuned bf16 kernels of gemm.hip / gemm256.hip in their X3 form (the K loop walks three plane pairs into the same
// accumulators; no operand copy is made).  bias / side operands / outputs are fp32; an output that only feeds another X3 GEMM
// can be written directly as two planes (C_lo != NULL), which saves that consumer's split pass.
// Measured logits error of the whole model against a float64 evaluation: 1e-6 (fp32 mode: 5e-7, bf16 mode: 2e-2).
#include "common.h"
#include <stdlib.h>

int mmrca_gemm_k1s_x3(const void* A_hi, const void* A_lo, const void* B_hi, const void* B_lo, void* C, void* C_lo, const void* bias,
                      const void* addend, void* preact, float* colsum, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                      int64_t ldc, int a_layout, int b_layout, int act, int accum, hipStream_t st, int pre16);
bool mmrca_gemm256_x3_ok(int64_t M, int64_t N, int64_t K, int a_layout, int act, bool has_addend, bool has_colsum);
int mmrca_gemm256_x3(const void* A_hi, const void* A_lo, const void* B_hi, const void* B_lo, void* C, void* C_lo, const void* bias,
                     void* preact, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int b_layout, int act,
                     hipStream_t st, int pre16);

int mmrca_gemm256_x3_streamk_split(int64_t M, int64_t N, int64_t K, int b_layout, bool has_a_lo, bool has_b_lo, void* stream);
static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static const int g_x3_tail_pct = getenv("MMRCA_AUTO256_TAIL_PCT") ? atoi(getenv("MMRCA_AUTO256_TAIL_PCT")) : 60;

extern "C" int mmrca_gemm_x3(const void* A_hi, const void* A_lo, const void* B_hi, const void* B_lo, void* C, void* C_lo,
                             const void* bias, const void* addend, void* preact, float* colsum, int64_t M, int64_t N, int64_t K,
                             int64_t lda, int64_t ldb, int64_t ldc, int a_layout, int b_layout, int act, int out_f32_accum,
                             int impl, void* stream) {
  // pass sets: A_lo and B_lo given = the full three-pass product; B_lo == NULL = two passes (A_hi B_hi + A_lo B_hi: the B operand is
  // taken at bf16 precision); A_lo == NULL too = one pass (a plain bf16 product with fp32 epilogue / outputs).  The engine's
  // backward may drop passes (MMRCA_X3_DGRAD_PASSES / MMRCA_X3_WGRAD_PASSES); the forward never does.
  MMRCA_REQUIRE(A_hi && B_hi && C && (A_lo || !B_lo), "gemm_x3: null operand (a B lo plane needs the A lo plane)");
  MMRCA_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_x3: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  MMRCA_REQUIRE((a_layout == MMRCA_ROWK || a_layout == MMRCA_KROW) && (b_layout == MMRCA_ROWK || b_layout == MMRCA_KROW), "gemm_x3: bad layout");
  // MMRCA_ACT_GELU_SAVE_GRAD_BF16: MMRCA_ACT_GELU_SAVE_GRAD with gelu' stored as bf16 (the operand of a bf16 backward: bf16x3f mode)
  const int pre16 = act == MMRCA_ACT_GELU_SAVE_GRAD_BF16;
  if (pre16) act = MMRCA_ACT_GELU_SAVE_GRAD;
  MMRCA_REQUIRE(act >= MMRCA_ACT_NONE && act <= MMRCA_ACT_MUL, "gemm_x3: bad activation");
  MMRCA_REQUIRE(act < MMRCA_ACT_GELU_BWD || preact, "gemm_x3: this activation needs the `preact` buffer");
  MMRCA_REQUIRE(lda >= (a_layout == MMRCA_ROWK ? K : M) && ldb >= (b_layout == MMRCA_ROWK ? K : N) && ldc >= N, "gemm_x3: leading dimension too small");
  MMRCA_REQUIRE(!(out_f32_accum && (addend || preact || bias || colsum || C_lo || act != MMRCA_ACT_NONE)), "gemm_x3: accumulate mode takes no epilogue");
  MMRCA_REQUIRE(N % 128 == 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && (a_layout == MMRCA_ROWK || M % 128 == 0),
                "gemm_x3: needs N %% 128 == 0, K %% 64 == 0 (and M %% 128 == 0 for a KROW A), lda/ldb %% 8 == 0, ldc %% 4 == 0 "
                "(got M=%lld N=%lld K=%lld)", (long long)M, (long long)N, (long long)K);
  MMRCA_REQUIRE(al16(A_hi) && al16(A_lo) && al16(B_hi) && al16(B_lo) && al16(C) && al16(C_lo) && al16(bias) && al16(addend) && al16(preact),
                "gemm_x3: operands must be 16-byte aligned");
  MMRCA_REQUIRE(impl == MMRCA_GEMM_AUTO || impl == MMRCA_GEMM_MFMA256 || impl == MMRCA_GEMM_MFMA_1STAGE, "gemm_x3: impl must be AUTO, MFMA256 or MFMA_1STAGE");
  hipStream_t st = (hipStream_t)stream;
  const bool ok256 = !out_f32_accum && mmrca_gemm256_x3_ok(M, N, K, a_layout, act, addend != nullptr, colsum != nullptr) &&
                     M * lda * 2 < (1ll << 32) && (b_layout == MMRCA_KROW ? K * ldb : N * ldb) * 2 < (1ll << 32) && M * ldc * 4 < (1ll << 32);
  if (impl == MMRCA_GEMM_MFMA256) {
    // the persistent kernel streams whole 256-row tiles of A: an explicit request must come with whole tiles (AUTO checks the same)
    MMRCA_REQUIRE(ok256 && M % 256 == 0, "gemm_x3: shape M=%lld N=%lld K=%lld / epilogue does not qualify for the 256x256 kernel (needs M %% 256 == 0, "
                  "N %% 256 == 0, no side operand)", (long long)M, (long long)N, (long long)K);
    return mmrca_gemm256_x3(A_hi, A_lo, B_hi, B_lo, C, C_lo, bias, preact, M, N, K, lda, ldb, ldc, b_layout, act, st, pre16);
  }
  if (impl == MMRCA_GEMM_AUTO && ok256 && M % 256 == 0 && (M / 256) * (N / 256) >= 256) {
    // whole rounds of one 256x256 tile per CU on the persistent kernel, a 25-60 % partial round on the 128x128 kernel (gemm.hip AUTO)
    int ncu = 256, devi = 0;
    (void)hipGetDevice(&devi);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, devi) != hipSuccess || ncu < 8) ncu = 256;
    ncu &= ~7;
    const int64_t tm = M / 256, tn = N / 256, tiles = tm * tn;
    const int64_t rounds = tiles / ncu, rem = tiles - rounds * ncu;
    const int64_t m_split = rounds * ncu / tn;
    // (a stream-K workspace on this stream keeps the partial round inside the persistent launch: gemm256.hip, round 6)
    const bool sk_tail = mmrca_gemm256_x3_streamk_split(M, N, K, b_layout, A_lo != nullptr, B_lo != nullptr, stream) >= 2;
    if (!sk_tail && rounds >= 1 && rem * 100 >= 25 * (int64_t)ncu && rem * 100 < (int64_t)g_x3_tail_pct * ncu && m_split >= 1 && m_split < tm) {
      const int64_t M1 = m_split * 256;
      if (int rc = mmrca_gemm256_x3(A_hi, A_lo, B_hi, B_lo, C, C_lo, bias, preact, M1, N, K, lda, ldb, ldc, b_layout, act, st, pre16)) return rc;
      const int64_t csz = C_lo ? 2 : 4;
      return mmrca_gemm_k1s_x3((const char*)A_hi + M1 * lda * 2, A_lo ? (const char*)A_lo + M1 * lda * 2 : nullptr, B_hi, B_lo, (char*)C + M1 * ldc * csz,
                               C_lo ? (char*)C_lo + M1 * ldc * 2 : nullptr, bias, nullptr, preact ? (char*)preact + M1 * ldc * (pre16 ? 2 : 4) : nullptr,
                               nullptr, M - M1, N, K, lda, ldb, ldc, a_layout, b_layout, act, 0, st, pre16);
    }
    return mmrca_gemm256_x3(A_hi, A_lo, B_hi, B_lo, C, C_lo, bias, preact, M, N, K, lda, ldb, ldc, b_layout, act, st, pre16);
  }
  return mmrca_gemm_k1s_x3(A_hi, A_lo, B_hi, B_lo, C, C_lo, bias, addend, preact, colsum, M, N, K, lda, ldb, ldc, a_layout, b_layout,
                           act, out_f32_accum, st, pre16);
}
