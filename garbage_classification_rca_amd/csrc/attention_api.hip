// K3 entry points: dispatch between the reference-grade kernels (attention_ref.hip) and the fused MFMA kernels
// (attention_mfma.hip).
#include "common.h"

int mmrca_mha_fwd_ref(const void*, const int32_t*, void*, float*, int, int, int, int, float, float, uint64_t, const int32_t*, int, hipStream_t);
int mmrca_mha_bwd_ref(const void*, const int32_t*, const void*, const void*, const float*, void*, int, int, int, int, float, float, uint64_t, const int32_t*, int, hipStream_t);
bool mmrca_mha_mfma_ok(int S, int dh, int dtype);
// fp32 on the fp32 matrix cores (attention_f32.hip): what the fp32 / bf16x3 modes run for head dim 64, S <= 208
bool mmrca_mha_f32m_ok(int S, int dh, int dtype);
int mmrca_mha_fwd_f32m(const void*, const int32_t*, void*, float*, int, int, int, int, float, float, uint64_t, const int32_t*, hipStream_t, void*, void*, const void*);
int mmrca_mha_bwd_f32m(const void*, const int32_t*, const void*, const void*, const float*, void*, int, int, int, int, float, float, uint64_t, const int32_t*, hipStream_t);
int mmrca_mha_fwd_mfma(const void*, const int32_t*, void*, float*, int, int, int, int, float, float, uint64_t, const int32_t*, hipStream_t);
int mmrca_mha_bwd_mfma(const void*, const int32_t*, const void*, const void*, const float*, void*, int, int, int, int, float, float, uint64_t, const int32_t*, hipStream_t);

extern "C" int mmrca_mha_fwd(const void* qkv, const int32_t* key_mask, void* out, float* lse, int B, int H, int S, int dh,
                             float scale, float drop_p, uint64_t drop_seed, const int32_t* cu_seqlens, int dtype, int impl,
                             void* stream) {
  MMRCA_REQUIRE(qkv && out && lse, "mha_fwd: null pointer");
  MMRCA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "mha_fwd: dropout p must be in [0,1)");
  MMRCA_REQUIRE(B > 0 && H > 0 && S > 0 && dh > 0, "mha_fwd: bad shape");
  const bool ok = mmrca_mha_mfma_ok(S, dh, dtype);
  if (impl == MMRCA_GEMM_MFMA && !ok && !mmrca_mha_f32m_ok(S, dh, dtype)) return mmrca_fail(-3, "mha_fwd: S=%d dh=%d dtype=%d does not qualify for the MFMA kernel", S, dh, dtype);
  if (ok && impl != MMRCA_GEMM_REF) return mmrca_mha_fwd_mfma(qkv, key_mask, out, lse, B, H, S, dh, scale, drop_p, drop_seed, cu_seqlens, (hipStream_t)stream);
  if (impl != MMRCA_GEMM_REF && mmrca_mha_f32m_ok(S, dh, dtype))
    return mmrca_mha_fwd_f32m(qkv, key_mask, out, lse, B, H, S, dh, scale, drop_p, drop_seed, cu_seqlens, (hipStream_t)stream, nullptr, nullptr, nullptr);
  return mmrca_mha_fwd_ref(qkv, key_mask, out, lse, B, H, S, dh, scale, drop_p, drop_seed, cu_seqlens, dtype, (hipStream_t)stream);
}

// fp32 attention forward that also writes the context as two bf16 planes (out_hi + out_lo = out to 2^-17): the operand form of the
// bf16x3 out-projection GEMM (mmrca_gemm_x3).  fp32 operands, head dim 64, S <= 208 (the fp32-matrix-core kernels); other shapes:
// call mmrca_mha_fwd and mmrca_split_f32.
extern "C" int mmrca_mha_fwd_planes(const void* qkv, const int32_t* key_mask, void* out, void* out_hi, void* out_lo, float* lse,
                                    int B, int H, int S, int dh, float scale, float drop_p, uint64_t drop_seed,
                                    const int32_t* cu_seqlens, void* stream) {
  MMRCA_REQUIRE(qkv && out && out_hi && out_lo && lse, "mha_fwd_planes: null pointer");
  MMRCA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "mha_fwd_planes: dropout p must be in [0,1)");
  MMRCA_REQUIRE(B > 0 && H > 0 && mmrca_mha_f32m_ok(S, dh, MMRCA_F32), "mha_fwd_planes: needs head dim 64 and 1 <= S <= 208 (got S=%d dh=%d)", S, dh);
  return mmrca_mha_fwd_f32m(qkv, key_mask, out, lse, B, H, S, dh, scale, drop_p, drop_seed, cu_seqlens, (hipStream_t)stream, out_hi, out_lo, nullptr);
}

// The same with q|k|v ALSO given as two bf16 planes (qkv_hi + qkv_lo = the fp32 projection to 2^-17: what mmrca_gemm_x3 writes
// with C_lo != NULL): the bf16x3f mode, whose bf16 backward reads the hi plane as its bf16 q|k|v -- no fp32 copy, no cast pass.
extern "C" int mmrca_mha_fwd_planes_in(const void* qkv_hi, const void* qkv_lo, const int32_t* key_mask, void* out, void* out_hi,
                                       void* out_lo, float* lse, int B, int H, int S, int dh, float scale, float drop_p,
                                       uint64_t drop_seed, const int32_t* cu_seqlens, void* stream) {
  MMRCA_REQUIRE(qkv_hi && qkv_lo && out_hi && out_lo && lse, "mha_fwd_planes_in: null pointer");      // out (fp32) is optional
  MMRCA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "mha_fwd_planes_in: dropout p must be in [0,1)");
  MMRCA_REQUIRE(B > 0 && H > 0 && mmrca_mha_f32m_ok(S, dh, MMRCA_F32), "mha_fwd_planes_in: needs head dim 64 and 1 <= S <= 208 (got S=%d dh=%d)", S, dh);
  MMRCA_REQUIRE((((uintptr_t)qkv_lo) & 7) == 0, "mha_fwd_planes_in: planes must be 8-byte aligned");
  return mmrca_mha_fwd_f32m(qkv_hi, key_mask, out, lse, B, H, S, dh, scale, drop_p, drop_seed, cu_seqlens, (hipStream_t)stream, out_hi, out_lo, qkv_lo);
}

// The same arithmetic on the bf16 matrix cores: every product three-pass over the planes (attention_mfma.hip::mha_fwd_x3_k), S <= 224.
// q|k|v AND the context as two bf16 planes; no fp32 tensor is read or written.  What the bf16x3f mode runs.
int mmrca_mha_fwd_x3_launch(const void*, const void*, const int32_t*, void*, void*, float*, int, int, int, int, float, float, uint64_t, const int32_t*, hipStream_t);
extern "C" int mmrca_mha_fwd_x3(const void* qkv_hi, const void* qkv_lo, const int32_t* key_mask, void* out_hi, void* out_lo, float* lse,
                                int B, int H, int S, int dh, float scale, float drop_p, uint64_t drop_seed, const int32_t* cu_seqlens,
                                void* stream) {
  MMRCA_REQUIRE(qkv_hi && qkv_lo && out_hi && out_lo && lse, "mha_fwd_x3: null pointer");
  MMRCA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "mha_fwd_x3: dropout p must be in [0,1)");
  MMRCA_REQUIRE(B > 0 && H > 0, "mha_fwd_x3: bad shape");
  return mmrca_mha_fwd_x3_launch(qkv_hi, qkv_lo, key_mask, out_hi, out_lo, lse, B, H, S, dh, scale, drop_p, drop_seed, cu_seqlens, (hipStream_t)stream);
}

static int mha_bwd_impl(const void* qkv, const int32_t* key_mask, const void* out, const void* dout, const float* lse,
                        void* dqkv, float* dqkv_colsum, int64_t rows, int B, int H, int S, int dh, float scale, float drop_p,
                        uint64_t drop_seed, const int32_t* cu_seqlens, int dtype, int impl, void* stream) {
  MMRCA_REQUIRE(qkv && out && dout && lse && dqkv, "mha_bwd: null pointer");
  MMRCA_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "mha_bwd: dropout p must be in [0,1)");
  MMRCA_REQUIRE(B > 0 && H > 0 && S > 0 && dh > 0, "mha_bwd: bad shape");
  const bool ok = mmrca_mha_mfma_ok(S, dh, dtype);
  if (impl == MMRCA_GEMM_MFMA && !ok && !mmrca_mha_f32m_ok(S, dh, dtype)) return mmrca_fail(-3, "mha_bwd: S=%d dh=%d dtype=%d does not qualify for the MFMA kernel", S, dh, dtype);
  // (reducing the bias column sums inside the MFMA kernels was measured: +41 us per ViT layer against 32 us for this pass --
  //  256 blocks contend on every one of the 2,304 fp32 atomics addresses)
  int rc;
  if (ok && impl != MMRCA_GEMM_REF)
    rc = mmrca_mha_bwd_mfma(qkv, key_mask, out, dout, lse, dqkv, B, H, S, dh, scale, drop_p, drop_seed, cu_seqlens, (hipStream_t)stream);
  else if (impl != MMRCA_GEMM_REF && mmrca_mha_f32m_ok(S, dh, dtype))
    rc = mmrca_mha_bwd_f32m(qkv, key_mask, out, dout, lse, dqkv, B, H, S, dh, scale, drop_p, drop_seed, cu_seqlens, (hipStream_t)stream);
  else
    rc = mmrca_mha_bwd_ref(qkv, key_mask, out, dout, lse, dqkv, B, H, S, dh, scale, drop_p, drop_seed, cu_seqlens, dtype, (hipStream_t)stream);
  if (rc) return rc;
  if (dqkv_colsum) return mmrca_colsum_accum(dqkv, dqkv_colsum, rows, 3LL * H * dh, 3LL * H * dh, dtype, stream);
  return 0;
}

extern "C" int mmrca_mha_bwd(const void* qkv, const int32_t* key_mask, const void* out, const void* dout, const float* lse,
                             void* dqkv, int B, int H, int S, int dh, float scale, float drop_p, uint64_t drop_seed,
                             const int32_t* cu_seqlens, int dtype, int impl, void* stream) {
  return mha_bwd_impl(qkv, key_mask, out, dout, lse, dqkv, nullptr, 0, B, H, S, dh, scale, drop_p, drop_seed, cu_seqlens, dtype,
                      impl, stream);
}

// total_rows: number of token rows of dqkv (B*S in the padded layout, cu_seqlens[B] in the packed one)
extern "C" int mmrca_mha_bwd_colsum(const void* qkv, const int32_t* key_mask, const void* out, const void* dout, const float* lse,
                                    void* dqkv, float* dqkv_colsum, int64_t total_rows, int B, int H, int S, int dh, float scale,
                                    float drop_p, uint64_t drop_seed, const int32_t* cu_seqlens, int dtype, int impl, void* stream) {
  MMRCA_REQUIRE(dqkv_colsum && total_rows > 0, "mha_bwd_colsum: null dqkv_colsum / bad total_rows");
  return mha_bwd_impl(qkv, key_mask, out, dout, lse, dqkv, dqkv_colsum, total_rows, B, H, S, dh, scale, drop_p, drop_seed,
                      cu_seqlens, dtype, impl, stream);
}
