// K3 (fast path): fused multi-head attention with MFMA.  Placeholder until the kernel lands: reports "not qualified"
// so that the dispatcher uses the reference-grade kernels.
#include "common.h"
bool mmrca_mha_mfma_ok(int S, int dh, int dtype) { (void)S; (void)dh; (void)dtype; return false; }
int mmrca_mha_fwd_mfma(const void*, const int32_t*, void*, float*, int, int, int, int, float, hipStream_t) { return mmrca_fail(-3, "mha mfma: not built"); }
int mmrca_mha_bwd_mfma(const void*, const int32_t*, const void*, const void*, const float*, void*, int, int, int, int, float, hipStream_t) { return mmrca_fail(-3, "mha mfma: not built"); }
