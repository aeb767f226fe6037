#!/usr/bin/env python
"""Under-filled forward products (few 128x128 tiles, long contraction): which 128x128 kernel?  AUTO / two-stage / single-stage / 32-deep."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from garbage_classification_rca_amd import lib as L
L.load()
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for bl in (L.ROWK, L.KROW):
    for (M, N, K) in [(3600, 384, 1824), (3600, 304, 1824), (3600, 512, 3072), (14400, 384, 1824), (14400, 304, 1824), (14400, 512, 3072), (14400, 256, 1056), (14400, 176, 1056), (57600, 256, 1056), (57600, 176, 1056)]:
        rp = (M + 255) // 256 * 256
        X = [torch.randn(rp, K, device="cuda").bfloat16() for _ in range(4)]
        W = torch.randn(N, K, device="cuda").bfloat16() if bl == L.ROWK else torch.randn(K, N, device="cuda").bfloat16()
        Z = torch.zeros(rp, N, device="cuda", dtype=torch.bfloat16)
        out = []
        for name, impl in (("auto", L.IMPL_AUTO), ("2stage", L.IMPL_MFMA), ("1stage", L.IMPL_MFMA_1STAGE), ("bk32", L.IMPL_MFMA_BK32)):
            i = [0]
            def f():
                i[0] += 1
                L.gemm(X[i[0] % 4], W, Z, M=M, N=N, K=K, lda=K, ldb=(K if bl == L.ROWK else N), ldc=N, a_layout=L.ROWK, b_layout=bl, dtype=L.BF16, impl=impl)
            try:
                out.append(f"{name} {timeit(f):6.1f}")
            except L.MmrcaError as e:
                out.append(f"{name}   n/a ")
        print(("fwd  " if bl == L.ROWK else "dgrad"), (M, N, K), " | ".join(out), flush=True)
