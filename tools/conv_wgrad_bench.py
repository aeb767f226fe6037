#!/usr/bin/env python
"""The conv layers' 1x1 weight gradients (EfficientNetV2-M @ 480): mmrca_gemm's accumulate mode (128x128 tiles, fp32 atomics) against
mmrca_gemm_splitk on the ragged output shape (256x256 tiles, slab partials + reduce).  usage: conv_wgrad_bench.py [batch ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garbage_classification_rca_amd import lib as L

L.load()
SH = [(176, 1056, 900), (1056, 176, 900), (304, 1824, 225), (1824, 304, 225), (48, 192, 14400), (192, 48, 14400), (160, 640, 900), (640, 160, 900),
      (3072, 512, 225), (512, 3072, 225), (80, 320, 3600), (320, 80, 3600), (96, 216, 14400), (192, 432, 3600)]
ws = torch.empty(L.SPLITK_WS_BYTES, dtype=torch.uint8, device="cuda")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in [int(a) for a in sys.argv[1:]] or [64, 16]:
    print(f"batch {B}: (M, N, K)  atomic us  slab us  bytes MB  floor us @4TB/s")
    for M, N, hw in SH:
        K = (B * hw + 63) // 64 * 64
        A = torch.randn(K, M, device="cuda").bfloat16()
        Bm = torch.randn(K, N, device="cuda").bfloat16()
        C = torch.zeros(M, N, device="cuda")
        # rotate over several operand copies so that the Infinity Cache does not hold them (as in the step)
        copies = max(1, min(8, int(600e6 // ((M + N) * K * 2))))
        As, Bs = [A.clone() for _ in range(copies)], [Bm.clone() for _ in range(copies)]
        i = [0]

        def atomic():
            i[0] += 1
            L.gemm(As[i[0] % copies], Bs[i[0] % copies], C, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, a_layout=L.KROW, b_layout=L.KROW, accum=True, dtype=L.BF16)

        def slab():
            i[0] += 1
            L.gemm_splitk(As[i[0] % copies], Bs[i[0] % copies], C, ws, M=M, N=N, K=K, lda=M, ldb=N, ldc=N)

        ta, ts = timeit(atomic), timeit(slab)
        mb = (M + N) * K * 2 / 1e6
        print(f"  ({M:5d}, {N:5d}, {K:7d})  {ta:8.1f}  {ts:8.1f}  {mb:8.1f}  {mb / 4.0:8.1f}")
