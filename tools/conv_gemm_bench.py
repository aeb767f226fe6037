#!/usr/bin/env python
"""The conv layers' 1x1 products in isolation (EfficientNetV2-M @ 480): forward x[rows, K] . W[N, K]^T and the input gradient
dz[rows, K'] . W[K', N'] (B K-major), against their byte floor.  usage: conv_gemm_bench.py [batch ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garbage_classification_rca_amd import lib as L

L.load()
# (cin, cout, H*W) of the expand / project 1x1 convolutions by stage
LAYERS = [(96, 48, 14400), (192, 48, 14400), (192, 80, 3600), (320, 80, 3600), (80, 320, 3600), (320, 160, 900), (160, 640, 900), (640, 160, 900),
          (160, 960, 900), (960, 176, 900), (176, 1056, 900), (1056, 176, 900), (1056, 304, 225), (304, 1824, 225), (1824, 304, 225),
          (1824, 512, 225), (512, 3072, 225), (3072, 512, 225), (512, 1280, 225)]


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
    print(f"batch {B}: (rows, cin -> cout)   fwd us  TB/s | dgrad us  TB/s | MB   floor us @4TB/s")
    for cin, cout, hw in LAYERS:
        rows = B * hw
        rp = (rows + 255) // 256 * 256
        mb = rows * (cin + cout) * 2 / 1e6
        copies = max(1, min(8, int(600e6 // (rp * (cin + cout) * 2))))
        X = [torch.randn(rp, cin, device="cuda").bfloat16() for _ in range(copies)]
        Z = [torch.randn(rp, cout, device="cuda").bfloat16() for _ in range(copies)]
        W = torch.randn(cout, cin, device="cuda").bfloat16()
        i = [0]

        def fwd():
            i[0] += 1
            L.gemm(X[i[0] % copies], W, Z[i[0] % copies], M=rows, N=cout, K=cin, lda=cin, ldb=cin, ldc=cout, a_layout=L.ROWK, b_layout=L.ROWK, dtype=L.BF16)

        def dgrad():
            i[0] += 1
            L.gemm(Z[i[0] % copies], W, X[i[0] % copies], M=rows, N=cin, K=cout, lda=cout, ldb=cin, ldc=cin, a_layout=L.ROWK, b_layout=L.KROW, dtype=L.BF16)

        tf, tg = timeit(fwd), timeit(dgrad)
        print(f"  ({rows:7d}, {cin:5d} -> {cout:5d})  {tf:7.1f}  {mb / tf:5.2f} | {tg:7.1f}  {mb / tg:5.2f} | {mb:7.1f}  {mb / 4.0:7.1f}")
