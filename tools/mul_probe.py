#!/usr/bin/env python
"""ACT_MUL (x gelu') epilogue of the FFN2 input gradient: persistent 256x256 kernel vs the 128x128 kernel AUTO uses for it, error against an
fp32 product and time per launch, on the ViT shape and on small / ragged shapes in both B layouts.  (Round 4 used it to develop a split-phase
form of the side-operand loads; see DESIGN K2, "the gelu' factor".)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from garbage_classification_rca_amd import lib as L
L.load()
for (M, N, K) in ((788, 512, 320), (2048, 512, 768), (50432, 3072, 768)):
    Mp = (M + 255) // 256 * 256
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.zeros(Mp, K, device="cuda", dtype=torch.bfloat16); X[:M] = (torch.randn(M, K, device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    Wm = (torch.randn(N, K, device="cuda", generator=g) * 0.2).to(torch.bfloat16)
    G = torch.ones(Mp, N, device="cuda", dtype=torch.bfloat16); G[:M] = (torch.rand(M, N, device="cuda", generator=g) + 0.25).to(torch.bfloat16)
    ref = (X[:M].float() @ Wm.float().t()) * G[:M].float()
    for bl in (0, 1):
        W = Wm if bl == 0 else Wm.t().contiguous()
        for impl, name in ((L.IMPL_MFMA256, "persistent 256x256"), (L.IMPL_MFMA_1STAGE, "128x128")):
            C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
            kw = dict(preact=G, M=M, N=N, K=K, lda=K, ldb=W.shape[1], ldc=N, b_layout=bl, act=L.ACT_MUL, dtype=L.BF16, impl=impl)
            if impl == L.IMPL_MFMA256:
                kw["rows_readable"] = (Mp, Mp)
            L.gemm(X, W, C, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                L.gemm(X, W, C, **kw)
            e1.record(); torch.cuda.synchronize()
            err = float((C.float() - ref).abs().max() / ref.abs().max())
            print(f"M {M} N {N} K {K} B layout {bl} {name:18s}: {e0.elapsed_time(e1) * 100:6.0f} us  rel err {err:.3e}", flush=True)
