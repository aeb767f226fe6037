"""General-shape GEMM (fp32 matrix cores, gemm_gen_k) vs the VALU reference kernel on fp32 encoder shapes and on bf16 conv
shapes the bf16 MFMA kernels cannot take."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from garbage_classification_rca_amd import lib as L

SHAPES = [("fp32 vit fwd qkv", 50432, 2304, 768, torch.float32, 0, 0, 0), ("fp32 vit dgrad ffn1", 50432, 768, 3072, torch.float32, 0, 1, 0),
          ("fp32 vit wgrad qkv", 2304, 768, 50432, torch.float32, 1, 1, 1),
          ("bf16 conv3x3 24->96 @120", 128 * 120 * 120, 96, 216, torch.bfloat16, 0, 0, 0),
          ("bf16 conv1x1 96->24", 128 * 120 * 120, 24, 96, torch.bfloat16, 0, 0, 0),
          ("bf16 conv1x1 1824->304 @15", 128 * 15 * 15, 304, 1824, torch.bfloat16, 0, 0, 0)]
for name, m, n, k, dt, al, bl, acc in SHAPES:
    A = torch.randn((m, k) if al == 0 else (k, m), device="cuda").to(dt)
    B = torch.randn((n, k) if bl == 0 else (k, n), device="cuda").to(dt)
    C = torch.zeros(m, n, device="cuda", dtype=torch.float32 if acc else dt)
    ldt = L.F32 if dt == torch.float32 else L.BF16
    out = {}
    for iname, impl in (("gen", L.IMPL_AUTO), ("ref", L.IMPL_REF)):
        def run():
            L.gemm(A, B, C, M=m, N=n, K=k, lda=A.shape[1], ldb=B.shape[1], ldc=n, a_layout=al, b_layout=bl, accum=bool(acc), dtype=ldt, impl=impl)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 if iname == "ref" else 10
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[iname] = (ms, 2.0 * m * n * k / ms / 1e9)
    print(f"{name:32s} M={m:8d} N={n:5d} K={k:6d}  gen {out['gen'][0]:8.2f} ms ({out['gen'][1]:6.1f} TF)   ref {out['ref'][0]:8.2f} ms ({out['ref'][1]:6.1f} TF)")
