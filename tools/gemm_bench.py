#!/usr/bin/env python
"""Per-shape timing of the MFMA GEMM kernels on the encoder shapes of BASELINE configs[1] (B=256).
Random operands (zero-filled data reads 15-20% high on this chip), interleaved rounds in one process."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd import lib as L

M = 50432
SHAPES = [  # (name, M, N, K, a_layout, b_layout, accum)
    ("fwd qkv", M, 2304, 768, 0, 0, 0), ("fwd out", M, 768, 768, 0, 0, 0), ("fwd ffn1", M, 3072, 768, 0, 0, 0), ("fwd ffn2", M, 768, 3072, 0, 0, 0),
    ("dgrad qkv", M, 768, 2304, 0, 1, 0), ("dgrad ffn1", M, 768, 3072, 0, 1, 0), ("dgrad ffn2", M, 3072, 768, 0, 1, 0),
    ("wgrad qkv", 2304, 768, M, 1, 1, 1), ("wgrad out", 768, 768, M, 1, 1, 1), ("wgrad ffn1", 3072, 768, M, 1, 1, 1), ("wgrad ffn2", 768, 3072, M, 1, 1, 1),
    ("epi K64", M, 3072, 64, 0, 0, 0), ("epi K128", M, 3072, 128, 0, 0, 0), ("epi K1536", M, 3072, 1536, 0, 0, 0),
    ("epi ffn1 preact", M, 3072, 768, 0, 0, 0), ("epi ffn1 gelu", M, 3072, 768, 0, 0, 0), ("epi ffn1 gelu preact", M, 3072, 768, 0, 0, 0),
    ("epi out addend", M, 768, 768, 0, 0, 0), ("epi ffn2 addend", M, 768, 3072, 0, 0, 0),
    ("text fwd qkv", 16384, 2304, 768, 0, 0, 0), ("text fwd out", 16384, 768, 768, 0, 0, 0),
    ("text wgrad qkv", 2304, 768, 9344, 1, 1, 1), ("text wgrad out", 768, 768, 9344, 1, 1, 1), ("text wgrad ffn1", 3072, 768, 9344, 1, 1, 1),
    ("cube 4096", 4096, 4096, 4096, 0, 0, 0), ("cube 8192", 8192, 8192, 8192, 0, 0, 0),
]
impls = {"auto": L.IMPL_AUTO, "1stage": L.IMPL_MFMA_1STAGE, "bk32": L.IMPL_MFMA_BK32, "2stage": L.IMPL_MFMA}
if os.environ.get("MMRCA_BENCH_256"):
    impls["m256"] = L.IMPL_MFMA256
DBG = [int(x) for x in os.environ.get("MMRCA_DBG", "0").split(",")]
only = sys.argv[1:] 
L.load()
dev = "cuda"
res = {}
for name, m, n, k, al, bl, acc in SHAPES:
    if only and not any(o in name for o in only):
        continue
    Mp = (m + 127) // 128 * 128
    if al == 0:
        A = torch.randn(Mp, k, device=dev).bfloat16(); lda = k
    else:
        A = torch.randn(k, m, device=dev).bfloat16(); lda = m
    if bl == 0:
        B = torch.randn(n, k, device=dev).bfloat16(); ldb = k
    else:
        B = torch.randn(k, n, device=dev).bfloat16(); ldb = n
    C = torch.zeros(Mp, n, device=dev, dtype=torch.float32 if acc else torch.bfloat16)
    bias = None if acc else torch.randn(n, device=dev).bfloat16()
    row = {}
    epi = dict(preact=torch.empty(Mp, n, device=dev, dtype=torch.bfloat16) if "preact" in name else None,
               addend=torch.randn(Mp, n, device=dev).bfloat16() if "addend" in name else None,
               act=(L.ACT_GELU_SAVE_GRAD if "gelu preact" in name else L.ACT_GELU) if "gelu" in name else L.ACT_NONE)
    variants = [(a, b, 0) for a, b in impls.items()] + [(f"256dbg{d}", L.IMPL_MFMA256, d) for d in DBG if 0 < d < 8]
    variants += [(f"128tgt{d >> 8}", L.IMPL_MFMA, d) for d in DBG if d >= 256]
    variants += [(f"128dbg{d}", L.IMPL_MFMA, d) for d in DBG if d == 16]
    variants += [(f"bk32tgt{d >> 8}", L.IMPL_MFMA_BK32, d) for d in DBG if d >= 256]
    runs = {}
    for iname, impl, dbg in variants:
        if impl == L.IMPL_MFMA256 and (acc or n % 256 or k < 128 or al == 1 or
                                       (epi["act"] == L.ACT_NONE and epi["preact"] is not None)):
            continue
        def run(impl=impl, dbg=dbg):
            L.load().mmrca_debug_set(dbg)
            L.gemm(A, B, C, bias=bias, M=m, N=n, K=k, lda=lda, ldb=ldb, ldc=n, a_layout=al, b_layout=bl, accum=bool(acc), dtype=L.BF16, impl=impl, **epi)
        runs[iname] = run
    if acc and L.gemm_splitk_ok(m, n, k, L.BF16):
        ws = torch.empty(L.SPLITK_WS_BYTES, dtype=torch.uint8, device=dev)
        runs["splitk256"] = lambda ws=ws: L.gemm_splitk(A, B, C, ws, M=m, N=n, K=k, lda=lda, ldb=ldb, ldc=n, a_layout=al, b_layout=bl)
        for d in DBG:
            if 0 < d < 8:
                def run_d(ws=ws, d=d):
                    L.load().mmrca_debug_set(d)
                    L.gemm_splitk(A, B, C, ws, M=m, N=n, K=k, lda=lda, ldb=ldb, ldc=n, a_layout=al, b_layout=bl)
                    L.load().mmrca_debug_set(0)
                runs[f"splitk_dbg{d}"] = run_d
    if os.environ.get("MMRCA_COLD") and not acc:   # the same launch over rotating A / C buffers (> 1 GB in total): operands not in L2 / MALL
        nb = 6
        As = [A.clone() for _ in range(nb)]
        Cs = [torch.empty_like(C) for _ in range(nb)]
        state = {"i": 0}
        def run_cold():
            i = state["i"] = (state["i"] + 1) % nb
            L.load().mmrca_debug_set(0)
            L.gemm(As[i], B, Cs[i], bias=bias, M=m, N=n, K=k, lda=lda, ldb=ldb, ldc=n, a_layout=al, b_layout=bl, accum=False, dtype=L.BF16,
                   impl=L.IMPL_AUTO, **epi)
        runs["auto_cold"] = run_cold
    if os.environ.get("MMRCA_YARDSTICK"):      # calibration only: the vendor library (hipBLASLt through torch.matmul) on the same operands
        At = A[:m] if al == 0 else A.t()
        Bt = B.t() if bl == 0 else B
        Cy = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        runs["hipblaslt"] = lambda At=At, Bt=Bt, Cy=Cy: torch.matmul(At, Bt, out=Cy)
        if not acc:
            bf = bias.float().bfloat16()
            runs["hipblaslt+bias"] = lambda At=At, Bt=Bt, Cy=Cy, bf=bf: torch.addmm(bf, At, Bt, out=Cy)
            if "gelu" in name:                 # the vendor's fused epilogue is the tanh approximation (not the erf GELU the reference computes)
                runs["hipblaslt+bias+gelu(tanh)"] = lambda At=At, Bt=Bt, bf=bf: torch._addmm_activation(bf, At, Bt, use_gelu=True)
    best = {k2: 1e9 for k2 in runs}
    for k2, fn in runs.items():
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    for rnd in range(5):                       # interleaved rounds: every variant sees the same cache / clock state
        for k2, fn in runs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            best[k2] = min(best[k2], e0.elapsed_time(e1) / 5)
    for k2 in runs:
        row[k2] = round(2.0 * m * n * k / (best[k2] * 1e-3) / 1e12, 1)
    L.load().mmrca_debug_set(0)
    res[name] = row
    print(f"{name:14s} M={m:6d} N={n:5d} K={k:6d}  " + "  ".join(f"{a}={b:7.1f} TF ({2.0*m*n*k/b/1e6:.0f}us)" for a, b in row.items()), flush=True)
