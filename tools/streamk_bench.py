#!/usr/bin/env python
"""A/B of the stream-K tail of the persistent 256x256 GEMM (csrc/gemm256.hip) on the ViT shapes of BASELINE configs[1] whose tile
count is not a whole number of rounds (N = 768 at M = 50,432: 591 tiles = 2.31 rounds of 256 CUs).  `off` = the workspace
unregistered: AUTO's round-5 choice (whole rounds on the persistent kernel + the rest on the 128x128 kernel, two launches).
Operands rotate over four copies so that no launch finds its A in the memory-side cache.  MMRCA_SK_MAX caps the ranges per tile."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd import lib as L

M = 50432
SHAPES = [("fwd out", 768, 768, L.ROWK, None), ("fwd ffn2", 768, 3072, L.ROWK, None), ("dgrad out", 768, 768, L.KROW, None),
          ("dgrad qkv", 768, 2304, L.KROW, None), ("dgrad ffn1", 768, 3072, L.KROW, None),
          ("fwd qkv (9.2 rounds)", 2304, 768, L.ROWK, None), ("fwd ffn1 gelu (9.2 rounds)", 3072, 768, L.ROWK, L.ACT_GELU_SAVE_GRAD)]
L.load()
L.load().mmrca_gemm_streamk_config(int(os.environ.get("MMRCA_SK_MAX", "4")), 0, 1)      # every K: the A/B is the point
res = {}
for name, N, K, bl, act in SHAPES:
    As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(4)]
    B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    Bd = B if bl == L.ROWK else B.t().contiguous()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    P = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if act else None
    bias = torch.randn(N, device="cuda").bfloat16() if bl == L.ROWK else None

    def run(i):
        L.gemm(As[i & 3], Bd, C, bias=bias, preact=P, M=M, N=N, K=K, lda=K, ldb=Bd.shape[1], ldc=N, b_layout=bl, dtype=L.BF16,
               act=act or L.ACT_NONE)
    L.streamk_workspace(M, N, As[0].device, force=True)
    run(0)
    torch.cuda.synchronize()
    key = (torch.cuda.current_device(), L.stream_ptr())
    ws = L._STREAMK_WS[key]
    out = {}
    for rnd in range(3):
        for mode in ("on", "off"):
            L._check(L.load().mmrca_gemm_streamk_workspace(L.ptr(ws) if mode == "on" else None, ws.numel(), L.stream_ptr()), "toggle")
            for i in range(4):
                run(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(40):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            out.setdefault(mode, []).append(e0.elapsed_time(e1) / 40 * 1e3)
    L._check(L.load().mmrca_gemm_streamk_workspace(L.ptr(ws), ws.numel(), L.stream_ptr()), "toggle")
    fl = 2.0 * M * N * K
    res[name] = {m: {"us": round(min(v), 1), "TFLOPs": round(fl / min(v) / 1e6, 1)} for m, v in out.items()}
    print(f"{name:28s} N={N:5d} K={K:5d}  " + "  ".join(f"{m}: {r['us']:7.1f} us {r['TFLOPs']:7.1f} TF/s" for m, r in res[name].items()), flush=True)
print(json.dumps({"sk_max": os.environ.get("MMRCA_SK_MAX", "4"), "shapes": res}))
