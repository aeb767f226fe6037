#!/usr/bin/env python
"""The text encoder's GEMM shapes at BASELINE configs[1] (packed captions: M = 9,280 rows = 36.25 row tiles of 256) on AUTO (the 128x128
kernels: these shapes have a ragged M and 111 - 444 tiles of 256x256) against the persistent 256x256 kernel with and without its
stream-K tail (rows_readable: the engine's buffers are padded to 256 rows)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd import lib as L

M = int(os.environ.get("TEXT_M", "9280"))
Mp = (M + 255) // 256 * 256
SHAPES = [("fwd qkv", 2304, 768, L.ROWK, L.ACT_NONE), ("fwd out", 768, 768, L.ROWK, L.ACT_NONE), ("fwd ffn1 gelu", 3072, 768, L.ROWK, L.ACT_GELU_SAVE_GRAD),
          ("fwd ffn2", 768, 3072, L.ROWK, L.ACT_NONE), ("dgrad qkv", 768, 2304, L.KROW, L.ACT_NONE), ("dgrad out", 768, 768, L.KROW, L.ACT_NONE),
          ("dgrad ffn1", 768, 3072, L.KROW, L.ACT_NONE)]
L.load()
lib = L.load()
res = {}
for name, N, K, bl, act in SHAPES:
    As = [torch.randn(Mp, K, device="cuda").bfloat16() for _ in range(6)]
    B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    Bd = B if bl == L.ROWK else B.t().contiguous()
    C = torch.empty(Mp, N, device="cuda", dtype=torch.bfloat16)
    P = torch.empty(Mp, N, device="cuda", dtype=torch.bfloat16) if act == L.ACT_GELU_SAVE_GRAD else None
    bias = torch.randn(N, device="cuda").bfloat16() if bl == L.ROWK else None
    L.streamk_workspace(65536, 256, As[0].device, force=True)
    ws = L._STREAMK_WS[(torch.cuda.current_device(), L.stream_ptr())]

    def run(i, impl, rr):
        L.gemm(As[i % 6], Bd, C, bias=bias, preact=P, M=M, N=N, K=K, lda=K, ldb=Bd.shape[1], ldc=N, b_layout=bl, dtype=L.BF16, act=act, impl=impl,
               rows_readable=rr)
    modes = {"auto": (L.IMPL_AUTO, None, 0), "p256": (L.IMPL_MFMA256, (Mp, 0), 0), "p256+sk2": (L.IMPL_MFMA256, (Mp, 0), 2), "p256+sk4": (L.IMPL_MFMA256, (Mp, 0), 4)}
    out = {}
    for rnd in range(3):
        for m, (impl, rr, sk) in modes.items():
            L._check(lib.mmrca_gemm_streamk_workspace(L.ptr(ws) if sk else None, ws.numel(), L.stream_ptr()), "toggle")
            lib.mmrca_gemm_streamk_config(max(sk, 2), 4, 1)
            for i in range(4):
                run(i, impl, rr)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(40):
                run(i, impl, rr)
            e1.record()
            torch.cuda.synchronize()
            out.setdefault(m, []).append(e0.elapsed_time(e1) / 40 * 1e3)
    fl = 2.0 * M * N * K
    res[name] = {m: {"us": round(min(v), 1), "TFLOPs": round(fl / min(v) / 1e6, 1)} for m, v in out.items()}
    print(f"{name:16s} N={N:5d} K={K:5d}  " + "  ".join(f"{m}: {r['us']:6.1f} us {r['TFLOPs']:6.1f} TF" for m, r in res[name].items()), flush=True)
lib.mmrca_gemm_streamk_config(4, 24, 0)
print(json.dumps({"M": M, "shapes": res}))
