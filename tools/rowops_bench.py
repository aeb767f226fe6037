#!/usr/bin/env python
"""Timing of the memory-bound row kernels at the ViT-B/16 B=256 shape (50432 x 768 rows): achieved GB/s over the
algorithmic bytes of each kernel.  Operands are rotated over several buffers so the 256 MB MALL cannot hold them."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd import lib as L

L.load()
dev = "cuda"
R, D, F = 50432, 768, 3072
NBUF = 4
bf = torch.bfloat16


def rnd(*s):
    return [torch.randn(*s, device=dev).to(bf) for _ in range(NBUF)]


dy, s, dres, ds, x, y = rnd(R, D), rnd(R, D), rnd(R, D), rnd(R, D), rnd(R, D), rnd(R, D)
gamma, beta = torch.randn(D, device=dev).to(bf), torch.randn(D, device=dev).to(bf)
mean, rstd = torch.randn(R, device=dev), torch.rand(R, device=dev) + 0.5
dgam, dbet, dcol = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
dg, h, dh = rnd(R, F), rnd(R, F), rnd(R, F)
dbf = torch.zeros(F, device=dev)

cases = {
    "ln_bwd (dy,s,dres->ds)": (4 * R * D * 2, lambda i: L.layernorm_bwd(dy[i], s[i], gamma, mean, rstd, dres[i], ds[i], dgam, dbet, R, D, D, D, D, L.BF16)),
    "ln_bwd +dcol": (4 * R * D * 2, lambda i: L.layernorm_bwd(dy[i], s[i], gamma, mean, rstd, dres[i], ds[i], dgam, dbet, R, D, D, D, D, L.BF16, dcol=dcol)),
    "ln_bwd no dres": (3 * R * D * 2, lambda i: L.layernorm_bwd(dy[i], s[i], gamma, mean, rstd, None, ds[i], dgam, dbet, R, D, D, D, D, L.BF16)),
    "ln_fwd (x->y)": (2 * R * D * 2, lambda i: L.add_layernorm_fwd(x[i], None, gamma, beta, None, y[i], mean, rstd, R, D, D, D, 1e-6, L.BF16)),
    "gelu_bwd_colsum": (3 * R * F * 2, lambda i: L.gelu_bwd_colsum(dg[i], h[i], dh[i], dbf, R, F, F, L.BF16)),
    "colsum 768": (R * D * 2, lambda i: L.colsum_accum(dy[i], dcol, R, D, D, L.BF16)),
    "colsum 2304 (ld 2304)": (R * 2304 * 2, lambda i: L.colsum_accum(dg[i], dbf, R, 2304, 2304, L.BF16)),
}
only = sys.argv[1:]
for name, (nbytes, fn) in cases.items():
    if only and not any(o in name for o in only):
        continue
    for i in range(NBUF):
        fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for rnd_ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(NBUF):
            fn(i)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / NBUF)
    print(f"{name:28s} {best * 1e3:8.1f} us  {nbytes / best / 1e6:8.0f} GB/s", flush=True)
