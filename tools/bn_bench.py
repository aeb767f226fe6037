#!/usr/bin/env python
"""BatchNorm passes of the conv path on one [rows, C] bf16 tensor, 20 calls each -- run under `rocprofv3 --kernel-trace --stats` to get the
per-kernel time; prints the algorithmic bytes of each pass.   usage: bn_bench.py rows C"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from garbage_classification_rca_amd import lib as L

rows, C = int(sys.argv[1]), int(sys.argv[2])
L.load()
dev = "cuda"
x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
dy = torch.randn(rows, C, device=dev).to(torch.bfloat16)
y, dx = torch.empty_like(x), torch.empty_like(x)
stats = torch.zeros(2, C, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
g, b = torch.ones(C, device=dev, dtype=torch.bfloat16), torch.zeros(C, device=dev, dtype=torch.bfloat16)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
scratch = torch.zeros(1, 2 * C, device=dev)
ws = torch.empty(4 << 20, device=dev) if os.environ.get("BN_WS", "1") == "1" else None      # BN_WS=0: the slice-per-workgroup reductions
unit = rows * C * 2
for _ in range(20):
    L.bn_stats(x, stats[0], stats[1], rm, rv, rows, C, C, 1e-3, 0.1, True, L.BF16, ws=ws)
    L.bn_act_fwd(x, stats[0], stats[1], g, b, y, rows, C, 1, L.BF16)
    L.bn_act_bwd(dy, x, stats[0], stats[1], g, b, dx, dg, db, scratch, rows, C, 1, True, L.BF16, ws=ws)
torch.cuda.synchronize()
print(f"rows {rows} C {C}: one tensor = {unit / 1e6:.1f} MB; moments read 1, forward 2, backward reduce 2, backward apply 3 tensors")
