#!/usr/bin/env python
"""Timing of the fused attention kernels on the cfg-2 shapes (B=256: ViT 12 heads x 197 tokens; DistilBERT 12 x 64)."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd import lib as L
L.load()
for name, B, H, S, masked in (("vit", 256, 12, 197, False), ("text", 256, 12, 64, True)):
    dh = 64
    qkv = torch.randn(B * S, 3 * H * dh, device="cuda").bfloat16()
    out = torch.empty(B * S, H * dh, device="cuda", dtype=torch.bfloat16)
    dout = torch.randn_like(out)
    dqkv = torch.empty_like(qkv)
    lse = torch.empty(B, H, S, device="cuda")
    mask = None
    if masked:
        mask = (torch.arange(S, device="cuda")[None, :] < torch.randint(8, S + 1, (B, 1), device="cuda")).int().contiguous()
    sc = 1 / math.sqrt(dh)
    def fwd(): L.mha_fwd(qkv, mask, out, lse, B, H, S, dh, sc, L.BF16)
    def bwd(): L.mha_bwd(qkv, mask, out, dout, lse, dqkv, B, H, S, dh, sc, L.BF16)
    db = torch.zeros(3 * H * dh, device="cuda")
    def bwd_cs(): L.mha_bwd(qkv, mask, out, dout, lse, dqkv, B, H, S, dh, sc, L.BF16, colsum=db)
    for fn, nm, mult in ((fwd, "fwd", 1.0), (bwd, "bwd", 2.5), (bwd_cs, "bwd+bias colsum", 2.5)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        fl = 4.0 * B * H * S * S * dh * mult
        print(f"{name} {nm}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF (algorithmic)", flush=True)
