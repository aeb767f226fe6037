"""Times the implicit-GEMM 3x3 convolution kernels (csrc/conv_igemm.hip) at EfficientNetV2-L's FusedMBConv shapes (B = 128,
480 px) next to the im2row + GEMM path they replace.  python tools/conv_igemm_bench.py [--batch 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd import lib as L   # noqa: E402

BF = torch.bfloat16


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    B = a.batch
    L.load()
    for name, H, Cin, Cout in [("stage1 32->32 @240", 240, 32, 32), ("stage2 64->256 @120", 120, 64, 256), ("stage3 96->384 @60", 60, 96, 384),
                               ("stage2 dgrad 256->64", 120, 256, 64), ("stage3 dgrad 384->96", 60, 384, 96), ("stage1 dgrad 32->32", 240, 32, 32)]:
        W = H
        P = B * H * W
        x = torch.randn(P, Cin, device="cuda").to(BF)
        w = (torch.randn(Cout, 9 * Cin, device="cuda") * 0.05).to(BF)
        z = torch.empty(P, Cout, device="cuda", dtype=BF)
        ns = L.conv3x3_stat_slots(B, H, W)
        parts = (torch.empty(ns, Cout, device="cuda"), torch.empty(ns, Cout, device="cuda"), torch.empty(ns, device="cuda"))
        mean, rstd = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")
        flop = 2.0 * P * 9 * Cin * Cout
        t0 = timed(lambda: L.conv3x3_fwd(x, w, z, B, H, W, Cin, Cout, L.BF16))
        t1 = timed(lambda: L.conv3x3_fwd(x, w, z, B, H, W, Cin, Cout, L.BF16, parts))
        t2 = timed(lambda: L.conv_bn_finish(parts, B, H, W, mean, rstd, None, None, Cout, 1e-3, 0.0))
        byt = 2.0 * P * (Cin + Cout)
        print(f"{name}: fwd {t0:.0f} us ({flop / t0 / 1e6:.0f} TF/s, {byt / t0 / 1e6:.2f} TB/s algorithmic), with moments {t1:.0f} us, finish {t2:.0f} us", flush=True)
        if "dgrad" in name:
            continue
        K = 9 * Cin
        Kp = (K + 127) // 128 * 128 if Cout % 128 == 0 else K
        col = torch.empty(P, Kp, device="cuda", dtype=BF)
        wp = torch.zeros(Cout, Kp, device="cuda", dtype=BF)
        wp[:, :K] = w
        t3 = timed(lambda: L.im2row3x3_tap(x, col, B, H, W, Cin, 1, Kp, L.BF16))
        t4 = timed(lambda: L.gemm(col, wp, z, M=P, N=Cout, K=Kp, lda=Kp, ldb=Kp, ldc=Cout, dtype=L.BF16))
        t5 = timed(lambda: L.bn_stats(z, mean, rstd, None, None, P, Cout, Cout, 1e-3, 0.0, True, L.BF16))
        print(f"    im2row {t3:.0f} us + gemm {t4:.0f} us + bn_stats {t5:.0f} us = {t3 + t4 + t5:.0f} us", flush=True)
        dz = torch.randn(P, Cout, device="cuda").to(BF)
        dw = torch.zeros(Cout, K, device="cuda")
        t6 = timed(lambda: L.conv3x3_wgrad(dz, x, dw, B, H, W, Cin, Cout, L.BF16))
        dwp = torch.zeros(Cout, Kp, device="cuda")
        Pk = (P + 63) // 64 * 64
        t7 = timed(lambda: L.gemm(dz, col, dwp, M=Cout, N=Kp, K=Pk, lda=Cout, ldb=Kp, ldc=Kp, a_layout=L.KROW, b_layout=L.KROW, accum=True, dtype=L.BF16))
        print(f"    wgrad {t6:.0f} us ({flop / t6 / 1e6:.0f} TF/s)   [im2row {t3:.0f} + gemm {t7:.0f} = {t3 + t7:.0f} us]", flush=True)


if __name__ == "__main__":
    main()
