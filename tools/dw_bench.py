"""Times the depthwise 3x3 / BatchNorm / squeeze-excitation kernels of the conv path at EfficientNetV2-L's MBConv shapes
(B = 128, 480 px input) with their algorithmic HBM rate.  python tools/dw_bench.py"""
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
    L.load()
    B = 128
    for name, H, C in [("stage4 30x30x768", 30, 768), ("stage5 30x30x1344", 30, 1344), ("stage6 15x15x2304", 15, 2304), ("stage7 15x15x3840", 15, 3840)]:
        P = B * H * H
        x = torch.randn(P, C, device="cuda").to(BF)
        dy = torch.randn(P, C, device="cuda").to(BF)
        w = torch.randn(C, 9, device="cuda").to(BF)
        y = torch.empty_like(x)
        dx = torch.empty_like(x)
        dw = torch.zeros(C, 9, device="cuda")
        mb = P * C * 2 / 1e6
        t0 = timed(lambda: L.dwconv3x3_fwd(x, w, y, B, H, H, C, 1, L.BF16))
        t1 = timed(lambda: L.dwconv3x3_bwd(dy, x, w, dx, None, B, H, H, C, 1, L.BF16))
        ws = torch.empty(16 << 20, device="cuda")
        t2 = timed(lambda: L.dwconv3x3_bwd(dy, x, w, None, dw, B, H, H, C, 1, L.BF16, ws=ws))
        mean, rstd = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        t3 = timed(lambda: L.bn_stats(x, mean, rstd, None, None, P, C, C, 1e-3, 0.0, True, L.BF16))
        print(f"{name}: {mb:.0f} MB/tensor  fwd {t0:.0f} us ({2 * mb / t0:.2f} TB/s)  bwd-data {t1:.0f} us ({2 * mb / t1:.2f})  "
              f"bwd-weight {t2:.0f} us ({2 * mb / t2:.2f})  bn moments {t3:.0f} us ({mb / t3:.2f})", flush=True)


if __name__ == "__main__":
    main()
