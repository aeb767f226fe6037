"""What does a READ-ONLY pass reach on this part?  (DESIGN 3d claims ~4.2 TB/s whatever the access pattern; this checks it against
torch's own reduction / copy kernels and the library's column sums on a 1 GiB bf16 tensor.)  python tools/hbm_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd import lib as L

L.load()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


R, C = 1 << 19, 1024          # 1 GiB of bf16
x = torch.randn(R, C, device="cuda").bfloat16()
y = torch.empty_like(x)
xf = torch.randn(R // 2, C, device="cuda")           # 1 GiB of fp32
nbytes = x.numel() * 2
out = torch.zeros(C, device="cuda")
mean, rstd = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
tests = {
    "torch.sum(bf16 -> fp32), all elements (read only)": (lambda: torch.sum(x, dtype=torch.float32), nbytes),
    "torch.sum(fp32), all elements (read only)": (lambda: torch.sum(xf), xf.numel() * 4),
    "torch.sum(bf16, dim 0) column sums (read only)": (lambda: torch.sum(x, dim=0, dtype=torch.float32), nbytes),
    "torch.amax(bf16) (read only)": (lambda: torch.amax(x), nbytes),
    "y.copy_(x) (read + write)": (lambda: y.copy_(x), 2 * nbytes),
    "y.fill_(0) (write only)": (lambda: y.fill_(0), nbytes),
    "mmrca_colsum_accum bf16 (read only)": (lambda: L.colsum_accum(x, out, R, C, C, L.BF16), nbytes),
    "mmrca_bn_stats one pass bf16 (read only)": (lambda: L.bn_stats(x, mean, rstd, rm, rv, R, C, C, 1e-3, 0.1, True, L.BF16), nbytes),
}
for name, (fn, nb) in tests.items():
    t = timeit(fn)
    print(f"{name:60s} {t * 1e6:8.1f} us   {nb / t / 1e12:5.2f} TB/s")
