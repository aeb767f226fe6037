"""Squeeze-excitation MLP: the fused kernels (mmrca_se_mlp_fwd / _bwd) against the GEMM + bias_act sequence they replace, isolated.
python tools/se_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd import lib as L

L.load()
dt, dc = torch.bfloat16, L.BF16


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, c, sq in [(64, 320, 20), (64, 640, 40), (64, 1056, 44), (64, 1824, 76), (64, 3072, 128), (128, 1344, 56), (128, 2304, 96), (128, 3840, 160)]:
    r = lambda *s: (torch.randn(*s, device="cuda") * 0.3).to(dt)
    pooled, w1, b1, w2, b2, ds = r(B, c), r(sq, c), r(sq), r(c, sq), r(c), r(B, c)
    h_pre, h, s_pre, s = r(B, sq), r(B, sq), r(B, c), r(B, c)
    ds_pre, dh_pre, dpool, dh = r(B, c), r(B, sq), r(B, c), r(B, sq)
    gw1, gb1, gw2, gb2 = (torch.zeros(*x, device="cuda") for x in ((sq, c), (sq,), (c, sq), (c,)))

    def fwd_f():
        L.se_mlp_fwd(pooled, w1, b1, w2, b2, h_pre, h, s_pre, s, B, c, sq, dc)

    def bwd_f():
        L.se_mlp_bwd(ds, pooled, h_pre, h, s_pre, w1, b1, w2, b2, ds_pre, dh_pre, dpool, gw1, gb1, gw2, gb2, B, c, sq, dc)

    def fwd_g():
        L.gemm(pooled, w1, h_pre, M=B, N=sq, K=c, lda=c, ldb=c, ldc=sq, dtype=dc)
        L.bias_act_fwd(h_pre, b1, h, B, sq, L.CONV_SILU, dc)
        L.gemm(h, w2, s_pre, M=B, N=c, K=sq, lda=sq, ldb=sq, ldc=c, dtype=dc)
        L.bias_act_fwd(s_pre, b2, s, B, c, L.CONV_SIGMOID, dc)

    def bwd_g():
        L.bias_act_bwd(ds, s_pre, b2, ds_pre, gb2, B, c, L.CONV_SIGMOID, dc)
        L.gemm(ds_pre, h, gw2, M=c, N=sq, K=64 * ((B + 63) // 64), lda=c, ldb=sq, ldc=sq, a_layout=L.KROW, b_layout=L.KROW, accum=True, dtype=dc)
        L.gemm(ds_pre, w2, dh, M=B, N=sq, K=c, lda=c, ldb=sq, ldc=sq, a_layout=L.ROWK, b_layout=L.KROW, dtype=dc)
        L.bias_act_bwd(dh, h_pre, b1, dh_pre, gb1, B, sq, L.CONV_SILU, dc)
        L.gemm(dh_pre, pooled, gw1, M=sq, N=c, K=64 * ((B + 63) // 64), lda=sq, ldb=c, ldc=c, a_layout=L.KROW, b_layout=L.KROW, accum=True, dtype=dc)
        L.gemm(dh_pre, w1, dpool, M=B, N=c, K=sq, lda=sq, ldb=c, ldc=c, a_layout=L.ROWK, b_layout=L.KROW, dtype=dc)

    print(f"B={B} c={c} sq={sq}: forward fused {timeit(fwd_f):7.1f} us  GEMMs {timeit(fwd_g):7.1f} us | backward fused {timeit(bwd_f):7.1f} us  GEMMs {timeit(bwd_g):7.1f} us")
