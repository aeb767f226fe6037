"""Implicit-GEMM 3x3 convolutions (csrc/conv_igemm.hip: no patch matrix) against torch.nn.functional.conv2d on the same bf16
operands: forward, the BatchNorm moments collected in its epilogue, the input gradient (the same kernel on dz with flipped,
transposed weights) and the weight gradient.  Reference op: torchvision's Conv2dNormActivation 3x3 of FusedMBConv
(multimodal_model.py:113-126).  Needs an MI355X."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from garbage_classification_rca_amd import lib as L                     # noqa: E402

BF = torch.bfloat16


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available()
    L.load()


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def rows(t):      # NCHW -> [B*H*W, C]
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()


def nchw(r, B, H, W):
    return r.view(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


def tap_major(w, pad=False):          # [Cout, Cin, 3, 3] -> [Cout, 9*Cin], column tap*Cin + ci (pad: Cin rounded up to 32 per tap, zeros)
    t = w.permute(0, 2, 3, 1)
    if pad and w.shape[1] % 32:
        t = F.pad(t, (0, 32 - w.shape[1] % 32))
    return t.reshape(w.shape[0], -1).contiguous()


# (B, H, W, Cin, Cout): patch shapes 8x16 / 4x32 / 2x64, ragged patches, ragged channel tiles (NJ = 1..4), several channel
# tiles, halo in one piece (Cin <= 96) and in 96- / 64- / 32-channel pieces
SHAPES = [(2, 12, 20, 32, 32), (1, 9, 33, 64, 256), (2, 16, 16, 96, 384), (1, 30, 30, 64, 72), (3, 7, 5, 32, 16),
          (1, 24, 60, 256, 64), (2, 11, 13, 384, 96), (1, 8, 64, 160, 40), (1, 5, 70, 32, 136),
          (2, 12, 20, 24, 24), (1, 9, 33, 48, 192), (2, 10, 10, 80, 320), (1, 6, 18, 8, 16)]         # EfficientNetV2-M's 24 / 48 / 80 channels


@pytest.mark.parametrize("B,H,W,Cin,Cout", SHAPES + [(10, 64, 64, 32, 32)])           # the last: 640 slots, two merge levels
def test_forward_and_batchnorm_moments(B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin)
    x = (torch.randn(B, Cin, H, W, generator=g) + 0.3).to(BF)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(BF)
    ref = F.conv2d(x.float(), w.float(), None, 1, 1)
    z = torch.full((B * H * W, Cout), float("nan"), device="cuda", dtype=BF)
    ns = L.conv3x3_stat_slots(B, H, W)
    parts = (torch.full((ns, Cout), float("nan"), device="cuda"), torch.full((ns, Cout), float("nan"), device="cuda"),
             torch.full((ns,), float("nan"), device="cuda"))
    L.conv3x3_fwd(rows(x).cuda(), tap_major(w, pad=True).cuda(), z, B, H, W, Cin, Cout, L.BF16, parts)
    torch.cuda.synchronize()
    e = rel(nchw(z.float().cpu(), B, H, W), ref)
    assert e < 6e-3, e                                   # one bf16 rounding of the output
    assert float(torch.nan_to_num(parts[2]).sum()) == B * H * W          # (second-level slots are still untouched)
    mean, rstd = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    L.conv_bn_finish(parts, B, H, W, mean, rstd, rm, rv, Cout, 1e-3, 0.1)
    zf = z.float()
    n = B * H * W
    assert rel(mean, zf.mean(0)) < 2e-5
    var = zf.var(0, unbiased=False)
    assert rel(rstd, (var + 1e-3).rsqrt()) < 2e-5
    assert rel(rm, 0.1 * zf.mean(0)) < 2e-5
    assert rel(rv, 0.9 + 0.1 * var * n / (n - 1)) < 2e-5
    # without the statistics: same outputs
    z2 = torch.empty_like(z)
    L.conv3x3_fwd(rows(x).cuda(), tap_major(w, pad=True).cuda(), z2, B, H, W, Cin, Cout, L.BF16)
    assert torch.equal(z2, z)


def test_moments_of_a_far_off_centre_channel():
    """(count, mean, M2) per wave + Chan merge needs no shift: mean 50, sigma 0.1 (the case ADVICE r2 raised for the one-pass sums)"""
    B, H, W, Cin, Cout = 2, 40, 40, 32, 32
    g = torch.Generator().manual_seed(5)
    x = torch.zeros(B, Cin, H + 2, W + 2)
    x[:, 0] = 1.0                                          # a constant plane: the centre tap of channel 0 carries the offset ...
    x[:, 1:] = torch.randn(B, Cin - 1, H + 2, W + 2, generator=g)
    x = x[:, :, 1:-1, 1:-1].contiguous().to(BF)            # (interior only: borders see the zero padding like everything else)
    w = torch.zeros(Cout, Cin, 3, 3)
    w[:, 0, 1, 1] = 50.0
    w[:, 1, 1, 1] = 0.1
    w = w.to(BF)
    z = torch.empty(B * H * W, Cout, device="cuda", dtype=BF)
    ns = L.conv3x3_stat_slots(B, H, W)
    parts = (torch.empty(ns, Cout, device="cuda"), torch.empty(ns, Cout, device="cuda"), torch.empty(ns, device="cuda"))
    L.conv3x3_fwd(rows(x).cuda(), tap_major(w).cuda(), z, B, H, W, Cin, Cout, L.BF16, parts)
    mean, rstd = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")
    L.conv_bn_finish(parts, B, H, W, mean, rstd, None, None, Cout, 0.0, 0.0)
    zd = z.double()
    assert rel(mean, zd.mean(0)) < 1e-6
    assert rel(rstd, zd.var(0, unbiased=False).rsqrt()) < 1e-4


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 12, 20, 32, 32), (1, 9, 33, 64, 256), (2, 16, 16, 96, 384), (1, 24, 60, 64, 64),
                                            (2, 12, 20, 24, 24), (1, 9, 33, 48, 192), (2, 10, 10, 80, 40)])
def test_input_gradient_is_the_same_kernel_on_flipped_weights(B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).to(BF).float().requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(BF).float()
    dz = torch.randn(B, Cout, H, W, generator=g).to(BF).float()
    F.conv2d(x, w, None, 1, 1).backward(dz)
    wflip = w.view(Cout, Cin, 9).flip(2).permute(1, 2, 0)                                               # [Cin, 9, Cout]: tap' * Cout_p + co
    if Cout % 32:
        wflip = F.pad(wflip, (0, 32 - Cout % 32))
    wflip = wflip.reshape(Cin, -1).contiguous().to(BF)
    dx = torch.empty(B * H * W, Cin, device="cuda", dtype=BF)
    L.conv3x3_fwd(rows(dz).to(BF).cuda(), wflip.cuda(), dx, B, H, W, Cout, Cin, L.BF16)
    assert rel(nchw(dx.float().cpu(), B, H, W), x.grad) < 6e-3


@pytest.mark.parametrize("B,H,W,Cin,Cout", SHAPES + [(4, 50, 50, 64, 256), (2, 3, 3, 8, 8)])
def test_weight_gradient(B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).to(BF).float()
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    dz = torch.randn(B, Cout, H, W, generator=g).to(BF).float()
    F.conv2d(x, w, None, 1, 1).backward(dz)
    ref = tap_major(w.grad)
    dw = torch.full((Cout, 9 * Cin), 0.5, device="cuda")                    # accumulates: += on top of what is there
    L.conv3x3_wgrad(rows(dz).to(BF).cuda(), rows(x).to(BF).cuda(), dw, B, H, W, Cin, Cout, L.BF16)
    assert rel(dw.cpu() - 0.5, ref) < 2e-5


@pytest.mark.parametrize("H,Cin,Cout", [(240, 32, 32), (120, 64, 256)])
def test_full_size_layers_agree_with_the_patch_matrix_path(H, Cin, Cout):
    """BASELINE configs[2] sizes (B = 128, EfficientNetV2-L stage 1 / stage 2: 7.4 M / 1.8 M pixels): the implicit-GEMM forward,
    weight gradient and input gradient against im2row + mmrca_gemm (+ col2im) on the same bf16 operands -- two bf16 paths, so the
    bound is a few bf16 roundings; what it guards is the index arithmetic at full size (pixel x channel offsets past 2^31 bytes)."""
    B = 128
    W = H
    P = B * H * W
    g = torch.Generator(device="cuda").manual_seed(H)
    x = torch.randn(P, Cin, device="cuda", generator=g).to(BF)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(BF)
    wt = tap_major(w)
    K = 9 * Cin
    col = torch.empty(P, K, device="cuda", dtype=BF)
    L.im2row3x3_tap(x, col, B, H, W, Cin, 1, K, L.BF16)
    z_ref = torch.empty(P, Cout, device="cuda", dtype=BF)
    L.gemm(col, wt, z_ref, M=P, N=Cout, K=K, lda=K, ldb=K, ldc=Cout, dtype=L.BF16)
    z = torch.empty(P, Cout, device="cuda", dtype=BF)
    ns = L.conv3x3_stat_slots(B, H, W)
    parts = (torch.empty(ns, Cout, device="cuda"), torch.empty(ns, Cout, device="cuda"), torch.empty(ns, device="cuda"))
    L.conv3x3_fwd(x, wt, z, B, H, W, Cin, Cout, L.BF16, parts)
    assert rel(z[-4096:], z_ref[-4096:]) < 1.5e-2 and rel(z[:4096], z_ref[:4096]) < 1.5e-2
    assert float((z.float() - z_ref.float()).abs().max()) / float(z_ref.float().abs().max()) < 1.5e-2
    mean, rstd = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")
    L.conv_bn_finish(parts, B, H, W, mean, rstd, None, None, Cout, 1e-3, 0.0)
    assert rel(mean, z.float().mean(0)) < 1e-4
    # weight gradient (fp32 accumulation in both paths)
    dz = torch.randn(P, Cout, device="cuda", generator=g).to(BF)
    dw = torch.zeros(Cout, K, device="cuda")
    L.conv3x3_wgrad(dz, x, dw, B, H, W, Cin, Cout, L.BF16)
    dw_ref = torch.zeros(Cout, K, device="cuda")
    Pk = (P + 63) // 64 * 64
    dzp = torch.zeros(Pk, Cout, device="cuda", dtype=BF); dzp[:P] = dz
    colp = torch.zeros(Pk, K, device="cuda", dtype=BF); colp[:P] = col
    L.gemm(dzp, colp, dw_ref, M=Cout, N=K, K=Pk, lda=Cout, ldb=K, ldc=K, a_layout=L.KROW, b_layout=L.KROW, accum=True, dtype=L.BF16)
    assert rel(dw, dw_ref) < 2e-3
    del colp, dzp
    # input gradient: the forward kernel on dz with flipped weights vs GEMM + col2im
    wflip = w.float().view(Cout, Cin, 9).flip(2).permute(1, 2, 0).reshape(Cin, 9 * Cout).contiguous().to(BF)
    dx = torch.empty(P, Cin, device="cuda", dtype=BF)
    L.conv3x3_fwd(dz, wflip, dx, B, H, W, Cout, Cin, L.BF16)
    L.gemm(dz, wt, col, M=P, N=K, K=Cout, lda=Cout, ldb=K, ldc=K, a_layout=L.ROWK, b_layout=L.KROW, dtype=L.BF16)
    dx_ref = torch.empty(P, Cin, device="cuda", dtype=BF)
    L.col2im3x3_tap(col, dx_ref, B, H, W, Cin, 1, K, L.BF16)
    assert float((dx.float() - dx_ref.float()).abs().max()) / float(dx_ref.float().abs().max()) < 3e-2     # (col2im sums nine bf16-rounded terms)
