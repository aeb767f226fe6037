"""Kernel-level parity: every libmmrca entry point, called through the C ABI, against a plain PyTorch fp32
statement of the same op (and the oracle for the fused head).  Needs an MI355X."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from garbage_classification_rca_amd import lib as L   # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "these tests need a GPU"
    L.load()
    torch.manual_seed(0)


def dev(x, dt=torch.float32):
    return x.to("cuda", dt).contiguous()


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


# ------------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------------
def _gemm_case(M, N, K, a_layout, b_layout, dt, impl, act=L.ACT_NONE, bias=False, addend=False, preact=False, accum=False):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    Kp = (K + 63) // 64 * 64 if a_layout == L.KROW or b_layout == L.KROW else K
    A = torch.randn(M, K, generator=g) * 0.5
    Bm = torch.randn(N, K, generator=g) * 0.5
    Ad = dev(A, dt) if a_layout == L.ROWK else dev(F.pad(A.t(), (0, 0, 0, Kp - K)), dt)      # [Kp, M]
    Bd = dev(Bm, dt) if b_layout == L.ROWK else dev(F.pad(Bm.t(), (0, 0, 0, Kp - K)), dt)    # [Kp, N]
    Af = (Ad if a_layout == L.ROWK else Ad[:K].t()).float()
    Bf = (Bd if b_layout == L.ROWK else Bd[:K].t()).float()
    bias_d = dev(torch.randn(N, generator=g), dt) if bias else None
    add_d = dev(torch.randn(M, N, generator=g), dt) if addend else None
    pre_d = torch.empty(M, N, device="cuda", dtype=dt) if preact else None
    ref = Af.double() @ Bf.double().t()
    if bias:
        ref = ref + bias_d.double()
    ref_pre = ref.clone()
    if act == L.ACT_GELU:
        ref = F.gelu(ref)
    if addend:
        ref = ref + add_d.double()
    if accum:
        Cd = dev(torch.randn(M, N, generator=g))
        ref = ref + Cd.double()
    else:
        Cd = torch.empty(M, N, device="cuda", dtype=dt)
    Kcall = Kp if (a_layout == L.KROW or b_layout == L.KROW) else K
    if a_layout == L.ROWK and Kcall != K:
        Ad = dev(F.pad(A, (0, Kcall - K)), dt)
    if impl == L.IMPL_MFMA256 and a_layout == L.ROWK:
        Ad = _pad256(Ad)
        add_call = None if add_d is None else _pad256(add_d)
    else:
        add_call = add_d
    if b_layout == L.ROWK and Kcall != K:
        Bd = dev(F.pad(Bm, (0, Kcall - K)), dt)
    rr = None
    if impl == L.IMPL_MFMA256 and a_layout == L.ROWK and M % 256:     # ragged M on the 256x256 kernel: state the readable rows
        rr = (Ad.shape[0], add_call.shape[0] if add_call is not None else 0)
    L.gemm(Ad, Bd, Cd, bias=bias_d, addend=add_call, preact=pre_d, M=M, N=N, K=Kcall,
           lda=Ad.shape[1], ldb=Bd.shape[1], ldc=N, a_layout=a_layout, b_layout=b_layout, act=act, accum=accum,
           dtype=L.dtype_code(dt), impl=impl, rows_readable=rr)
    torch.cuda.synchronize()
    tol = TOL[dt] * (1 if dt == torch.float32 else 1.0)
    assert rel_err(Cd, ref) < tol, (M, N, K, a_layout, b_layout, dt, impl)
    if preact:
        assert rel_err(pre_d, ref_pre) < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layouts", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_ref_layouts(dt, layouts):
    _gemm_case(100, 72, 52, layouts[0], layouts[1], dt, L.IMPL_REF, bias=True)
    _gemm_case(64, 4, 3072, layouts[0], layouts[1], dt, L.IMPL_REF)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_ref_epilogues(dt):
    _gemm_case(70, 68, 64, 0, 0, dt, L.IMPL_REF, act=L.ACT_GELU, bias=True, addend=True, preact=True)
    _gemm_case(70, 68, 128, 1, 1, dt, L.IMPL_REF, accum=True)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layouts", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_general_kernel_on_the_fp32_matrix_cores(dt, layouts):
    """gemm_gen_k = what AUTO runs for fp32 and for bf16 shapes outside the bf16 MFMA kernels: odd shapes (edge tiles in
    M, N and K), all operand layouts, every epilogue, accumulate mode, and the split-K path of long-contraction weight
    gradients (conv: K = B*H*W)."""
    al, bl = layouts
    _gemm_case(100, 72, 52, al, bl, dt, L.IMPL_AUTO, bias=True)                       # one partial tile in every dimension
    _gemm_case(257, 96, 216, al, bl, dt, L.IMPL_AUTO, bias=True, addend=True)         # conv 3x3, 24 -> 96 channels
    _gemm_case(64, 24, 96, al, bl, dt, L.IMPL_AUTO)                                   # conv 1x1, N below one tile
    _gemm_case(300, 304, 1824, al, bl, dt, L.IMPL_AUTO, act=L.ACT_GELU, bias=True, preact=True)   # K > 1024: third summation level
    _gemm_case(70, 68, 128, al, bl, dt, L.IMPL_AUTO, accum=True)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_general_kernel_split_k_weight_gradient(dt):
    """Few output tiles, long contraction, accumulate mode -> K ranges over blockIdx.y with fp32 atomics."""
    M, N, K = 96, 216, 40_000
    g = torch.Generator().manual_seed(3)
    A = dev(torch.randn(K, M, generator=g) * 0.5, dt)       # KROW operands: dY [rows, cout], X [rows, 9 cin]
    Bm = dev(torch.randn(K, N, generator=g) * 0.5, dt)
    C0 = torch.randn(M, N, generator=g)
    Cd = dev(C0)
    L.gemm(A, Bm, Cd, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, a_layout=L.KROW, b_layout=L.KROW, accum=True, dtype=L.dtype_code(dt),
           impl=L.IMPL_AUTO)
    ref = C0.double() + A.double().cpu().t() @ Bm.double().cpu()
    assert rel_err(Cd.cpu(), ref) < 2e-5, rel_err(Cd.cpu(), ref)     # the bf16 INPUTS are taken as given; products and sums are fp32


def test_gemm_mfma_exact_integers():
    """A = I-like / asymmetric small integers: catches swapped fragment maps exactly (no rounding)."""
    M, N, K = 256, 256, 128
    A = torch.zeros(M, K); A[torch.arange(M), torch.arange(M) % K] = 1.0; A[:, 3] += 2.0
    B = (torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 5) % 7 - 3.0
    for (al, bl) in [(0, 0), (0, 1), (1, 1), (1, 0)]:
        Ad = dev(A if al == 0 else A.t(), torch.bfloat16)
        Bd = dev(B if bl == 0 else B.t(), torch.bfloat16)
        Cd = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        L.gemm(Ad, Bd, Cd, M=M, N=N, K=K, lda=Ad.shape[1], ldb=Bd.shape[1], ldc=N, a_layout=al, b_layout=bl,
               dtype=L.BF16, impl=L.IMPL_MFMA)
        torch.cuda.synchronize()
        assert torch.equal(Cd.float().cpu(), A @ B.t()), (al, bl)


@pytest.mark.parametrize("layouts", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_mfma_layouts(layouts):
    al, bl = layouts
    M = 384 if al == L.KROW else 300          # ragged M only for a row-major A (the forward / dgrad cases)
    _gemm_case(M, 256, 192, al, bl, torch.bfloat16, L.IMPL_MFMA, bias=True)
    _gemm_case(M, 128, 64, al, bl, torch.bfloat16, L.IMPL_MFMA)


def test_gemm_mfma_epilogues_and_wgrad():
    _gemm_case(200, 384, 256, 0, 0, torch.bfloat16, L.IMPL_MFMA, act=L.ACT_GELU, bias=True, addend=True, preact=True)
    _gemm_case(788, 768, 768, 0, 1, torch.bfloat16, L.IMPL_MFMA, addend=True)      # dgrad, ViT rows at B=4
    # wgrad: dW[N,K] += dY^T X, contraction over 788 rows padded to 832 with zeros; split-K + fp32 atomics
    _gemm_case(768, 768, 788, 1, 1, torch.bfloat16, L.IMPL_MFMA, accum=True)
    _gemm_case(256, 128, 5000, 1, 1, torch.bfloat16, L.IMPL_MFMA, accum=True)


@pytest.mark.parametrize("layouts", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_bk32_kernel(layouts):
    """128x128x32 kernel (four blocks per CU): exact integers first (new 64-byte-row LDS image), then random data, ragged M,
    odd number of 32-deep K steps, epilogues, split-K accumulate."""
    al, bl = layouts
    M, N, K = 256, 256, 96
    A = ((torch.arange(M)[:, None] * 7 + torch.arange(K)[None, :] * 3) % 5 - 2.0)
    B = ((torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 5) % 7 - 3.0)
    Kp = 128
    Ap, Bp = F.pad(A, (0, Kp - K)), F.pad(B, (0, Kp - K))
    Ad, Bd = dev(Ap if al == 0 else Ap.t(), torch.bfloat16), dev(Bp if bl == 0 else Bp.t(), torch.bfloat16)
    Cd = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    L.gemm(Ad, Bd, Cd, M=M, N=N, K=Kp, lda=Ad.shape[1], ldb=Bd.shape[1], ldc=N, a_layout=al, b_layout=bl, dtype=L.BF16, impl=L.IMPL_MFMA_BK32)
    torch.cuda.synchronize()
    assert torch.equal(Cd.float().cpu(), (A @ B.t()).bfloat16().float()), (al, bl)
    Mr = 384 if al == L.KROW else 300
    _gemm_case(Mr, 256, 192, al, bl, torch.bfloat16, L.IMPL_MFMA_BK32, bias=True)
    _gemm_case(Mr, 384, 256, al, bl, torch.bfloat16, L.IMPL_MFMA_BK32, act=L.ACT_GELU, bias=True, addend=True, preact=True)
    _gemm_case(768, 768, 788, 1, 1, torch.bfloat16, L.IMPL_MFMA_BK32, accum=True)


@pytest.mark.parametrize("layouts", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_single_stage_kernel(layouts):
    al, bl = layouts
    Mr = 384 if al == L.KROW else 300
    _gemm_case(Mr, 256, 192, al, bl, torch.bfloat16, L.IMPL_MFMA_1STAGE, bias=True)
    _gemm_case(Mr, 384, 256, al, bl, torch.bfloat16, L.IMPL_MFMA_1STAGE, act=L.ACT_GELU, bias=True, addend=True, preact=True)
    _gemm_case(768, 768, 788, 1, 1, torch.bfloat16, L.IMPL_MFMA_1STAGE, accum=True)


def test_gemm_removed_experimental_kernels_fail_loudly():
    """The four experimental bf16 kernels of round 1 that never made it into AUTO (persistent 128x128, 256x128 tall, 256x256 with
    sixteen / four waves) were deleted in round 2; their impl codes stay reserved and are rejected."""
    for impl in (L.IMPL_MFMA_PERSIST, L.IMPL_MFMA_TALL, L.IMPL_MFMA_256W, L.IMPL_MFMA_256X4):
        with pytest.raises(L.MmrcaError, match="removed"):
            _gemm_case(256, 256, 128, 0, 0, torch.bfloat16, impl)


def _pad256(t):
    """contract of the 256x256 kernel: round_up(M, 256) readable rows of A and of the side operand (content irrelevant: NaN)"""
    M = t.shape[0]
    return t if M % 256 == 0 else torch.cat([t, torch.full((256 - M % 256, t.shape[1]), float("nan"), device=t.device, dtype=t.dtype)])


@pytest.mark.parametrize("bl", [0, 1])
def test_gemm_mfma256_layouts_and_epilogues(bl):
    """Persistent 256x256 kernel (operand stream in flight across barriers AND across tile boundaries): exact integers
    first, then random data with K-tile counts from 2 (tail-only path) upward, odd and even (LDS stage parity at the tile
    boundary), a ragged M, and enough tiles that workgroups walk several tiles each."""
    al = 0
    M, N, K = 512, 256, 192
    A = ((torch.arange(M)[:, None] * 7 + torch.arange(K)[None, :] * 3) % 5 - 2.0)
    B = ((torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 5) % 7 - 3.0)
    Ad, Bd = dev(A, torch.bfloat16), dev(B if bl == 0 else B.t(), torch.bfloat16)
    Cd = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    L.gemm(Ad, Bd, Cd, M=M, N=N, K=K, lda=Ad.shape[1], ldb=Bd.shape[1], ldc=N, a_layout=al, b_layout=bl, dtype=L.BF16, impl=L.IMPL_MFMA256)
    torch.cuda.synchronize()
    assert torch.equal(Cd.float().cpu(), (A @ B.t()).bfloat16().float()), (al, bl)
    for K in (128, 192, 320, 768, 832):
        _gemm_case(700, 512, K, al, bl, torch.bfloat16, L.IMPL_MFMA256, bias=True)
    _gemm_case(1000, 256, 256, al, bl, torch.bfloat16, L.IMPL_MFMA256, bias=True, addend=True)
    # 26 x 12 = 312 tiles on 256 workgroups (two tiles for some), then 1,170 tiles (four to five each), odd / even K-tile counts
    _gemm_case(6600, 3072, 192, al, bl, torch.bfloat16, L.IMPL_MFMA256, bias=True, addend=True)
    _gemm_case(33200, 2304, 128, al, bl, torch.bfloat16, L.IMPL_MFMA256, bias=True)
    _gemm_case(33200, 2304, 320, al, bl, torch.bfloat16, L.IMPL_MFMA256)


@pytest.mark.parametrize("M,N,K,al,bl,accum,bias", [
    (1000, 224, 192, 0, 0, False, True),      # ragged N (EfficientNetV2-L's 1x1 projections: 224 / 1344 / 192 channels)
    (777, 1344, 224, 0, 0, False, False),     # ragged N and a contraction that is a multiple of 32, not of 64
    (1000, 224, 1344, 0, 1, False, False),    # input gradient: B is [K, N] with ragged N
    (640, 192, 768, 0, 1, False, False),
    (1344, 224, 1024, 1, 1, True, False),     # weight gradient: A is [K, M] with ragged M, B is [K, N] with ragged N
    (224, 1344, 960, 1, 1, True, False),
    (128, 56, 1344, 0, 0, False, True),       # squeeze-excitation FC (56 output channels)
    # round 5: contractions that are multiples of 8 but not of 32 -- EfficientNetV2-M (the reference's default image model,
    # multimodal_model.py:113-126): 80 / 176 / 304 channels -- on the 32-deep kernel's edge step (B chunks past K fetched as zeros)
    (14400, 1824, 304, 0, 0, False, False),   # stage-6 expand, forward
    (1000, 1056, 176, 0, 0, False, True),     # stage-5 expand (ragged M)
    (777, 320, 80, 0, 0, False, False),       # stage-4 expand: 2.5 steps
    (14400, 1824, 304, 0, 1, False, False),   # input gradient of a 1824 -> 304 projection: B is [K = 304, N = 1824]
    (900, 960, 176, 0, 1, False, False),
    (512, 128, 40, 0, 0, False, False),       # one and a quarter steps
    (512, 128, 40, 0, 1, False, False),
    (14400, 304, 1824, 0, 1, False, False),   # K = 28.5 steps of 64: the single-stage 64-deep kernel's edge step (AUTO since round 5)
    (2000, 176, 1056, 0, 0, False, True),
    (48, 192, 4096, 1, 1, True, False),       # weight gradient of a 192 -> 48 projection: M = 48 rows of output (was the general kernel)
    (16, 64, 1024, 1, 1, True, False),
])
def test_gemm_ragged_shapes_run_on_the_mfma_kernels(M, N, K, al, bl, accum, bias):
    """channel counts that are multiples of 8 but not of 128 (and contractions that are multiples of 32 but not of 64) on the
    128x128 bf16 MFMA kernels: edge tiles read a clamped chunk and never store it.  AUTO against the fp64 product; the same call
    with MMRCA_GEMM_RAGGED=0 semantics (impl REF) is the checker's checker."""
    _gemm_case(M, N, K, al, bl, torch.bfloat16, L.IMPL_AUTO, bias=bias, accum=accum)
    if K % 32 and not accum:                  # the 32-deep kernel's edge step, explicitly (AUTO takes it too)
        _gemm_case(M, N, K, al, bl, torch.bfloat16, L.IMPL_MFMA_BK32, bias=bias, accum=accum)
    if K % 64 and K >= 64 and not accum:      # ... and the single-stage 64-deep kernel's (MMRCA_GEMM_KEDGE64, off by default)
        _gemm_case(M, N, K, al, bl, torch.bfloat16, L.IMPL_MFMA_1STAGE, bias=bias, accum=accum)
    # the output columns past N and rows past M are untouched: a canary frame around C
    g = torch.Generator().manual_seed(5)
    A = dev(torch.randn((M, K) if al == 0 else (K, M), generator=g) * 0.5, torch.bfloat16)
    B = dev(torch.randn((N, K) if bl == 0 else (K, N), generator=g) * 0.5, torch.bfloat16)
    if accum:
        return
    ldc = N + 8
    C = torch.full((M + 3, ldc), 7.0, device="cuda", dtype=torch.bfloat16)
    L.gemm(A, B, C, M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], ldc=ldc, a_layout=al, b_layout=bl, dtype=L.BF16)
    torch.cuda.synchronize()
    assert float((C[:, N:] - 7.0).abs().max()) == 0.0 and float((C[M:] - 7.0).abs().max()) == 0.0
    ref = (A.float() if al == 0 else A.float().t()) @ (B.float().t() if bl == 0 else B.float())
    assert rel_err(C[:M, :N], ref) < TOL[torch.bfloat16]
    if K % 32:
        # the edge step must not depend on what lies behind the last valid chunk: poison the operands' neighbourhood (A as a column
        # window of a wider NaN-filled buffer, so that a read past K would hit NaN) and compare bit for bit with the clean run
        if al == 0:
            Aw = torch.full((M, K + 32), float("nan"), device="cuda", dtype=torch.bfloat16)
            Aw[:, :K] = A
            C2 = torch.full((M + 3, ldc), 7.0, device="cuda", dtype=torch.bfloat16)
            L.gemm(Aw, B, C2, M=M, N=N, K=K, lda=K + 32, ldb=B.shape[1], ldc=ldc, a_layout=al, b_layout=bl, dtype=L.BF16)
            torch.cuda.synchronize()
            assert torch.equal(C2, C)


@pytest.mark.parametrize("bl", [0, 1])
def test_gemm_mfma256_fused_gelu_epilogues_and_colsum(bl):
    """The 256x256 kernel's compile-time epilogues: GELU with gelu' saved (forward FFN1), multiply by the saved gelu' with
    the column sums of the stored result (input gradient of FFN2 + FFN1 bias gradient), residual addend; ragged M (the
    padded rows of A and of the side operand hold NaN)."""
    dt = torch.bfloat16
    M, N, K = 788, 512, 320
    g = torch.Generator().manual_seed(11)
    X = dev(torch.randn(M, K, generator=g) * 0.3, dt)
    Xp = _pad256(X)
    Wm = torch.randn(N, K, generator=g) * 0.2
    Wd = dev(Wm if bl == 0 else Wm.t(), dt)
    Wf = (Wd if bl == 0 else Wd.t()).double()
    bias, add = dev(torch.randn(N, generator=g) * 0.1, dt), dev(torch.randn(M, N, generator=g), dt)
    Y, Gp = torch.empty(M, N, device="cuda", dtype=dt), torch.empty(1024, N, device="cuda", dtype=dt)
    L.gemm(Xp, Wd, Y, bias=bias, preact=Gp, M=M, N=N, K=K, lda=K, ldb=Wd.shape[1], ldc=N, b_layout=bl, act=L.ACT_GELU_SAVE_GRAD,
           dtype=L.BF16, impl=L.IMPL_MFMA256, rows_readable=(Xp.shape[0], 0))
    pre = (X.double() @ Wf.t() + bias.double()).float().requires_grad_(True)
    yr = F.gelu(pre); yr.backward(torch.ones_like(yr))
    assert rel_err(Y, yr.detach()) < TOL[dt] and rel_err(Gp[:M], pre.grad) < TOL[dt]
    C = torch.empty(M, N, device="cuda", dtype=dt)
    db = dev(torch.randn(N, generator=g)); db0 = db.clone()
    L.gemm(Xp, Wd, C, preact=Gp, M=M, N=N, K=K, lda=K, ldb=Wd.shape[1], ldc=N, b_layout=bl, act=L.ACT_MUL,
           dtype=L.BF16, impl=L.IMPL_MFMA256, colsum=db, rows_readable=(Xp.shape[0], Gp.shape[0]))
    assert rel_err(C, (X.double() @ Wf.t()) * Gp[:M].double()) < TOL[dt]
    assert rel_err(db - db0, C.float().sum(0)) < 1e-4
    C2 = torch.empty(M, N, device="cuda", dtype=dt)          # residual addend (the out-projection / FFN2 forward epilogue)
    L.gemm(Xp, Wd, C2, bias=bias, addend=_pad256(add), M=M, N=N, K=K, lda=K, ldb=Wd.shape[1], ldc=N, b_layout=bl, dtype=L.BF16,
           impl=L.IMPL_MFMA256, rows_readable=(Xp.shape[0], 1024))
    # the row contract of the 256x256 kernel is checked before any launch (one call per precondition):
    with pytest.raises(L.MmrcaError, match="mmrca_gemm_rows"):           # ragged M without a statement of the readable rows
        L.gemm(Xp, Wd, C2, bias=bias, M=M, N=N, K=K, lda=K, ldb=Wd.shape[1], ldc=N, b_layout=bl, dtype=L.BF16, impl=L.IMPL_MFMA256)
    with pytest.raises(L.MmrcaError, match="rows of A"):                 # A shorter than the tiles the kernel streams
        L.gemm(X, Wd, C2, bias=bias, M=M, N=N, K=K, lda=K, ldb=Wd.shape[1], ldc=N, b_layout=bl, dtype=L.BF16, impl=L.IMPL_MFMA256,
               rows_readable=(M, 0))
    with pytest.raises(L.MmrcaError, match="side operand"):              # side operand shorter than the tiles
        L.gemm(Xp, Wd, C2, bias=bias, addend=add, M=M, N=N, K=K, lda=K, ldb=Wd.shape[1], ldc=N, b_layout=bl, dtype=L.BF16,
               impl=L.IMPL_MFMA256, rows_readable=(Xp.shape[0], M))
    assert rel_err(C2, X.double() @ Wf.t() + bias.double() + add.double()) < TOL[dt]


@pytest.mark.parametrize("M", [256 * 150, 256 * 150 + 100])
def test_gemm_mfma256_mul_epilogue_over_several_tiles_per_workgroup(M):
    """ACT_MUL on the persistent kernel with MORE tiles than workgroups (453 on 256 CUs): the gelu' factor is fetched one pass ahead by asm
    loads with counted waits, next to the stores of the passes and the operand stream of the NEXT tile -- every tile, whole and
    ragged, must come out right, and so must the column sums that ride on it."""
    dt = torch.bfloat16
    N, K = 768, 768
    g = torch.Generator(device="cuda").manual_seed(M)
    Mp = (M + 255) // 256 * 256
    Xp = torch.full((Mp, K), float("nan"), device="cuda", dtype=dt)
    Xp[:M] = (torch.randn(M, K, device="cuda", generator=g) * 0.3).to(dt)
    W = (torch.randn(K, N, device="cuda", generator=g) * 0.2).to(dt)            # KROW: the input-gradient form, dX = dY W
    Gp = torch.full((Mp, N), float("nan"), device="cuda", dtype=dt)
    Gp[:M] = (torch.rand(M, N, device="cuda", generator=g) + 0.25).to(dt)
    C = torch.empty(M, N, device="cuda", dtype=dt)
    db = torch.zeros(N, device="cuda")
    L.gemm(Xp, W, C, preact=Gp, M=M, N=N, K=K, lda=K, ldb=N, ldc=N, b_layout=L.KROW, act=L.ACT_MUL, dtype=L.BF16, impl=L.IMPL_MFMA256,
           colsum=db, rows_readable=(Mp, Mp))
    torch.cuda.synchronize()
    ref = (Xp[:M].float() @ W.float()) * Gp[:M].float()
    err = (C.float() - ref).abs().max() / ref.abs().max()
    assert float(err) < TOL[dt], float(err)
    # every 256-row tile on its own (an error confined to one tile must not hide behind the largest entry of the whole matrix)
    per_tile = ((C.float() - ref).abs().view(-1, N)[: (M // 256) * 256].view(-1, 256, N).amax(dim=(1, 2)) /
                ref.abs().view(-1, N)[: (M // 256) * 256].view(-1, 256, N).amax(dim=(1, 2)))
    assert float(per_tile.max()) < TOL[dt], float(per_tile.max())
    assert rel_err(db, C.float().sum(0)) < 1e-4


@pytest.mark.parametrize("dt,impl", [(torch.float32, L.IMPL_REF), (torch.bfloat16, L.IMPL_REF), (torch.bfloat16, L.IMPL_MFMA),
                                     (torch.bfloat16, L.IMPL_MFMA_BK32), (torch.bfloat16, L.IMPL_AUTO)])
def test_gemm_fused_bias_gradient_and_gelu_backward(dt, impl):
    """wgrad with the bias gradient riding on the same pass, and dgrad with gelu'(h) in the epilogue."""
    M, N, K = 788, 768, 256            # dY [M,N], X [M,K]
    Mp = 832
    g = torch.Generator().manual_seed(3)
    dY = torch.zeros(Mp, N); dY[:M] = torch.randn(M, N, generator=g)
    X = torch.zeros(Mp, K); X[:M] = torch.randn(M, K, generator=g)
    dYd, Xd = dev(dY, dt), dev(X, dt)
    dW, db = dev(torch.randn(N, K, generator=g)), dev(torch.randn(N, generator=g))
    refW = dW.double() + dYd.double().t() @ Xd.double()
    refb = db.double() + dYd.double().sum(0)
    L.gemm(dYd, Xd, dW, bias=db, M=N, N=K, K=Mp, lda=N, ldb=K, ldc=K, a_layout=L.KROW, b_layout=L.KROW, accum=True,
           dtype=L.dtype_code(dt), impl=impl)
    assert rel_err(dW, refW) < 1e-4 and rel_err(db, refb) < 1e-4
    W = dev(torch.randn(N, K, generator=g) * 0.1, dt)
    H = dev(torch.randn(M, K, generator=g), dt)
    dX = torch.empty(M, K, device="cuda", dtype=dt)
    L.gemm(dYd, W, dX, preact=H, M=M, N=K, K=N, lda=N, ldb=K, ldc=K, a_layout=L.ROWK, b_layout=L.KROW, act=L.ACT_GELU_BWD,
           dtype=L.dtype_code(dt), impl=impl)
    hr = H.float().clone().requires_grad_(True)
    F.gelu(hr).backward(torch.ones_like(hr))
    ref = (dYd[:M].double() @ W.double()) * hr.grad.double()
    assert rel_err(dX, ref) < TOL[dt]
    # forward emits gelu'(v) (ACT_GELU_SAVE_GRAD), backward multiplies by it (ACT_MUL)
    Xf, Wf = dev(torch.randn(M, N, generator=g) * 0.2, dt), dev(torch.randn(K, N, generator=g) * 0.2, dt)
    Y, Gp = torch.empty(M, K, device="cuda", dtype=dt), torch.empty(M, K, device="cuda", dtype=dt)
    L.gemm(Xf, Wf, Y, preact=Gp, M=M, N=K, K=N, lda=N, ldb=N, ldc=K, act=L.ACT_GELU_SAVE_GRAD, dtype=L.dtype_code(dt), impl=impl)
    pre = (Xf.double() @ Wf.double().t()).float().requires_grad_(True)
    yr = F.gelu(pre); yr.backward(torch.ones_like(yr))
    assert rel_err(Y, yr.detach()) < TOL[dt] and rel_err(Gp, pre.grad) < TOL[dt]
    dX2 = torch.empty(M, K, device="cuda", dtype=dt)
    L.gemm(dYd, W, dX2, preact=Gp, M=M, N=K, K=N, lda=N, ldb=K, ldc=K, a_layout=L.ROWK, b_layout=L.KROW, act=L.ACT_MUL,
           dtype=L.dtype_code(dt), impl=impl)
    assert rel_err(dX2, (dYd[:M].double() @ W.double()) * Gp.double()) < TOL[dt]


@pytest.mark.parametrize("dt,impl", [(torch.float32, L.IMPL_REF), (torch.bfloat16, L.IMPL_AUTO), (torch.bfloat16, L.IMPL_MFMA),
                                     (torch.bfloat16, L.IMPL_MFMA_1STAGE)])
def test_gemm_colsum_rides_on_the_input_gradient(dt, impl):
    """mmrca_gemm_colsum: dH = (dY W) * gelu'(h) and db += column sums of the stored dH (ragged M, several row tiles)."""
    M, N, K = 788, 384, 256            # dY [M,K], W [K,N] (KROW), C [M,N]
    g = torch.Generator().manual_seed(5)
    dY, W = dev(torch.randn(M, K, generator=g), dt), dev(torch.randn(K, N, generator=g) * 0.1, dt)
    Gp, add = dev(torch.rand(M, N, generator=g), dt), dev(torch.randn(M, N, generator=g), dt)
    for act, pre, addend in ((L.ACT_MUL, Gp, None), (L.ACT_NONE, None, add), (L.ACT_MUL, Gp, add)):
        C = torch.empty(M, N, device="cuda", dtype=dt)
        db = dev(torch.randn(N, generator=g))
        db0 = db.clone()
        L.gemm(dY, W, C, preact=pre, addend=addend, M=M, N=N, K=K, lda=K, ldb=N, ldc=N, a_layout=L.ROWK, b_layout=L.KROW,
               act=act, dtype=L.dtype_code(dt), impl=impl, colsum=db)
        ref = dY.double() @ W.double()
        if act == L.ACT_MUL:
            ref = ref * Gp.double()
        if addend is not None:
            ref = ref + add.double()
        assert rel_err(C, ref) < TOL[dt]
        assert rel_err(db - db0, C.float().sum(0)) < 1e-4          # sums of what was stored


def test_gemm_auto_dispatch_random_shape_sweep():
    """AUTO dispatch (single-stage forward / input-gradient kernel, 128x128x32 and two-stage weight-gradient kernels) over a
    seeded sweep of ragged shapes: every layout, K from one step to many, M not a multiple of anything."""
    rng = np.random.RandomState(2024)
    for case in range(24):
        al, bl = int(rng.randint(0, 2)), int(rng.randint(0, 2))
        N = 128 * int(rng.randint(1, 7))
        K = 64 * int(rng.randint(1, 14))
        M = 128 * int(rng.randint(1, 9)) if al == L.KROW else int(rng.randint(1, 1100))
        accum = bool(al == L.KROW and bl == L.KROW and rng.randint(0, 2))
        if accum:
            _gemm_case(M, N, K, al, bl, torch.bfloat16, L.IMPL_AUTO, accum=True)
        else:
            _gemm_case(M, N, K, al, bl, torch.bfloat16, L.IMPL_AUTO, bias=bool(rng.randint(0, 2)), addend=bool(rng.randint(0, 2)),
                       act=(L.ACT_GELU if rng.randint(0, 2) else L.ACT_NONE), preact=bool(rng.randint(0, 2)))


def _splitk_case(M, N, K, al=1, bl=1, exact=False):
    """mmrca_gemm_splitk (256x256 tiles, slab partials, reduce launch) vs fp64 on the rounded operands"""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    if exact:
        A = ((torch.arange(M)[:, None] * 7 + torch.arange(K)[None, :] * 3) % 5 - 2.0)
        B = ((torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 5) % 7 - 3.0)
    else:
        A, B = torch.randn(M, K, generator=g) * 0.5, torch.randn(N, K, generator=g) * 0.5
    Ad = dev(A if al == 0 else A.t(), torch.bfloat16)
    Bd = dev(B if bl == 0 else B.t(), torch.bfloat16)
    C0 = torch.randint(-8, 9, (M, N), generator=g).float() if exact else torch.randn(M, N, generator=g)
    Cd = dev(C0)
    ws = torch.empty(L.SPLITK_WS_BYTES, dtype=torch.uint8, device="cuda")
    L.gemm_splitk(Ad, Bd, Cd, ws, M=M, N=N, K=K, lda=Ad.shape[1], ldb=Bd.shape[1], ldc=N, a_layout=al, b_layout=bl)
    torch.cuda.synchronize()
    Af = (Ad if al == 0 else Ad.t()).double()
    Bf = (Bd if bl == 0 else Bd.t()).double()
    ref = (Af @ Bf.t()).cpu() + C0.double()
    if exact:
        assert torch.equal(Cd.cpu().double(), ref), (M, N, K, al, bl)
    else:
        assert rel_err(Cd.cpu(), ref) < 2e-5, (M, N, K, al, bl, rel_err(Cd.cpu(), ref))
    return Cd


@pytest.mark.parametrize("layouts", [(1, 1), (0, 0), (0, 1), (1, 0)])
def test_gemm_splitk_256_tiles(layouts):
    al, bl = layouts
    _splitk_case(256, 256, 256, al, bl, exact=True)          # one tile, one split of four K steps
    _splitk_case(512, 256, 64 * 37, al, bl, exact=True)       # ragged last split
    _splitk_case(768, 512, 64 * 101, al, bl)


@pytest.mark.parametrize("shape", [(176, 1056, 64 * 45), (1056, 176, 64 * 45), (48, 192, 64 * 40), (304, 1824, 64 * 9), (8, 8, 128),
                                   (264, 256, 192), (256, 520, 64 * 7), (3072, 512, 64 * 3)])
def test_gemm_splitk_on_output_shapes_that_are_not_multiples_of_256(shape):
    """the conv layers' weight gradients (both operands K-major): edge tiles fetch column 0 in place of the columns past the operands'
    edge -- the operands here are EXACT-size allocations, and integer-valued, so any stray read or a stored out-of-range element shows
    as an inequality (C carries a guard band of rows that must stay untouched)"""
    M, N, K = shape
    assert L.gemm_splitk_ragged_ok(M, N, K, L.BF16)
    _splitk_case(M, N, K, 1, 1, exact=True)
    _splitk_case(M, N, K, 1, 1)
    # a wider row pitch for B (a column window of a wider activation: ShuffleNet's in-place split) and for C, guard rows behind C
    g = torch.Generator().manual_seed(5)
    A = torch.randint(-2, 3, (K, M), generator=g).float()
    Bw = torch.randint(-3, 4, (K, N + 24), generator=g).float()
    Ad, Bd = dev(A, torch.bfloat16), dev(Bw, torch.bfloat16)
    Cd = torch.full((M + 3, N + 8), 7.0, device="cuda")
    ws = torch.empty(L.SPLITK_WS_BYTES, dtype=torch.uint8, device="cuda")
    L.gemm_splitk(Ad, Bd[:, 8:], Cd, ws, M=M, N=N, K=K, lda=M, ldb=N + 24, ldc=N + 8)
    torch.cuda.synchronize()
    ref = torch.full((M + 3, N + 8), 7.0, dtype=torch.float64)
    ref[:M, :N] += A.double().t() @ Bw[:, 8:8 + N].double()
    assert torch.equal(Cd.cpu().double(), ref)


def test_gemm_splitk_rejects_ragged_shapes_in_row_major_layouts():
    A, B, C = dev(torch.zeros(176, 256), torch.bfloat16), dev(torch.zeros(256, 256), torch.bfloat16), dev(torch.zeros(176, 256))
    ws = torch.empty(L.SPLITK_WS_BYTES, dtype=torch.uint8, device="cuda")
    with pytest.raises(L.MmrcaError):
        L.gemm_splitk(A, B, C, ws, M=176, N=256, K=256, lda=256, ldb=256, ldc=256, a_layout=L.ROWK, b_layout=L.ROWK)


def test_gemm_splitk_is_bitwise_reproducible_and_matches_the_atomic_kernel():
    a = _splitk_case(768, 768, 64 * 131)
    b = _splitk_case(768, 768, 64 * 131)
    assert torch.equal(a, b)


# ---- stream-K tail of the persistent 256x256 kernel (round 6): the leftover tiles' K loops run as ranges on every CU
@pytest.fixture
def streamk():
    """register a stream-K workspace for the current stream (the library's opt-in, MMRCA_SK=1 in production) and remove it afterwards"""
    L.streamk_workspace(65536, 256, torch.device("cuda", torch.cuda.current_device()), force=True)
    _streamk_toggle(True)
    L.load().mmrca_gemm_streamk_config(4, 4, 1)       # (production: no tail below 24 K steps, fused bf16x3 products only)
    yield
    L.load().mmrca_gemm_streamk_config(4, 24, 0)
    for (d, st), ws in list(L._STREAMK_WS.items()):
        L._check(L.load().mmrca_gemm_streamk_workspace(None, ws.numel(), st), "streamk off")
    L._STREAMK_WS.clear()


def _streamk_ws_counters():
    key = (torch.cuda.current_device(), L.stream_ptr())
    ws = L._STREAMK_WS.get(key)
    return None if ws is None else ws[:4096].view(torch.int32)


def _streamk_toggle(on):
    """(un)register the current stream's workspace with the library: off = the kernel as it was (whole leftover tiles)"""
    key = (torch.cuda.current_device(), L.stream_ptr())
    ws = L._STREAMK_WS[key]
    L._check(L.load().mmrca_gemm_streamk_workspace(L.ptr(ws) if on else None, ws.numel(), L.stream_ptr()), "streamk toggle")


@pytest.mark.parametrize("shape", [(22272, 768, 256, 2), (25600, 768, 768, 4), (50432, 768, 768, 3), (32000, 2304, 512, 2), (25500, 768, 1024, 4)])
@pytest.mark.parametrize("bl", [L.ROWK, L.KROW])
def test_gemm_streamk_tail_matches_fp64_and_the_whole_tile_walk_and_is_bitwise_reproducible(shape, bl, streamk, monkeypatch):
    """M x N = whole rounds of 256 tiles + L leftover tiles whose K loops are cut into `split` ranges (22272 x 768: 261 tiles, L = 5;
    25600 x 768: 300, L = 44; 50432 x 768 = the ViT's out-projection at B = 256: 591, L = 79; 32000 x 2304: 1125, L = 101 -- two ranges per tile; 25500: a ragged last row tile inside a split tile)"""
    M, N, K, split = shape
    g = torch.Generator().manual_seed(M + N + K)
    Mp = (M + 255) // 256 * 256
    X = torch.zeros(Mp, K, device="cuda", dtype=torch.bfloat16)
    X[:M] = dev(torch.randn(M, K, generator=g) * 0.5, torch.bfloat16)
    W = dev(torch.randn(N, K, generator=g) * 0.1, torch.bfloat16)
    Wd = W if bl == L.ROWK else W.t().contiguous()
    bias = dev(torch.randn(N, generator=g) * 0.2, torch.bfloat16)
    ref = X[:M].double() @ W.double().t() + bias.double()

    def run():
        Y = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        L.gemm(X, Wd, Y, bias=bias, M=M, N=N, K=K, lda=K, ldb=Wd.shape[1], ldc=N, b_layout=bl, dtype=L.BF16, impl=L.IMPL_MFMA256,
               rows_readable=(Mp, 0))
        return Y
    a = run()
    torch.cuda.synchronize()
    ctr = _streamk_ws_counters()
    assert ctr is not None and int(ctr.abs().sum()) == 0, "every tile counter is back at zero after the launch"
    tiles = (Mp // 256) * (N // 256)
    left = tiles % 256
    assert split == min(256 // left, K // 64 // 2, 4)           # what the host picks (mmrca_gemm256_streamk_split)
    assert rel_err(a, ref) < 1e-2
    for _ in range(3):
        assert torch.equal(run(), a), "partials are added in a fixed order: bitwise reproducible"
    _streamk_toggle(False)
    try:
        b = run()
    finally:
        _streamk_toggle(True)
    assert rel_err(b, ref) < 1e-2
    # the two walks add the same fp32 products in a different order: equal to within bf16 rounding of the output
    assert rel_err(a, b) < 8e-3 and (a != b).float().mean().item() < 0.05


def test_gemm_streamk_tail_under_every_epilogue_of_the_persistent_kernel(streamk):
    """GELU + saved gelu', the gelu' factor with the bias-gradient column sums, a residual addend: the unit that finishes a split
    tile runs the same epilogue code on the summed partials (M = 25,600 = 100 row tiles x 3: 300 tiles, 44 of them split four ways)"""
    M, N, K = 25600, 768, 1024
    g = torch.Generator().manual_seed(11)
    X = dev(torch.randn(M, K, generator=g) * 0.5, torch.bfloat16)
    W = dev(torch.randn(N, K, generator=g) * 0.05, torch.bfloat16)
    bias = dev(torch.randn(N, generator=g) * 0.2, torch.bfloat16)
    h = X.double() @ W.double().t() + bias.double()
    Y, P = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    L.gemm(X, W, Y, bias=bias, preact=P, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, act=L.ACT_GELU_SAVE_GRAD, dtype=L.BF16, impl=L.IMPL_MFMA256)
    assert rel_err(Y, F.gelu(h)) < 1e-2
    hh = h.detach().clone().requires_grad_(True)
    F.gelu(hh).sum().backward()
    assert rel_err(P, hh.grad) < 1e-2
    # input gradient x gelu' with the column sums: dH = (dY W2) * gelu'(h), db1 = column sums of dH
    W2 = dev(torch.randn(K, N, generator=g) * 0.05, torch.bfloat16)          # [K2 = 1024 rows of the contraction][N]  (KROW)
    dY = dev(torch.randn(M, K, generator=g) * 0.5, torch.bfloat16)
    dH = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    db = torch.zeros(N, device="cuda")
    L.gemm(dY, W2, dH, preact=P, M=M, N=N, K=K, lda=K, ldb=N, ldc=N, b_layout=L.KROW, act=L.ACT_MUL, dtype=L.BF16, impl=L.IMPL_MFMA256, colsum=db)
    ref = (dY.double() @ W2.double()) * P.double()
    assert rel_err(dH, ref) < 1e-2
    assert rel_err(db, dH.double().sum(0)) < 1e-3
    R = dev(torch.randn(M, N, generator=g), torch.bfloat16)
    Z = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    L.gemm(X, W, Z, bias=bias, addend=R, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, dtype=L.BF16, impl=L.IMPL_MFMA256)
    assert rel_err(Z, h + R.double()) < 1e-2
    assert int(_streamk_ws_counters().abs().sum()) == 0


@pytest.mark.parametrize("K,act", [(768, L.ACT_NONE), (3072, L.ACT_NONE), (768, L.ACT_GELU_SAVE_GRAD)])
def test_gemm_streamk_tail_under_the_fused_bf16x3_form(K, act, streamk, monkeypatch):
    """the three-pass product in its fused four-plane form (K steps of 32 of all four planes) with the leftover tiles' K loops cut into
    ranges: fp32-accurate against float64 (1e-5 of the largest entry), bitwise reproducible, and within fp32 rounding of the whole-tile
    walk; AUTO (mmrca_gemm_x3) keeps the partial round inside the launch when the stream has a workspace"""
    monkeypatch.setattr(L, "STREAMK", True)
    M, N = 50432, 768
    g = torch.Generator(device="cuda").manual_seed(K)
    X = torch.randn(M, K, device="cuda", generator=g) * 0.5
    W = torch.randn(N, K, device="cuda", generator=g) * 0.05
    bias = torch.randn(N, device="cuda", generator=g) * 0.2
    pl = lambda t: (t.bfloat16(), (t - t.bfloat16().float()).bfloat16())
    Xp, Wp = pl(X), pl(W)
    Xv, Wv = Xp[0].double() + Xp[1].double(), Wp[0].double() + Wp[1].double()
    h = Xv @ Wv.t() + bias.double()
    ref = F.gelu(h) if act == L.ACT_GELU_SAVE_GRAD else h

    def run():
        Y = torch.empty(M, N, device="cuda")
        P = torch.empty(M, N, device="cuda") if act == L.ACT_GELU_SAVE_GRAD else None
        L.gemm_x3(Xp, Wp, Y, bias=bias, preact=P, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, act=act)
        return Y
    a = run()
    torch.cuda.synchronize()
    assert int(_streamk_ws_counters().abs().sum()) == 0
    print(f"fused bf16x3 product with a stream-K tail, K={K} act={act}: {rel_err(a, ref):.2e} of the largest entry from float64")
    assert rel_err(a, ref) < 2e-5        # (observed 3.0e-6 at K = 3072; a bf16 product sits at ~3e-3)
    assert torch.equal(run(), a)
    _streamk_toggle(False)
    try:
        b = run()
    finally:
        _streamk_toggle(True)
    # the two walks add the same fp32 terms in a different order: observed 2.2e-6 of the largest entry at K = 3072 (each is within
    # 3e-6 of float64); bound = 4x that
    assert rel_err(b, ref) < 2e-5 and rel_err(a, b) < 1e-5
    assert not torch.equal(a, b), "the two walks add in a different order: identical outputs would mean the tail did not run"


def test_gemm_streamk_tail_on_a_side_stream_and_through_auto(streamk, monkeypatch):
    monkeypatch.setattr(L, "STREAMK", True)
    monkeypatch.setattr(L, "STREAMK_BF16", True)
    """one workspace per stream (two streams run GEMMs concurrently in the engine); AUTO keeps the partial round inside the launch
    instead of handing it to the 128x128 kernel"""
    M, N, K = 50432, 768, 768
    g = torch.Generator().manual_seed(3)
    X = dev(torch.randn(M, K, generator=g) * 0.5, torch.bfloat16)
    W = dev(torch.randn(N, K, generator=g) * 0.05, torch.bfloat16)
    ref = X.double() @ W.double().t()
    outs = []
    side = torch.cuda.Stream()
    for st in (torch.cuda.current_stream(), side):
        with torch.cuda.stream(st):
            Y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            L.gemm(X, W, Y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, dtype=L.BF16)
            outs.append(Y)
    torch.cuda.synchronize()
    assert len(L._STREAMK_WS) >= 2
    assert rel_err(outs[0], ref) < 1e-2 and torch.equal(outs[0], outs[1])


def test_gemm_at_the_benchmarked_sizes():
    """BASELINE configs[1]: 50,432 token rows (B=256 x 197).  Forward / input-gradient GEMM at M = 50,432 (XCD remap at
    2,364 tiles, grouped rastering) and the weight gradient with a 50,432-row contraction (split-K over every resident
    block slot), each against fp64 on the same bf16-rounded operands."""
    M = 50432
    g = torch.Generator().manual_seed(5)
    X = dev(torch.randn(M, 768, generator=g) * 0.5, torch.bfloat16)
    W = dev(torch.randn(3072, 768, generator=g) * 0.05, torch.bfloat16)
    bias = dev(torch.randn(3072, generator=g) * 0.1, torch.bfloat16)
    Y = torch.empty(M, 3072, device="cuda", dtype=torch.bfloat16)
    L.gemm(X, W, Y, bias=bias, M=M, N=3072, K=768, lda=768, ldb=768, ldc=3072, dtype=L.BF16)           # forward FFN1
    ref = X.double() @ W.double().t() + bias.double()
    assert rel_err(Y, ref) < 1e-2
    dX = torch.empty(M, 768, device="cuda", dtype=torch.bfloat16)
    L.gemm(Y, W, dX, M=M, N=768, K=3072, lda=3072, ldb=768, ldc=768, a_layout=L.ROWK, b_layout=L.KROW, dtype=L.BF16)   # dgrad FFN1
    ref = Y.double() @ W.double()
    assert rel_err(dX, ref) < 1e-2
    del ref
    for impl in ("atomic", "splitk"):
        dW = torch.zeros(3072, 768, device="cuda")
        if impl == "atomic":
            L.gemm(Y, X, dW, M=3072, N=768, K=M, lda=3072, ldb=768, ldc=768, a_layout=L.KROW, b_layout=L.KROW, accum=True, dtype=L.BF16)
        else:
            ws = torch.empty(L.SPLITK_WS_BYTES, dtype=torch.uint8, device="cuda")
            L.gemm_splitk(Y, X, dW, ws, M=3072, N=768, K=M, lda=3072, ldb=768, ldc=768)
        ref = Y.double().t() @ X.double()
        assert rel_err(dW, ref) < 1e-5, (impl, rel_err(dW, ref))


def test_gemm_rejects_bad_arguments():
    a = torch.zeros(4, 4, device="cuda")
    with pytest.raises(L.MmrcaError):
        L.gemm(a, a, a, M=4, N=4, K=4, lda=2, ldb=4, ldc=4, dtype=L.F32)
    with pytest.raises(L.MmrcaError):
        L.gemm(a, a, a, M=4, N=4, K=4, lda=4, ldb=4, ldc=4, dtype=L.BF16, impl=L.IMPL_MFMA)
    with pytest.raises(L.MmrcaError):
        L.gemm(torch.zeros(4, 4), a, a, M=4, N=4, K=4, lda=4, ldb=4, ldc=4, dtype=L.F32)     # host tensor


# ------------------------------------------------------------------------------------------------------
# LayerNorm, GELU', column sums
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D", [768, 1024, 96, 1408])
def test_layernorm_fwd_bwd(dt, D):
    rows, eps = 37, 1e-6
    x, r = dev(torch.randn(rows, D), dt), dev(torch.randn(rows, D), dt)
    g, b = dev(1 + 0.1 * torch.randn(D), dt), dev(0.1 * torch.randn(D), dt)
    s_out, y = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    L.add_layernorm_fwd(x, r, g, b, s_out, y, mean, rstd, rows, D, D, D, eps, L.dtype_code(dt))
    s_ref = (x.float() + r.float())
    if dt == torch.bfloat16:
        s_ref = s_ref.bfloat16().float()
    sr = s_ref.clone().requires_grad_(True)
    gr, br = g.float().clone().requires_grad_(True), b.float().clone().requires_grad_(True)
    y_ref = F.layer_norm(sr, (D,), gr, br, eps)
    assert rel_err(s_out, s_ref) < TOL[dt] and rel_err(y, y_ref) < TOL[dt]
    dy, dres = dev(torch.randn(rows, D), dt), dev(torch.randn(rows, D), dt)
    ds = torch.empty_like(x)
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    L.layernorm_bwd(dy, s_out, g, mean, rstd, dres, ds, dg, db, rows, D, D, D, D, L.dtype_code(dt))
    y_ref.backward(dy.float())
    assert rel_err(ds, sr.grad + dres.float()) < TOL[dt]
    assert rel_err(dg, gr.grad) < TOL[dt] and rel_err(db, br.grad) < TOL[dt]


@pytest.mark.parametrize("D", [768, 384, 1408])
@pytest.mark.parametrize("form", ["res+sum", "plain", "res only", "res+sum+dropout", "plain+out dropout"])
def test_layernorm_fwd_bf16_dispatch_branches(D, form):
    """the bf16 forward's four instantiations per width (csrc/rowops.hip::add_ln_fwd_bf16_k<NV2, RP, PACK, DROP>): the packed-row
    form (a stored sum, or a plain x: the normalised values are bf16 numbers) and the fp32-row form (residual without a stored sum: the
    fp32 sum is normalised), with and without the counter-based dropout sites; enough rows for several grid-stride rounds and a
    ragged last one.  Dropout masks are rebuilt on the host from the same hash."""
    from garbage_classification_rca_amd.procedural import counter_uniform
    rows, eps = 8 * 1024 + 13, 1e-12
    gen = torch.Generator().manual_seed(D + len(form))
    x, r = torch.randn(rows, D, generator=gen).bfloat16(), torch.randn(rows, D, generator=gen).bfloat16()
    g, b = (1 + 0.1 * torch.randn(D, generator=gen)).bfloat16(), (0.1 * torch.randn(D, generator=gen)).bfloat16()
    use_res, use_sum = form.startswith("res"), "sum" in form
    p_in, p_out = (0.1 if form == "res+sum+dropout" else 0.0), (0.1 if form == "plain+out dropout" else 0.0)
    xs = x.float()
    idx = np.arange(rows * D, dtype=np.uint64)
    if p_in:
        keep = torch.from_numpy(counter_uniform(11, idx) >= np.float32(p_in)).view(rows, D)
        xs = torch.where(keep, xs * (1.0 / (1.0 - p_in)), torch.zeros(()))
    s_ref = xs + r.float() if use_res else xs
    if use_sum:
        s_ref = s_ref.bfloat16().float()
    y_ref = F.layer_norm(s_ref, (D,), g.float(), b.float(), eps)
    if p_out:
        keep = torch.from_numpy(counter_uniform(22, idx) >= np.float32(p_out)).view(rows, D)
        y_ref = torch.where(keep, y_ref * (1.0 / (1.0 - p_out)), torch.zeros(()))
    xd, rd, gd, bd = x.cuda(), r.cuda(), g.cuda(), b.cuda()
    s_out = torch.full_like(xd, float("nan")) if use_sum else None
    y = torch.full_like(xd, float("nan"))
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    L.add_layernorm_fwd(xd, rd if use_res else None, gd, bd, s_out, y, mean, rstd, rows, D, D, D, eps, L.BF16,
                        in_drop=(p_in, 11) if p_in else (0.0, 0), out_drop=(p_out, 22) if p_out else (0.0, 0))
    torch.cuda.synchronize()
    if use_sum:
        assert torch.equal(s_out.float().cpu(), s_ref)                       # the stored sum: bit-exact
    assert rel_err(mean.cpu(), s_ref.mean(1)) < 1e-5 and rel_err(rstd.cpu(), (s_ref.var(1, unbiased=False) + eps).rsqrt()) < 1e-5
    d = (y.float().cpu() - y_ref).abs()
    assert float(d.max()) < 0.04 and float(d.mean()) < 2e-3, (float(d.max()), float(d.mean()))     # bf16 rounding of |y| <~ 5


def test_layernorm_strided_rows_class_token_only():
    B, Tn, D = 3, 5, 768
    x = dev(torch.randn(B * Tn, D))
    g, b = dev(torch.ones(D)), dev(torch.zeros(D))
    y = torch.empty(B, D, device="cuda")
    mean, rstd = torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    L.add_layernorm_fwd(x, None, g, b, None, y, mean, rstd, B, D, Tn * D, D, 1e-6, L.F32)
    assert rel_err(y, F.layer_norm(x.view(B, Tn, D)[:, 0], (D,), g, b, 1e-6)) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gelu_bwd_and_colsum(dt):
    n = 64 * 772
    h, dg = dev(torch.randn(n) * 2, dt), dev(torch.randn(n), dt)
    dh = torch.empty_like(h)
    L.gelu_bwd(dg, h, dh, n, L.dtype_code(dt))
    hr = h.float().clone().requires_grad_(True)
    F.gelu(hr).backward(dg.float())
    assert rel_err(dh, hr.grad) < TOL[dt]
    Mg, Ng = 1003, 3072
    hg, dgg = dev(torch.randn(Mg, Ng) * 2, dt), dev(torch.randn(Mg, Ng), dt)
    dhg, dbg = torch.empty_like(hg), dev(torch.randn(Ng))
    hr2 = hg.float().clone().requires_grad_(True)
    F.gelu(hr2).backward(dgg.float())
    refb = dbg + hr2.grad.sum(0)
    L.gelu_bwd_colsum(dgg, hg, dhg, dbg, Mg, Ng, Ng, L.dtype_code(dt))
    assert rel_err(dhg, hr2.grad) < TOL[dt] and rel_err(dbg, refb) < (1e-4 if dt == torch.float32 else 3e-3)
    M, N = 1003, 772
    dY = dev(torch.randn(M, N), dt)
    db = dev(torch.randn(N))
    ref = db + dY.float().sum(0)
    L.colsum_accum(dY, db, M, N, N, L.dtype_code(dt))
    assert rel_err(db, ref) < 1e-4


# ------------------------------------------------------------------------------------------------------
# embeddings, patchify, assemble
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,structured", [(50, False), (4100, False), (64 * 70, True)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_embed_fwd_bwd(dt, rows, structured):
    # rows > 1024: a wave of embed_bwd walks several rows, keeping position / token-type sums in registers
    V, Pn, D = 1000, 66, 768
    word, pos, typ = dev(torch.randn(V, D), dt), dev(torch.randn(Pn, D), dt), dev(torch.randn(D), dt)
    ids = torch.randint(0, V, (rows,), dtype=torch.int32, device="cuda"); ids[:5] = 7
    pid = torch.randint(0, Pn, (rows,), dtype=torch.int32, device="cuda")
    if structured:
        pid = (torch.arange(rows, device="cuda") % 64).int()
    out = torch.empty(rows, D, device="cuda", dtype=dt)
    L.embed_fwd(ids, pid, word, pos, typ, out, rows, D, L.dtype_code(dt))
    ref = word.float()[ids.long()] + pos.float()[pid.long()] + typ.float()
    assert rel_err(out, ref) < TOL[dt]
    dout = dev(torch.randn(rows, D), dt)
    dw, dp, dty = torch.zeros(V, D, device="cuda"), torch.zeros(Pn, D, device="cuda"), torch.zeros(D, device="cuda")
    dout[3::7] = 0                      # all-zero gradient rows (masked positions) are skipped inside the kernel
    L.embed_bwd(dout, ids, pid, dw, dp, dty, rows, D, L.dtype_code(dt))
    rw = torch.zeros(V, D, device="cuda").index_add_(0, ids.long(), dout.float())
    rp = torch.zeros(Pn, D, device="cuda").index_add_(0, pid.long(), dout.float())
    assert rel_err(dw, rw) < 1e-5 and rel_err(dp, rp) < 1e-5 and rel_err(dty, dout.float().sum(0)) < 1e-5
    # padding_idx: the pad row of the word table (id 7 here) / of the position table (position 1) takes no gradient
    dw.zero_(); dp.zero_(); dty.zero_()
    L.embed_bwd(dout, ids, pid, dw, dp, dty, rows, D, L.dtype_code(dt), pad_id=7, pos_pad_id=1)
    rw[7] = 0; rp[1] = 0
    assert float(dw[7].abs().max()) == 0.0 and float(dp[1].abs().max()) == 0.0
    assert rel_err(dw, rw) < 1e-5 and rel_err(dp, rp) < 1e-5 and rel_err(dty, dout.float().sum(0)) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_patchify_matches_conv_and_assemble(dt):
    B, P, Himg, D = 2, 16, 64, 768
    nP = (Himg // P) ** 2
    img = dev(torch.randn(B, 3, Himg, Himg))
    patches = torch.empty(B * nP, 3 * P * P, device="cuda", dtype=dt)
    L.patchify_fwd(img, patches, B, 3, Himg, Himg, P, L.dtype_code(dt))
    w = dev(torch.randn(D, 3, P, P) * 0.05)
    ref = F.conv2d(img, w, stride=P).reshape(B, D, nP).permute(0, 2, 1).reshape(B * nP, D)
    got = patches.float() @ w.reshape(D, -1).t()
    assert rel_err(got, ref) < (1e-4 if dt == torch.float32 else 2e-2)
    proj, cls, pos = dev(torch.randn(B * nP, D), dt), dev(torch.randn(D), dt), dev(torch.randn(nP + 1, D), dt)
    x = torch.empty(B * (nP + 1), D, device="cuda", dtype=dt)
    L.vit_assemble_fwd(proj, cls, pos, x, B, nP, D, L.dtype_code(dt))
    xr = torch.cat([cls.float().expand(B, 1, D), proj.float().view(B, nP, D)], 1) + pos.float()
    assert rel_err(x, xr.reshape(-1, D)) < TOL[dt]
    dx = dev(torch.randn(B * (nP + 1), D), dt)
    dproj = torch.empty_like(proj)
    dcls, dpos = torch.zeros(D, device="cuda"), torch.zeros(nP + 1, D, device="cuda")
    L.vit_assemble_bwd(dx, dproj, dcls, dpos, B, nP, D, L.dtype_code(dt))
    d3 = dx.float().view(B, nP + 1, D)
    assert rel_err(dproj, d3[:, 1:].reshape(-1, D)) < 1e-6
    assert rel_err(dpos, d3.sum(0)) < 1e-5 and rel_err(dcls, d3[:, 0].sum(0)) < 1e-5


# ------------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------------
def _attn_ref(qkv, mask, B, H, S, dh, dout=None):
    qkv = qkv.float().clone().requires_grad_(True)
    q, k, v = (t.view(B, S, H, dh).transpose(1, 2) for t in qkv.view(B, S, 3 * H * dh).split(H * dh, dim=-1))
    sc = q @ k.transpose(2, 3) / math.sqrt(dh)
    if mask is not None:
        km = mask.bool()[:, None, None, :]
        valid = km.any(-1, keepdim=True)
        sc = sc.masked_fill(~km, float("-inf"))
        a = torch.softmax(torch.where(valid, sc, torch.zeros_like(sc)), -1) * valid
    else:
        a = torch.softmax(sc, -1)
    out = (a @ v).transpose(1, 2).reshape(B * S, H * dh)
    if dout is None:
        return out, None
    out.backward(dout.float())
    return out, qkv.grad


@pytest.mark.parametrize("dt,impl", [(torch.float32, L.IMPL_REF), (torch.float32, L.IMPL_AUTO), (torch.bfloat16, L.IMPL_REF), (torch.bfloat16, L.IMPL_AUTO)])
@pytest.mark.parametrize("S,masked", [(64, True), (197, False), (40, True), (300, True), (512, False), (1, False), (17, True)])
def test_mha_fwd_bwd(dt, impl, S, masked):
    # S = 512 is the text encoders' max_position_embeddings (the largest sequence the path can see); S = 1 the smallest
    B, H, dh = 3, 4, 64
    qkv = dev(torch.randn(B * S, 3 * H * dh), dt)
    mask = None
    if masked:
        mask = torch.ones(B, S, dtype=torch.int32)
        mask[0, S // 2:] = 0
        mask[2, :] = 0            # fully masked caption (modality dropout)
        mask = mask.cuda()
    out = torch.empty(B * S, H * dh, device="cuda", dtype=dt)
    lse = torch.empty(B, H, S, device="cuda")
    L.mha_fwd(qkv, mask, out, lse, B, H, S, dh, 1 / math.sqrt(dh), L.dtype_code(dt), impl)
    dout = dev(torch.randn(B * S, H * dh), dt)
    ref, dref = _attn_ref(qkv, mask, B, H, S, dh, dout)
    tol = 1e-4 if dt == torch.float32 else 2e-2
    assert torch.isfinite(out.float()).all()
    assert rel_err(out, ref) < tol
    dqkv = torch.zeros_like(qkv)
    L.mha_bwd(qkv, mask, out, dout, lse, dqkv, B, H, S, dh, 1 / math.sqrt(dh), L.dtype_code(dt), impl)
    assert torch.isfinite(dqkv.float()).all()
    assert rel_err(dqkv, dref) < (2e-4 if dt == torch.float32 else 3e-2)
    # the same backward with the in-projection bias gradient (column sums of the stored dqkv) reduced inside the kernels
    dqkv2, db = torch.zeros_like(qkv), dev(torch.randn(3 * H * dh))
    db0 = db.clone()
    L.mha_bwd(qkv, mask, out, dout, lse, dqkv2, B, H, S, dh, 1 / math.sqrt(dh), L.dtype_code(dt), impl, colsum=db)
    assert torch.equal(dqkv2, dqkv)
    assert rel_err(db - db0, dqkv.float().sum(0)) < 1e-4


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("S,dh,masked,p", [(64, 64, True, 0.0), (197, 64, False, 0.0), (40, 64, True, 0.1), (300, 32, False, 0.1)])
def test_class_token_attention_equals_row0_of_full_attention(dt, S, dh, masked, p):
    """mmrca_mha_cls_* = row 0 of the full kernels (forward), and the full backward fed a gradient that is zero
    outside row 0 -- same masks, same dropout counters."""
    B, H, seed = 3, 4, 99
    qkv = dev(torch.randn(B * S, 3 * H * dh), dt)
    mask = None
    if masked:
        mask = torch.ones(B, S, dtype=torch.int32)
        mask[0, S // 2:] = 0
        mask[2, :] = 0
        mask = mask.cuda()
    sc, code = 1 / math.sqrt(dh), L.dtype_code(dt)
    out = torch.empty(B * S, H * dh, device="cuda", dtype=dt)
    lse = torch.empty(B, H, S, device="cuda")
    L.mha_fwd(qkv, mask, out, lse, B, H, S, dh, sc, code, L.IMPL_REF, drop_p=p, drop_seed=seed)
    out_c = torch.empty(B, H * dh, device="cuda", dtype=dt)
    lse_c = torch.empty(B, H, device="cuda")
    L.mha_cls_fwd(qkv, mask, out_c, lse_c, B, H, S, dh, sc, code, drop_p=p, drop_seed=seed)
    tol = 1e-5 if dt == torch.float32 else 1e-2
    assert rel_err(out_c, out.view(B, S, H * dh)[:, 0]) < tol
    fin = torch.isfinite(lse[:, :, 0])
    assert torch.equal(fin, torch.isfinite(lse_c)) and rel_err(lse_c[fin], lse[:, :, 0][fin]) < 1e-5
    dout_c = dev(torch.randn(B, H * dh), dt)
    dout = torch.zeros(B * S, H * dh, device="cuda", dtype=dt)
    dout.view(B, S, H * dh)[:, 0] = dout_c
    dqkv = torch.zeros_like(qkv)
    L.mha_bwd(qkv, mask, out, dout, lse, dqkv, B, H, S, dh, sc, code, L.IMPL_REF, drop_p=p, drop_seed=seed)
    dqkv_c = torch.full_like(qkv, float("nan"))          # the class-token backward must write every element
    L.mha_cls_bwd(qkv, mask, out_c, dout_c, lse_c, dqkv_c, B, H, S, dh, sc, code, drop_p=p, drop_seed=seed)
    assert torch.isfinite(dqkv_c.float()).all()
    assert rel_err(dqkv_c, dqkv) < (1e-4 if dt == torch.float32 else 2e-2)
    assert float(dqkv_c.view(B, S, 3, H * dh)[:, 1:, 0].abs().max()) == 0.0


@pytest.mark.parametrize("dt,impl", [(torch.float32, L.IMPL_REF), (torch.float32, L.IMPL_AUTO), (torch.bfloat16, L.IMPL_REF), (torch.bfloat16, L.IMPL_AUTO)])
@pytest.mark.parametrize("S,p", [(64, 0.0), (64, 0.1), (200, 0.0)])
def test_attention_packed_layout_reproduces_the_padded_run(dt, impl, S, p):
    """cu_seqlens: captions stored back to back without their padding give, on every kept row, what the padded layout
    with a prefix key mask gives (same dropout counters); also for the class-token kernels and the fused bias column sums."""
    B, H, dh, seed = 5, 3, 64, 21
    lens = torch.tensor([S, 1, S // 2, 7, S - 3])
    mask = (torch.arange(S)[None, :] < lens[:, None]).int().cuda()
    keep = mask.bool().view(-1)
    cu = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).int().cuda()
    Mv = int(lens.sum())
    sc, code = 1 / math.sqrt(dh), L.dtype_code(dt)
    qkv = dev(torch.randn(B * S, 3 * H * dh), dt)
    qkv_p = torch.zeros(Mv + 128, 3 * H * dh, device="cuda", dtype=dt); qkv_p[:Mv] = qkv[keep]
    out, lse = torch.zeros(B * S, H * dh, device="cuda", dtype=dt), torch.zeros(B, H, S, device="cuda")
    L.mha_fwd(qkv, mask, out, lse, B, H, S, dh, sc, code, impl, drop_p=p, drop_seed=seed)
    out_p, lse_p = torch.zeros(Mv + 128, H * dh, device="cuda", dtype=dt), torch.zeros(B, H, S, device="cuda")
    L.mha_fwd(qkv_p, None, out_p, lse_p, B, H, S, dh, sc, code, impl, drop_p=p, drop_seed=seed, cu=cu)
    tol = 1e-5 if dt == torch.float32 else 1e-2
    assert rel_err(out_p[:Mv], out[keep]) < tol
    valid_q = mask.bool()[:, None, :].expand(B, H, S)
    assert rel_err(lse_p[valid_q], lse[valid_q]) < 1e-5
    dout = dev(torch.randn(B * S, H * dh), dt) * keep[:, None]          # no gradient enters at padding rows
    dout_p = torch.zeros(Mv + 128, H * dh, device="cuda", dtype=dt); dout_p[:Mv] = dout[keep]
    dqkv, db = torch.zeros_like(qkv), torch.zeros(3 * H * dh, device="cuda")
    L.mha_bwd(qkv, mask, out, dout, lse, dqkv, B, H, S, dh, sc, code, impl, drop_p=p, drop_seed=seed, colsum=db)
    dqkv_p, db_p = torch.full_like(qkv_p, float("nan")), torch.zeros(3 * H * dh, device="cuda")
    L.mha_bwd(qkv_p, None, out_p, dout_p, lse_p, dqkv_p, B, H, S, dh, sc, code, impl, drop_p=p, drop_seed=seed, colsum=db_p,
              cu=cu, rows=Mv)
    gtol = 1e-4 if dt == torch.float32 else 2e-2
    assert torch.isfinite(dqkv_p[:Mv].float()).all()
    assert rel_err(dqkv_p[:Mv], dqkv[keep]) < gtol
    assert rel_err(db_p, dqkv[keep].float().sum(0)) < (1e-4 if dt == torch.float32 else 1e-2)
    # class-token kernels in the packed layout
    first = cu[:-1].long()
    oc, lc = torch.zeros(B, H * dh, device="cuda", dtype=dt), torch.zeros(B, H, device="cuda")
    L.mha_cls_fwd(qkv_p, None, oc, lc, B, H, S, dh, sc, code, drop_p=p, drop_seed=seed, cu=cu)
    assert rel_err(oc, out_p[first]) < tol and rel_err(lc, lse_p[:, :, 0]) < 1e-5
    dout_c = dev(torch.randn(B, H * dh), dt)
    dfull = torch.zeros_like(dout_p); dfull[first] = dout_c
    ref = torch.zeros_like(qkv_p)
    L.mha_bwd(qkv_p, None, out_p, dfull, lse_p, ref, B, H, S, dh, sc, code, impl, drop_p=p, drop_seed=seed, cu=cu)
    got = torch.full_like(qkv_p, float("nan"))
    L.mha_cls_bwd(qkv_p, None, oc, dout_c, lc, got, B, H, S, dh, sc, code, drop_p=p, drop_seed=seed, cu=cu)
    assert torch.isfinite(got[:Mv].float()).all() and rel_err(got[:Mv], ref[:Mv]) < gtol


# ------------------------------------------------------------------------------------------------------
# train-mode dropout of the text encoder (counter-based masks shared by forward and backward)
# ------------------------------------------------------------------------------------------------------
def _keep(seed, shape, p):
    from garbage_classification_rca_amd.procedural import counter_uniform
    idx = np.arange(int(np.prod(shape)), dtype=np.uint64).reshape(shape)
    return torch.from_numpy((counter_uniform(seed, idx) >= p).astype(np.float32) / (1.0 - p)).cuda()


def test_layernorm_dropout_fwd_bwd_match_masked_reference():
    rows, D, eps, p = 33, 768, 1e-12, 0.1
    x, r = dev(torch.randn(rows, D)), dev(torch.randn(rows, D))
    g, b = dev(1 + 0.1 * torch.randn(D)), dev(0.1 * torch.randn(D))
    s_out, y = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    L.add_layernorm_fwd(x, r, g, b, s_out, y, mean, rstd, rows, D, D, D, eps, L.F32, in_drop=(p, 11), out_drop=(p, 22))
    k_in, k_out = _keep(11, (rows, D), p), _keep(22, (rows, D), p)
    assert abs(float((k_in > 0).float().mean()) - 0.9) < 0.01
    xr = x.clone().requires_grad_(True)
    s_ref = xr * k_in + r
    y_ref = F.layer_norm(s_ref, (D,), g, b, eps) * k_out
    assert rel_err(s_out, s_ref.detach()) < 1e-5 and rel_err(y, y_ref.detach()) < 1e-5
    dy = dev(torch.randn(rows, D))
    y_ref.backward(dy)
    # backward: ds (residual-stream gradient), dbranch (gradient of the dropped branch)
    ds, dbr = torch.empty_like(x), torch.empty_like(x)
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    L.layernorm_bwd(dy, s_out, g, mean, rstd, None, ds, dg, db, rows, D, D, D, D, L.F32, dy_drop=(p, 22), branch_drop=(p, 11), dbranch=dbr)
    assert rel_err(dbr, xr.grad) < 1e-4
    assert rel_err(ds * k_in, xr.grad) < 1e-4


@pytest.mark.parametrize("S,masked", [(64, True), (128, True), (197, False), (33, True)])
def test_fp32_matrix_core_attention_with_dropout_matches_the_reference_kernel(S, masked):
    """attention_f32.hip (what the fp32 / bf16x3 modes run: fp32 MFMA, AUTO) against the row-per-wave fp32 kernel (REF):
    same masks from the same counter hash, so forward, lse and all three gradients agree to fp32 rounding"""
    B, H, dh, p, seed = 3, 5, 64, 0.1, 91
    qkv = dev(torch.randn(B * S, 3 * H * dh))
    mask = None
    if masked:
        mask = torch.ones(B, S, dtype=torch.int32); mask[1, S // 3:] = 0; mask[2, :] = 0; mask = mask.cuda()
    sc = 1 / math.sqrt(dh)
    dout = dev(torch.randn(B * S, H * dh, generator=torch.Generator().manual_seed(1)))
    res = {}
    for impl in (L.IMPL_REF, L.IMPL_AUTO):
        out, lse = torch.empty(B * S, H * dh, device="cuda"), torch.empty(B, H, S, device="cuda")
        L.mha_fwd(qkv, mask, out, lse, B, H, S, dh, sc, L.F32, impl, drop_p=p, drop_seed=seed)
        dqkv = torch.full_like(qkv, float("nan"))
        L.mha_bwd(qkv, mask, out, dout, lse, dqkv, B, H, S, dh, sc, L.F32, impl, drop_p=p, drop_seed=seed)
        res[impl] = (out, lse, dqkv)
    (o0, l0, g0), (o1, l1, g1) = res[L.IMPL_REF], res[L.IMPL_AUTO]
    assert torch.isfinite(o1).all() and torch.isfinite(g1).all()
    assert rel_err(o1, o0) < 1e-5
    fin = torch.isfinite(l0)
    assert torch.equal(fin, torch.isfinite(l1)) and rel_err(l1[fin], l0[fin]) < 1e-5
    assert rel_err(g1, g0) < 5e-5


@pytest.mark.parametrize("S,masked", [(64, True), (197, False)])
def test_attention_dropout_mfma_matches_reference_kernel_and_torch(S, masked):
    B, H, dh, p, seed = 2, 3, 64, 0.1, 77
    qkv = dev(torch.randn(B * S, 3 * H * dh), torch.bfloat16)
    mask = None
    if masked:
        mask = torch.ones(B, S, dtype=torch.int32); mask[1, S // 3:] = 0; mask = mask.cuda()
    sc = 1 / math.sqrt(dh)
    outs = {}
    for impl in (L.IMPL_REF, L.IMPL_AUTO):
        out = torch.empty(B * S, H * dh, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(B, H, S, device="cuda")
        L.mha_fwd(qkv, mask, out, lse, B, H, S, dh, sc, L.BF16, impl, drop_p=p, drop_seed=seed)
        dout = dev(torch.randn(B * S, H * dh, generator=torch.Generator().manual_seed(1)), torch.bfloat16)
        dqkv = torch.zeros_like(qkv)
        L.mha_bwd(qkv, mask, out, dout, lse, dqkv, B, H, S, dh, sc, L.BF16, impl, drop_p=p, drop_seed=seed)
        outs[impl] = (out.float(), dqkv.float())
    assert rel_err(outs[L.IMPL_AUTO][0], outs[L.IMPL_REF][0]) < 2e-2
    assert rel_err(outs[L.IMPL_AUTO][1], outs[L.IMPL_REF][1]) < 3e-2
    # torch statement with the same mask
    keep = _keep(seed, (B * H, S, S), p).view(B, H, S, S)
    q3 = qkv.float().clone().requires_grad_(True)
    q, k, v = (t.view(B, S, H, dh).transpose(1, 2) for t in q3.view(B, S, 3 * H * dh).split(H * dh, dim=-1))
    s_ = q @ k.transpose(2, 3) * sc
    if mask is not None:
        s_ = s_.masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
    a = torch.softmax(s_, -1) * keep
    o = (a @ v).transpose(1, 2).reshape(B * S, H * dh)
    o.backward(dout.float())
    assert rel_err(outs[L.IMPL_REF][0], o.detach()) < 2e-2
    assert rel_err(outs[L.IMPL_REF][1], q3.grad) < 3e-2


# ------------------------------------------------------------------------------------------------------
# fused head vs the oracle, loss, optimizers
# ------------------------------------------------------------------------------------------------------
def _head_setup(d_img, d_txt, mode, reverse, n_classes=4):
    from oracle import model as O
    from garbage_classification_rca_amd import spec as S
    from garbage_classification_rca_amd.procedural import proc_tensor
    m = O.OracleMMRCA(n_classes, 0.0, 0.0, 0.7, torch.nn.Identity(), torch.nn.Identity(), d_img, d_txt, reverse,
                      mode == 1, mode == 2)
    sd = {k: torch.from_numpy(proc_tensor(k, tuple(v.shape))) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    names = {"sai": "self_attention_image", "sat": "self_attention_text", "c1": "cross_attention_1", "c2": "cross_attention_2"}
    leaf = {"wq": "W_query.weight", "bq": "W_query.bias", "wk": "W_key.weight", "bk": "W_key.bias",
            "wv": "W_value.weight", "bv": "W_value.bias", "g": "norm.weight", "b": "norm.bias"}
    fin = ["final_with_everything", "final_features_only_linear", "cross_attention_only_linear"][mode]
    keys = {f"{b}_{l}": f"{names[b]}.{leaf[l]}" for b in names for l in leaf}
    keys["fin_w"], keys["fin_b"] = fin + ".weight", fin + ".bias"
    wt = {f: sd[k].cuda().contiguous() for f, k in keys.items()}
    gt = {f: torch.zeros_like(t) for f, t in wt.items()}
    W, G = L.HeadPtrs(), L.HeadPtrs()
    for f in L.HEAD_FIELDS:
        setattr(W, f, wt[f].data_ptr()); setattr(G, f, gt[f].data_ptr())
    return m, keys, wt, gt, W, G


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("reverse", [True, False])
@pytest.mark.parametrize("dims", [(1280, 768), (768, 768), (2048, 768)])
def test_head_fwd_bwd_vs_oracle(mode, reverse, dims):
    d_img, d_txt = dims
    B = 5
    m, keys, wt, gt, W, G = _head_setup(d_img, d_txt, mode, reverse)
    g = torch.Generator().manual_seed(5)
    img, txt = torch.randn(B, d_img, generator=g), torch.randn(B, d_txt, generator=g)
    imgd, txtd = img.cuda(), txt.cuda()
    logits = torch.empty(B, 4, device="cuda")
    L.head_fwd(imgd, txtd, W, logits, B, d_img, d_txt, 4, reverse, mode, 0.0, 0, L.F32)
    ir, tr = img.clone().requires_grad_(True), txt.clone().requires_grad_(True)
    for p in m.parameters():
        p.requires_grad_(True)
    ref = m.head(tr, ir)
    assert rel_err(logits.cpu(), ref.detach()) < 1e-4
    dl = torch.randn(B, 4, generator=g)
    ref.backward(dl)
    dimg, dtxt = torch.empty_like(imgd), torch.empty_like(txtd)
    L.head_bwd(dl.cuda(), imgd, txtd, W, G, dimg, dtxt, B, d_img, d_txt, 4, reverse, mode, 0.0, 0, L.F32)
    assert rel_err(dimg.cpu(), ir.grad) < 1e-3 and rel_err(dtxt.cpu(), tr.grad) < 1e-3   # fp32 vs torch-CPU fp32, amplified test weights
    named = dict(m.named_parameters())
    for f, k in keys.items():
        gref = named[k].grad
        if gref is None:
            assert float(gt[f].abs().max()) == 0.0, f
        else:
            # key-projection biases have an exactly-zero gradient (softmax is shift invariant): absolute floor
            assert (gt[f].cpu() - gref).abs().max() <= 1e-3 * gref.abs().max() + 1e-6, f


def test_head_dropout_is_consistent_between_fwd_and_bwd():
    """With dropout the forward is linear in the kept columns; check d logits / d fin_b and the mask's keep rate."""
    d_img, d_txt, B = 768, 768, 64
    m, keys, wt, gt, W, G = _head_setup(d_img, d_txt, 1, True)
    wt["fin_w"].fill_(1.0); wt["fin_b"].zero_()
    img, txt = torch.rand(B, d_img).cuda() + 0.5, torch.rand(B, d_txt).cuda() + 0.5
    lg0, lg1 = torch.empty(B, 4, device="cuda"), torch.empty(B, 4, device="cuda")
    L.head_fwd(img, txt, W, lg0, B, d_img, d_txt, 4, True, 1, 0.0, 11, L.F32)
    L.head_fwd(img, txt, W, lg1, B, d_img, d_txt, 4, True, 1, 0.6, 11, L.F32)
    # E[dropout(x)] = x: with 1536 columns per sample the means agree to a few percent
    assert abs((lg1.mean() / lg0.mean()).item() - 1.0) < 0.05
    lg2 = torch.empty_like(lg1)
    L.head_fwd(img, txt, W, lg2, B, d_img, d_txt, 4, True, 1, 0.6, 11, L.F32)
    assert torch.equal(lg1, lg2)                      # counter-based mask: same seed, same mask
    dl = torch.ones(B, 4, device="cuda")
    L.head_bwd(dl, img, txt, W, G, None, None, B, d_img, d_txt, 4, True, 1, 0.6, 11, L.F32)
    # sum over classes/columns of dW = sum_b sum_col dropped(x)[b,col] * 1 = sum of logits of one class
    assert rel_err(gt["fin_w"][0].sum(), lg1[:, 0].sum()) < 1e-4


def test_xent_matches_torch():
    B, C = 37, 4
    z = torch.randn(B, C) * 2
    y = torch.randint(0, C, (B,))
    w = torch.tensor([0.5, 2.0, 1.0, 1.5])
    for eps in (0.0, 0.1):
        for cw in (None, w):
            zr = z.clone().requires_grad_(True)
            ref = torch.nn.CrossEntropyLoss(weight=cw, label_smoothing=eps)(zr, y)
            ref.backward()
            loss, dz = torch.empty(1, device="cuda"), torch.empty(B, C, device="cuda")
            L.xent_fwd_bwd(z.cuda(), y.int().cuda(), None if cw is None else cw.cuda(), eps, loss, dz, B, C)
            assert abs(loss.item() - ref.item()) < 1e-5 * max(1, abs(ref.item()))
            assert rel_err(dz.cpu(), zr.grad) < 1e-5


def test_optimizers_match_torch():
    n = 4096 + 64
    p0, g = torch.randn(n), torch.randn(n)
    for kind in ("sgd", "adamw"):
        pt = p0.clone().requires_grad_(True)
        opt = torch.optim.SGD([pt], lr=0.01, weight_decay=0.01) if kind == "sgd" else torch.optim.AdamW([pt], lr=0.01, weight_decay=0.01)
        pd, lp = p0.clone().cuda(), torch.empty(n, device="cuda", dtype=torch.bfloat16)
        mom, var = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        for step in range(1, 4):
            pt.grad = g.clone() * step
            opt.step()
            if kind == "sgd":
                L.sgd_step(pd, (g * step).cuda(), lp, n, 0.01, 0.01)
            else:
                L.adamw_step(pd, (g * step).cuda(), mom, var, lp, n, 0.01, 0.9, 0.999, 1e-8, 0.01, step)
        assert rel_err(pd.cpu(), pt.detach()) < 1e-5
        assert torch.equal(lp.cpu(), pd.cpu().bfloat16())


# ------------------------------------------------------------------------------------------------------
# K7: GPU image preprocessing
# ------------------------------------------------------------------------------------------------------
def test_gpu_image_pipeline_matches_oracle_validation_pipeline():
    from oracle import transforms as T
    from garbage_classification_rca_amd.preprocess import GpuImagePipeline
    rng = np.random.RandomState(11)
    sizes = [(300, 400), (400, 300), (224, 224), (97, 531), (640, 64), (31, 29), (512, 512)]
    imgs = [rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for h, w in sizes]
    flips = [(False, False), (True, False), (False, True), (True, True), (False, False), (True, True), (False, True)]
    pipe = GpuImagePipeline(224, 224, max_batch=8, max_pixels=512 * 512)
    for rep in range(3):                       # alternates the two staging slots
        out = pipe(imgs, flips).cpu().numpy()
        assert out.shape == (len(imgs), 3, 224, 224)
        for b, img in enumerate(imgs):
            ref = T.validation_pipeline(img, 224, 224, flip_v=flips[b][0], flip_h=flips[b][1])
            d = np.abs(out[b] - ref)
            # same float32 formula; a tap sum that lands within rounding of x.5 may round to the neighbouring uint8 step
            assert d.max() <= 1.0 / 255 / 0.224 + 1e-5 and (d > 1e-5).mean() < 1e-3, (sizes[b], d.max(), (d > 1e-5).mean())
    out2 = GpuImagePipeline(480, 384, max_batch=2, max_pixels=400 * 400)(imgs[:2]).cpu().numpy()     # non-square target
    one = GpuImagePipeline(224, 224, max_batch=1, max_pixels=512 * 512)       # --batch_size 1: the constructor's warm-up must fit
    assert np.array_equal(one(imgs[:1]).cpu().numpy(), pipe(imgs[:1]).cpu().numpy())
    for b in range(2):
        ref = T.validation_pipeline(imgs[b], 480, 384)
        assert np.abs(out2[b] - ref).max() <= 1.0 / 255 / 0.224 + 1e-5
