"""The bf16x3 mode (csrc/gemm_x3.hip): fp32-accurate nn.Linear products on the bf16 matrix cores -- every operand carried as two
bf16 planes (hi + lo), three MFMA passes per product -- and the engine mode built on it (--dtype bf16x3), which must meet the
north-star bound (logits within 1e-3 relative of the reference, which is fp32 end to end: multimodal_model.py:651-726) at several
times the speed of the fp32-matrix-core mode.  Needs an MI355X; every call goes through the C ABI.

Tolerances: a bf16x3 product against a float64 product of the ORIGINAL fp32 operands: 2e-5 of the largest output (a plain bf16
product of the same operands is ~3e-3 away); engine logits <= 1e-3 and parameter gradients <= 2e-3 of the group's largest
against the float64 oracle (the same bounds as the fp32 mode's test)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from garbage_classification_rca_amd import lib as L            # noqa: E402
from oracle import model as O                                  # noqa: E402


def planes(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


def test_split_kernel_is_bit_exact_and_carries_16_bits():
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(4096 * 33 * 4, device="cuda", generator=g) * torch.exp(torch.randn(4096 * 33 * 4, device="cuda", generator=g) * 4)
    hi, lo = torch.empty_like(x, dtype=torch.bfloat16), torch.empty_like(x, dtype=torch.bfloat16)
    L.split_f32(x, hi, lo, x.numel())
    rh, rl = planes(x)
    assert torch.equal(hi, rh) and torch.equal(lo, rl)
    err = ((hi.float() + lo.float()) - x).abs() / x.abs().clamp_min(1e-30)
    assert float(err.max()) <= 2.0 ** -16


def _ref(A, B, a_layout, b_layout):
    A64, B64 = A.double(), B.double()
    if a_layout == L.KROW:
        A64 = A64.t()
    if b_layout == L.KROW:
        B64 = B64.t()
    return A64 @ B64.t()


CASES = [
    # (M, N, K, a_layout, b_layout, epilogue, impl)
    (512, 768, 768, L.ROWK, L.ROWK, "bias", L.IMPL_AUTO),
    (200, 256, 128, L.ROWK, L.ROWK, "bias", L.IMPL_AUTO),                 # ragged rows
    (384, 3072, 768, L.ROWK, L.ROWK, "gelu_save", L.IMPL_AUTO),
    (384, 768, 3072, L.ROWK, L.ROWK, "bias_addend", L.IMPL_AUTO),
    (512, 768, 2304, L.ROWK, L.KROW, "addend", L.IMPL_AUTO),              # input gradient + residual-stream gradient
    (512, 3072, 768, L.ROWK, L.KROW, "mul_colsum", L.IMPL_AUTO),          # FFN2 input gradient x gelu' + FFN1 bias gradient
    (1024, 768, 768, L.ROWK, L.ROWK, "bias", L.IMPL_MFMA256),             # persistent 256x256 kernel
    (1024, 3072, 768, L.ROWK, L.ROWK, "gelu_save", L.IMPL_MFMA256),
    (2048, 768, 3072, L.ROWK, L.KROW, "none", L.IMPL_MFMA256),
    (1024, 768, 768, L.ROWK, L.ROWK, "bias_planes", L.IMPL_MFMA256),      # output written as two bf16 planes
    (1024, 3072, 768, L.ROWK, L.ROWK, "gelu_save_planes", L.IMPL_MFMA256),
    (384, 768, 768, L.ROWK, L.ROWK, "bias_planes", L.IMPL_AUTO),
    (512, 3072, 768, L.ROWK, L.KROW, "mul_colsum_planes", L.IMPL_AUTO),
]


@pytest.mark.parametrize("M,N,K,al,bl,ep,impl", CASES)
def test_gemm_x3_against_float64(M, N, K, al, bl, ep, impl):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K) if al == L.ROWK else (K, M), device="cuda", generator=g)
    B = torch.randn((N, K) if bl == L.ROWK else (K, N), device="cuda", generator=g) * 0.05
    rows_pad = (M + 255) // 256 * 256           # the persistent kernel streams whole 256-row tiles of A
    Ap = [torch.zeros(rows_pad, K, dtype=torch.bfloat16, device="cuda") for _ in range(2)] if al == L.ROWK else None
    if Ap is not None:
        h, l = planes(A)
        Ap[0][:M], Ap[1][:M] = h, l
    else:
        Ap = list(planes(A))
    Bp = planes(B)
    bias = torch.randn(N, device="cuda", generator=g) if "bias" in ep or "gelu" in ep else None
    addend = torch.randn(M, N, device="cuda", generator=g) if "addend" in ep else None
    side = torch.randn(M, N, device="cuda", generator=g) if "mul" in ep else None
    preact = side.clone() if side is not None else (torch.empty(M, N, device="cuda") if "gelu" in ep else None)
    colsum = torch.zeros(N, device="cuda") if "colsum" in ep else None
    act = L.ACT_GELU_SAVE_GRAD if "gelu" in ep else (L.ACT_MUL if "mul" in ep else L.ACT_NONE)
    out_planes = "planes" in ep
    if out_planes:
        C, C_lo = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"), torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    else:
        C, C_lo = torch.zeros(M, N, device="cuda"), None
    L.gemm_x3(Ap, Bp, C, C_lo=C_lo, bias=bias, addend=addend, preact=preact, colsum=colsum, M=M, N=N, K=K,
              lda=(K if al == L.ROWK else M), ldb=(K if bl == L.ROWK else N), ldc=N, a_layout=al, b_layout=bl, act=act, impl=impl)
    torch.cuda.synchronize()
    ref = _ref(A, B, al, bl)
    if bias is not None:
        ref = ref + bias.double()
    if act == L.ACT_GELU_SAVE_GRAD:
        z = ref
        Phi = 0.5 * (1 + torch.erf(z / 2 ** 0.5))
        gp = Phi + z * torch.exp(-0.5 * z * z) / (2 * torch.pi) ** 0.5
        assert float((preact.double() - gp).abs().max()) < 2e-5 * float(z.abs().max()) + 2e-6      # |gelu''| <= 1.13: the product's own error
        ref = z * Phi
    if act == L.ACT_MUL:
        ref = ref * side.double()
    if addend is not None:
        ref = ref + addend.double()
    got = (C.float() + C_lo.float()).double() if out_planes else C.double()
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max()) / scale
    bf16_err = float(((A.bfloat16().float().double() if al == L.ROWK else A.bfloat16().float().double().t()) @
                      (B.bfloat16().float().double().t() if bl == L.ROWK else B.bfloat16().float().double()) - _ref(A, B, al, bl)).abs().max()) / float(_ref(A, B, al, bl).abs().max())
    print(f"x3 {ep} M={M} N={N} K={K}: rel err {err:.2e} (a single bf16 product: {bf16_err:.2e})")
    assert err < (4e-5 if out_planes else 2e-5)
    if colsum is not None:
        cs_ref = got.sum(0)
        assert float((colsum.double() - cs_ref).abs().max()) <= 1e-4 * float(cs_ref.abs().max()) + 1e-3


@pytest.mark.parametrize("M,N,K,splitk", [(768, 768, 4096, True), (2304, 768, 1600 * 64 // 64, True), (768, 3072, 50432, True), (256, 128, 640, False),
                                          (768, 768, 448, False)])
def test_weight_gradient_x3_accumulates_into_fp32(M, N, K, splitk):
    """dW[M,N] += dY^T X with dY [K, M], X [K, N] (both KROW): the split-K 256x256 form and the 128x128 atomic form"""
    g = torch.Generator(device="cuda").manual_seed(K)
    dY = torch.randn(K, M, device="cuda", generator=g) * 0.1
    X = torch.randn(K, N, device="cuda", generator=g)
    C0 = torch.randn(M, N, device="cuda", generator=g)
    C = C0.clone()
    if splitk:
        assert L.gemm_splitk_ok(M, N, K, L.BF16)
        ws = torch.empty(L.SPLITK_WS_BYTES, dtype=torch.uint8, device="cuda")
        L.gemm_splitk_x3(planes(dY), planes(X), C, ws, M=M, N=N, K=K, lda=M, ldb=N, ldc=N)
        C2 = C0.clone()
        L.gemm_splitk_x3(planes(dY), planes(X), C2, ws, M=M, N=N, K=K, lda=M, ldb=N, ldc=N)
        torch.cuda.synchronize()
        assert torch.equal(C, C2)                      # no atomics: bitwise reproducible
    else:
        L.gemm_x3(planes(dY), planes(X), C, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, a_layout=L.KROW, b_layout=L.KROW, accum=True)
    torch.cuda.synchronize()
    ref = C0.double() + dY.double().t() @ X.double()
    err = float((C.double() - ref).abs().max()) / float(ref.abs().max())
    print(f"x3 wgrad M={M} N={N} K={K}: rel err {err:.2e}")
    assert err < 2e-5


@pytest.mark.parametrize("passes", [2, 1])
@pytest.mark.parametrize("kind", ["p256", "k1s", "splitk", "atomic"])
def test_reduced_pass_sets_compute_exactly_the_stated_products(passes, kind):
    """B_lo == None: two passes = (A_hi + A_lo) . B_hi; A_lo == None as well: one pass = A_hi . B_hi -- the opt-in cheaper backward of the
    bf16x3 mode (MMRCA_X3_WGRAD_PASSES / MMRCA_X3_DGRAD_PASSES; the forward always runs all three).  Checked against float64 products of
    the planes that are supposed to take part, on each kernel that has the bf16x3 form."""
    g = torch.Generator(device="cuda").manual_seed(passes * 7 + len(kind))
    if kind in ("p256", "k1s"):
        M, N, K = (1024, 768, 768) if kind == "p256" else (384, 256, 1280)
        A = torch.randn(M, K, device="cuda", generator=g)
        B = torch.randn(N, K, device="cuda", generator=g) * 0.05
        (ah, al), (bh, bl) = planes(A), planes(B)
        C = torch.empty(M, N, device="cuda")
        L.gemm_x3((ah, al if passes == 2 else None), (bh, None), C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N,
                  impl=L.IMPL_MFMA256 if kind == "p256" else L.IMPL_MFMA_1STAGE)
        a_used = ah.double() + (al.double() if passes == 2 else 0.0)
        ref = a_used @ bh.double().t()
        full = A.double() @ B.double().t()
    else:
        M, N, K = (768, 768, 4096) if kind == "splitk" else (256, 128, 640)
        dY = torch.randn(K, M, device="cuda", generator=g) * 0.1
        X = torch.randn(K, N, device="cuda", generator=g)
        (ah, al), (bh, bl) = planes(dY), planes(X)
        C = torch.zeros(M, N, device="cuda")
        a_pl, b_pl = (ah, al if passes == 2 else None), (bh, None)
        if kind == "splitk":
            ws = torch.empty(L.SPLITK_WS_BYTES, dtype=torch.uint8, device="cuda")
            L.gemm_splitk_x3(a_pl, b_pl, C, ws, M=M, N=N, K=K, lda=M, ldb=N, ldc=N)
        else:
            L.gemm_x3(a_pl, b_pl, C, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, a_layout=L.KROW, b_layout=L.KROW, accum=True)
        a_used = ah.double() + (al.double() if passes == 2 else 0.0)
        ref = a_used.t() @ bh.double()
        full = dY.double().t() @ X.double()
    torch.cuda.synchronize()
    scale = float(full.abs().max())
    assert float((C.double() - ref).abs().max()) / scale < 3e-6            # fp32 accumulation of exactly these planes
    dropped = float((C.double() - full).abs().max()) / scale
    print(f"{kind} with {passes} pass(es): {dropped:.2e} from the full product")
    assert 1e-4 < dropped < 2e-2                                             # and visibly not the three-pass product


def test_explicit_256_kernel_rejects_a_ragged_row_count():
    A = [torch.zeros(512, 128, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
    B = [torch.zeros(256, 128, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
    C = torch.zeros(512, 256, device="cuda")
    with pytest.raises(L.MmrcaError):
        L.gemm_x3(A, B, C, M=300, N=256, K=128, lda=128, ldb=128, ldc=256, impl=L.IMPL_MFMA256)


def test_optimizer_steps_refresh_both_planes():
    n = 4096
    g = torch.Generator(device="cuda").manual_seed(3)
    p = torch.randn(n, device="cuda", generator=g)
    gr = torch.randn(n, device="cuda", generator=g)
    hi, lo = torch.empty(n, dtype=torch.bfloat16, device="cuda"), torch.empty(n, dtype=torch.bfloat16, device="cuda")
    L.sgd_step(p, gr, hi, n, 1e-2, 1e-2, lp_lo=lo)
    rh, rl = planes(p)
    assert torch.equal(hi, rh) and torch.equal(lo, rl)
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    L.adamw_step(p, gr, m, v, hi, n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 1, lp_lo=lo)
    rh, rl = planes(p)
    assert torch.equal(hi, rh) and torch.equal(lo, rl)


# ---------------------------------------------------------------------------------------------------------------------------
# the engine mode
# ---------------------------------------------------------------------------------------------------------------------------
def _pair(text="distilbert", image="transformer_B16", mode=0):
    from tests.test_engine_gpu import _build_pair
    return _build_pair("bf16x3", text, image, mode)


def _pair_init_weights(mode, seed=0):
    """the engine's own initialisation (what bench.py trains from: encoders N(0, 0.02), LayerNorm (1, 0), torch defaults in the
    head) copied into the oracle"""
    from garbage_classification_rca_amd.engine import MMRCAEngine
    eng = MMRCAEngine("distilbert", "transformer_B16", 4, True, mode, "bf16x3")
    eng.init_parameters(seed)
    sd = {k: eng.arena.view(k).detach().cpu().clone() for k in eng.param_keys}
    orc = O.build_oracle("distilbert", "transformer_B16", True, mode == 1, mode == 2, drop_ratio=0.0, enc_dropout=0.0).eval()
    orc.text_model.load_flat(sd, "text_model.")
    orc.image_model.load_flat(sd, "image_model.")
    orc.load_state_dict({k: v for k, v in sd.items() if not k.startswith(("text_model.", "image_model."))}, strict=False)
    return eng, orc, sd


@pytest.mark.parametrize("weights,mode,gtol", [("init", 0, 2e-3), ("init", 2, 2e-3), ("amplified", 0, 5e-2), ("amplified", 2, 5e-2)])
def test_x3_logits_and_gradients_match_the_float64_oracle(weights, mode, gtol):
    """Logits <= 1e-3 (north-star bound) and every parameter gradient within gtol of its tensor's largest entry, against the oracle
    evaluated in float64.  "init": the initialisation the model trains from -- the fp32 mode's bound (2e-3) holds with two orders of
    margin (measured ~3e-5).  "amplified": the procedural stress weights of test_engine_gpu.py (large, structured), on which the
    backward is ill-conditioned -- the fp32 engine itself is 5e-4 .. 8e-4 away from float64 there with products exact to 1e-7
    (gradients of the head's attention projections are differences of nearly equal terms).  A bf16x3 product is exact to ~3e-6 (two
    bf16 planes carry 17-18 significant bits), i.e. ~30x the fp32 product's error, and lands at 3e-3 .. 3e-2 of a tensor's largest
    entry there (worst: cross_attention_1.W_query.weight under --cross_attention_only, a gradient of size 8e-4 that is the
    difference of O(1) terms); bound 5e-2, stated rather than hidden.  The logits stay at 2-3e-5 on these weights too."""
    from tests.test_engine_gpu import _inputs, rel
    B, S_len = 3, 24
    eng, orc, sd = _pair(mode=mode) if weights == "amplified" else _pair_init_weights(mode)
    assert eng.x3 and eng.dtype == torch.float32
    ids, mask, images = _inputs(B, S_len)
    logits = eng.forward(ids.cuda(), mask.cuda(), images.cuda())
    orc = orc.double()
    for p in orc.parameters():
        p.requires_grad_(True)
    ref = orc(ids, mask, images.double(), eval=True)
    e = rel(logits, ref.detach())
    print(f"bf16x3 [{weights}] logits relative error vs the float64 oracle:", e)
    assert e < 1e-3
    labels = torch.tensor([0, 1, 2][:B])
    cw = torch.tensor([0.7, 1.3, 0.9, 1.1])
    O.cross_entropy(ref, labels, cw.double(), 0.1).backward()
    loss, dl = torch.empty(1, device="cuda"), torch.empty(B, 4, device="cuda")
    L.xent_fwd_bwd(logits, labels.int().cuda(), cw.cuda(), 0.1, loss, dl, B, 4)
    eng.arena.g.zero_()
    eng.backward(dl)
    torch.cuda.synchronize()
    named = {"text_model." + k.replace("/", "."): p for k, p in orc.text_model.params.items()}
    named.update({"image_model." + k.replace("/", "."): p for k, p in orc.image_model.params.items()})
    named.update({k: p for k, p in orc.named_parameters() if not k.startswith(("text_model.", "image_model."))})
    worst, worst_k = 0.0, None
    gmax = max(float(p.grad.abs().max()) for p in named.values() if p.grad is not None)
    for k in eng.param_keys:
        got = eng.arena.view(k, "g").cpu().double()
        gr = named[k].grad
        if gr is None:
            assert float(got.abs().max()) == 0.0, k
            continue
        err = (got - gr.view_as(got)).abs().max().item()
        scale = max(gr.abs().max().item(), 1e-3 * gmax)
        if err / scale > worst:
            worst, worst_k = err / scale, k
        assert err <= gtol * scale, (k, err, scale)
    print(f"worst relative gradient error (bf16x3 engine [{weights}] vs float64 oracle):", worst, worst_k)
    eng.release_buffers()


def test_x3_at_the_benchmarked_shape_b256_and_configs3():
    """BASELINE configs[1] at B = 256 with packed captions and the class-token tail (what bench.py --dtype bf16x3 times): logits
    <= 1e-3 against the oracle; then configs[3]'s pairing (ViT-L/16 + BERT-base, --cross_attention_only, S = 128) at a small batch."""
    import os
    from garbage_classification_rca_amd import engine as E
    from garbage_classification_rca_amd.procedural import proc_input, synth_captions
    from tests.test_engine_gpu import rel
    B, S_len = 256, 64
    eng, orc, sd = _pair()
    ids_np, mask_np = synth_captions(B, S_len, seed=4321)
    images = torch.from_numpy(proc_input("b256.images", (B, 3, 224, 224)))
    ids, mask = torch.from_numpy(ids_np), torch.from_numpy(mask_np)
    pack = E.make_text_pack(mask_np, "cuda")
    logits = eng.forward(ids.cuda(), mask.cuda(), images.cuda(), text_pack=pack)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    n = 64                                   # the oracle on the first 64 pairs (CPU time)
    with torch.no_grad():
        ref = torch.cat([orc(ids[i:i + 32], mask[i:i + 32], images[i:i + 32], eval=True) for i in range(0, n, 32)])
    e = rel(logits[:n], ref)
    print("B=256 bf16x3 logits relative error vs oracle:", e)
    assert e < 1e-3
    eng.release_buffers()
    del eng
    torch.cuda.empty_cache()
    eng, orc, sd = _pair("bert", "transformer_L16", mode=2)
    ids_np, mask_np = synth_captions(4, 128, seed=7)
    images = torch.from_numpy(proc_input("cfg3.images", (4, 3, 224, 224)))
    logits = eng.forward(torch.from_numpy(ids_np).cuda(), torch.from_numpy(mask_np).cuda(), images.cuda())
    with torch.no_grad():
        ref = orc(torch.from_numpy(ids_np), torch.from_numpy(mask_np), images, eval=True)
    e = rel(logits, ref)
    print("configs[3] pairing, bf16x3 logits relative error vs oracle:", e)
    assert e < 1e-3
    eng.release_buffers()


def test_x3_train_steps_track_the_fp32_mode():
    """three fused train steps (SGD) in bf16x3 and in fp32 mode from the same weights on the same batches: same losses, same weights"""
    from garbage_classification_rca_amd.multimodal_model import MM_RCA
    from garbage_classification_rca_amd.optim import FlatSGD
    from garbage_classification_rca_amd.procedural import synth_captions
    from garbage_classification_rca_amd.training import FusedCrossEntropy, hip_train_step
    import contextlib, io
    B = 8
    ids_np, mask_np = synth_captions(B, 32, seed=11)
    ids, mask = torch.from_numpy(ids_np).cuda(), torch.from_numpy(mask_np).cuda()
    images = torch.randn(B, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    labels = (torch.arange(B, device="cuda") % 4).int()
    runs = {}
    for dt in ("bf16x3", torch.float32):
        with contextlib.redirect_stdout(io.StringIO()):
            m = MM_RCA(4, 0.0, 0.0, 0.7, 256, "distilbert", B, True, False, False, image_model_name="transformer_B16", dtype=dt, init_seed=0)
        m.enc_dropout = 0.0
        m.train()
        for p in m.parameters():
            p.requires_grad = True
        opt = FlatSGD(m, lr=1e-2, weight_decay=1e-2)
        crit = FusedCrossEntropy(None, 0.0)
        losses = [float(hip_train_step(m, ids, mask, images, labels, crit, opt).item()) for _ in range(3)]
        runs[str(dt)] = (losses, m.engine.arena.p.clone())
        if dt == "bf16x3":       # the hi + lo planes the next forward reads are the fused optimizer's
            ar = m.engine.arena
            h, l = planes(ar.p)
            assert torch.equal(ar.lp, h) and torch.equal(ar.lp_lo, l)
        m.engine.release_buffers()
    (l3, p3), (l32, p32) = runs["bf16x3"], runs["torch.float32"]
    print("losses bf16x3:", l3, "fp32:", l32)
    assert max(abs(a - b) for a, b in zip(l3, l32)) < 2e-4
    assert float((p3 - p32).abs().max()) < 2e-5
