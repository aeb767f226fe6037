"""The bf16x3f mode (--dtype bf16x3f): the bf16x3 FORWARD -- fp32 residual stream / LayerNorm / attention, every nn.Linear as a
three-pass split-bf16 product, logits within the north-star bound of the fp32 reference (CVPR_code/multimodal_model.py:651-726) --
behind the bf16 mode's BACKWARD (single-pass bf16 products and the bf16 attention backward on the hi planes of the saved
activations).  Needs an MI355X; every call goes through the C ABI.

Tolerances: logits <= 1e-3 against the float64 oracle (observed 1e-6 .. 4e-5); parameter gradients are bf16-backward gradients:
cosine >= 0.999 per tensor against the float64 oracle and no worse than the bf16 mode's on the same weights."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from garbage_classification_rca_amd import lib as L            # noqa: E402
from garbage_classification_rca_amd.engine import MMRCAEngine  # noqa: E402
from oracle import model as O                                  # noqa: E402


def _planes(x):
    hi = x.to(torch.bfloat16)
    return hi.contiguous(), (x - hi.float()).to(torch.bfloat16).contiguous()


@pytest.mark.parametrize("rows,D,dres,branch,dcol", [(300, 768, True, False, True), (64, 1024, False, True, True), (513, 768, True, True, False),
                                                       (7, 512, False, False, False)])
def test_layernorm_bwd_mixed_matches_fp32_autograd(rows, D, dres, branch, dcol):
    """bf16 gradients / gamma against an fp32 saved sum: compared with torch autograd in fp32 on the SAME (bf16-rounded) gradient
    inputs; the outputs are bf16, so 2^-8 of the largest entry bounds them"""
    g = torch.Generator(device="cuda").manual_seed(rows + D)
    s = torch.randn(rows, D, device="cuda", generator=g) * 2 + 0.3
    gamma = (1 + 0.1 * torch.randn(D, device="cuda", generator=g)).bfloat16()
    dy = (torch.randn(rows, D, device="cuda", generator=g) * 0.1).bfloat16()
    dr = (torch.randn(rows, D, device="cuda", generator=g) * 0.1).bfloat16() if dres else None
    mean, var = s.mean(1), s.var(1, unbiased=False)
    rstd = (var + 1e-6).rsqrt()
    ds = torch.zeros(rows, D, dtype=torch.bfloat16, device="cuda")
    dbr = torch.zeros(rows, D, dtype=torch.bfloat16, device="cuda") if branch else None
    dgam, dbet = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    dc = torch.zeros(D, device="cuda") if dcol else None
    dcb = torch.zeros(D, device="cuda") if (branch and dcol) else None
    L.layernorm_bwd_mixed(dy, s, gamma, mean.contiguous(), rstd.contiguous(), dr, ds, dgam, dbet, rows, D, D, D, D, dbranch=dbr, dcol=dc, dcol_branch=dcb)
    torch.cuda.synchronize()
    sr = s.clone().requires_grad_(True)
    gm = gamma.float().requires_grad_(True)
    bt = torch.zeros(D, device="cuda", requires_grad=True)
    y = torch.nn.functional.layer_norm(sr, (D,), gm, bt, 1e-6)
    y.backward(dy.float())
    ref = sr.grad
    tol = 2.0 ** -7
    if branch:
        assert float((dbr.float() - ref).abs().max()) <= tol * float(ref.abs().max())
    want = ref + dr.float() if dres else ref
    assert float((ds.float() - want).abs().max()) <= tol * float(want.abs().max())
    assert float((dgam - gm.grad).abs().max()) <= 1e-4 * float(gm.grad.abs().max()) + 1e-5
    assert float((dbet - bt.grad).abs().max()) <= 1e-4 * float(bt.grad.abs().max()) + 1e-5
    if dcol:
        assert float((dc - ds.float().sum(0)).abs().max()) <= 2e-3 * float(ds.float().sum(0).abs().max()) + 1e-3
    if dcb is not None:
        assert float((dcb - dbr.float().sum(0)).abs().max()) <= 2e-3 * float(dbr.float().sum(0).abs().max()) + 1e-3


@pytest.mark.parametrize("M,impl", [(1024, L.IMPL_MFMA256), (384, L.IMPL_AUTO), (2304, L.IMPL_AUTO)])
def test_gemm_x3_writes_the_gelu_derivative_as_bf16(M, impl):
    """MMRCA_ACT_GELU_SAVE_GRAD_BF16: the same product and gelu as MMRCA_ACT_GELU_SAVE_GRAD, with gelu' stored as bf16 (what the
    bf16 backward multiplies by) -- bit-equal to rounding the fp32 form's gelu'"""
    N, K = 3072, 768
    g = torch.Generator(device="cuda").manual_seed(M)
    A, B = torch.randn(M, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g) * 0.05
    bias = torch.randn(N, device="cuda", generator=g)
    rows_pad = (M + 255) // 256 * 256
    Ap = [torch.zeros(rows_pad, K, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
    Ap[0][:M], Ap[1][:M] = _planes(A)
    outs = {}
    for act, pdt in ((L.ACT_GELU_SAVE_GRAD, torch.float32), (L.ACT_GELU_SAVE_GRAD_BF16, torch.bfloat16)):
        C, C_lo = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"), torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        pre = torch.zeros(M, N, dtype=pdt, device="cuda")
        L.gemm_x3(Ap, _planes(B), C, C_lo=C_lo, bias=bias, preact=pre, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, act=act, impl=impl)
        outs[act] = (C, C_lo, pre)
    torch.cuda.synchronize()
    (c0, l0, p0), (c1, l1, p1) = outs[L.ACT_GELU_SAVE_GRAD], outs[L.ACT_GELU_SAVE_GRAD_BF16]
    assert torch.equal(c0, c1) and torch.equal(l0, l1)
    assert torch.equal(p0.bfloat16(), p1)


@pytest.mark.parametrize("B,H,S,packed", [(3, 12, 197, False), (5, 12, 64, True), (2, 16, 128, False)])
def test_attention_forward_from_planes_equals_fp32_input(B, H, S, packed):
    """mmrca_mha_fwd_planes_in (q|k|v as hi + lo bf16 planes) against mmrca_mha_fwd_planes on the fp32 values the planes encode"""
    dh = 64
    D = H * dh
    g = torch.Generator(device="cuda").manual_seed(S)
    if packed:
        lens = torch.tensor([S, 9, 33, 1, 50][:B])
        cu = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).int().cuda()
        rows = int(lens.sum())
        mask = torch.ones(rows, dtype=torch.int32, device="cuda")
    else:
        cu, rows = None, B * S
        mask = (torch.rand(rows, device="cuda", generator=g) > 0.2).int()
        mask.view(B, S)[:, 0] = 1
    hi, lo = _planes(torch.randn(rows, 3 * D, device="cuda", generator=g))
    qkv = hi.float() + lo.float()            # exactly what the planes encode
    out = {}
    for name in ("f32", "planes"):
        o = torch.zeros(rows, D, device="cuda")
        pl = (torch.zeros(rows, D, dtype=torch.bfloat16, device="cuda"), torch.zeros(rows, D, dtype=torch.bfloat16, device="cuda"))
        lse = torch.zeros(B * H * S, device="cuda")
        if name == "f32":
            L.mha_fwd_planes(qkv, mask, o, pl, lse, B, H, S, dh, dh ** -0.5, cu=cu)
        else:
            L.mha_fwd_planes_in((hi, lo), mask, None, pl, lse, B, H, S, dh, dh ** -0.5, cu=cu)
        out[name] = (o, pl, lse)
    torch.cuda.synchronize()
    assert torch.equal(out["f32"][1][0], out["planes"][1][0]) and torch.equal(out["f32"][1][1], out["planes"][1][1])
    assert torch.equal(out["f32"][2], out["planes"][2])
    assert float(out["planes"][0].abs().max()) == 0.0          # the fp32 context is optional and was not asked for


@pytest.mark.parametrize("B,H,S", [(3, 12, 197), (24, 12, 197), (70, 12, 197), (30, 16, 224), (23, 12, 193)])
def test_attention_forward_x3_persistent_form_for_the_vit(B, H, S):
    """mha_fwd_x3_p_k (round 6): no mask, no dropout, padded layout, 193 <= S <= 224 -- a 16-wave workgroup per CU walks heads, the K and
    V image pairs time-share the staging pipeline (36 heads: one per workgroup; 288 / 840 / 480 / 276: two to four per workgroup, a
    ragged last round).  Against a float64 softmax(QK^T)V on the values the planes encode: context within 3e-5 of the largest entry (4x the
    worst observed), log-sum-exp within 1e-5 relative; and bit-identical when run again (no atomics, no order dependence)."""
    dh = 64
    D = H * dh
    rows = B * S
    g = torch.Generator(device="cuda").manual_seed(S + B)
    hi, lo = _planes(torch.randn(rows, 3 * D, device="cuda", generator=g))
    qkv = hi.float() + lo.float()

    def run():
        pl = (torch.zeros(rows, D, dtype=torch.bfloat16, device="cuda"), torch.zeros(rows, D, dtype=torch.bfloat16, device="cuda"))
        lse = torch.zeros(B * H * S, device="cuda")
        L.mha_fwd_x3((hi, lo), None, pl, lse, B, H, S, dh, dh ** -0.5)
        return pl, lse
    pl, lse = run()
    got = pl[0].float() + pl[1].float()
    x = qkv.double().view(B, S, 3, H, dh)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    sc = (q @ k.transpose(2, 3)) * dh ** -0.5
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(rows, D).float()
    lref = torch.logsumexp(sc, -1).reshape(-1).float()           # [B, H, S]
    torch.cuda.synchronize()
    e = float((got - ref).abs().max()) / float(ref.abs().max())
    el = float((lse - lref).abs().max())
    print(f"bf16x3 persistent attention forward B={B} H={H} S={S}: context error {e:.2e}, lse error {el:.2e}")
    # observed over the five shapes: context 5.0e-6 .. 7.2e-6, log-sum-exp 1.9e-6 .. 4.3e-6 (seeded inputs, no atomics: repeatable)
    assert e < 3e-5 and el <= 1e-5 * max(1.0, float(lref.abs().max()))
    pl2, lse2 = run()
    assert torch.equal(pl2[0], pl[0]) and torch.equal(pl2[1], pl[1]) and torch.equal(lse2, lse)


@pytest.mark.parametrize("B,H,S,packed,drop", [(3, 12, 197, False, 0.0), (5, 12, 64, True, 0.0), (2, 16, 128, False, 0.1), (4, 12, 64, False, 0.1),
                                               (2, 12, 224, False, 0.0), (3, 12, 17, True, 0.0)])
def test_attention_forward_x3_on_the_bf16_matrix_cores_matches_the_fp32_kernels(B, H, S, packed, drop):
    """mmrca_mha_fwd_x3 (three-pass products over the q|k|v planes, P split in registers) against the fp32-matrix-core forward
    on the values the planes encode -- and, for S <= 208 only (the fp32 kernel's limit), a float64 softmax(QK^T)V otherwise:
    context within 2e-5 of the largest entry, log-sum-exp within 1e-5; masks, packed layout and dropout (same counter masks)"""
    dh = 64
    D = H * dh
    g = torch.Generator(device="cuda").manual_seed(S + B)
    if packed:
        lens = torch.tensor([S, 9, 33, 1, 50][:B]).clamp(max=S)
        cu = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).int().cuda()
        rows = int(lens.sum())
        mask = torch.ones(rows, dtype=torch.int32, device="cuda")
    else:
        cu, rows = None, B * S
        mask = (torch.rand(rows, device="cuda", generator=g) > 0.2).int()
        mask.view(B, S)[:, 0] = 1
    hi, lo = _planes(torch.randn(rows, 3 * D, device="cuda", generator=g))
    qkv = hi.float() + lo.float()
    pl = (torch.zeros(rows, D, dtype=torch.bfloat16, device="cuda"), torch.zeros(rows, D, dtype=torch.bfloat16, device="cuda"))
    lse = torch.zeros(B * H * S, device="cuda")
    L.mha_fwd_x3((hi, lo), mask, pl, lse, B, H, S, dh, dh ** -0.5, drop_p=drop, drop_seed=77, cu=cu)
    got = pl[0].float() + pl[1].float()
    if S <= 208:
        o = torch.zeros(rows, D, device="cuda")
        pr = (torch.zeros(rows, D, dtype=torch.bfloat16, device="cuda"), torch.zeros(rows, D, dtype=torch.bfloat16, device="cuda"))
        lr = torch.zeros(B * H * S, device="cuda")
        L.mha_fwd_planes(qkv, mask, o, pr, lr, B, H, S, dh, dh ** -0.5, drop_p=drop, drop_seed=77, cu=cu)
        torch.cuda.synchronize()
        ref = o
        live = torch.isfinite(lr)
        assert torch.equal(torch.isfinite(lse), live)
        assert float((lse[live] - lr[live]).abs().max()) <= 1e-5 * max(1.0, float(lr[live].abs().max()))
    else:
        x = qkv.double().view(B, S, 3, H, dh)
        q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
        sc = (q @ k.transpose(2, 3)) * dh ** -0.5
        sc = sc.masked_fill(mask.view(B, 1, 1, S) == 0, float("-inf"))
        ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(rows, D).float()
    torch.cuda.synchronize()
    e = float((got - ref).abs().max()) / float(ref.abs().max())
    print(f"bf16x3 attention forward B={B} H={H} S={S} packed={packed} drop={drop}: context error {e:.2e}")
    assert e < 2e-5


def _grads_vs_oracle(eng, orc, B=3, S_len=24):
    from tests.test_engine_gpu import _inputs, rel
    ids, mask, images = _inputs(B, S_len)
    logits = eng.forward(ids.cuda(), mask.cuda(), images.cuda())
    orc = orc.double()
    for p in orc.parameters():
        p.requires_grad_(True)
        p.grad = None
    ref = orc(ids, mask, images.double(), eval=True)
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
    gmax = max(float(p.grad.abs().max()) for p in named.values() if p.grad is not None)
    worst, cos_min = 0.0, 1.0
    for k in eng.param_keys:
        gr = named[k].grad
        got = eng.arena.view(k, "g").cpu().double()
        if gr is None:
            assert float(got.abs().max()) == 0.0, k
            continue
        gr = gr.view_as(got)
        worst = max(worst, float((got - gr).abs().max()) / max(float(gr.abs().max()), 1e-3 * gmax))
        if float(gr.abs().max()) > 1e-3 * gmax and gr.numel() >= 64:
            cos_min = min(cos_min, float(torch.nn.functional.cosine_similarity(got.flatten(), gr.flatten(), dim=0)))
    return rel(logits, ref.detach()), worst, cos_min


@pytest.mark.parametrize("mode", [0, 2])
def test_x3f_forward_is_the_x3_forward_and_backward_is_a_bf16_backward(mode):
    """On the initialisation the model trains from: logits <= 1e-3 against the float64 oracle (the bf16x3 forward: ~1e-6), every
    parameter gradient a faithful bf16-precision gradient (cosine >= 0.999, worst entry within 6e-2 of the tensor's largest) and
    not worse than the bf16 mode's own gradients on the same weights."""
    from tests.test_x3_gpu import _pair_init_weights
    eng3, orc, sd = _pair_init_weights(mode)
    eng3.release_buffers()
    del eng3
    res = {}
    for name, dt in (("bf16x3f", "bf16x3f"), ("bf16", torch.bfloat16)):
        eng = MMRCAEngine("distilbert", "transformer_B16", 4, True, mode, dt)
        eng.load_arrays(sd)
        res[name] = _grads_vs_oracle(eng, orc)
        eng.release_buffers()
        del eng
        torch.cuda.empty_cache()
    (lf, wf, cf), (lb, wb, cb) = res["bf16x3f"], res["bf16"]
    print(f"bf16x3f: logits {lf:.2e}, worst gradient entry {wf:.2e}, min cosine {cf:.5f};  bf16: logits {lb:.2e}, worst {wb:.2e}, min cosine {cb:.5f}")
    assert lf < 1e-3 and lf < 1e-4
    assert cf >= 0.999 and wf <= 6e-2
    assert wf <= 1.5 * wb + 1e-3


def test_x3f_at_the_benchmarked_shape_b256_and_configs3():
    """BASELINE configs[1] at B = 256 with packed captions and the class-token tail (what bench.py's `compliant` leg times) and
    configs[3]'s pairing (ViT-L/16 + BERT-base, --cross_attention_only, S = 128): logits <= 1e-3 against the float64 oracle."""
    import os
    from garbage_classification_rca_amd import engine as E
    from garbage_classification_rca_amd.procedural import proc_input, synth_captions
    from tests.test_engine_gpu import rel, _build_pair
    B, S_len = 256, 64
    eng, orc, sd = _build_pair("bf16x3f", "distilbert", "transformer_B16", 0)
    assert eng.x3f
    ids_np, mask_np = synth_captions(B, S_len, seed=4321)
    images = torch.from_numpy(proc_input("b256.images", (B, 3, 224, 224)))
    ids, mask = torch.from_numpy(ids_np), torch.from_numpy(mask_np)
    logits = eng.forward(ids.cuda(), mask.cuda(), images.cuda(), text_pack=E.make_text_pack(mask_np, "cuda"))
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    n = 32
    orc = orc.double()
    with torch.no_grad():
        ref = orc(ids[:n], mask[:n], images[:n].double(), eval=True)
    e = rel(logits[:n], ref)
    print("B=256 bf16x3f logits relative error vs the float64 oracle:", e)
    assert e < 1e-3
    eng.release_buffers()
    del eng
    torch.cuda.empty_cache()
    eng, orc, sd = _build_pair("bf16x3f", "bert", "transformer_L16", 2)
    ids_np, mask_np = synth_captions(4, 128, seed=7)
    images = torch.from_numpy(proc_input("cfg3.images", (4, 3, 224, 224)))
    logits = eng.forward(torch.from_numpy(ids_np).cuda(), torch.from_numpy(mask_np).cuda(), images.cuda())
    with torch.no_grad():
        ref = orc.double()(torch.from_numpy(ids_np), torch.from_numpy(mask_np), images.double(), eval=True)
    e = rel(logits, ref)
    print("configs[3] pairing, bf16x3f logits relative error vs the float64 oracle:", e)
    assert e < 1e-3
    eng.release_buffers()


def test_x3f_train_steps_follow_the_fp32_mode_and_keep_the_planes():
    """three fused SGD steps with text-encoder dropout ON (the backward must regenerate the forward's masks) from the same weights in
    bf16x3f and fp32: first-step loss equal to fp32 accuracy (same forward), later losses within bf16-gradient drift, and the
    hi / lo planes the next forward reads are the fused optimizer's"""
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
    for dt in ("bf16x3f", torch.float32):
        with contextlib.redirect_stdout(io.StringIO()):
            m = MM_RCA(4, 0.0, 0.0, 0.7, 256, "distilbert", B, True, False, False, image_model_name="transformer_B16", dtype=dt, init_seed=0)
        m.train()
        for p in m.parameters():
            p.requires_grad = True
        opt = FlatSGD(m, lr=1e-2, weight_decay=1e-2)
        crit = FusedCrossEntropy(None, 0.0)
        losses = [float(hip_train_step(m, ids, mask, images, labels, crit, opt).item()) for _ in range(3)]
        runs[str(dt)] = (losses, m.engine.arena.p.clone())
        if dt == "bf16x3f":
            ar = m.engine.arena
            h, l = _planes(ar.p)
            assert torch.equal(ar.lp, h) and torch.equal(ar.lp_lo, l)
        m.engine.release_buffers()
    (l3, p3), (l32, p32) = runs["bf16x3f"], runs["torch.float32"]
    print("losses bf16x3f:", l3, "fp32:", l32)
    assert abs(l3[0] - l32[0]) < 2e-5                  # the forward is the fp32-accurate one
    assert max(abs(a - b) for a, b in zip(l3, l32)) < 5e-3
    assert float((p3 - p32).abs().max()) < 2e-3


@pytest.mark.parametrize("text_model,image_model,size,B", [("distilbert", "eff_v2_medium", 224, 4), ("roberta", "eff_v2_large", 160, 3)])
def test_x3f_with_a_conv_backbone_runs_the_conv_kernels_in_bf16_and_meets_the_bound(text_model, image_model, size, B):
    """bf16x3f over a conv image backbone (the reference's default model, multimodal_model.py:11-36, 113-126, and configs[2]'s pairing):
    the text encoder is the fp32-accurate bf16x3 forward, the conv backbone runs its bf16 kernels on the hi planes of the weights.
    Logits <= 1e-3 against the oracle (eval-mode BatchNorm) -- the bf16 mode itself does not meet it on these weights because of its
    TEXT encoder -- and a train-mode step reaches every parameter group with finite gradients that match the bf16 mode's conv gradients."""
    from garbage_classification_rca_amd.procedural import synth_captions
    from tests.test_engine_gpu import rel
    eng = MMRCAEngine(text_model, image_model, 4, True, 0, "bf16x3f", image_size=size)
    assert eng.x3f and eng.conv_dtype == torch.bfloat16 and eng.conv.cdtype == torch.bfloat16
    eng.init_parameters(0)
    sd = {k: eng.arena.view(k).detach().cpu().clone() for k in eng.param_keys}
    orc = O.build_oracle(text_model, image_model, True, False, False, drop_ratio=0.0, enc_dropout=0.0).eval()
    orc.text_model.load_flat(sd, "text_model.")
    orc.image_model.load_state_dict({k[len("image_model."):]: v for k, v in sd.items() if k.startswith("image_model.")}, strict=False)
    orc.load_state_dict({k: v for k, v in sd.items() if not k.startswith(("text_model.", "image_model."))}, strict=False)
    ids, mask = (torch.from_numpy(a) for a in synth_captions(B, 32, seed=1))
    if text_model == "roberta":
        ids = ids.clone(); ids[mask == 0] = 1
    images = torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(3))
    logits = eng.forward(ids.cuda(), mask.cuda(), images.cuda(), save=False, bn_train=False)
    with torch.no_grad():
        ref = orc(ids, mask, images, eval=True)
    e = rel(logits, ref)
    e16 = MMRCAEngine(text_model, image_model, 4, True, 0, torch.bfloat16, image_size=size)
    e16.load_arrays(sd)
    eb = rel(e16.forward(ids.cuda(), mask.cuda(), images.cuda(), save=False, bn_train=False), ref)
    print(f"{image_model} + {text_model} {size}^2: logits vs oracle bf16x3f {e:.2e}, bf16 {eb:.2e}")
    # (the bf16 mode's error is printed for comparison only: at full size it is several times larger and misses the bound because of its text
    # encoder -- bench.py's parity object, configs[2]: 2.1e-3 vs 4.6e-4 -- at this test's small images the conv backbone's bf16 rounding
    # dominates both modes and they come out alike, 4-5e-4)
    assert e < 1e-3
    # one train-mode step: every group is reached with finite gradients; and for the SAME gradient at the backbone's output the conv
    # gradients of the two modes coincide (the same bf16 kernels on the same bf16 weights and activations)
    dl = torch.randn(B, 4, generator=torch.Generator().manual_seed(5)).cuda() * 0.1
    dfeat = torch.randn(B, eng.d_img, generator=torch.Generator().manual_seed(6)).cuda() * 0.1
    grads = {}
    for name, en in (("x3f", eng), ("bf16", e16)):
        en.conv.injected_keep = torch.ones(sum(1 for b in en.conv.blocks if b.get("res") and b.get("sd", 0) > 0), B)
        en.forward(ids.cuda(), mask.cuda(), images.cuda(), drop_p=0.0, seed=5, save=True, bn_train=True)
        en.arena.g.zero_()
        en.backward(dl)
        torch.cuda.synchronize()
        assert torch.isfinite(en.arena.g).all()
        for grp in ("image_emb", "text_emb", "head"):
            lo, hi = en.groups[grp]
            assert float(en.arena.g[lo:hi].abs().max()) > 0, (name, grp)
        # (BatchNorm on running statistics for the comparison: with batch statistics a randomly initialised EfficientNetV2 amplifies
        # the summation-order noise of its own fp32 atomics ~30x -- test_fullsize_gpu.py -- so two runs of ONE engine already differ)
        en.forward(ids.cuda(), mask.cuda(), images.cuda(), drop_p=0.0, seed=5, save=True, bn_train=False)
        en.arena.g.zero_()
        en.conv.backward(dfeat)
        torch.cuda.synchronize()
        grads[name] = en.arena.g.clone()
    lo, hi = eng.image_span[0], eng.groups["head"][0]
    cos = float(torch.nn.functional.cosine_similarity(grads["x3f"][lo:hi].double(), grads["bf16"][lo:hi].double(), dim=0))
    print("conv-backbone gradients for the same upstream gradient, bf16x3f vs bf16 mode: cosine", cos)
    assert cos > 0.999            # (fp32 atomics order + the bf16 rounding of the other mode's upstream activations)
    eng.release_buffers(); e16.release_buffers()
