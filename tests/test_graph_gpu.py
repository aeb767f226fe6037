"""The small-batch path: a train step captured in a HIP graph (training.GraphedTrainStep) against the eager step it replaces
(reference seam: the per-batch body of run_one_epoch, main_both.py:95-126 -- the reference launches every op from Python; a graph
replay must compute what those launches compute, including fresh dropout masks per step)."""
import contextlib
import io

import pytest
import numpy as np
import torch

pytestmark = pytest.mark.gpu


def _model(image_model, B, size, seed=0, dtype=torch.bfloat16):
    from garbage_classification_rca_amd.multimodal_model import MM_RCA
    with contextlib.redirect_stdout(io.StringIO()):
        m = MM_RCA(4, 0.6, 0.0, 0.7, 256, "distilbert", B, True, False, False, image_model_name=image_model, dtype=dtype,
                   device=torch.device("cuda", 0), init_seed=seed, image_size=size)
    m.train()
    for p in m.parameters():
        p.requires_grad = True
    return m


def _batches(n, B, size, T=16):
    from garbage_classification_rca_amd.procedural import synth_captions
    out = []
    for k in range(n):
        ids, mask = (torch.from_numpy(a).cuda() for a in synth_captions(B, T, seed=100 + k))
        g = torch.Generator(device="cuda").manual_seed(200 + k)
        out.append((ids, mask, torch.randn(B, 3, size, size, device="cuda", generator=g), (torch.arange(B, device="cuda") % 4).to(torch.int32)))
    return out


def test_mask_epoch_equals_a_seed_advanced_by_the_step_stride():
    """every dropout site of the step (embedding / attention probabilities / hidden states in rowops + attention kernels, the head's
    feature dropout): forward with (seed 7, epoch 3) == forward with (seed 10, epoch 0), bit for bit; and epoch 3 != epoch 0"""
    from garbage_classification_rca_amd import lib as L
    from garbage_classification_rca_amd.engine import MMRCAEngine
    eng = MMRCAEngine("distilbert", "transformer_B16", 4, True, 0, torch.bfloat16)
    eng.init_parameters(0)
    ids, mask, images, _ = _batches(1, 2, 224)[0]
    try:
        base = eng.forward(ids, mask, images, drop_p=0.6, seed=7, enc_drop_p=0.1, save=False).clone()
        L.seed_epoch_set(3)
        shifted = eng.forward(ids, mask, images, drop_p=0.6, seed=7, enc_drop_p=0.1, save=False).clone()
        ctr = torch.full((1,), 3, dtype=torch.int64, device="cuda")
        L.seed_epoch_set(device_value=ctr)
        from_hbm = eng.forward(ids, mask, images, drop_p=0.6, seed=7, enc_drop_p=0.1, save=False).clone()
    finally:
        L.seed_epoch_set(0)
    want = eng.forward(ids, mask, images, drop_p=0.6, seed=10, enc_drop_p=0.1, save=False).clone()
    torch.cuda.synchronize()
    assert torch.equal(shifted, want) and torch.equal(from_hbm, want)
    assert not torch.equal(base, want)
    eng.release_buffers()


@pytest.mark.parametrize("image_model,size,dtype", [("shuffle_net", 224, torch.bfloat16), ("transformer_B16", 224, torch.bfloat16),
                                                   ("transformer_B16", 224, "bf16x3f"), ("eff_v2_medium", 128, torch.bfloat16),
                                                   ("eff_v2_medium", 128, "bf16x3f"), ("eff_v2_medium+side", 128, "bf16x3f")])
def test_graphed_step_computes_the_eager_step(image_model, size, dtype, monkeypatch):
    """two models from the same seed, the same six batches: eager hip_train_step vs GraphedTrainStep (2 eager calls, 1 capture, 3
    replays).  Same losses step by step -- which requires every replay to draw the masks of ITS step (feature dropout 0.6: frozen
    masks move the loss by tenths) -- same parameters at the end, same host-side step counters; then one more EAGER step on both
    (the graph leaves the mask epoch at 0)."""
    from garbage_classification_rca_amd.optim import FlatSGD
    from garbage_classification_rca_amd.training import FusedCrossEntropy, GraphedTrainStep, hip_train_step
    from garbage_classification_rca_amd import lib as L
    B, n = 4, 6
    data = _batches(n + 1, B, size)
    if image_model.endswith("+side"):        # the conv weight gradients on their side stream: the capture records the fork / join as graph edges
        from garbage_classification_rca_amd import conv_engine as CE
        monkeypatch.setattr(CE, "SIDE_WGRAD", True)
        image_model = image_model[:-5]
    ma, mb, mc = (_model(image_model, B, size, dtype=dtype) for _ in range(3))
    # (stochastic depth of the EfficientNetV2 blocks is drawn, not injected: its keep masks are counter-based draws of the step seed like
    # the dropout masks -- round 5, mmrca_sd_rowscale -- so the replay of step s and the eager step s keep the same blocks)
    oa, ob, oc = (FlatSGD(m, lr=2e-3, weight_decay=1e-2) for m in (ma, mb, mc))
    crit = FusedCrossEntropy(None, 0.0)
    graphed = GraphedTrainStep(mb, crit, ob, warmup=2)
    la, lb, lc = [], [], []
    with contextlib.redirect_stdout(io.StringIO()):
        for k in range(n):
            la.append(hip_train_step(ma, *data[k], crit, oa, None, text_pack=None))
            lb.append(graphed(*data[k]))
            lc.append(hip_train_step(mc, *data[k], crit, oc, None, text_pack=None))       # the eager twin: the yardstick of run-to-run noise
        la.append(hip_train_step(ma, *data[n], crit, oa, None, text_pack=None))
        lb.append(hip_train_step(mb, *data[n], crit, ob, None, text_pack=None))
        lc.append(hip_train_step(mc, *data[n], crit, oc, None, text_pack=None))
    torch.cuda.synchronize()
    la, lb, lc = [float(x) for x in la], [float(x) for x in lb], [float(x) for x in lc]
    print("eager  ", [round(x, 4) for x in la])
    print("graphed", [round(x, 4) for x in lb])
    print("eager 2", [round(x, 4) for x in lc])
    assert graphed.replays == n - 2 and len(graphed._graphs) == 1
    assert ma._fwd_count == mb._fwd_count == n + 1
    assert L.seed_epoch_host() == 0
    # two EAGER runs already differ -- the fp32 atomics behind the BatchNorm / LayerNorm / bias sums land in another order, and train-mode
    # BatchNorm at B = 4 amplifies it from step to step: the graphed run must sit within that noise of the eager one (feature dropout
    # 0.6: frozen or shifted masks move the loss by tenths, see the control below)
    dev, noise = max(abs(a - b) for a, b in zip(la, lb)), max(abs(a - c) for a, c in zip(la, lc))
    print("largest loss difference: graphed vs eager", dev, " eager vs eager", noise)
    assert dev <= 3.0 * noise + 2e-3, (la, lb, lc)
    assert len({round(x, 3) for x in lb}) > 3                      # (the losses do move from step to step)
    pa, pb, pc = ma.engine.arena.p, mb.engine.arena.p, mc.engine.arena.p
    rel, noise = float((pa - pb).norm() / pa.norm()), float((pa - pc).norm() / pa.norm())
    print("relative distance of the parameters after", n + 1, "steps: graphed vs eager", rel, " eager vs eager", noise)
    assert rel <= 3.0 * noise + 2e-5
    if ma.engine.conv is not None:
        assert ma.engine.conv.n_train_forwards == mb.engine.conv.n_train_forwards == n + 1
        run = lambda m: torch.cat([t.flatten() for k, t in sorted(m.engine.conv.buffers.items()) if k.endswith(("running_mean", "running_var"))])
        ra, rb, rc = run(ma), run(mb), run(mc)
        rel, noise = float((ra - rb).norm() / ra.norm()), float((ra - rc).norm() / ra.norm())
        print("running statistics: graphed vs eager", rel, " eager vs eager", noise)
        assert rel <= 3.0 * noise + 1e-5
    for m in (ma, mb, mc):
        m.engine.release_buffers()


def test_frozen_masks_would_be_caught():
    """the control of the test above: the same comparison with the mask epoch NOT advanced (the captured step's seeds replayed) fails
    the tolerance -- so the agreement above is evidence that replays draw fresh masks"""
    from garbage_classification_rca_amd.optim import FlatSGD
    from garbage_classification_rca_amd.training import FusedCrossEntropy, GraphedTrainStep, hip_train_step
    B, n, size = 4, 6, 224
    data = _batches(n, B, size)
    ma, mb = _model("transformer_B16", B, size), _model("transformer_B16", B, size)
    oa, ob = FlatSGD(ma, lr=1e-2, weight_decay=1e-2), FlatSGD(mb, lr=1e-2, weight_decay=1e-2)
    crit = FusedCrossEntropy(None, 0.0)
    graphed = GraphedTrainStep(mb, crit, ob, warmup=2)
    la, lb = [], []
    with contextlib.redirect_stdout(io.StringIO()):
        for k in range(n):
            la.append(hip_train_step(ma, *data[k], crit, oa, None, text_pack=None))
            if k >= 3:
                mb._fwd_count = 2          # rewind the step counter: every replay then asks for epoch 0 = the captured step's masks
            lb.append(graphed(*data[k]))
    torch.cuda.synchronize()
    la, lb = [float(x) for x in la], [float(x) for x in lb]
    assert max(abs(a - b) for a, b in zip(la[3:], lb[3:])) > 2e-2, (la, lb)
    ma.engine.release_buffers()
    mb.engine.release_buffers()


def test_graphed_step_draws_stochastic_depth_inside_the_graph():
    """EfficientNetV2 under capture: the keep masks are drawn by a captured launch from (step seed, mask epoch) -- the step captures,
    replays, and keeps producing finite, moving losses; and the masks of replay r are the masks the host mirror of the hash gives for
    eager step r"""
    from garbage_classification_rca_amd.optim import FlatSGD
    from garbage_classification_rca_amd.training import FusedCrossEntropy, GraphedTrainStep
    B, size = 4, 128
    m = _model("eff_v2_medium", B, size)
    graphed = GraphedTrainStep(m, FusedCrossEntropy(None, 0.0), FlatSGD(m, lr=1e-3, weight_decay=1e-2), warmup=1)
    data = _batches(2, B, size)
    with contextlib.redirect_stdout(io.StringIO()):
        losses = [float(graphed(*data[k % 2])) for k in range(6)]
    assert graphed.replays == 5 and all(l == l and abs(l) < 1e3 for l in losses), losses
    assert len({round(l, 4) for l in losses}) >= 4, losses
    from garbage_classification_rca_amd.procedural import counter_uniform
    conv, eng = m.engine.conv, m.engine
    n_sd = int(conv._sd_p.numel())
    got = conv._bufs[("sd.rowscale", n_sd, B, torch.float32)][:n_sd].cpu().numpy()          # the masks of the LAST replay (step 6)
    p = conv._sd_p.cpu().numpy()
    u = counter_uniform(eng._site_seed(m._drop_seed + m._fwd_count, 254, 0), np.arange(n_sd * B, dtype=np.uint64)).reshape(n_sd, B)
    want = np.where(u >= p, 1.0 / (1.0 - p), 0.0).astype(np.float32)
    assert m._fwd_count == 6 and np.array_equal(got, want) and 0 < (got == 0).sum() < got.size
    # the engine frees its buffers (a new image size, an evaluation at another batch size ...): the graphs that point into them are
    # dropped and the step warms up and captures again
    m.engine.release_buffers()
    with contextlib.redirect_stdout(io.StringIO()):
        more = [float(graphed(*data[k % 2])) for k in range(3)]
    assert graphed.replays == 7 and len(graphed._graphs) == 1 and all(l == l for l in more), more
    m.engine.release_buffers()


def test_a_capture_error_surfaced_by_a_launch_check_falls_back_to_the_eager_step(monkeypatch):
    """a capture-illegal call that surfaces through MMRCA_CHECK_LAUNCH arrives as an MmrcaError whose message is the HIP
    capture error; GraphedTrainStep must treat it like torch's own capture failure -- roll the host state back, keep that batch shape
    eager, and compute the same step -- while any OTHER library error still raises."""
    from garbage_classification_rca_amd import training as TR
    from garbage_classification_rca_amd import lib as L
    from garbage_classification_rca_amd.optim import FlatSGD
    B, size, n = 4, 224, 4
    data = _batches(n, B, size)
    ma, mb = (_model("transformer_B16", B, size) for _ in range(2))
    oa, ob = (FlatSGD(m, lr=2e-3, weight_decay=1e-2) for m in (ma, mb))
    crit = TR.FusedCrossEntropy(None, 0.0)
    graphed = TR.GraphedTrainStep(mb, crit, ob, warmup=1)
    real = TR._enqueue_step
    message = ["mmrca_gemm: launch failed: operation not permitted when stream is capturing"]

    def flaky(*a, **k):
        if torch.cuda.is_current_stream_capturing():
            raise L.MmrcaError(message[0])
        return real(*a, **k)

    monkeypatch.setattr(TR, "_enqueue_step", flaky)
    la, lb = [], []
    with contextlib.redirect_stdout(io.StringIO()):
        for k in range(n):
            la.append(float(TR.hip_train_step(ma, *data[k], crit, oa, None, text_pack=None)))
            lb.append(float(graphed(*data[k])))
    assert graphed.replays == 0 and len(graphed._graphs) == 0 and len(graphed._no_graph) == 1
    assert ma._fwd_count == mb._fwd_count == n and L.seed_epoch_host() == 0
    assert max(abs(a - b) for a, b in zip(la, lb)) < 5e-3, (la, lb)
    # any other library error is an error
    graphed2 = TR.GraphedTrainStep(mb, crit, ob, warmup=0)
    message[0] = "mmrca_gemm: K must be a multiple of 8"
    with pytest.raises(L.MmrcaError, match="multiple of 8"), contextlib.redirect_stdout(io.StringIO()):
        graphed2(*data[0])
    assert not torch.cuda.is_current_stream_capturing()
    for m in (ma, mb):
        m.engine.release_buffers()
