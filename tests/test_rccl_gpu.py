"""The RCCL code path of the data-parallel exchange, executed on the one GPU a test box has: backend "nccl" (= RCCL on ROCm) with
world size 1.  Every multi-rank test of GradSync runs on gloo (CPU), which takes the SUM + div_ branch; this drives what an
8-GPU run takes -- ReduceOp.AVG, the bf16 wire, spans produced on two HIP streams, the device-bound process group -- through
the real library (reference seam: nn.DataParallel at main_both.py:386-388, replaced by one process per GPU).  No scaling claim:
one rank exchanges with itself."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    import torch.distributed as dist
    assert torch.cuda.is_available()
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("wire", [None, torch.bfloat16])
def test_gradsync_on_rccl_world1(nccl_world1, wire):
    """spans reported in the engine's order from two producer streams; AVG over one rank is the identity (fp32 wire: the arena is
    bit-unchanged; bf16 wire: every entry is its bf16 rounding), and the launch / byte counters say what was put on the wire"""
    from garbage_classification_rca_amd.distributed import GradSync
    dist = nccl_world1
    assert dist.get_backend() == "nccl"
    n = 6 << 20                                       # 24 MB of fp32 "gradients"
    g = torch.Generator(device="cuda").manual_seed(1)
    flat = torch.randn(n, device="cuda", generator=g)
    want = flat.clone() if wire is None else flat.to(torch.bfloat16).float()
    sync = GradSync(flat, world=1, bucket_bytes=8 << 20, wire_dtype=wire, force=True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cuts = [n, 5 << 20, 4 << 20, 3 << 20, 1 << 20, 0]          # descending spans, as the backward finishes them
    main = torch.cuda.current_stream()
    for i in range(len(cuts) - 1):
        st = s1 if i % 2 == 0 else s2
        st.wait_stream(main)
        with torch.cuda.stream(st):
            flat[cuts[i + 1]:cuts[i]].mul_(1.0)                 # "produce" the span on this stream
            sync.span_ready(cuts[i + 1], cuts[i], flush=(i == len(cuts) - 2))
    main.wait_stream(s1)
    main.wait_stream(s2)
    sync.finish()
    torch.cuda.synchronize()
    assert torch.equal(flat, want)
    assert not sync.pending
    assert sync.bytes_reduced == n * (4 if wire is None else 2)
    # 2 M-element (8 MB) buckets: [5M, 6M) + [4M, 5M) merge and flush at 2 M elements; [3M, 4M) + [1M, 3M) flush at 3 M; [0, 1M) on the final flush
    assert sync.launches == 3


def test_engine_backward_hands_every_group_to_rccl(nccl_world1):
    """one small MM-RCA step with the engine's grad_sync hooked to RCCL (world 1): every parameter group is reported exactly once,
    the spans tile the arena, and the averaged gradients equal the local ones"""
    from garbage_classification_rca_amd.distributed import GradSync
    from garbage_classification_rca_amd.engine import MMRCAEngine
    from garbage_classification_rca_amd import lib as L
    from garbage_classification_rca_amd.procedural import synth_captions
    B = 2
    eng = MMRCAEngine("distilbert", "transformer_B16", 4, True, 0, torch.bfloat16)
    eng.init_parameters(0)
    ids, mask = (torch.from_numpy(a).cuda() for a in synth_captions(B, 16, seed=3))
    images = torch.randn(B, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    dl = torch.full((B, 4), 0.1, device="cuda")

    def grads(sync):
        eng.grad_sync = sync
        eng.arena.g.zero_()
        eng.forward(ids, mask, images)
        eng.backward(dl)
        if sync is not None:
            sync.finish()
        torch.cuda.synchronize()
        return eng.arena.g.clone()
    local = grads(None)
    seen = []
    sync = GradSync(eng.arena.g, world=1, force=True)
    orig = sync.span_ready
    sync.span_ready = lambda lo, hi, flush=False: (seen.append((lo, hi)), orig(lo, hi, flush))[1]
    synced = grads(sync)
    eng.grad_sync = None
    assert sorted(seen) == sorted(eng.groups.values())          # every group once; together they tile the arena
    assert sync.bytes_reduced == eng.arena.total * 4 and sync.launches >= 3
    # AVG over one rank is the identity; two runs of the backward differ only by the summation order of the fp32 atomics behind
    # the LayerNorm / bias / embedding gradients
    diff = float((local - synced).abs().max())
    print("largest difference between the local and the RCCL-averaged gradients:", diff, "of", float(local.abs().max()))
    assert diff <= 1e-5 * float(local.abs().max())
    eng.release_buffers()


def test_conv_backbone_reports_its_stages_one_by_one(nccl_world1):
    """the conv image encoder used to be ONE span after its whole backward.  Its stages are now handed over as they
    finish (conv_engine.backward -> engine._ready), last stage first, and tile the image span of the arena."""
    from garbage_classification_rca_amd.distributed import GradSync
    from garbage_classification_rca_amd.engine import MMRCAEngine
    from garbage_classification_rca_amd.procedural import synth_captions
    B = 2
    eng = MMRCAEngine("distilbert", "shuffle_net", 4, True, 0, torch.bfloat16, image_size=64)
    eng.init_parameters(0)
    ids, mask = (torch.from_numpy(a).cuda() for a in synth_captions(B, 16, seed=3))
    images = torch.randn(B, 3, 64, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    seen = []
    sync = GradSync(eng.arena.g, world=1, force=True)
    orig = sync.span_ready
    sync.span_ready = lambda lo, hi, flush=False: (seen.append((lo, hi)), orig(lo, hi, flush))[1]
    eng.grad_sync = sync
    eng.forward(ids, mask, images, bn_train=True)
    eng.backward(torch.full((B, 4), 0.1, device="cuda"))
    sync.finish()
    torch.cuda.synchronize()
    eng.grad_sync = None
    img_lo, img_hi = eng.image_span[0], eng.groups["head"][0]
    img = [sp for sp in seen if img_lo <= sp[0] < img_hi]
    names = [k for k in eng.groups if k.startswith("image_")]
    assert len(img) == len(names) >= 5                           # conv5, stage4, stage3, stage2, stem
    assert [sp[0] for sp in img] == sorted((sp[0] for sp in img), reverse=True)      # last stage first
    assert sorted(img)[0][0] == img_lo and sorted(img)[-1][1] == img_hi
    assert all(a[1] == b[0] for a, b in zip(sorted(img), sorted(img)[1:]))
    eng.release_buffers()


def test_graphed_step_captures_the_rccl_exchange_and_replays_the_eager_step(nccl_world1):
    """training.GraphedTrainStep with a live exchange (GradSync(force=True) on the world-1 RCCL group): the bucketed all-reduces and
    the wait in front of the optimizer are captured INSIDE the step's HIP graph; replays compute what the eager step with the same
    exchange computes (same losses within the eager-vs-eager noise, same parameters), the exchange ran on every replay (a gradient
    poisoned before a replay comes back averaged = overwritten by the replay's backward, finite), and an accumulating micro-batch
    (do_step=False: no exchange) is a different graph.  Reference seam: run_one_epoch's per-batch body (main_both.py:95-126) under
    the data-parallel replacement of nn.DataParallel (:386-388)."""
    import contextlib, io
    from garbage_classification_rca_amd.distributed import GradSync
    from garbage_classification_rca_amd.multimodal_model import MM_RCA
    from garbage_classification_rca_amd.optim import FlatSGD
    from garbage_classification_rca_amd.training import FusedCrossEntropy, GraphedTrainStep, hip_train_step
    from garbage_classification_rca_amd.procedural import synth_captions
    B, n = 4, 6

    def model():
        with contextlib.redirect_stdout(io.StringIO()):
            m = MM_RCA(4, 0.6, 0.0, 0.7, 256, "distilbert", B, True, False, False, image_model_name="transformer_B16", dtype=torch.bfloat16,
                       device=torch.device("cuda", 0), init_seed=0)
        m.train()
        for p in m.parameters():
            p.requires_grad = True
        return m

    data = []
    for k in range(n):
        ids, mask = (torch.from_numpy(a).cuda() for a in synth_captions(B, 16, seed=50 + k))
        data.append((ids, mask, torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(60 + k)).cuda(), torch.arange(B).cuda() % 4))
    ma, mb = model(), model()
    oa, ob = FlatSGD(ma, lr=2e-3, weight_decay=1e-2), FlatSGD(mb, lr=2e-3, weight_decay=1e-2)
    crit = FusedCrossEntropy(None, 0.0)
    sa = GradSync(ma.engine.arena.g, world=1, bucket_bytes=32 << 20, force=True)
    sb = GradSync(mb.engine.arena.g, world=1, bucket_bytes=32 << 20, force=True)
    assert sb.active() and sb.capturable()
    graphed = GraphedTrainStep(mb, crit, ob, grad_sync=sb, warmup=2)
    la, lb = [], []
    with contextlib.redirect_stdout(io.StringIO()):
        for k in range(n):
            la.append(hip_train_step(ma, *data[k], crit, oa, sa, text_pack=None))
            lb.append(graphed(*data[k]))
    torch.cuda.synchronize()
    la, lb = [float(x) for x in la], [float(x) for x in lb]
    print("eager + RCCL  ", [round(x, 4) for x in la])
    print("graphed + RCCL", [round(x, 4) for x in lb])
    assert graphed.replays == n - 2 and len(graphed._graphs) == 1 and not sb.pending
    launches_captured = sb.launches
    assert launches_captured > 2 * 1 and sa.launches > sb.launches          # (the eager twin launches its collectives every step)
    assert max(abs(a - b) for a, b in zip(la, lb)) < 2e-2, (la, lb)
    rel = float((ma.engine.arena.p - mb.engine.arena.p).norm() / ma.engine.arena.p.norm())
    print("parameters after", n, "steps, graphed + RCCL vs eager + RCCL:", rel)
    assert rel < 2e-4
    # an accumulating micro-batch exchanges nothing and is its own graph; the stepping one that follows replays the first graph
    with contextlib.redirect_stdout(io.StringIO()):
        for k in range(3):
            graphed(*data[k], do_step=False)
        assert len(graphed._graphs) == 2 and sb.launches == launches_captured
        l_last = graphed(*data[0])
    torch.cuda.synchronize()
    assert torch.isfinite(l_last) and torch.isfinite(mb.engine.arena.p).all() and len(graphed._graphs) == 2
    ma.engine.release_buffers()
    mb.engine.release_buffers()
