"""GPU parity tests of the BLIP-2 Q-Former path: the K3x attention kernels against torch, the HIP
engine against oracle/qformer.py (pinned to transformers 5.15.0's Blip2VisionModel / Blip2QFormerModel), and the training
loop of q_former_training.py:274-309 against stock torch AdamW / CrossEntropyLoss."""
import numpy as np
import pytest
import torch

from garbage_classification_rca_amd import lib as L
from garbage_classification_rca_amd import q_former as QF
from garbage_classification_rca_amd.procedural import proc_tensor, proc_input, counter_uniform
from oracle import qformer as OQ

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _attn_ref(q, k, v, H, mask=None):
    B, Sq, D = q.shape
    return OQ.attention(q.float().cpu(), k.float().cpu(), v.float().cpu(), H, mask)


def _run_cross(q, k, v, B, H, Sq, Skv, dh, dtype, impl, drop_p=0.0, seed=0):
    out = torch.zeros(B * Sq, H * dh, dtype=dtype, device=DEV)
    L.mha_cross_fwd(q, q.stride(0), k, k.stride(0), v, v.stride(0), out, H * dh, B, H, Sq, Skv, dh, dh ** -0.5, L.dtype_code(dtype), impl,
                    drop_p=drop_p, drop_seed=seed)
    torch.cuda.synchronize()
    return out.float().cpu().view(B, Sq, H * dh)


@pytest.mark.parametrize("B,H,S,dh", [(2, 16, 257, 88), (1, 4, 17, 24), (3, 2, 32, 64), (2, 3, 100, 96), (1, 2, 288, 128)])
@pytest.mark.parametrize("dtype,impl", [(torch.bfloat16, L.IMPL_MFMA), (torch.bfloat16, L.IMPL_REF), (torch.float32, L.IMPL_AUTO)])
def test_cross_attention_self_fused_qkv(B, H, S, dh, dtype, impl):
    """the vision tower's use: q / k / v are the three column blocks of one fused [rows, 3*H*dh] buffer (head dim 88 at S=257)"""
    g = torch.Generator().manual_seed(S * 7 + dh)
    D = H * dh
    qkv = (torch.randn(B * S, 3 * D, generator=g) * 0.8).to(dtype).to(DEV)
    got = _run_cross(qkv, qkv[:, D:], qkv[:, 2 * D:], B, H, S, S, dh, dtype, impl)
    x = qkv.float().cpu().view(B, S, 3 * D)
    exp = _attn_ref(x[..., :D], x[..., D:2 * D], x[..., 2 * D:], H)
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    assert (got - exp).abs().max().item() < tol * max(1.0, exp.abs().max().item())


@pytest.mark.parametrize("B,H,Sq,Skv,dh", [(3, 12, 32, 257, 64), (2, 2, 8, 17, 32), (1, 3, 33, 50, 40), (2, 4, 70, 224, 64)])
@pytest.mark.parametrize("dtype,impl", [(torch.bfloat16, L.IMPL_MFMA), (torch.float32, L.IMPL_AUTO)])
def test_cross_attention_separate_operands_and_dropout(B, H, Sq, Skv, dh, dtype, impl):
    """the Q-Former's use: 32 queries over 257 image tokens, K | V fused in one buffer; the dropout mask is the documented
    counter hash over ((b*H+h)*Sq + i)*Skv + j"""
    g = torch.Generator().manual_seed(Sq * 13 + Skv)
    D = H * dh
    q = (torch.randn(B * Sq, D, generator=g) * 0.7).to(dtype).to(DEV)
    kv = (torch.randn(B * Skv, 2 * D, generator=g) * 0.7).to(dtype).to(DEV)
    qf, kvf = q.float().cpu().view(B, Sq, D), kv.float().cpu().view(B, Skv, 2 * D)
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    for p, seed in ((0.0, 0), (0.1, 12345), (0.5, 7)):
        got = _run_cross(q, kv, kv[:, D:], B, H, Sq, Skv, dh, dtype, impl, drop_p=p, seed=seed)
        mask = OQ.keep_mask(seed, (B * H, Sq, Skv), p) if p > 0 else None
        exp = _attn_ref(qf, kvf[..., :D], kvf[..., D:], H, mask)
        assert (got - exp).abs().max().item() < tol * max(1.0, exp.abs().max().item()), (p, seed)


@pytest.mark.parametrize("B,H,Sq,Skv,dh", [(2, 16, 257, 257, 88), (3, 12, 32, 257, 64), (2, 12, 32, 32, 64), (1, 3, 33, 50, 40), (1, 2, 50, 300, 128),
                                           (2, 2, 8, 17, 32), (1, 2, 384, 161, 96)])
def test_cross_attention_x3_matches_fp32_arithmetic(B, H, Sq, Skv, dh):
    """mmrca_mha_cross_fwd_x3 (fp32 operands, three-pass products on the bf16 matrix cores, keys walked in chunks with running
    softmax statistics, context as two bf16 planes) against float64 attention of the same fp32 operands: the error of fp32 arithmetic
    (~1e-6), not of bf16 (~1e-2).  The ViT-g shape (one, two and -- with 160-key chunks -- partially filled chunks), the Q-Former's
    self- and cross-attention with dropout, ragged tiles, head dims 32 .. 128, three query tiles per wave (S_q = 384)."""
    g = torch.Generator().manual_seed(Sq * 13 + Skv + dh)
    D = H * dh
    q = (torch.randn(B * Sq, D, generator=g) * 0.9).to(DEV)
    kv = (torch.randn(B * Skv, 2 * D, generator=g) * 0.9).to(DEV)
    qf, kvf = q.double().cpu().view(B, Sq, D), kv.double().cpu().view(B, Skv, 2 * D)
    for p, seed in ((0.0, 0), (0.1, 12345)):
        hi = torch.full((B * Sq, D), float("nan"), dtype=torch.bfloat16, device=DEV)
        lo = torch.full_like(hi, float("nan"))
        L.mha_cross_fwd_x3(q, D, kv, 2 * D, kv[:, D:], 2 * D, (hi, lo), D, B, H, Sq, Skv, dh, dh ** -0.5, drop_p=p, drop_seed=seed)
        torch.cuda.synchronize()
        got = (hi.float() + lo.float()).cpu().view(B, Sq, D).double()
        mask = OQ.keep_mask(seed, (B * H, Sq, Skv), p) if p > 0 else None
        exp = _attn_ref(qf, kvf[..., :D], kvf[..., D:], H, mask)
        err = (got - exp).abs().max().item() / max(1.0, exp.abs().max().item())
        assert err < 3e-5, (p, err)           # two bf16 planes carry 16 bits: 1.5e-5 of the largest entry is the output format's own floor


def test_cross_attention_rejects_bad_arguments():
    q = torch.zeros(64, 64, dtype=torch.float32, device=DEV)
    with pytest.raises(L.MmrcaError):       # fp32 cannot be forced onto the bf16 MFMA kernel
        L.mha_cross_fwd(q, 64, q, 64, q, 64, q, 64, 2, 1, 32, 32, 64, 0.125, L.F32, L.IMPL_MFMA)
    with pytest.raises(L.MmrcaError):       # row stride smaller than H*dh
        L.mha_cross_fwd(q, 32, q, 64, q, 64, q, 64, 2, 1, 32, 32, 64, 0.125, L.F32, L.IMPL_AUTO)
    with pytest.raises(L.MmrcaError):
        L.mha_cross_fwd(q, 64, q, 64, q, 64, q, 64, 2, 1, 32, 32, 64, 0.125, L.F32, L.IMPL_AUTO, drop_p=1.0)


def test_persistent_gemm_plain_gelu_epilogue():
    """FFN1 of the frozen towers: bias + GELU without the saved derivative, on the persistent 256x256 kernel (AUTO picks it
    at this shape) against the fp32 reference kernel"""
    M, N, K = 3072, 6144, 1408
    g = torch.Generator().manual_seed(11)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    outs = []
    for impl in (L.IMPL_AUTO, L.IMPL_MFMA256, L.IMPL_REF):
        c = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        L.gemm(a, w, c, bias=b, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, act=L.ACT_GELU, dtype=L.BF16, impl=impl)
        outs.append(c.float())
    exp = torch.nn.functional.gelu(a.float() @ w.float().t() + b.float())
    for c in outs:
        assert (c - exp).abs().max().item() < 2e-2 * exp.abs().max().item()
    assert torch.equal(outs[0], outs[1])          # AUTO is the 256x256 kernel here


# ------------------------------------------------------------------------------------------------------------------
# engine vs oracle
# ------------------------------------------------------------------------------------------------------------------
TINY = QF.Blip2Spec(v_dim=96, v_layers=2, v_heads=4, v_mlp=192, image_size=56, patch=14, q_dim=64, q_layers=3, q_heads=2,
                    q_mlp=128, cross_freq=2, n_query=8, n_classes=4)
# real widths (K = 1408, head dim 88, 257 tokens, 32 queries x 257 keys) at two layers each: the shapes the full model runs
WIDE = QF.Blip2Spec(v_layers=2, q_layers=2)


def _state(spec):
    sd = {k: proc_tensor(k, shp) for k, shp in QF.blip2_params(spec)}
    sd["query_tokens"] = sd["query_tokens"] * np.float32(20.0)
    cls = {"classifier.weight": proc_tensor("classifier.weight", (spec.n_classes, spec.q_dim)) * np.float32(4.0),
           "classifier.bias": proc_tensor("classifier.bias", (spec.n_classes,))}
    return sd, cls


def _cfg(spec):
    return dict(v_layers=spec.v_layers, v_heads=spec.v_heads, patch=spec.patch, q_layers=spec.q_layers, q_heads=spec.q_heads,
                cross_freq=spec.cross_freq, hidden_drop=spec.hidden_drop, attn_drop=spec.attn_drop)


def _engine(spec, dtype, sd, cls):
    eng = QF.Blip2QFormerEngine(spec, dtype=dtype, device=DEV)
    eng.load_state_dict(sd, cls)
    return eng


@pytest.mark.parametrize("spec,B", [(TINY, 3), (WIDE, 2)], ids=["tiny", "wide"])
def test_engine_eval_logits_match_oracle(spec, B):
    sd, cls = _state(spec)
    px = proc_input("qf_px_%d" % spec.v_dim, (B, 3, spec.image_size, spec.image_size))
    exp, hs = OQ.forward_logits(sd, cls, torch.from_numpy(px), _cfg(spec), train=False)
    scale = exp.abs().max().item()
    eng = _engine(spec, torch.float32, sd, cls).eval()
    got = eng.forward(torch.from_numpy(px).to(DEV)).cpu()
    rel = (got - exp).abs().max().item() / scale
    assert rel < 1e-3, rel                      # north_star: logits within 1e-3 relative of the reference
    eng16 = _engine(spec, torch.bfloat16, sd, cls).eval()
    got16 = eng16.forward(torch.from_numpy(px).to(DEV)).cpu()
    rel16 = (got16 - exp).abs().max().item() / scale
    assert rel16 < 5e-2, rel16                  # bf16 storage of 1408-wide activations through 2 + 2 layers
    assert (got16.argmax(1) == exp.argmax(1)).all() or rel16 < 1e-2


def test_engine_bf16x3f_mode_matches_oracle_in_eval_and_train_mode():
    """the compliant mode of configs[4] (fp32 values as two bf16 planes, three matrix-core passes per product in every nn.Linear and
    in both attention products): real widths (K = 1408, head dim 88, 257 tokens, 32 queries x 257 keys) at two layers each, eval
    mode and train mode (the Q-Former's dropout sites with host-rebuilt masks) against the fp32 oracle -- fp32-grade agreement"""
    spec, B = WIDE, 2
    sd, cls = _state(spec)
    px = proc_input("qf_px_%d" % spec.v_dim, (B, 3, spec.image_size, spec.image_size))
    eng = _engine(spec, "bf16x3f", sd, cls).eval()
    assert eng.x3 and eng.mode == "bf16x3f" and eng.dtype == torch.float32
    exp, _ = OQ.forward_logits(sd, cls, torch.from_numpy(px), _cfg(spec), train=False)
    got = eng.forward(torch.from_numpy(px).to(DEV)).cpu()
    rel = (got - exp).abs().max().item() / exp.abs().max().item()
    print("bf16x3f Q-Former path, eval, 2 + 2 layers at real widths: logits vs oracle", rel)
    assert rel < 1e-4, rel
    eng.train()
    got_t = eng.forward(torch.from_numpy(px).to(DEV), drop_seed=77).cpu()
    exp_t, _ = OQ.forward_logits(sd, cls, torch.from_numpy(px), _cfg(spec), train=True, drop_seed=77)
    rel_t = (got_t - exp_t).abs().max().item() / exp_t.abs().max().item()
    assert rel_t < 1e-4, rel_t
    assert (exp_t - exp).abs().max().item() > 1e-3 * exp.abs().max().item()       # the masks do act
    eng.release_buffers()


def test_engine_train_mode_dropout_matches_oracle_masks():
    """train mode (q_former_training.py:276): the Q-Former's five dropout sites per layer + the embedding dropout, with the
    masks rebuilt on the host from the counter hash"""
    spec = TINY
    sd, cls = _state(spec)
    px = proc_input("qf_px_train", (4, 3, spec.image_size, spec.image_size))
    eng = _engine(spec, torch.float32, sd, cls).train()
    got = eng.forward(torch.from_numpy(px).to(DEV), drop_seed=77).cpu()
    exp, _ = OQ.forward_logits(sd, cls, torch.from_numpy(px), _cfg(spec), train=True, drop_seed=77)
    assert (got - exp).abs().max().item() < 1e-3 * exp.abs().max().item()
    ev, _ = OQ.forward_logits(sd, cls, torch.from_numpy(px), _cfg(spec), train=False)
    assert (exp - ev).abs().max().item() > 1e-2 * ev.abs().max().item()          # the masks do act
    # a new forward pass draws new masks
    a = eng.forward(torch.from_numpy(px).to(DEV)).cpu()
    b = eng.forward(torch.from_numpy(px).to(DEV)).cpu()
    assert not torch.equal(a, b)


def test_training_loop_matches_reference_loop():
    """nine iterations of q_former_training.py:279-309 (accumulation 8: one step at iteration 8, one for the remainder):
    classifier weights, per-iteration losses and the epoch's avg_loss against torch.optim.AdamW / CrossEntropyLoss"""
    spec = TINY
    sd, cls = _state(spec)
    n_it, B = 9, 5
    rng = np.random.default_rng(3)
    batches = [{"pixel_values": torch.from_numpy(proc_input("qf_loop_%d" % i, (B, 3, spec.image_size, spec.image_size))),
                "labels": torch.from_numpy(rng.integers(0, 4, size=(B, 1)).astype(np.int64))} for i in range(n_it)]
    eng = _engine(spec, torch.float32, sd, cls)
    opt = QF.ClassifierAdamW(eng)
    avg = QF.run_one_epoch(eng, opt, batches, DEV)
    lin = torch.nn.Linear(spec.q_dim, spec.n_classes)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(cls["classifier.weight"])); lin.bias.copy_(torch.from_numpy(cls["classifier.bias"]))
    count = [0]

    def feats(px):        # the product numbers its forward passes 1, 2, ... and seeds the dropout sites with that count
        count[0] += 1
        _, hs = OQ.forward_logits(sd, cls, px, _cfg(spec), train=True, drop_seed=count[0])
        return hs[:, 0, :]

    exp_avg, exp_losses = OQ.reference_loop(feats, lin, [(b["pixel_values"], b["labels"]) for b in batches])
    got = eng.classifier_state_dict()
    assert (got["classifier.weight"] - lin.weight.detach()).abs().max().item() < 2e-5
    assert (got["classifier.bias"] - lin.bias.detach()).abs().max().item() < 2e-5
    assert (got["classifier.weight"] - torch.from_numpy(cls["classifier.weight"])).abs().max().item() > 5e-4     # two steps were taken
    assert abs(avg - exp_avg) < 1e-4 * max(1.0, abs(exp_avg))
    acc = QF.calculate_acc(eng, batches, DEV)
    assert 0.0 <= acc <= 1.0 and not eng.training


def test_load_state_dict_accepts_peft_prefix_and_split_qkv_bias():
    spec = TINY
    sd, cls = _state(spec)
    old = {}
    for k, v in sd.items():
        if k.endswith("self_attn.qkv.bias"):
            D = spec.v_dim
            v = v.copy(); v[D:2 * D] = 0
            sd[k] = v
            old["base_model.model." + k[:-len("qkv.bias")] + "q_bias"] = v[:D]
            old["base_model.model." + k[:-len("qkv.bias")] + "v_bias"] = v[2 * D:]
        else:
            old["base_model.model." + k] = v
    old["base_model.model.language_model.model.decoder.layers.0.self_attn.q_proj.lora_A.default.weight"] = np.zeros((32, 8), np.float32)
    a, b = _engine(spec, torch.float32, sd, cls), _engine(spec, torch.float32, old, cls)
    assert torch.equal(a.store.w, b.store.w)
    with pytest.raises(KeyError):
        QF.Blip2QFormerEngine(spec, dtype=torch.float32, device=DEV).load_state_dict({"query_tokens": sd["query_tokens"]})


def test_q_former_training_script_end_to_end(tmp_path):
    """folder -> q_former_training.main (two epochs, tiny widths) -> Classifier_epoch_*.pth -> reload and score"""
    import glob
    import os
    from PIL import Image
    from garbage_classification_rca_amd import q_former_training as QT
    rng = np.random.default_rng(1)
    for split in ("Train", "Val"):
        for c, cname in enumerate(("Blue", "Green", "Black", "TTR")):
            os.makedirs(tmp_path / split / cname, exist_ok=True)
            for i in range(5 if split == "Train" else 2):
                a = rng.integers(0, 80, size=(44, 60, 3)).astype(np.uint8)
                a[..., c % 3] += np.uint8(120 + 20 * c)                      # a learnable colour cue per class
                Image.fromarray(a).save(tmp_path / split / cname / f"item_{i}7.png")
    hist = QT.main(["--dataset_folder_name", str(tmp_path / "Train"), "--dataset_folder_name_val", str(tmp_path / "Val"),
                    "--batch_size", "4", "--epochs", "2", "--num_workers", "0", "--dtype", "fp32"], spec=TINY, out_dir=str(tmp_path))
    assert len(hist) == 2 and all(np.isfinite(h["train_loss_avg"]) for h in hist)
    files = glob.glob(str(tmp_path / "Classifier_epoch_*_acc_*.pth"))
    assert files, "the first epoch always beats max_val_accuracy = 0 unless accuracy is exactly 0"
    sd = torch.load(files[0])
    assert set(sd) == {"classifier.weight", "classifier.bias"} and sd["classifier.weight"].shape == (4, TINY.q_dim)
    eng = QF.Blip2QFormerEngine(TINY, dtype=torch.float32, device=DEV)
    eng.init_parameters(seed=0)
    eng.load_state_dict({}, sd, strict=False)
    ds = QT.ImageCaptioningDataset(sorted(glob.glob(str(tmp_path / "Val") + "/*/*")), image_size=TINY.image_size)
    batches = [QT.collate_fn([ds[i] for i in range(len(ds))])]
    acc = QF.calculate_acc(eng, batches, DEV)
    assert 0.0 <= acc <= 1.0
    # the test-split script on the same checkpoint: the reference's report files and its accuracy formula (correct / 2000)
    from garbage_classification_rca_amd import q_former_test_set as QS
    res = QS.main(["--dataset_folder_name", str(tmp_path / "Val"), "--classifier_weights", files[0], "--num_workers", "0",
                   "--dtype", "fp32"], spec=TINY, out_dir=str(tmp_path))
    assert abs(res["accuracy"] - acc) < 1e-9                                   # same weights (seed-0 towers), same images
    assert abs(res["test_accuracy_reference_formula"] - 100 * acc * 8 / 2000) < 1e-9
    assert os.path.exists(res["report_csv"]) and int(np.asarray(res["confusion_matrix"]).sum()) == 8


def test_two_rank_rehearsal_of_the_qformer_bench_keeps_classifiers_identical():
    """N>1 path of `bench.py --workload qformer`: two ranks share this one GPU (gloo standing in for RCCL), each runs its own
    images through the frozen towers, the 3,076 classifier gradients are all-reduced before the optimizer step"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MMRCA_DIST_BACKEND="gloo", MMRCA_CHECK_REPLICAS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--workload", "qformer", "--gpus", "2", "--batch", "4", "--steps", "8",
           "--warmup", "0", "--no_cpu_baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "classifier replicas identical after" in r.stderr and " 0 optimizer steps" not in r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak"
