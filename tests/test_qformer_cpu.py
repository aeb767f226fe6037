"""CPU tests of the BLIP-2 Q-Former path: the oracle against the transformers-generated fixture,
the product's parameter list against transformers' state_dict keys, and the reference loop's semantics."""
import copy
import os

import numpy as np
import pytest
import torch

from garbage_classification_rca_amd import q_former as QF
from garbage_classification_rca_amd.procedural import proc_tensor, checksum
from oracle import qformer as OQ

GOLD = os.path.join(os.path.dirname(__file__), "golden", "qformer_tiny.npz")


def tiny_spec():
    g = np.load(GOLD)
    cfg = dict(kv.split("=") for kv in g["cfg"])
    return QF.Blip2Spec(**{k: int(v) for k, v in cfg.items()})


def tiny_state(spec):
    sd = {k: proc_tensor(k, shp) for k, shp in QF.blip2_params(spec)}
    sd["query_tokens"] = sd["query_tokens"] * np.float32(20.0)          # as make_qformer_golden.py
    return sd


def oracle_cfg(spec):
    return dict(v_layers=spec.v_layers, v_heads=spec.v_heads, patch=spec.patch, q_layers=spec.q_layers, q_heads=spec.q_heads,
                cross_freq=spec.cross_freq, hidden_drop=spec.hidden_drop, attn_drop=spec.attn_drop)


def test_param_list_is_the_transformers_state_dict():
    """every frozen key the product allocates is a key of Blip2VisionModel / Blip2QFormerModel (+ query_tokens), and none is missing"""
    g = np.load(GOLD)
    spec = tiny_spec()
    ours = sorted(k for k, _ in QF.blip2_params(spec))
    assert ours == sorted(str(k) for k in g["keys"])
    sd = tiny_state(spec)
    assert abs(checksum([sd[k] for k in sorted(sd)[:8]]) - float(g["weight_checksum"])) < 1e-6 * max(1.0, abs(float(g["weight_checksum"])))


def test_default_spec_is_blip2_opt_2_7b():
    s = QF.BLIP2_OPT_2_7B
    assert (s.v_dim, s.v_layers, s.v_heads, s.v_mlp, s.patch, s.v_tokens) == (1408, 39, 16, 6144, 14, 257)
    assert (s.q_dim, s.q_layers, s.q_heads, s.q_mlp, s.n_query, s.cross_freq) == (768, 12, 12, 3072, 32, 2)
    n = sum(int(np.prod(shp)) for _, shp in QF.blip2_params(s))
    assert 1.08e9 < n < 1.10e9          # 986 M vision tower + 105 M Q-Former + query tokens


def test_oracle_matches_transformers_fixture():
    g = np.load(GOLD)
    spec = tiny_spec()
    sd = tiny_state(spec)
    px = torch.from_numpy(g["pixel_values"])
    emb = OQ.vision_forward(sd, px, layers=spec.v_layers, heads=spec.v_heads, patch=spec.patch)
    np.testing.assert_allclose(emb.numpy(), g["image_embeds"], rtol=0, atol=2e-5)
    hs = OQ.qformer_forward(sd, emb, layers=spec.q_layers, heads=spec.q_heads, cross_freq=spec.cross_freq)
    np.testing.assert_allclose(hs.numpy(), g["qformer_last_hidden_state"], rtol=0, atol=2e-5)


def test_oracle_matches_live_transformers():
    pytest.importorskip("transformers")
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_qformer_golden as MG
    vision, qformer, sd = MG.build()
    px = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        emb = vision(pixel_values=px).last_hidden_state
        hs = qformer(query_embeds=torch.from_numpy(sd["query_tokens"]).expand(2, -1, -1), encoder_hidden_states=emb,
                     encoder_attention_mask=torch.ones(emb.shape[:-1], dtype=torch.long)).last_hidden_state
    spec = tiny_spec()
    emb_o = OQ.vision_forward(sd, px, layers=spec.v_layers, heads=spec.v_heads, patch=spec.patch)
    hs_o = OQ.qformer_forward(sd, emb_o, layers=spec.q_layers, heads=spec.q_heads, cross_freq=spec.cross_freq)
    assert (emb_o - emb).abs().max().item() < 2e-5
    assert (hs_o - hs).abs().max().item() < 2e-5


def test_oracle_dropout_masks_rate_and_scale():
    m = OQ.keep_mask(OQ.site_seed(3, 2, 4), (64, 128), 0.1)
    vals = set(np.unique(m.numpy()).tolist())
    assert vals == {0.0, np.float32(1.0 / 0.9).item()}
    assert abs((m == 0).float().mean().item() - 0.1) < 0.02
    # distinct sites draw distinct masks
    assert not torch.equal(m, OQ.keep_mask(OQ.site_seed(3, 2, 5), (64, 128), 0.1))


def test_reference_loop_applies_only_the_last_micro_batch():
    """q_former_training.py:283 zeroes the gradient at the top of every iteration, so the step taken at iteration 8 sees the
    gradient of batch 8 alone (scaled 1/8); iterations 1..7 leave the classifier untouched."""
    torch.manual_seed(0)
    feats = [torch.randn(5, 16) for _ in range(9)]
    ys = [torch.randint(0, 4, (5, 1)) for _ in range(9)]
    lin = torch.nn.Linear(16, 4)
    ref = copy.deepcopy(lin)
    it = iter(feats)
    avg, losses = OQ.reference_loop(lambda px: next(it), lin, [(None, y) for y in ys], accumulation_steps=8)
    # expected: one AdamW step from batch index 7's gradient / 8, then (9 % 8 != 0) one more from batch index 8's
    opt = torch.optim.AdamW(ref.parameters(), lr=5e-4, eps=1e-5)
    for j in (7, 8):
        opt.zero_grad()
        (torch.nn.functional.cross_entropy(ref(feats[j]), ys[j].view(-1)) / 8).backward()
        opt.step()
    assert torch.allclose(ref.weight, lin.weight, atol=1e-7) and torch.allclose(ref.bias, lin.bias, atol=1e-7)
    assert len(losses) == 9 and abs(avg - sum(losses) / 8) < 1e-9            # :306 divides by the last index (8), not the count (9)


def test_engine_needs_the_hip_library_and_a_gpu():
    """the product path has no CPU fallback: building the engine on a CPU device must not silently compute anything"""
    from garbage_classification_rca_amd import lib as L
    if not os.path.exists(L.LIB_PATH):
        pytest.skip("libmmrca.so not built")
    eng = QF.Blip2QFormerEngine(tiny_spec(), dtype=torch.float32, device="cpu")
    with pytest.raises(L.MmrcaError):
        eng.forward(torch.zeros(1, 3, 56, 56))


def _make_folder(root, n_per_class=3, size=(50, 40)):
    from PIL import Image
    rng = np.random.default_rng(0)
    paths = []
    for cname in ("Blue", "Green", "Black", "TTR"):
        os.makedirs(os.path.join(root, cname), exist_ok=True)
        for i in range(n_per_class):
            fn = os.path.join(root, cname, f"plastic_bottle_{i}12.png")
            Image.fromarray(rng.integers(0, 256, size=(size[1], size[0], 3), dtype=np.uint8)).save(fn)
            paths.append(fn)
    return paths


def test_dataset_labels_collate_and_image_transform(tmp_path):
    """ImageCaptioningDataset / collate_fn of q_former_training.py:62-122 for the keys on the path, and the image half of the
    processor against transformers' BlipImageProcessor at blip2-opt-2.7b's settings"""
    from garbage_classification_rca_amd import q_former_training as QT
    paths = _make_folder(str(tmp_path))
    ds = QT.ImageCaptioningDataset(sorted(paths), image_size=56)
    assert len(ds) == 12 and ds.item_text(0) == "plastic bottle"                    # digits, extension, underscores removed (:72)
    batch = QT.collate_fn([ds[i] for i in range(len(ds))])
    assert batch["pixel_values"].shape == (12, 3, 56, 56) and batch["labels"].shape == (12, 1)
    by_dir = {p.split("/")[-2]: int(batch["labels"][i, 0]) for i, p in enumerate(sorted(paths))}
    assert by_dir == {"Blue": 0, "Green": 1, "Black": 2, "TTR": 3}                  # TTR -> Yellow -> 3 (:86-89, 259-263)
    pytest.importorskip("transformers")
    # (the default BlipImageProcessor backend needs torchvision, absent here; the PIL backend is the same pipeline)
    from transformers.models.blip.image_processing_pil_blip import BlipImageProcessorPil
    from PIL import Image
    proc = BlipImageProcessorPil(size={"height": 224, "width": 224})
    img = Image.open(sorted(paths)[0])
    ref = proc(images=img, return_tensors="pt")["pixel_values"][0]
    got = QT.Blip2ImageTransform(224)(img)
    assert got.shape == ref.shape and (got - ref).abs().max().item() < 0.05       # same PIL bicubic; 1/255 steps are 0.015 wide
    assert (got - ref).abs().mean().item() < 2e-3


def test_test_set_report_keeps_the_reference_quirks():
    """q_former_test_set.py:168-235: accuracy over a hard-coded 2000, classification_report fed (predictions, truth)"""
    from garbage_classification_rca_amd import q_former_test_set as QS

    class Fake:
        def eval(self):
            return self

        def forward(self, px):
            return torch.nn.functional.one_hot(px.view(-1).long(), 4).float()      # "pixel_values" carry the wanted prediction

    pred = torch.tensor([0, 0, 1, 2, 3, 3, 3, 1])
    true = torch.tensor([0, 1, 1, 2, 3, 0, 3, 2])
    loader = [{"pixel_values": pred[:5], "labels": true[:5].view(-1, 1)}, {"pixel_values": pred[5:], "labels": true[5:].view(-1, 1)}]
    acc2000, report, rd, conf, acc = QS.calculate_acc(Fake(), loader, "cpu", verbose=False)
    assert acc == 5 / 8 and acc2000 == 100 * 5 / 2000
    assert conf.sum() == 8 and conf[0, 3] == 1 and conf[1, 0] == 1              # rows = ground truth, columns = prediction
    # class id 3 (printed under the reference's fourth name, "Yellow"): 3 predicted, 2 true, 2 right.  With the arguments
    # swapped the report's "precision" is the true recall (2/2) and its "recall" the true precision (2/3)
    assert abs(rd["Yellow"]["precision"] - 1.0) < 1e-12 and abs(rd["Yellow"]["recall"] - 2 / 3) < 1e-12
    assert list(rd)[:4] == ["Black", "Blue", "Green", "Yellow"]
