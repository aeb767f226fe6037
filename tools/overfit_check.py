#!/usr/bin/env python
"""Sanity of the training dynamics of the whole HIP path (bf16, all fusions, packed captions, class-token tail): a fixed
batch of synthetic pairs must be memorised within a few dozen AdamW steps."""
import contextlib, io, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd.multimodal_model import MM_RCA
from garbage_classification_rca_amd.optim import FlatAdamW
from garbage_classification_rca_amd.procedural import synth_captions
from garbage_classification_rca_amd.training import FusedCrossEntropy, hip_train_step
from garbage_classification_rca_amd.engine import make_text_pack

B, S = 32, 64
dev = torch.device("cuda")
with contextlib.redirect_stdout(io.StringIO()):
    m = MM_RCA(4, 0.0, 0.0, 0.7, 256, "distilbert", B, True, False, False, image_model_name="transformer_B16",
               dtype=torch.bfloat16, device=dev, init_seed=0)
m.train(); m.enc_dropout = 0.0
for p in m.parameters():
    p.requires_grad = True
opt, crit = FlatAdamW(m, lr=1e-4, weight_decay=0.0), FusedCrossEntropy(None, 0.0)
ids, mask = synth_captions(B, S, seed=1)
pack = make_text_pack(mask, dev)
ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
images = torch.randn(B, 3, 224, 224, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
labels = (torch.arange(B, device=dev) % 4).to(torch.int32)
losses = []
with contextlib.redirect_stdout(io.StringIO()):
    for i in range(60):
        losses.append(float(hip_train_step(m, ids, mask, images, labels, crit, opt, None, text_pack=pack)))
print("loss every 10 steps:", [round(l, 4) for l in losses[::10]], "last", round(losses[-1], 4))
assert losses[-1] < 0.2 * losses[0], "the fixed batch was not memorised"
print("ok")
