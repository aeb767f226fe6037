#!/usr/bin/env python
"""Feasibility probe for VERDICT r3 #9: capture one MM-RCA train step (configs[0]: ShuffleNetV2 + DistilBERT, B = 4) in a HIP graph
and compare the replay with the eager step.   usage: graph_probe.py [image_model] [batch] [image_size]"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from garbage_classification_rca_amd import lib as L
from garbage_classification_rca_amd.multimodal_model import MM_RCA
from garbage_classification_rca_amd.optim import FlatSGD
from garbage_classification_rca_amd.procedural import synth_captions
from garbage_classification_rca_amd.training import FusedCrossEntropy, hip_train_step

image_model = sys.argv[1] if len(sys.argv) > 1 else "shuffle_net"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
size = int(sys.argv[3]) if len(sys.argv) > 3 else 224
dev = torch.device("cuda", 0)
L.load()
with contextlib.redirect_stdout(io.StringIO()):
    model = MM_RCA(4, 0.6, 0.0, 0.7, 256, "distilbert", B, True, False, False, image_model_name=image_model, dtype=torch.bfloat16,
                   device=dev, init_seed=0, image_size=size)
model.train()
for p in model.parameters():
    p.requires_grad = True
opt = FlatSGD(model, lr=1e-3, weight_decay=1e-2)
crit = FusedCrossEntropy(None, 0.0)
ids, mask_host = synth_captions(B, 64, seed=4321)
ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask_host).to(dev)
images = torch.randn(B, 3, size, size, device=dev)
labels = (torch.arange(B, device=dev) % 4).to(torch.int32)


def step():
    return hip_train_step(model, ids, mask, images, labels, crit, opt, None, text_pack=None)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    step()
print("eager ms/step:", round(timed(step, 30), 3), flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        loss = step()
except Exception as e:
    print("CAPTURE FAILED:", type(e).__name__, str(e)[:2000], flush=True)
    raise
torch.cuda.synchronize()
print("captured", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("loss after replay:", float(loss), flush=True)
print("graph ms/step:", round(timed(g.replay, 30), 3), flush=True)
