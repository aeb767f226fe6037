#!/usr/bin/env python
"""Input-path throughput: the GPU image pipeline (pinned staging + H2D + one kernel per batch)
against the per-sample DataLoader-worker form of the same pipeline (main_both.Transforms), on synthetic decoded images."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd.preprocess import GpuImagePipeline
from garbage_classification_rca_amd.main_both import Transforms     # the per-sample DataLoader-worker form of the same pipeline
from PIL import Image

B, H, W = 256, 384, 512
rng = np.random.RandomState(0)
imgs = [rng.randint(0, 256, size=(H - (i % 5) * 16, W - (i % 7) * 16, 3)).astype(np.uint8) for i in range(B)]
pipe = GpuImagePipeline(224, 224, max_batch=B, max_pixels=H * W)
for _ in range(3):
    out = pipe(imgs)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    out = pipe(imgs)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
# kernel alone (inputs already on the device): re-launch on the last slot
from garbage_classification_rca_amd import lib as L
import ctypes as C
k = (pipe._i - 1) & 1
e0.record()
for _ in range(20):
    L._check(L.load().mmrca_image_preprocess(L.ptr(pipe._dev[k]), L.ptr(pipe._dev_desc[k]), L.ptr(out), B, 224, 224,
                                             C.cast(pipe._mean, C.c_void_p), C.cast(pipe._std, C.c_void_p), L.stream_ptr()), "pre")
e1.record(); torch.cuda.synchronize()
kus = e0.elapsed_time(e1) / 20 * 1e3
src_mb = sum(i.size for i in imgs) / 1e6
print(f"GPU pipeline (host pack + pinned H2D + kernel): {B / dt:9.0f} images/s  ({dt * 1e3:.2f} ms per batch of {B}, {src_mb:.0f} MB of pixels)")
print(f"kernel alone: {kus:.1f} us per batch = {B / kus * 1e6:.0f} images/s, {(src_mb * 1e6 + B * 3 * 224 * 224 * 4) / kus / 1e6:.2f} TB/s (source read once + output)")
torch.set_num_threads(1)
tf = Transforms(224, 224)
pil = [Image.fromarray(i) for i in imgs[:32]]
t0 = time.perf_counter()
for im in pil:
    tf(im)
dc = (time.perf_counter() - t0) / 32
print(f"per-sample CPU form (main_both.Transforms), one core: {1 / dc:9.0f} images/s")
