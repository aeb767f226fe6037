"""GPU side of the input path: the reference's validation image pipeline
(main_both.py:433-440) and the V/H flips of its train pipeline (:416-417) as ONE kernel launch per batch.

    decoded uint8 HWC images (any sizes)  --pinned staging + one async H2D-->  mmrca_image_preprocess  -->  fp32 [B,3,H,W]

``plan_padding`` restates ``PadToMaintainAR.apply`` (keep_aspect_ratio.py:24-50) as numbers instead of a padded copy --
including its axis naming: ``img.shape`` is (H, W, C) but the transform calls size[0] "width", so it compares H/W with
the target ratio and pads the axis that makes a non-matching image LONGER (pinned by tests/golden/pad_goldens.npz).
The remaining albumentations augmentations (Rotate, GaussianBlur, BrightnessContrast, Sharpen, Perspective,
ShiftScaleRotate) are not built.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as L

MEAN = (0.485, 0.456, 0.406)       # main_both.py:400
STD = (0.229, 0.224, 0.225)        # main_both.py:401

DESC_DTYPE = np.dtype([("offset", np.int64), ("h", np.int32), ("w", np.int32), ("pad_top", np.int32), ("pad_left", np.int32),
                       ("ph", np.int32), ("pw", np.int32), ("flip_v", np.int32), ("flip_h", np.int32)])
assert DESC_DTYPE.itemsize == 40       # == sizeof(MmrcaImageDesc)


def plan_padding(h: int, w: int, aspect_ratio: float) -> Tuple[int, int, int, int]:
    """(pad_top, pad_left, padded_h, padded_w) of PadToMaintainAR(aspect_ratio) for an (h, w) image."""
    current = h / w
    if current == aspect_ratio:
        return 0, 0, h, w
    if current < aspect_ratio:
        half = int((int(aspect_ratio * w) - h) / 2)
        return 0, half, h, w + 2 * half
    half = int((int(h / aspect_ratio) - w) / 2)
    return half, 0, h + 2 * half, w


class GpuImagePipeline:
    """Reusable staging for batches of up to ``max_batch`` images of at most ``max_pixels`` pixels each.  Two pinned host
    buffers and two device buffers alternate, the copy runs on its own stream, so batch i+1 can be staged and copied
    while batch i is being consumed (``__call__`` returns a tensor that is ordered on the caller's current stream)."""

    def __init__(self, out_h: int, out_w: int, max_batch: int, max_pixels: int, device="cuda"):
        self.out_h, self.out_w, self.max_batch = int(out_h), int(out_w), int(max_batch)
        self.device = torch.device(device)
        cap = int(max_batch) * int(max_pixels) * 3
        self._host = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._host_desc = [torch.empty(max_batch * DESC_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._dev = [torch.empty(cap, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self._dev_desc = [torch.empty(max_batch * DESC_DTYPE.itemsize, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._slot_free = [None, None]          # event: the kernel that consumed this slot has finished
        self._i = 0
        self._mean = (C.c_float * 3)(*MEAN)
        self._std = (C.c_float * 3)(*STD)

    def __call__(self, images: Sequence[np.ndarray], flips: Optional[Sequence[Tuple[bool, bool]]] = None) -> torch.Tensor:
        B = len(images)
        if B == 0 or B > self.max_batch:
            raise L.MmrcaError(f"GpuImagePipeline: batch of {B} images (capacity {self.max_batch})")
        k = self._i & 1
        self._i += 1
        if self._slot_free[k] is not None:
            self._slot_free[k].synchronize()               # the pinned buffer may still be read by the previous copy / kernel
        host = self._host[k].numpy()
        desc = np.zeros(B, dtype=DESC_DTYPE)
        ar = self.out_w / self.out_h
        off = 0
        for b, img in enumerate(images):
            if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
                raise L.MmrcaError("GpuImagePipeline: images must be uint8 HWC with 3 channels")
            h, w = img.shape[:2]
            n = h * w * 3
            if off + n > host.size:
                raise L.MmrcaError("GpuImagePipeline: staging buffer too small for this batch")
            host[off:off + n] = np.ascontiguousarray(img).reshape(-1)
            pt, pl, ph, pw = plan_padding(h, w, ar)
            fv, fh = (flips[b] if flips is not None else (False, False))
            desc[b] = (off, h, w, pt, pl, ph, pw, int(fv), int(fh))
            off += (n + 15) // 16 * 16
        self._host_desc[k].numpy()[:B * DESC_DTYPE.itemsize] = desc.view(np.uint8).reshape(-1)
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._copy_stream):
            self._dev[k][:off].copy_(self._host[k][:off], non_blocking=True)
            self._dev_desc[k][:B * DESC_DTYPE.itemsize].copy_(self._host_desc[k][:B * DESC_DTYPE.itemsize], non_blocking=True)
        main.wait_stream(self._copy_stream)
        out = torch.empty(B, 3, self.out_h, self.out_w, dtype=torch.float32, device=self.device)
        L._check(L.load().mmrca_image_preprocess(L.ptr(self._dev[k]), L.ptr(self._dev_desc[k]), L.ptr(out), B, self.out_h, self.out_w,
                                                 C.cast(self._mean, C.c_void_p), C.cast(self._std, C.c_void_p), L.stream_ptr()),
                 "mmrca_image_preprocess")
        ev = torch.cuda.Event()
        ev.record(main)
        self._slot_free[k] = ev
        return out
