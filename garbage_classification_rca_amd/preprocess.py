"""GPU side of the input path: the reference's validation image pipeline
(main_both.py:433-440) and the V/H flips of its train pipeline (:416-417) as ONE kernel launch per batch.

    decoded uint8 HWC images (any sizes)  --pinned staging + one async H2D-->  mmrca_image_preprocess  -->  fp32 [B,3,H,W]

``plan_padding`` restates ``PadToMaintainAR.apply`` (keep_aspect_ratio.py:24-50) as numbers instead of a padded copy --
including its axis naming: ``img.shape`` is (H, W, C) but the transform calls size[0] "width", so it compares H/W with
the target ratio and pads the axis that makes a non-matching image LONGER (pinned by tests/golden/pad_goldens.npz).
The six remaining albumentations augmentations of the TRAIN_PIPELINE (:408-427: Rotate, GaussianBlur,
RandomBrightnessContrast, Sharpen, Perspective, ShiftScaleRotate) run on the GPU too (csrc/augment.hip): the host draws
which ones fire and their parameters (``sample_train_params``: albumentations' distributions) and turns them into
per-image descriptors (``plan_*``); the kernels are deterministic functions of those descriptors.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as L

MEAN = (0.485, 0.456, 0.406)       # main_both.py:400
STD = (0.229, 0.224, 0.225)        # main_both.py:401

DESC_DTYPE = np.dtype([("offset", np.int64), ("h", np.int32), ("w", np.int32), ("pad_top", np.int32), ("pad_left", np.int32),
                       ("ph", np.int32), ("pw", np.int32), ("flip_v", np.int32), ("flip_h", np.int32)])
assert DESC_DTYPE.itemsize == 40       # == sizeof(MmrcaImageDesc)


ROT_DTYPE = np.dtype([("src_offset", np.int64), ("dst_offset", np.int64), ("h", np.int32), ("w", np.int32), ("dh", np.int32),
                      ("dw", np.int32), ("x_min", np.int32), ("y_min", np.int32), ("enabled", np.int32), ("pad_", np.int32),
                      ("inv", np.float32, (6,))])
AUG_DTYPE = np.dtype([("blur", np.float32, (7,)), ("blur_k", np.int32), ("flip_v", np.int32), ("flip_h", np.int32), ("has_bc", np.int32),
                      ("bc_alpha", np.float32), ("bc_beta", np.float32), ("has_sharp", np.int32), ("sharp", np.float32, (9,)),
                      ("has_persp", np.int32), ("persp", np.float32, (9,)), ("has_scale", np.int32), ("scale", np.float32, (6,))])
assert ROT_DTYPE.itemsize == 72 and AUG_DTYPE.itemsize == 160       # == sizeof(MmrcaRotateDesc), sizeof(MmrcaAugDesc)

# cv2.getGaussianKernel(k, sigma <= 0) for k <= 7 returns these fixed taps (A.GaussianBlur(sigma_limit=0) -> sigma 0)
_BLUR_TAPS = {3: (0.25, 0.5, 0.25), 5: (0.0625, 0.25, 0.375, 0.25, 0.0625),
              7: (0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125)}


def plan_rotation(h: int, w: int, angle_deg: float):
    """A.Rotate(crop_border=True) of an (h, w) image: (inv[6], x_min, y_min, dh, dw).  ``inv`` maps a pixel of the rotated
    frame to the source pixel (cv2.warpAffine inverts getRotationMatrix2D((w/2-.5, h/2-.5), angle, 1)); the crop is
    albumentations' largest axis-aligned rectangle without border pixels."""
    t = math.radians(angle_deg)
    c, s_ = math.cos(t), math.sin(t)
    cx, cy = w / 2 - 0.5, h / 2 - 0.5
    inv = (c, -s_, cx - c * cx + s_ * cy, s_, c, cy - s_ * cx - c * cy)
    long_is_w = w >= h
    long_side, short_side = (w, h) if long_is_w else (h, w)
    sa, ca = abs(s_), abs(c)
    if short_side <= 2.0 * sa * ca * long_side or abs(sa - ca) < 1e-10:
        half = 0.5 * short_side
        wr, hr = (half / sa, half / ca) if long_is_w else (half / ca, half / sa)
    else:
        c2 = ca * ca - sa * sa
        wr, hr = (w * ca - h * sa) / c2, (h * ca - w * sa) / c2
    x_min, x_max = max(0, int(w / 2 - wr / 2)), min(w, int(w / 2 + wr / 2))
    y_min, y_max = max(0, int(h / 2 - hr / 2)), min(h, int(h / 2 + hr / 2))
    return inv, x_min, y_min, y_max - y_min, x_max - x_min


def plan_perspective(h: int, w: int, jitter) -> np.ndarray:
    """A.Perspective(keep_size=True): 3x3 map output pixel -> input pixel for the corner jitter [4,2] (fractions of the
    image size, corners in the order tl, tr, br, bl before re-ordering), the keep_size resize folded in."""
    q = np.asarray(jitter, dtype=np.float64).copy()
    q[1, 0] = 1.0 - q[1, 0]
    q[2] = 1.0 - q[2]
    q[3, 1] = 1.0 - q[3, 1]
    q *= np.array([w, h], dtype=np.float64)
    order = np.argsort(q[:, 0], kind="stable")
    left, right = q[order[:2]], q[order[2:]]
    left = left[np.argsort(left[:, 1], kind="stable")]
    tl, bl = left[0], left[1]
    far = np.hypot(*(right - tl).T)
    br, tr = (right[0], right[1]) if far[0] >= far[1] else (right[1], right[0])
    mw = max(int(np.hypot(*(br - bl))), int(np.hypot(*(tr - tl))), 2)
    mh = max(int(np.hypot(*(tr - br))), int(np.hypot(*(tl - bl))), 2)
    # rectangle [0,mw] x [0,mh] -> quadrilateral in closed form (Heckbert's square-to-quad mapping) -- not np.linalg.solve: a
    # threaded BLAS call per image from the training loop's main thread, next to 16 busy DataLoader workers, cost ~30 ms each
    (x0, y0), (x1, y1), (x2, y2), (x3, y3) = tl, tr, br, bl
    dx1, dx2, dx3 = x1 - x2, x3 - x2, x0 - x1 + x2 - x3
    dy1, dy2, dy3 = y1 - y2, y3 - y2, y0 - y1 + y2 - y3
    det = dx1 * dy2 - dx2 * dy1
    g = (dx3 * dy2 - dx2 * dy3) / det
    hh = (dx1 * dy3 - dx3 * dy1) / det
    unit = np.array([[x1 - x0 + g * x1, x3 - x0 + hh * x3, x0], [y1 - y0 + g * y1, y3 - y0 + hh * y3, y0], [g, hh, 1.0]], dtype=np.float64)
    hinv = unit @ np.diag([1.0 / mw, 1.0 / mh, 1.0])
    rx, ry = mw / w, mh / h
    resize = np.array([[rx, 0, 0.5 * rx - 0.5], [0, ry, 0.5 * ry - 0.5], [0, 0, 1]], dtype=np.float64)
    return (hinv @ resize).astype(np.float32).reshape(9)


def plan_scale(h: int, w: int, scale: float) -> np.ndarray:
    """A.ShiftScaleRotate(shift 0, rotate 0): output pixel -> input pixel, zoom about (w/2-.5, h/2-.5)."""
    cx, cy = w / 2 - 0.5, h / 2 - 0.5
    return np.array([1 / scale, 0, cx - cx / scale, 0, 1 / scale, cy - cy / scale], dtype=np.float64).astype(np.float32)


def sharpen_kernel(alpha: float, lightness: float) -> np.ndarray:
    k = np.full((3, 3), -alpha, dtype=np.float64)
    k[1, 1] = (1 - alpha) + alpha * (8 + lightness)
    return k.astype(np.float32).reshape(9)


def sample_train_params(rng: np.random.Generator, n: int, prob: float) -> List[dict]:
    """One parameter dict per image, drawn like albumentations 1.3 draws them for the reference's TRAIN_PIPELINE
    (main_both.py:407-429): every transform fires independently with probability ``prob`` (--prob_aug).
    Rotate: angle ~ U(-90, 90); GaussianBlur: ksize in {3,5,7}; RandomBrightnessContrast: alpha = 1 + U(-.2,.2),
    beta = U(-.2,.2); Sharpen: alpha ~ U(.2,.5), lightness ~ U(.5,1); Perspective: corner jitter |N(0, U(.05,.1))| mod 1;
    ShiftScaleRotate: scale ~ U(.5, 1.5)."""
    out = []
    for _ in range(n):
        p = {}
        if rng.random() < prob:
            p["angle"] = float(rng.uniform(-90, 90))
        if rng.random() < prob:
            p["blur_k"] = int(rng.choice((3, 5, 7)))
        p["flip_v"] = bool(rng.random() < prob)
        p["flip_h"] = bool(rng.random() < prob)
        if rng.random() < prob:
            p["bc"] = (float(1.0 + rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2)))
        if rng.random() < prob:
            p["sharpen"] = (float(rng.uniform(0.2, 0.5)), float(rng.uniform(0.5, 1.0)))
        if rng.random() < prob:
            p["persp"] = np.mod(np.abs(rng.normal(0.0, rng.uniform(0.05, 0.1), size=(4, 2))), 1.0)
        if rng.random() < prob:
            p["scale"] = float(rng.uniform(0.5, 1.5))
        out.append(p)
    return out


def pack_images(images: Sequence) -> dict:
    """Decoded uint8 HWC images (different sizes) -> ONE flat uint8 tensor in the staging layout (each image at the next
    16-byte boundary) + their [B,2] shapes.  Used as the DataLoader collate step of the GPU input path: a worker hands over one
    shared-memory segment per batch instead of one per image (256 segments per batch made the loader 40x slower than the
    decode itself), and with pin_memory=True the H2D copy reads it directly."""
    shapes = torch.tensor([[int(im.shape[0]), int(im.shape[1])] for im in images], dtype=torch.int32)
    sizes = [(int(h) * int(w) * 3 + 15) // 16 * 16 for h, w in shapes.tolist()]
    flat = torch.empty(sum(sizes), dtype=torch.uint8)
    off = 0
    for im, n in zip(images, sizes):
        t = im if isinstance(im, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(im))
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError("pack_images: images must be uint8 HWC with 3 channels")
        flat[off:off + t.numel()] = t.reshape(-1)
        off += n
    return {"flat": flat, "shapes": shapes}


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


def _pinned_no_fork(nbytes: int) -> torch.Tensor:
    """Pinned host buffer that forked children do NOT inherit (MADV_DONTFORK).  DataLoader workers are forked after these
    buffers exist; without this the parent's next write to the buffer is a copy-on-write of DMA-registered pages and the
    first H2D copy afterwards took 34 s (128 MB) on the MI355X box instead of 5 ms."""
    t = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    page = 4096
    lo = t.data_ptr() // page * page
    hi = (t.data_ptr() + nbytes + page - 1) // page * page
    libc = C.CDLL(None, use_errno=True)
    libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    if libc.madvise(C.c_void_p(lo), hi - lo, 10) != 0:          # MADV_DONTFORK
        print(f"[mmrca] madvise(MADV_DONTFORK) failed (errno {C.get_errno()}): the first batch after a DataLoader fork may be slow")
    return t


class GpuImagePipeline:
    """Reusable staging for batches of up to ``max_batch`` images of at most ``max_pixels`` pixels each.  Two pinned host
    buffers and two device buffers alternate, the copy runs on its own stream, so batch i+1 can be staged and copied
    while batch i is being consumed (``__call__`` returns a tensor that is ordered on the caller's current stream)."""

    def __init__(self, out_h: int, out_w: int, max_batch: int, max_pixels: int, device="cuda"):
        self.out_h, self.out_w, self.max_batch = int(out_h), int(out_w), int(max_batch)
        self.device = torch.device(device)
        self.max_pixels = int(max_pixels)
        cap = 2 * int(max_batch) * int(max_pixels) * 3          # second half: rotated + cropped copies (A.Rotate)
        self._cap = cap
        self._host = [_pinned_no_fork(cap) for _ in range(2)]
        self._host_desc = [_pinned_no_fork(max_batch * DESC_DTYPE.itemsize) for _ in range(2)]
        self._dev = [torch.empty(cap, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self._dev_desc = [torch.empty(max_batch * DESC_DTYPE.itemsize, dtype=torch.uint8, device=self.device) for _ in range(2)]
        extra = max_batch * (ROT_DTYPE.itemsize + AUG_DTYPE.itemsize)
        self._host_aug = [_pinned_no_fork(extra) for _ in range(2)]
        self._dev_aug = [torch.empty(extra, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self._u8 = torch.empty(2, self.max_batch, self.out_h, self.out_w, 3, dtype=torch.uint8, device=self.device)   # ping-pong images
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._slot_free = [None, None]          # event: the kernel that consumed this slot has finished
        self._keep = [None, None]               # pinned batch a slot's H2D copy reads from (kept alive until the slot is reused)
        self._i = 0
        self._mean = (C.c_float * 3)(*MEAN)
        self._std = (C.c_float * 3)(*STD)
        # One tiny batch through every stage NOW: loads the kernels' code objects and makes every device allocation before the
        # caller forks its DataLoader workers (the first launch from a process with 16 live forked children took 20 s on the
        # MI355X box; after this warm-up every call is ~10 ms).
        tiny = np.zeros((8, 8, 3), dtype=np.uint8)
        warm = [dict(angle=10.0, blur_k=3, bc=(1.0, 0.0), sharpen=(0.3, 1.0), persp=np.full((4, 2), 0.05), scale=0.9), {}][:min(2, self.max_batch)]
        self(tiny[None].repeat(len(warm), 0), aug=warm)          # (a --batch_size 1 pipeline has room for one image only)
        self([tiny])
        torch.cuda.synchronize(self.device)

    def __call__(self, images: Sequence[np.ndarray], flips: Optional[Sequence[Tuple[bool, bool]]] = None,
                 aug: Optional[Sequence[dict]] = None) -> torch.Tensor:
        """images: decoded uint8 HWC arrays (or tensors).  flips: (vertical, horizontal) per image (validation pipeline +
        flips in one launch).  aug: one ``sample_train_params`` dict per image -> the full TRAIN_PIPELINE."""
        packed = images if isinstance(images, dict) else None           # pack_images() output: already in the staging layout
        shapes = [tuple(x) for x in packed["shapes"].tolist()] if packed is not None else [(int(im.shape[0]), int(im.shape[1])) for im in images]
        B = len(shapes)
        if B == 0 or B > self.max_batch:
            raise L.MmrcaError(f"GpuImagePipeline: batch of {B} images (capacity {self.max_batch})")
        if aug is not None and (flips is not None or len(aug) != B):
            raise L.MmrcaError("GpuImagePipeline: pass either flips or one aug dict per image")
        k = self._i & 1
        self._i += 1
        if self._slot_free[k] is not None:
            self._slot_free[k].synchronize()               # the pinned buffer may still be read by the previous copy / kernel
        need = 2 * sum((h * w * 3 + 15) // 16 * 16 for h, w in shapes)
        if need > self._cap:                               # a batch of larger images than planned for: grow both slots once
            torch.cuda.synchronize(self.device)
            self._cap = need + need // 4
            self._host = [_pinned_no_fork(self._cap) for _ in range(2)]
            self._dev = [torch.empty(self._cap, dtype=torch.uint8, device=self.device) for _ in range(2)]
            self.max_pixels = max(self.max_pixels, max(h * w for h, w in shapes))
        host = self._host[k].numpy()
        desc = np.zeros(B, dtype=DESC_DTYPE)
        rot = np.zeros(B, dtype=ROT_DTYPE)
        augd = np.zeros(B, dtype=AUG_DTYPE)
        ar = self.out_w / self.out_h
        off = 0
        sizes = []
        for b, (h, w) in enumerate(shapes):
            n = h * w * 3
            if packed is None:
                img = images[b].numpy() if isinstance(images[b], torch.Tensor) else images[b]
                if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
                    raise L.MmrcaError("GpuImagePipeline: images must be uint8 HWC with 3 channels")
                host[off:off + n] = np.ascontiguousarray(img).reshape(-1)
            self.max_pixels = max(self.max_pixels, h * w)
            sizes.append((off, h, w))
            off += (n + 15) // 16 * 16
        src = self._host[k]
        if packed is not None:
            flat = packed["flat"]
            if flat.dtype != torch.uint8 or flat.numel() != off:
                raise L.MmrcaError("GpuImagePipeline: packed batch does not match its shapes")
            if flat.is_pinned():
                src = flat                                  # the DataLoader pinned it: copy to the device straight from there
                self._keep[k] = flat
            else:
                host[:off] = flat.numpy()
        src_end = off
        any_rot = False
        for b, (o, h, w) in enumerate(sizes):
            p = aug[b] if aug is not None else {}
            if p.get("angle") is not None:
                inv, x_min, y_min, dh, dw = plan_rotation(h, w, p["angle"])
                if dh > 0 and dw > 0:
                    rot[b] = (o, off, h, w, dh, dw, x_min, y_min, 1, 0, np.asarray(inv, dtype=np.float64).astype(np.float32))
                    o, h, w = off, dh, dw
                    off += (dh * dw * 3 + 15) // 16 * 16
                    any_rot = True
            pt, pl, ph, pw = plan_padding(h, w, ar)
            fv, fh = (flips[b] if flips is not None else (False, False))
            desc[b] = (o, h, w, pt, pl, ph, pw, int(fv), int(fh))
            if aug is not None:
                a = augd[b]
                if p.get("blur_k"):
                    a["blur_k"] = p["blur_k"]
                    a["blur"][:p["blur_k"]] = _BLUR_TAPS[p["blur_k"]]
                a["flip_v"], a["flip_h"] = int(bool(p.get("flip_v"))), int(bool(p.get("flip_h")))
                if p.get("bc") is not None:
                    a["has_bc"], a["bc_alpha"] = 1, np.float32(p["bc"][0])
                    a["bc_beta"] = np.float32(p["bc"][1]) * np.float32(255.0)
                if p.get("sharpen") is not None:
                    a["has_sharp"], a["sharp"] = 1, sharpen_kernel(*p["sharpen"])
                if p.get("persp") is not None:
                    a["has_persp"], a["persp"] = 1, plan_perspective(self.out_h, self.out_w, p["persp"])
                if p.get("scale") is not None:
                    a["has_scale"], a["scale"] = 1, plan_scale(self.out_h, self.out_w, p["scale"])
        nd = B * DESC_DTYPE.itemsize
        self._host_desc[k].numpy()[:nd] = desc.view(np.uint8).reshape(-1)
        nr, na = B * ROT_DTYPE.itemsize, B * AUG_DTYPE.itemsize
        if aug is not None:
            ha = self._host_aug[k].numpy()
            ha[:nr] = rot.view(np.uint8).reshape(-1)
            ha[nr:nr + na] = augd.view(np.uint8).reshape(-1)
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._copy_stream):
            self._dev[k][:src_end].copy_(src[:src_end], non_blocking=True)
            self._dev_desc[k][:nd].copy_(self._host_desc[k][:nd], non_blocking=True)
            if aug is not None:
                self._dev_aug[k][:nr + na].copy_(self._host_aug[k][:nr + na], non_blocking=True)
        main.wait_stream(self._copy_stream)
        out = torch.empty(B, 3, self.out_h, self.out_w, dtype=torch.float32, device=self.device)
        lib = L.load()
        mean, std = C.cast(self._mean, C.c_void_p), C.cast(self._std, C.c_void_p)
        if aug is None:
            L._check(lib.mmrca_image_preprocess(L.ptr(self._dev[k]), L.ptr(self._dev_desc[k]), L.ptr(out), B, self.out_h, self.out_w,
                                                mean, std, L.stream_ptr()), "mmrca_image_preprocess")
        else:
            if any_rot:
                L._check(lib.mmrca_image_rotate_crop(L.ptr(self._dev[k]), L.ptr(self._dev_aug[k]), B, self.max_pixels, L.stream_ptr()),
                         "mmrca_image_rotate_crop")
            L._check(lib.mmrca_image_resize_u8(L.ptr(self._dev[k]), L.ptr(self._dev_desc[k]), L.ptr(self._u8[0]), B, self.out_h,
                                               self.out_w, L.stream_ptr()), "mmrca_image_resize_u8")
            L._check(lib.mmrca_image_augment(L.ptr(self._u8[0]), L.ptr(self._u8[1]), self._dev_aug[k].data_ptr() + nr, L.ptr(out), B,
                                             self.out_h, self.out_w, mean, std, L.stream_ptr()), "mmrca_image_augment")
        ev = torch.cuda.Event()
        ev.record(main)
        self._slot_free[k] = ev
        return out
