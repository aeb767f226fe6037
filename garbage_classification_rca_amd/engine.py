"""Host-side sequencing of the MM-RCA hot path over libmmrca (HIP).

Data layout in HBM (sized for 288 GB: nothing is recomputed, every saved activation stays resident):

* ONE flat fp32 parameter arena (master weights) in the order  text encoder | vision encoder | fusion head,
  ONE flat fp32 gradient arena with the same offsets (data-parallel all-reduce runs over contiguous slices of
  it, no bucket copies) and, in bf16 mode, ONE bf16 working copy refreshed by the fused optimizer kernel.
  q/k/v projection weights of the text encoders sit adjacently, so the fused [3D, D] QKV GEMM needs no copy.
* activations: row-major [rows, features] in the compute dtype, rows padded to a multiple of 256 (ROWPAD) with zeros that
  no kernel ever writes (the persistent 256x256 GEMM streams whole 256-row tiles; the weight-gradient GEMM contracts over
  rows through transposed LDS reads and needs whole 64-row steps).
* no torch autograd inside: forward() saves what backward() needs; backward() accumulates (+=) into the gradient
  arena, which is exactly the reference's gradient-accumulation semantic (main_both.py:112-124).

PyTorch is used for memory, streams and a few index copies only; every arithmetic op is a libmmrca call.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as L
from . import spec as S
from .conv_engine import ConvEncoder, CONV_MODELS

import os

ALIGN = 64          # arena entries start on 256-byte boundaries
# Bias gradient as an extra ones-operand MFMA inside the weight-gradient GEMM: parity-tested, measured net-neutral
# (conditional MFMAs disturb the main loop), off by default.
FUSE_BIAS_GRAD = os.environ.get("MMRCA_FUSE_BIAS", "0") == "1"
# The same for the QKV in-projection only (the one linear whose bias gradient no neighbouring kernel produces) on the
# 128x128x32 weight-gradient kernel, instead of a 232 MB column-sum pass per layer: measured 1.5 % SLOWER end to end
# (5,720 vs 5,808 samples/s, same box, two rounds) -- the conditional MFMAs cost the main loop more than the pass. Off.
QKV_BIAS_IN_WGRAD = os.environ.get("MMRCA_QKV_BIAS_IN_WGRAD", "0") == "1"
# GELU backward fused into the GEMMs: FFN1's forward epilogue stores gelu'(h) instead of h, FFN2's input-gradient
# epilogue multiplies by it and emits the FFN1 bias gradient (column sums): +3.8 % end to end (no separate pass).
FUSE_GELU_GRAD = os.environ.get("MMRCA_FUSE_GELU", "1") == "1"
# weight-gradient GEMMs on a second HIP stream, concurrent with the input-gradient chain of the same layer.  Helped
# (+4 %) while the GEMMs left the chip half idle; with the current kernels two concurrent GEMMs only fight over LDS
# slots and L2 (-4 %, same-box A/B), so it is off by default.
SIDE_STREAM_WGRAD = os.environ.get("MMRCA_SIDE_STREAM", "0") == "1"
# text encoder and vision encoder are independent until the fusion head: run them on two streams
CONCURRENT_ENCODERS = os.environ.get("MMRCA_CONCURRENT_ENCODERS", "1") == "1"
# Only the class-token row of each encoder's LAST layer reaches the head (reference: hidden_state[:, 0] /
# torchvision's x[:, 0]), so everything of that layer after the attention mix -- out-proj, residual, LayerNorms, the
# FFN, and their backward -- runs on B rows instead of B*S.  Same logits and same gradients for every parameter (the
# pruned rows' outputs were never read; their gradients are exact zeros); "0" restores the full-row top layer.
CLS_TAIL = os.environ.get("MMRCA_CLS_TAIL", "1") == "1"
# weight gradients on the 256x256 split-K kernel (mmrca_gemm_splitk) wherever the shape qualifies; "0" = 128x128 + fp32 atomics
SPLITK_WGRAD = os.environ.get("MMRCA_SPLITK_WGRAD", "1") == "1"
# bf16x3 mode: plane pairs per backward product (3 = full; see _lin_bwd_x3).  The forward always runs all three.
X3_WGRAD_PASSES = int(os.environ.get("MMRCA_X3_WGRAD_PASSES", "3"))
X3_DGRAD_PASSES = int(os.environ.get("MMRCA_X3_DGRAD_PASSES", "3"))
# ViT residual adds ride on the NEXT LayerNorm (add_layernorm: s = x + res, y = LN(s)) instead of the GEMM epilogue: the
# out-projection and FFN2 forward GEMMs become bias-only and qualify for the persistent 256x256 kernel (1,050-1,150 TFLOP/s
# against 780 for the 128x128 kernel with its addend read); the LayerNorm reads one more operand and writes the sum.
LN_RESIDUAL = os.environ.get("MMRCA_LN_RESIDUAL", "1") == "1"
ROWPAD = 256         # (the persistent 256x256 GEMM reads whole 256-row tiles of its A operand)


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class Arena:
    def __init__(self, entries: List[Tuple[str, Tuple[int, ...]]], device, with_lp: bool, with_lo: bool = False):
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...], int]] = {}
        off = 0
        for k, shp in entries:
            n = 1
            for s in shp:
                n *= s
            off = _round_up(off, ALIGN)
            self.offsets[k] = (off, tuple(shp), n)
            off += n
        self.total = _round_up(off, ALIGN)
        self.p = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.g = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.lp = torch.zeros(self.total, dtype=torch.bfloat16, device=device) if with_lp else None
        # bf16x3 mode: p = lp + lp_lo (hi / lo bf16 planes of the fp32 masters, see csrc/gemm_x3.hip)
        self.lp_lo = torch.zeros(self.total, dtype=torch.bfloat16, device=device) if with_lo else None
        self.lp_valid = False
        self._views: Dict[Tuple[str, str], torch.Tensor] = {}

    def view(self, key, which="p"):
        """[shape] view of one parameter in plane `which` (p | g | lp | lp_lo).  Cached: the planes never move, and the engine asks
        for the same ~400 views every step (a slice + view costs ~5 us of host time each)."""
        v = self._views.get((key, which))
        if v is None:
            off, shp, n = self.offsets[key]
            v = self._views[(key, which)] = getattr(self, which)[off:off + n].view(shp)
        return v

    def span(self, first_key, last_key):
        a = self.offsets[first_key][0]
        off, _, n = self.offsets[last_key]
        return a, off + n


class Planes:
    """An fp32 [rows, cols] matrix stored as two bf16 planes, value = hi + lo (bf16x3 mode, csrc/gemm_x3.hip): what a GEMM epilogue
    writes when its only consumers are other bf16x3 GEMMs (saves the fp32 store + the consumers' split pass)."""

    def __init__(self, hi, lo):
        self.hi, self.lo = hi, lo


class TextPack:
    """Packed (unpadded) token layout of one caption batch: the rows the text encoder works on are the live tokens of
    every caption stored back to back -- padding rows are dead work under class-token pooling with masked keys (their
    outputs are never read, their gradients are exact zeros).  ``perm`` [M] = index of each packed row in the padded
    [B*T] layout, ``cu`` [B+1] = cumulative lengths, ``mask`` [M] = key mask per packed row, ``first`` [B] = row of
    each caption's class token.  M is rounded up to a multiple of 64 with padding rows of captions that have room
    (masked keys like in the padded layout), so the weight-gradient GEMMs see whole 64-row K steps."""

    def __init__(self, perm, cu, mask, first, M, B, T):
        self.perm, self.cu, self.mask, self.first, self.M, self.B, self.T = perm, cu, mask, first, M, B, T


def make_text_pack(attention_mask, device) -> Optional["TextPack"]:
    """Build the packed layout from a HOST attention mask [B,T] (numpy / CPU tensor: what a DataLoader hands over, so no
    device sync is involved).  Returns None when packing does not apply: a mask that is not a non-empty prefix of ones
    per caption (e.g. a caption zeroed by modality dropout), or B*T not a multiple of 64."""
    import numpy as np
    m = np.asarray(attention_mask.cpu() if isinstance(attention_mask, torch.Tensor) else attention_mask) != 0
    B, T = m.shape
    lens = m.sum(1)
    if (B * T) % 64 != 0 or (lens < 1).any() or not (m == (np.arange(T)[None, :] < lens[:, None])).all():
        return None
    total = int(lens.sum())
    extra = _round_up(total, 64) - total
    ext = lens.astype(np.int64).copy()
    for b in range(B - 1, -1, -1):
        if extra == 0:
            break
        take = min(extra, T - int(ext[b]))
        ext[b] += take
        extra -= take
    cu = np.zeros(B + 1, dtype=np.int64)
    cu[1:] = np.cumsum(ext)
    M = int(cu[-1])
    within = np.arange(M) - np.repeat(cu[:-1], ext)
    perm = np.repeat(np.arange(B) * T, ext) + within
    dev = torch.device(device)

    def to(a, dt):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dt)
        if dev.type == "cuda":                     # pinned staging + async copy: no stream synchronisation on the way
            return t.pin_memory().to(dev, non_blocking=True)
        return t.to(dev)

    return TextPack(to(perm, torch.int64), to(cu, torch.int32), to(m.reshape(-1)[perm], torch.int32), to(cu[:-1], torch.int64), M, B, T)


class MMRCAEngine:
    """Owns parameters, gradients and activations of one MM-RCA replica on one GPU."""

    def __init__(self, text_model: str, image_model: str, n_classes: int = 4, reverse: bool = True, mode: int = 0,
                 dtype=torch.bfloat16, device="cuda", gemm_impl: int = L.IMPL_AUTO,
                 attn_impl: int = L.IMPL_AUTO, image_size: int = 224):
        """dtype: torch.bfloat16 (the benchmarked mode) | torch.float32 (every GEMM on the fp32 matrix cores) | "bf16x3": fp32
        storage, residual stream, LayerNorm, attention and head, with every encoder nn.Linear (forward, input gradient, weight
        gradient) as a three-pass split-bf16 product on the bf16 matrix cores (csrc/gemm_x3.hip): the reference's fp32
        arithmetic (multimodal_model.py:651-726) to ~1e-6 on the logits at several times the fp32 mode's speed.
        "bf16x3f": the bf16x3 FORWARD (same logits) with the bf16 mode's BACKWARD -- single-pass bf16 products and the bf16 attention
        backward on the hi planes of the saved fp32 activations, bf16 gradient buffers, fp32 gradient accumulation: the fastest
        mode whose logits meet the north-star bound; its gradients are the bf16 mode's, evaluated at fp32-accurate activations."""
        L.load()
        name = dtype.lower() if isinstance(dtype, str) else ""
        self.x3 = name in ("bf16x3", "x3", "bf16x3f", "x3f")
        self.x3f = name in ("bf16x3f", "x3f")
        if self.x3f and not FUSE_GELU_GRAD:
            raise L.MmrcaError("bf16x3f needs the fused GELU gradient (MMRCA_FUSE_GELU=1)")
        if self.x3:
            dtype = torch.float32
        elif isinstance(dtype, str):
            dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "f32": torch.float32}[dtype.lower()]
        if text_model == "bart":
            # the reference accepts the name (multimodal_model.py:137-144, 182-183) but its head is built for 768-wide text
            # features (input_size_txt = 768, :257) while facebook/bart-large emits 1024: its forward fails on the first batch
            raise ValueError("Wrong text model: bart (facebook/bart-large is 1024 wide; the reference's MM_RCA head is hard-wired to "
                             "768-wide text features, multimodal_model.py:257, and cannot run it either)")
        if text_model not in S.TEXT_SPECS:
            raise ValueError(f"Wrong text model: {text_model}")
        if image_model not in S.VISION_SPECS and image_model not in CONV_MODELS:
            raise ValueError(f"Wrong image model: {image_model}")
        self.ts = S.TEXT_SPECS[text_model]
        self.vs = S.VISION_SPECS.get(image_model)
        self.n_classes, self.reverse, self.mode = n_classes, bool(reverse), int(mode)
        self.dtype, self.device = dtype, torch.device(device)
        self.dt = L.dtype_code(dtype)
        # dtype of the gradient buffers / of every backward kernel (bf16x3f: a bf16 backward behind an fp32-accurate forward)
        self.gdtype = torch.bfloat16 if self.x3f else dtype
        self.bdt = L.BF16 if self.x3f else self.dt
        self.gemm_impl, self.attn_impl = gemm_impl, attn_impl
        # bf16x3f with a conv image backbone: the conv kernels run in bf16 (on the hi planes of the weights) next to the fp32-accurate text
        # encoder.  Measured on the initialisation the model trains from (tools/conv_feature_error.py, 480 x 480): of the bf16 mode's
        # logits error (1.0e-3 EfficientNetV2-M, 1.5e-3 -L) the TEXT encoder's bf16 feature carries 7.8e-4 / 1.3e-3 and the conv backbone's
        # 3.9e-4 / 5.4e-4 -- BatchNorm renormalises every layer -- so this split meets the 1e-3 bound at nearly the bf16 mode's speed,
        # where the all-fp32 conv kernels of bf16x3 run at a fifth of it.
        self.conv_dtype = torch.bfloat16 if (self.x3f and image_model in CONV_MODELS) else None
        # conv image backbones (EfficientNetV2-M/L, ShuffleNetV2): conv_engine.ConvEncoder over csrc/conv.hip
        self.conv = ConvEncoder(image_model, self, image_size) if self.vs is None else None
        self.d_txt, self.d_img = self.ts.dim, (self.vs.dim if self.vs is not None else self.conv.dim)
        ents = [("text_model." + k, s) for k, s in S.text_params(self.ts)]
        vis_ents = S.vision_params(self.vs) if self.vs is not None else self.conv.param_entries()
        ents += [("image_model." + k, s) for k, s in vis_ents]
        self.head_keys = S.head_used_params(self.d_img, self.d_txt, n_classes, mode == 1, mode == 2)
        ents += self.head_keys
        self.arena = Arena(ents, self.device, with_lp=(dtype == torch.bfloat16 or self.x3), with_lo=self.x3)
        self.param_keys = [k for k, _ in ents]
        self.text_span = self.arena.span(ents[0][0], "text_model." + S.text_params(self.ts)[-1][0])
        self.image_span = self.arena.span("image_model." + vis_ents[0][0], "image_model." + vis_ents[-1][0])
        if self.conv is not None:
            self.conv.init_buffers(self.device)
        self.head_span = self.arena.span(self.head_keys[0][0], self.head_keys[-1][0])
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        # bf16x3: gelu(h) and its gradient only feed other bf16x3 GEMMs, so their epilogues write them as two bf16 planes
        self._g_planes = self.x3 and FUSE_GELU_GRAD and os.environ.get("MMRCA_X3_PLANES_OUT", "1") == "1"
        self._plane_valid = set()
        self._plane_fresh = set()
        self._ln_planes = self._g_planes
        # FFN1's epilogue: gelu + gelu' (bf16x3f: gelu' as bf16, what its bf16 backward multiplies by)
        self._gelu_act = (L.ACT_GELU_SAVE_GRAD_BF16 if self.x3f else L.ACT_GELU_SAVE_GRAD) if FUSE_GELU_GRAD else L.ACT_GELU
        self._hdt = torch.bfloat16 if self.x3f else None
        self._saved = None
        # parameter groups that become final together during backward; they tile the arena exactly (padding included)
        self.groups: Dict[str, Tuple[int, int]] = {}
        starts = []
        for k in self.param_keys:
            gname = self._group_of(k)
            if not starts or starts[-1][0] != gname:
                starts.append((gname, self.arena.offsets[k][0]))
        for j, (gname, lo) in enumerate(starts):
            hi = starts[j + 1][1] if j + 1 < len(starts) else self.arena.total
            self.groups[gname] = (lo, hi)
        self.grad_sync = None          # distributed.GradSync, set by the training driver
        on_gpu = self.device.type == "cuda"
        self._side_v = torch.cuda.Stream(device=self.device) if (SIDE_STREAM_WGRAD and on_gpu) else None
        self._side_t = torch.cuda.Stream(device=self.device) if (SIDE_STREAM_WGRAD and on_gpu) else None
        self._side = self._side_v          # the side stream of the encoder whose backward is being queued
        self._text_stream = torch.cuda.Stream(device=self.device) if (CONCURRENT_ENCODERS and on_gpu) else None
        self._first_wgrad_ev = None
        self._head_w = self._head_struct("p")
        self._head_g = self._head_struct("g")
        # fused QKV views of the text encoder must be contiguous in the arena
        for i in range(self.ts.layers):
            K = S.text_layer_keys(self.ts, i)
            o = [self.arena.offsets["text_model." + K[n] + ".weight"] for n in ("q", "k", "v")]
            assert o[1][0] == o[0][0] + o[0][2] and o[2][0] == o[1][0] + o[1][2], "qkv weights not adjacent"
            o = [self.arena.offsets["text_model." + K[n] + ".bias"] for n in ("q", "k", "v")]
            assert o[1][0] == o[0][0] + o[0][2] and o[2][0] == o[1][0] + o[1][2], "qkv biases not adjacent"

    @staticmethod
    def _group_of(key: str) -> str:
        import re
        m = re.match(r"text_model\.(?:transformer|encoder)\.layer\.(\d+)\.", key)
        if m:
            return f"text_layer_{int(m.group(1))}"
        if key.startswith("text_model.pooler"):
            return "text_tail"
        if key.startswith("text_model."):
            return "text_emb"
        m = re.match(r"image_model\.encoder\.layers\.encoder_layer_(\d+)\.", key)
        if m:
            return f"image_layer_{int(m.group(1))}"
        if key.startswith("image_model.encoder.ln"):
            return "image_ln"
        # conv backbones (conv_engine.py): one group per stage, so that the gradient exchange of stage k overlaps the backward of the
        # stages below it (the stem stays "image_emb": the group the encoder's backward finishes with)
        m = re.match(r"image_model\.(stem\.1|stage\d+|final_conv|conv5)\.", key)
        if m:
            return "image_stage_" + m.group(1)
        if key.startswith("image_model."):
            return "image_emb"
        return "head"

    def _ready(self, group: str, flush: bool = False):
        if self.grad_sync is not None and group in self.groups:
            lo, hi = self.groups[group]
            self.grad_sync.span_ready(lo, hi, flush)

    # ------------------------------------------------------------------ parameters
    def _head_struct(self, which) -> L.HeadPtrs:
        names = {"sai": "self_attention_image", "sat": "self_attention_text", "c1": "cross_attention_1", "c2": "cross_attention_2"}
        leaf = {"wq": "W_query.weight", "bq": "W_query.bias", "wk": "W_key.weight", "bk": "W_key.bias",
                "wv": "W_value.weight", "bv": "W_value.bias", "g": "norm.weight", "b": "norm.bias"}
        fin = ["final_with_everything", "final_features_only_linear", "cross_attention_only_linear"][self.mode]
        st = L.HeadPtrs()
        for b, nm in names.items():
            for l, suffix in leaf.items():
                setattr(st, f"{b}_{l}", self.arena.view(f"{nm}.{suffix}", which).data_ptr())
        st.fin_w = self.arena.view(fin + ".weight", which).data_ptr()
        st.fin_b = self.arena.view(fin + ".bias", which).data_ptr()
        return st

    def W(self, key):
        """parameter view in the compute dtype"""
        return self.arena.view(key, "lp" if self.dtype == torch.bfloat16 else "p")

    def convW(self, key):
        """parameter view in the conv encoder's compute dtype (bf16x3f: the bf16 hi plane while W() hands out the fp32 masters)"""
        return self.arena.view(key, "lp") if self.conv_dtype == torch.bfloat16 else self.W(key)

    def Wflat(self, key, numel):
        off = self.arena.offsets[key][0]
        src = self.arena.lp if self.dtype == torch.bfloat16 else self.arena.p
        return src[off:off + numel]

    def Wx3(self, key, numel=None):
        """(hi, lo) bf16 planes of a weight (bf16x3 mode)"""
        off, _, n = self.arena.offsets[key]
        n = n if numel is None else numel
        return self.arena.lp[off:off + n], self.arena.lp_lo[off:off + n]

    # ---- bf16x3: (hi, lo) bf16 planes of fp32 activations.  One pair of plane buffers per fp32 buffer (keyed by its address:
    # saved activations have per-layer buffers, so the planes a forward GEMM made of its input are still there for that
    # layer's weight-gradient GEMM); `_plane_valid` lists the ones made since the current forward began.
    def _planes_of(self, x):
        key = ("planes", x.data_ptr(), tuple(x.shape))
        pl = self._bufs.get(key)
        if pl is None:
            pl = (torch.zeros(x.shape, dtype=torch.bfloat16, device=self.device), torch.zeros(x.shape, dtype=torch.bfloat16, device=self.device))
            self._bufs[key] = pl
        return key, pl

    def planes_buf(self, name, rows, cols, layer=0) -> Planes:
        return Planes(self.buf(name + "_hi", rows, cols, torch.bfloat16, layer), self.buf(name + "_lo", rows, cols, torch.bfloat16, layer))

    def _split(self, x, rows, cols, reuse=False):
        if isinstance(x, Planes):           # produced as planes by a GEMM epilogue
            return x.hi, x.lo
        key, pl = self._planes_of(x)
        if key in self._plane_fresh:        # just written by the LayerNorm that produced x (one consumer in the forward)
            self._plane_fresh.discard(key)
            self._plane_valid.add(key)
            return pl
        if not (reuse and key in self._plane_valid):
            L.split_f32(x, pl[0], pl[1], rows * cols)
            self._plane_valid.add(key)
        return pl

    def _hi(self, x, rows, cols):
        """bf16 (hi-plane) form of a saved fp32 activation: the operand of the bf16x3f mode's bf16 backward.  The forward's GEMMs /
        LayerNorms / attention already wrote it for everything they consumed; anything else is rounded here."""
        if isinstance(x, Planes):
            return x.hi
        key, pl = self._planes_of(x)
        if key in self._plane_fresh:
            self._plane_fresh.discard(key)
            self._plane_valid.add(key)
        elif key not in self._plane_valid:
            L.cast_f32_to_bf16(x, pl[0], rows * cols)
            self._plane_valid.add(key)
        return pl[0]

    def Wb(self, key, numel=None):
        """parameter view in the dtype of the backward kernels (bf16x3f: the hi plane = bf16 rounding of the fp32 master)"""
        if not self.x3f:
            return self.W(key) if numel is None else self.Wflat(key, numel)
        off, shp, n = self.arena.offsets[key]
        return self.arena.lp[off:off + n].view(shp) if numel is None else self.arena.lp[off:off + numel]

    def G(self, key):
        return self.arena.view(key, "g")

    def Gflat(self, key, numel):
        off = self.arena.offsets[key][0]
        return self.arena.g[off:off + numel]

    def refresh_working_copy(self, force=False):
        if self.arena.lp is not None and (force or not self.arena.lp_valid):
            if self.x3:
                L.split_f32(self.arena.p, self.arena.lp, self.arena.lp_lo, self.arena.total)
            else:
                L.cast_f32_to_bf16(self.arena.p, self.arena.lp, self.arena.total)
            self.arena.lp_valid = True

    def init_parameters(self, seed: int = 0):
        """Random init (no pretrained weights offline): encoders N(0, 0.02) / LayerNorm (1, 0) as BERT/ViT do;
        head layers with torch's nn.Linear / nn.LayerNorm defaults (the reference's init, multimodal_model.py:199-292)."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.no_grad():
            for k in self.param_keys:
                v = self.arena.view(k)
                leaf = k.rsplit(".", 1)[-1]
                is_head = not k.startswith(("text_model.", "image_model."))
                lname = k.lower()
                is_norm = ("norm" in lname) or (".ln_" in lname) or lname.endswith(("encoder.ln.weight", "encoder.ln.bias"))
                if self.conv is not None and k.startswith("image_model."):
                    # torchvision's init: conv kaiming_normal_(fan_out), BatchNorm (1, 0), squeeze-excitation biases 0
                    if v.dim() == 4:
                        fan_out = v.shape[0] * v.shape[2] * v.shape[3]
                        v.copy_(torch.randn(v.shape, generator=g) * math.sqrt(2.0 / fan_out))
                    elif leaf == "weight":
                        v.fill_(1.0)
                    else:
                        v.zero_()
                elif is_norm:
                    v.fill_(1.0 if leaf == "weight" else 0.0)
                elif is_head:
                    if v.dim() == 2:
                        bound = 1.0 / math.sqrt(v.shape[1])
                        v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * bound)
                        self._last_fan_in = v.shape[1]
                    else:
                        bound = 1.0 / math.sqrt(self._last_fan_in)
                        v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * bound)
                elif leaf == "bias" or leaf.endswith("_bias"):
                    v.zero_()
                else:
                    v.copy_(torch.randn(v.shape, generator=g) * 0.02)
        self.arena.lp_valid = False

    def load_arrays(self, sd: Dict[str, "torch.Tensor"], strict=True):
        with torch.no_grad():
            for k in self.param_keys:
                if k in sd:
                    self.arena.view(k).copy_(torch.as_tensor(sd[k]).to(self.device, torch.float32).view(self.arena.offsets[k][1]))
                elif strict:
                    raise KeyError(k)
        self.arena.lp_valid = False

    # ------------------------------------------------------------------ buffers
    def _splitk_ws(self):
        """workspace of mmrca_gemm_splitk for the CURRENT stream (the text / vision / side streams run weight gradients
        concurrently, so each needs its own partial-tile slabs)"""
        key = ("splitk_ws", torch.cuda.current_stream().cuda_stream)
        t = self._bufs.get(key)
        if t is None:
            t = torch.empty(L.SPLITK_WS_BYTES, dtype=torch.uint8, device=self.device)
            self._bufs[key] = t
        return t

    def vec(self, name, n, dtype=torch.float32, layer=0):
        """a [1, n] vector (per-row statistics, log-sum-exps): NOT a GEMM operand, so no 256-row padding -- through buf() every
        [1, B*H*S] log-sum-exp buffer of the ViT was a 620 MB tensor"""
        key = (name, layer, "vec", n, dtype)
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.zeros(1, n, dtype=dtype, device=self.device)
        return t

    def buf(self, name, rows, cols, dtype=None, layer=0):
        dtype = dtype or self.dtype
        key = (name, layer, rows, cols, dtype)
        t = self._bufs.get(key)
        if t is None:
            t = torch.zeros(_round_up(max(rows, 1), ROWPAD), cols, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def release_buffers(self):
        self.buffer_generation = getattr(self, "buffer_generation", 0) + 1     # captured HIP graphs point into these buffers (GraphedTrainStep checks)
        self._bufs.clear()
        self._plane_valid.clear()
        self._plane_fresh.clear()
        self._saved = None
        if self.conv is not None:
            self.conv.release()

    # ------------------------------------------------------------------ op helpers
    def _lin_fwd(self, x, wkey, bkey, out, M, N, K, act=L.ACT_NONE, preact=None, addend=None, wnumel=None):
        if self.x3:
            b = self.arena.view(bkey) if wnumel is None else self.arena.p[self.arena.offsets[bkey][0]:self.arena.offsets[bkey][0] + N]
            po = isinstance(out, Planes)
            L.gemm_x3(self._split(x, M, K), self.Wx3(wkey, wnumel), out.hi if po else out, C_lo=(out.lo if po else None), bias=b, addend=addend,
                      preact=preact, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, a_layout=L.ROWK, b_layout=L.ROWK, act=act, impl=self.gemm_impl)
            return
        w = self.W(wkey) if wnumel is None else self.Wflat(wkey, wnumel)
        b = self.W(bkey) if wnumel is None else self.Wflat(bkey, N)
        L.gemm(x, w, out, bias=b, addend=addend, preact=preact, M=M, N=N, K=K, lda=K, ldb=K, ldc=N,
               a_layout=L.ROWK, b_layout=L.ROWK, act=act, dtype=self.dt, impl=self.gemm_impl)

    def _lin_bwd(self, dy, x, wkey, bkey, dx, M, N, K, addend=None, wnumel=None, gelu_h=None, bias_done=False, gelu_db=None,
                 fuse_db=False):
        """dy [M,N], x [M,K], weight [N,K]:  dW += dy^T x and db += colsum(dy) in ONE pass (the bias gradient rides
        on the weight-gradient GEMM); dx = dy W (+ addend), optionally times gelu'(gelu_h) in the epilogue."""
        Mk = _round_up(M, 64)
        gw = self.G(wkey) if wnumel is None else self.Gflat(wkey, wnumel)
        gb = self.G(bkey) if wnumel is None else self.Gflat(bkey, N)
        if self.x3 and not self.x3f:
            return self._lin_bwd_x3(dy, x, wkey, dx, M, N, K, Mk, gw, gb, addend, wnumel, gelu_h, bias_done, gelu_db)
        if self.x3f:
            x = self._hi(x, M, K)
        bdt = self.bdt

        def wgrad():
            fused = (FUSE_BIAS_GRAD or fuse_db) and not bias_done
            if SPLITK_WGRAD and not fused and self.gemm_impl == L.IMPL_AUTO and L.gemm_splitk_ok(N, K, Mk, bdt):
                # 256x256 tiles, partial tiles through a per-stream workspace, no atomics (1,020-1,150 vs 800-870 TFLOP/s)
                L.gemm_splitk(dy, x, gw, self._splitk_ws(), M=N, N=K, K=Mk, lda=N, ldb=K, ldc=K)
            else:
                L.gemm(dy, x, gw, bias=(gb if fused else None), M=N, N=K, K=Mk, lda=N, ldb=K, ldc=K, a_layout=L.KROW,
                       b_layout=L.KROW, accum=True, dtype=bdt, impl=self.gemm_impl)
            if not fused and not bias_done:     # bias_done: the producer of dy already accumulated its column sums
                L.colsum_accum(dy, gb, M, N, N, bdt)

        if self._side is None:
            wgrad()
        else:
            # dy and x are complete on the main stream here; the side stream reads them while the main stream goes on
            # with the input-gradient chain.  Buffers are only re-written in the next layer (after _layer_boundary) or,
            # for the incoming-gradient buffer, after _wait_first_wgrad().
            main = torch.cuda.current_stream()
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                wgrad()
                if self._first_wgrad_ev is None:
                    self._first_wgrad_ev = torch.cuda.Event()
                    self._first_wgrad_ev.record(self._side)
        if dx is not None:
            w = self.Wb(wkey, wnumel)
            fuse = gelu_h is not None and FUSE_GELU_GRAD
            # fused: dh = (dy W) * gelu'(h) in the epilogue (h holds gelu' then) and the FFN1 bias gradient = its column sums
            L.gemm(dy, w, dx, addend=addend, preact=(gelu_h if fuse else None), M=M, N=K, K=N, lda=N, ldb=K, ldc=K,
                   a_layout=L.ROWK, b_layout=L.KROW, act=(L.ACT_MUL if fuse else L.ACT_NONE), dtype=bdt,
                   impl=self.gemm_impl, colsum=(gelu_db if fuse else None))
            if gelu_h is not None and not fuse:
                if gelu_db is not None:     # dh = dg * gelu'(h) and the FFN1 bias gradient (column sums of dh) in one pass
                    L.gelu_bwd_colsum(dx, gelu_h, dx, gelu_db, M, K, K, bdt)
                else:
                    L.gelu_bwd(dx, gelu_h, dx, M * K, bdt)

    def _lin_bwd_x3(self, dy, x, wkey, dx, M, N, K, Mk, gw, gb, addend, wnumel, gelu_h, bias_done, gelu_db):
        """bf16x3 form of _lin_bwd: dy is split once for both products, x's planes are the ones its forward GEMM made"""
        dyp = self._split(dy, M, N)
        xp = self._split(x, M, K, reuse=True)
        # pass sets of the backward products (csrc/gemm_x3.hip): 3 = hi.hi + lo.hi + hi.lo; 2 drops the lo plane of the non-gradient
        # operand (x for the weight gradient, W for the input gradient: that operand enters at bf16 precision); 1 = hi.hi only
        wg_a, wg_b = (dyp, xp) if X3_WGRAD_PASSES >= 3 else ((dyp, (xp[0], None)) if X3_WGRAD_PASSES == 2 else ((dyp[0], None), (xp[0], None)))

        def wgrad():
            if SPLITK_WGRAD and self.gemm_impl == L.IMPL_AUTO and L.gemm_splitk_ok(N, K, Mk, L.BF16):
                L.gemm_splitk_x3(wg_a, wg_b, gw, self._splitk_ws(), M=N, N=K, K=Mk, lda=N, ldb=K, ldc=K)
            else:
                L.gemm_x3(wg_a, wg_b, gw, M=N, N=K, K=Mk, lda=N, ldb=K, ldc=K, a_layout=L.KROW, b_layout=L.KROW, accum=True)
            if not bias_done:
                if isinstance(dy, Planes):
                    raise L.MmrcaError("bf16x3: the bias gradient of a two-plane dY must come from its producer (bias_done)")
                L.colsum_accum(dy, gb, M, N, N, self.dt)

        if self._side is None:
            wgrad()
        else:
            main = torch.cuda.current_stream()
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                wgrad()
                if self._first_wgrad_ev is None:
                    self._first_wgrad_ev = torch.cuda.Event()
                    self._first_wgrad_ev.record(self._side)
        if dx is not None:
            fuse = gelu_h is not None and FUSE_GELU_GRAD
            po = isinstance(dx, Planes)
            if po and gelu_h is not None and not fuse:
                raise L.MmrcaError("bf16x3 mode with a two-plane FFN gradient needs the fused GELU gradient (MMRCA_FUSE_GELU=1)")
            wp = self.Wx3(wkey, wnumel)
            dg_a, dg_b = (dyp, wp) if X3_DGRAD_PASSES >= 3 else ((dyp, (wp[0], None)) if X3_DGRAD_PASSES == 2 else ((dyp[0], None), (wp[0], None)))
            L.gemm_x3(dg_a, dg_b, dx.hi if po else dx, C_lo=(dx.lo if po else None), addend=addend,
                      preact=(gelu_h if fuse else None), colsum=(gelu_db if fuse else None),
                      M=M, N=K, K=N, lda=N, ldb=K, ldc=K, a_layout=L.ROWK, b_layout=L.KROW, act=(L.ACT_MUL if fuse else L.ACT_NONE),
                      impl=self.gemm_impl)
            if gelu_h is not None and not fuse:
                if gelu_db is not None:
                    L.gelu_bwd_colsum(dx, gelu_h, dx, gelu_db, M, K, K, self.dt)
                else:
                    L.gelu_bwd(dx, gelu_h, dx, M * K, self.dt)

    def _ln_fwd(self, x, res, pfx, sum_out, y, mean, rstd, rows, D, eps, ld_x=None, ld_y=None, in_drop=(0.0, 0), out_drop=(0.0, 0),
                to_gemm=False):
        """to_gemm (bf16x3 mode): the output feeds a GEMM, so the kernel also writes it as two bf16 planes -- ONLY as planes when
        `y` is a Planes buffer (pre-LN encoders: nothing else reads it); the next _lin_fwd on `y` picks them up instead of
        running a split pass."""
        yp = None
        if isinstance(y, Planes):
            yp, y = (y.hi, y.lo), None
        elif self.x3 and to_gemm and self._ln_planes and (ld_y is None or ld_y == D):
            key, yp = self._planes_of(y)
            self._plane_fresh.add(key)
        L.add_layernorm_fwd(x, res, self.W(pfx + ".weight"), self.W(pfx + ".bias"), sum_out, y, mean, rstd, rows, D,
                            ld_x or D, ld_y or D, eps, self.dt, in_drop=in_drop, out_drop=out_drop, y_planes=yp)

    def _ln_bwd(self, dy, s, pfx, mean, rstd, dres, ds, rows, D, ld_dy=None, ld_s=None, ld_ds=None, dy_drop=(0.0, 0),
                branch_drop=(0.0, 0), dbranch=None, dcol=None, dcol_branch=None):
        if self.x3f:        # bf16 gradients against the fp32 residual stream the forward saved
            L.layernorm_bwd_mixed(dy, s, self.Wb(pfx + ".weight"), mean, rstd, dres, ds, self.G(pfx + ".weight"), self.G(pfx + ".bias"),
                                  rows, D, ld_dy or D, ld_s or D, ld_ds or D, dy_drop=dy_drop, branch_drop=branch_drop, dbranch=dbranch,
                                  dcol=dcol, dcol_branch=dcol_branch)
            return
        L.layernorm_bwd(dy, s, self.W(pfx + ".weight"), mean, rstd, dres, ds, self.G(pfx + ".weight"), self.G(pfx + ".bias"),
                        rows, D, ld_dy or D, ld_s or D, ld_ds or D, self.dt, dy_drop=dy_drop, branch_drop=branch_drop, dbranch=dbranch,
                        dcol=dcol, dcol_branch=dcol_branch)

    def _attn_bwd_operands(self, a, rows, D, name, cap, B):
        """(qkv, context) as the attention backward reads them: the saved tensors, or in bf16x3f their bf16 roundings (q|k|v through
        one scratch buffer of `cap` rows shared by all layers, the context's hi plane); a["tail"]: the class-token context [B, D]"""
        tail = bool(a.get("tail"))
        c = a["ctx_c"] if tail else a["ctx"]
        if not self.x3f:
            return a["qkv"], c
        if isinstance(a["qkv"], Planes):
            return a["qkv"].hi, self._hi(c, B if tail else rows, D)
        qkv16 = self.buf(name + "_qkv16", cap, 3 * D, torch.bfloat16)
        L.cast_f32_to_bf16(a["qkv"], qkv16, rows * 3 * D)
        return qkv16, self._hi(c, B if tail else rows, D)

    def _qkv_planes_ok(self, S, dh):
        """bf16x3f: the q|k|v projection is written as two bf16 planes -- the fp32 attention forward rebuilds the fp32 values from
        them and the bf16 attention backward reads the hi plane, so no fp32 copy and no cast pass exist"""
        return self.x3f and self._ln_planes and self.attn_impl == L.IMPL_AUTO and L.mha_fwd_planes_ok(S, dh)

    def _mha_fwd(self, qkv, mask32, ctx, lse, B, H, S, dh, drop_p=0.0, drop_seed=0, cu=None):
        if isinstance(qkv, Planes):
            key, pl = self._planes_of(ctx)
            if L.mha_fwd_x3_ok(S, dh):      # three-pass products on the bf16 matrix cores (the fp32 pipe is 1/16 of their rate)
                L.mha_fwd_x3((qkv.hi, qkv.lo), mask32, pl, lse, B, H, S, dh, dh ** -0.5, drop_p=drop_p, drop_seed=drop_seed, cu=cu)
            else:
                L.mha_fwd_planes_in((qkv.hi, qkv.lo), mask32, None, pl, lse, B, H, S, dh, dh ** -0.5, drop_p=drop_p, drop_seed=drop_seed, cu=cu)
            self._plane_fresh.add(key)
            return
        if self.x3 and self._ln_planes and self.attn_impl == L.IMPL_AUTO and L.mha_fwd_planes_ok(S, dh):
            # the context feeds the out-projection GEMM: written as two bf16 planes next to the fp32 copy the backward reads
            key, pl = self._planes_of(ctx)
            L.mha_fwd_planes(qkv, mask32, ctx, pl, lse, B, H, S, dh, dh ** -0.5, drop_p=drop_p, drop_seed=drop_seed, cu=cu)
            self._plane_fresh.add(key)
            return
        L.mha_fwd(qkv, mask32, ctx, lse, B, H, S, dh, dh ** -0.5, self.dt, self.attn_impl, drop_p=drop_p, drop_seed=drop_seed, cu=cu)

    def _layer_boundary(self):
        """All side-stream weight-gradient work of the finished layer must be done before the next layer re-writes the
        gradient buffers it read (and before its parameter group is handed to the gradient all-reduce)."""
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
            self._first_wgrad_ev = None

    def _wait_first_wgrad(self):
        """The first weight-gradient GEMM of a layer reads the incoming-gradient buffer that the layer's last op
        overwrites with the outgoing gradient."""
        if self._side is not None and self._first_wgrad_ev is not None:
            torch.cuda.current_stream().wait_event(self._first_wgrad_ev)

    @staticmethod
    def _site_seed(base: int, layer: int, site: int) -> int:
        return (base * 1000003 + layer * 16 + site + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF

    # ------------------------------------------------------------------ text encoder
    def _text_forward(self, ids, mask, save, drop_p: float = 0.0, drop_seed: int = 0, pack: Optional[TextPack] = None):
        s, P = self.ts, "text_model."
        B, T = ids.shape
        cap, D, Fd, H = B * T, s.dim, s.ffn, s.heads          # buffers are sized for the padded layout
        dh = D // H
        ids32 = ids.to(torch.int32).contiguous().view(-1)
        if s.pos_offset:
            nonpad = (ids != s.pad_id).to(torch.int64)
            pos = (torch.cumsum(nonpad, dim=1) * nonpad + s.pad_id).to(torch.int32).contiguous().view(-1)
        else:
            pos = torch.arange(T, device=ids.device, dtype=torch.int32).repeat(B)
        if pack is not None:
            if (pack.B, pack.T) != (B, T):
                raise L.MmrcaError(f"text pack was built for a [{pack.B},{pack.T}] batch, got [{B},{T}]")
            M, cu, first = pack.M, pack.cu, pack.first
            ids32, pos, mask32 = ids32[pack.perm].contiguous(), pos[pack.perm].contiguous(), pack.mask
        else:
            M, cu = cap, None
            first = torch.arange(B, device=ids.device, dtype=torch.int64) * T
            mask32 = mask.to(torch.int32).contiguous()
        fb = lambda name, cols, l=0, dt=None: self.buf("t_" + name, cap, cols, dt, l if save else 0)
        stat = lambda name, l=0: self.vec("t_" + name, _round_up(cap, ROWPAD), torch.float32, l if save else 0)
        emb = fb("emb", D)
        type_row = self.Wflat(P + "embeddings.token_type_embeddings.weight", D) if s.type_vocab else None
        L.embed_fwd(ids32, pos, self.W(P + "embeddings.word_embeddings.weight"), self.W(P + "embeddings.position_embeddings.weight"),
                    type_row, emb, M, D, self.dt)
        x = fb("x", D, 0)
        mean0, rstd0 = stat("mean0"), stat("rstd0")
        dp = float(drop_p)
        sd = lambda layer, site: self._site_seed(drop_seed, layer, site)
        post_attn_drop = dp if s.name != "distilbert" else 0.0     # BertSelfOutput drops the attention output, DistilBERT does not
        self._ln_fwd(emb, None, P + "embeddings.LayerNorm", None, x, mean0, rstd0, M, D, s.ln_eps, out_drop=(dp, sd(0, 0)), to_gemm=True)
        layers = []
        for i in range(s.layers):
            K = S.text_layer_keys(s, i)
            last_tail = CLS_TAIL and i == s.layers - 1          # (the class-token attention reads fp32 q|k|v)
            qkv = self.planes_buf("t_qkv", cap, 3 * D, i if save else 0) if (self._qkv_planes_ok(T, dh) and not last_tail) else fb("qkv", 3 * D, i)
            ctx, lse = fb("ctx", D, i), self.vec("t_lse", _round_up(B * H * T, ROWPAD), torch.float32, i if save else 0)
            self._lin_fwd(x, P + K["q"] + ".weight", P + K["q"] + ".bias", qkv, M, 3 * D, D, wnumel=3 * D * D)
            if CLS_TAIL and i == s.layers - 1:
                # class-token tail: the class-token query's attention, then rows b*T only (see CLS_TAIL above)
                cb = lambda name, cols, dt=None: self.buf("t_" + name + "_c", B, cols, dt, i if save else 0)
                ctx_c, x_c = cb("ctx", D), cb("xin", D)
                lse = self.vec("t_lse_c", _round_up(B * H, ROWPAD), torch.float32, i if save else 0)
                L.mha_cls_fwd(qkv, mask32, ctx_c, lse, B, H, T, dh, dh ** -0.5, self.dt, drop_p=dp, drop_seed=sd(i, 1), cu=cu)
                x_c[:B].copy_(x.index_select(0, first))
                att = cb("att", D)
                self._lin_fwd(ctx_c, P + K["o"] + ".weight", P + K["o"] + ".bias", att, B, D, D)
                s1, x1 = cb("s1", D), cb("x1", D)
                m1, r1 = stat("m1c", i), stat("r1c", i)
                self._ln_fwd(att, x_c, P + K["ln1"], s1, x1, m1, r1, B, D, s.ln_eps, in_drop=(post_attn_drop, sd(i, 2)), to_gemm=True)
                h, g = cb("h", Fd, self._hdt), (self.planes_buf("t_g_c", B, Fd, i if save else 0) if self._g_planes else cb("g", Fd))
                self._lin_fwd(x1, P + K["f1"] + ".weight", P + K["f1"] + ".bias", g, B, Fd, D, act=self._gelu_act, preact=h)
                f = cb("f", D)
                self._lin_fwd(g, P + K["f2"] + ".weight", P + K["f2"] + ".bias", f, B, D, Fd)
                s2, xn = cb("s2", D), cb("xout", D)
                m2, r2 = stat("m2c", i), stat("r2c", i)
                self._ln_fwd(f, x1, P + K["ln2"], s2, xn, m2, r2, B, D, s.ln_eps, in_drop=(dp, sd(i, 3)))
                layers.append(dict(x=x, qkv=qkv, ctx=ctx, lse=lse, s1=s1, x1=x1, m1=m1, r1=r1, h=h, g=g, s2=s2, m2=m2, r2=r2,
                                   ctx_c=ctx_c, tail=True))
                cls = xn[:B].clone()
                return cls, dict(B=B, T=T, M=M, cu=cu, first=first, ids32=ids32, pos=pos, mask32=mask32, emb=emb, mean0=mean0, rstd0=rstd0, layers=layers,
                                 drop_p=dp, drop_seed=drop_seed)
            self._mha_fwd(qkv, mask32, ctx, lse, B, H, T, dh, drop_p=dp, drop_seed=sd(i, 1), cu=cu)
            att = fb("tmpD", D)
            self._lin_fwd(ctx, P + K["o"] + ".weight", P + K["o"] + ".bias", att, M, D, D)
            s1, x1 = fb("s1", D, i), fb("x1", D, i)
            m1, r1 = stat("m1", i), stat("r1", i)
            self._ln_fwd(att, x, P + K["ln1"], s1, x1, m1, r1, M, D, s.ln_eps, in_drop=(post_attn_drop, sd(i, 2)), to_gemm=True)
            h, g = fb("h", Fd, i, self._hdt), (self.planes_buf("t_g", cap, Fd, i if save else 0) if self._g_planes else fb("g", Fd, i))
            self._lin_fwd(x1, P + K["f1"] + ".weight", P + K["f1"] + ".bias", g, M, Fd, D, act=self._gelu_act, preact=h)   # fused mode: h <- gelu'(pre-activation)
            f = fb("tmpD", D)
            self._lin_fwd(g, P + K["f2"] + ".weight", P + K["f2"] + ".bias", f, M, D, Fd)
            s2, xn = fb("s2", D, i), fb("x", D, i + 1)
            m2, r2 = stat("m2", i), stat("r2", i)
            self._ln_fwd(f, x1, P + K["ln2"], s2, xn, m2, r2, M, D, s.ln_eps, in_drop=(dp, sd(i, 3)), to_gemm=(i + 1 < s.layers))
            layers.append(dict(x=x, qkv=qkv, ctx=ctx, lse=lse, s1=s1, x1=x1, m1=m1, r1=r1, h=h, g=g, s2=s2, m2=m2, r2=r2))
            x = xn
        cls = x.index_select(0, first)
        return cls, dict(B=B, T=T, M=M, cu=cu, first=first, ids32=ids32, pos=pos, mask32=mask32, emb=emb, mean0=mean0, rstd0=rstd0, layers=layers,
                         drop_p=dp, drop_seed=drop_seed)

    def _text_backward(self, dcls, sv):
        s, P = self.ts, "text_model."
        B, T = sv["B"], sv["T"]
        M, D, Fd, H = sv["M"], s.dim, s.ffn, s.heads
        dh, cu, first = D // H, sv["cu"], sv["first"]
        gb = lambda name, cols: self.buf("tg_" + name, B * T, cols, self.gdtype)
        gplanes = self._g_planes and not self.x3f
        dx = gb("dxA", D)
        tail = bool(sv["layers"][-1].get("tail"))
        if not tail:
            dx[:M].zero_()
            dx.index_copy_(0, first, dcls.to(dx.dtype))
        for i in reversed(range(s.layers)):
            if tail and i == s.layers - 1:
                self._text_backward_tail(dcls, sv, dx)
                continue
            K, a = S.text_layer_keys(s, i), sv["layers"][i]
            ds2 = gb("ds2", D)
            dp, sd = sv["drop_p"], (lambda layer, site: self._site_seed(sv["drop_seed"], layer, site))
            post_attn_drop = dp if s.name != "distilbert" else 0.0
            df = gb("dbr_ffn", D) if dp > 0 else None      # gradient of the dropped FFN branch
            gb_f2, gb_o = self.G(P + K["f2"] + ".bias"), self.G(P + K["o"] + ".bias")
            # the LayerNorm backward also emits the column sums of its output = bias gradient of the linear that fed it
            self._ln_bwd(dx, a["s2"], P + K["ln2"], a["m2"], a["r2"], None, ds2, M, D, branch_drop=(dp, sd(i, 3)), dbranch=df,
                         dcol=(None if df is not None else gb_f2), dcol_branch=(gb_f2 if df is not None else None))
            dg = self.planes_buf("tg_dF", B * T, Fd) if gplanes else gb("dF", Fd)
            self._lin_bwd(df if df is not None else ds2, a["g"], P + K["f2"] + ".weight", P + K["f2"] + ".bias", dg, M, D, Fd,
                          gelu_h=a["h"], bias_done=True, gelu_db=self.G(P + K["f1"] + ".bias"))
            dx1 = gb("dxB", D)
            self._lin_bwd(dg, a["x1"], P + K["f1"] + ".weight", P + K["f1"] + ".bias", dx1, M, Fd, D, addend=ds2, bias_done=True)
            ds1 = gb("ds1", D)
            datt = gb("dbr_att", D) if post_attn_drop > 0 else None
            self._ln_bwd(dx1, a["s1"], P + K["ln1"], a["m1"], a["r1"], None, ds1, M, D, branch_drop=(post_attn_drop, sd(i, 2)), dbranch=datt,
                         dcol=(None if datt is not None else gb_o), dcol_branch=(gb_o if datt is not None else None))
            dctx = gb("dctx", D)
            self._lin_bwd(datt if datt is not None else ds1, a["ctx"], P + K["o"] + ".weight", P + K["o"] + ".bias", dctx, M, D, D,
                          bias_done=True)
            dqkv = gb("dqkv", 3 * D)
            # the attention backward also reduces the q|k|v bias gradients (adjacent in the arena) while it has the tiles
            qkv_b, ctx_b = self._attn_bwd_operands(a, M, D, "tg", B * T, B)
            L.mha_bwd(qkv_b, sv["mask32"], ctx_b, dctx, a["lse"], dqkv, B, H, T, dh, dh ** -0.5, self.bdt, self.attn_impl,
                      drop_p=dp, drop_seed=sd(i, 1), cu=cu)
            self._wait_first_wgrad()       # (the FFN2 weight gradient does not read dx in the post-LN layout; harmless)
            self._lin_bwd(dqkv, a["x"], P + K["q"] + ".weight", P + K["q"] + ".bias", dx, M, 3 * D, D, addend=ds1, wnumel=3 * D * D,
                          fuse_db=QKV_BIAS_IN_WGRAD)
            self._layer_boundary()
            self._ready(f"text_layer_{i}")
        ds0 = gb("ds2", D)
        self._ln_bwd(dx, sv["emb"], P + "embeddings.LayerNorm", sv["mean0"], sv["rstd0"], None, ds0, M, D,
                     dy_drop=(sv["drop_p"], self._site_seed(sv["drop_seed"], 0, 0)))
        dtype_row = self.Gflat(P + "embeddings.token_type_embeddings.weight", D) if s.type_vocab else None
        L.embed_bwd(ds0, sv["ids32"], sv["pos"], self.G(P + "embeddings.word_embeddings.weight"),
                    self.G(P + "embeddings.position_embeddings.weight"), dtype_row, M, D, self.bdt,
                    pad_id=s.pad_id, pos_pad_id=(s.pad_id if s.pos_offset else -1))
        self._layer_boundary()
        self._ready("text_emb", flush=True)

    def _text_backward_tail(self, dcls, sv, dx):
        """Backward of the LAST text layer when its post-attention part ran on the class-token rows only (CLS_TAIL)."""
        s, P = self.ts, "text_model."
        B, T = sv["B"], sv["T"]
        M, D, Fd, H = sv["M"], s.dim, s.ffn, s.heads
        dh, i, cu, first = D // H, s.layers - 1, sv["cu"], sv["first"]
        K, a = S.text_layer_keys(s, i), sv["layers"][i]
        gb = lambda name, cols: self.buf("tg_" + name, B * T, cols, self.gdtype)
        gc = lambda name, cols: self.buf("tg_" + name + "_c", B, cols, self.gdtype)
        dp, sd = sv["drop_p"], (lambda layer, site: self._site_seed(sv["drop_seed"], layer, site))
        post_attn_drop = dp if s.name != "distilbert" else 0.0
        dxc = gc("dx", D)
        dxc[:B].copy_(dcls)
        ds2 = gc("ds2", D)
        df = gc("dbr_ffn", D) if dp > 0 else None
        gb_f2, gb_o = self.G(P + K["f2"] + ".bias"), self.G(P + K["o"] + ".bias")
        self._ln_bwd(dxc, a["s2"], P + K["ln2"], a["m2"], a["r2"], None, ds2, B, D, branch_drop=(dp, sd(i, 3)), dbranch=df,
                     dcol=(None if df is not None else gb_f2), dcol_branch=(gb_f2 if df is not None else None))
        dg = self.planes_buf("tg_dF_c", B, Fd) if (self._g_planes and not self.x3f) else gc("dF", Fd)
        self._lin_bwd(df if df is not None else ds2, a["g"], P + K["f2"] + ".weight", P + K["f2"] + ".bias", dg, B, D, Fd,
                      gelu_h=a["h"], bias_done=True, gelu_db=self.G(P + K["f1"] + ".bias"))
        dx1 = gc("dxB", D)
        self._lin_bwd(dg, a["x1"], P + K["f1"] + ".weight", P + K["f1"] + ".bias", dx1, B, Fd, D, addend=ds2, bias_done=True)
        ds1 = gc("ds1", D)
        datt = gc("dbr_att", D) if post_attn_drop > 0 else None
        self._ln_bwd(dx1, a["s1"], P + K["ln1"], a["m1"], a["r1"], None, ds1, B, D, branch_drop=(post_attn_drop, sd(i, 2)), dbranch=datt,
                     dcol=(None if datt is not None else gb_o), dcol_branch=(gb_o if datt is not None else None))
        dctx_c = gc("dctx", D)
        self._lin_bwd(datt if datt is not None else ds1, a["ctx_c"], P + K["o"] + ".weight", P + K["o"] + ".bias", dctx_c, B, D, D,
                      bias_done=True)
        # back to all rows: the attention mixes the class-token gradient into every key / value row
        ds1_full = gb("ds1", D)
        ds1_full[:M].zero_()
        ds1_full.index_copy_(0, first, ds1[:B])
        dqkv = gb("dqkv", 3 * D)
        qkv_b, ctx_b = self._attn_bwd_operands(a, M, D, "tg", B * T, B)
        L.mha_cls_bwd(qkv_b, sv["mask32"], ctx_b, dctx_c, a["lse"], dqkv, B, H, T, dh, dh ** -0.5, self.bdt,
                      drop_p=dp, drop_seed=sd(i, 1), cu=cu)
        self._lin_bwd(dqkv, a["x"], P + K["q"] + ".weight", P + K["q"] + ".bias", dx, M, 3 * D, D, addend=ds1_full, wnumel=3 * D * D,
                      fuse_db=QKV_BIAS_IN_WGRAD)
        self._layer_boundary()
        self._ready(f"text_layer_{i}")

    # ------------------------------------------------------------------ vision encoder
    def _vision_forward(self, images, save):
        s, P = self.vs, "image_model."
        B = images.shape[0]
        if tuple(images.shape[1:]) != (3, s.image, s.image):
            raise ValueError(f"images must be [B,3,{s.image},{s.image}], got {tuple(images.shape)}")
        nP, Tn, D, Fd, H = s.tokens - 1, s.tokens, s.dim, s.ffn, s.heads
        dh, M, Kp = D // H, B * Tn, 3 * s.patch * s.patch
        fb = lambda name, rows, cols, l=0, dt=None: self.buf("v_" + name, rows, cols, dt, l if save else 0)
        stat = lambda name, l=0: self.vec("v_" + name, _round_up(M, ROWPAD), torch.float32, l if save else 0)
        images = images.to(torch.float32).contiguous()
        patches = fb("patches", B * nP, Kp)
        L.patchify_fwd(images, patches, B, 3, s.image, s.image, s.patch, self.dt)
        proj = fb("proj", B * nP, D)
        self._lin_fwd(patches, P + "conv_proj.weight", P + "conv_proj.bias", proj, B * nP, D, Kp, wnumel=D * Kp)
        x = fb("x", M, D, 0)
        L.vit_assemble_fwd(proj, self.W(P + "class_token"), self.W(P + "encoder.pos_embedding"), x, B, nP, D, self.dt)
        layers = []
        fuse_res = LN_RESIDUAL and (self.dt == L.BF16 or self.x3)
        pend = None                      # (ffn2 output, x1) of the previous layer whose sum -- this layer's input -- is still to be formed
        for i in range(s.layers):
            Lk = P + f"encoder.layers.encoder_layer_{i}."
            y1, m1, r1 = (self.planes_buf("v_y1", M, D, i if save else 0) if self._ln_planes else fb("y1", M, D, i)), stat("m1", i), stat("r1", i)
            if pend is None:
                self._ln_fwd(x, None, Lk + "ln_1", None, y1, m1, r1, M, D, s.ln_eps)
            else:
                x = fb("x", M, D, i)
                self._ln_fwd(pend[0], pend[1], Lk + "ln_1", x, y1, m1, r1, M, D, s.ln_eps)       # x = ffn2 + x1 (previous layer), y1 = LN(x)
                pend = None
            last_tail = CLS_TAIL and i == s.layers - 1          # (the class-token attention reads fp32 q|k|v)
            qkv = self.planes_buf("v_qkv", M, 3 * D, i if save else 0) if (self._qkv_planes_ok(Tn, dh) and not last_tail) else fb("qkv", M, 3 * D, i)
            ctx = fb("ctx", M, D, i)
            lse = self.vec("v_lse", _round_up(B * H * Tn, ROWPAD), torch.float32, i if save else 0)
            self._lin_fwd(y1, Lk + "self_attention.in_proj_weight", Lk + "self_attention.in_proj_bias", qkv, M, 3 * D, D)
            if CLS_TAIL and i == s.layers - 1:
                # class-token tail: the class-token query's attention, then rows b*Tn only (see CLS_TAIL above)
                cb = lambda name, cols, dt=None: fb(name + "_c", B, cols, i, dt)
                ctx_c, x_c = cb("ctx", D), cb("xin", D)
                lse = self.vec("v_lse_c", _round_up(B * H, ROWPAD), torch.float32, i if save else 0)
                L.mha_cls_fwd(qkv, None, ctx_c, lse, B, H, Tn, dh, dh ** -0.5, self.dt)
                x_c[:B].copy_(x[:M].view(B, Tn, D)[:, 0])
                x1 = cb("x1", D)
                self._lin_fwd(ctx_c, Lk + "self_attention.out_proj.weight", Lk + "self_attention.out_proj.bias", x1, B, D, D, addend=x_c)
                y2, m2, r2 = (self.planes_buf("v_y2_c", B, D, i if save else 0) if self._ln_planes else cb("y2", D)), stat("m2c", i), stat("r2c", i)
                self._ln_fwd(x1, None, Lk + "ln_2", None, y2, m2, r2, B, D, s.ln_eps)
                h, g = cb("h", Fd, self._hdt), (self.planes_buf("v_g_c", B, Fd, i if save else 0) if self._g_planes else cb("g", Fd))
                self._lin_fwd(y2, Lk + "mlp.0.weight", Lk + "mlp.0.bias", g, B, Fd, D, act=self._gelu_act, preact=h)
                xn = cb("xout", D)
                self._lin_fwd(g, Lk + "mlp.3.weight", Lk + "mlp.3.bias", xn, B, D, Fd, addend=x1)
                layers.append(dict(x=x, y1=y1, m1=m1, r1=r1, qkv=qkv, ctx=ctx, lse=lse, x1=x1, y2=y2, m2=m2, r2=r2, h=h, g=g,
                                   ctx_c=ctx_c, tail=True))
                feat = self.buf("v_feat", B, D)
                mf, rf = stat("mf"), stat("rf")
                self._ln_fwd(xn, None, P + "encoder.ln", None, feat, mf, rf, B, D, s.ln_eps)
                return feat[:B], dict(B=B, patches=patches, xL=xn, mf=mf, rf=rf, layers=layers)
            self._mha_fwd(qkv, None, ctx, lse, B, H, Tn, dh)
            x1 = fb("x1", M, D, i)
            y2, m2, r2 = (self.planes_buf("v_y2", M, D, i if save else 0) if self._ln_planes else fb("y2", M, D, i)), stat("m2", i), stat("r2", i)
            if fuse_res:
                ao = fb("attn_o", M, D, 0)
                self._lin_fwd(ctx, Lk + "self_attention.out_proj.weight", Lk + "self_attention.out_proj.bias", ao, M, D, D)
                self._ln_fwd(ao, x, Lk + "ln_2", x1, y2, m2, r2, M, D, s.ln_eps)                 # x1 = attention output + x, y2 = LN(x1)
            else:
                self._lin_fwd(ctx, Lk + "self_attention.out_proj.weight", Lk + "self_attention.out_proj.bias", x1, M, D, D, addend=x)
                self._ln_fwd(x1, None, Lk + "ln_2", None, y2, m2, r2, M, D, s.ln_eps)
            h, g = fb("h", M, Fd, i, self._hdt), (self.planes_buf("v_g", M, Fd, i if save else 0) if self._g_planes else fb("g", M, Fd, i))
            self._lin_fwd(y2, Lk + "mlp.0.weight", Lk + "mlp.0.bias", g, M, Fd, D, act=self._gelu_act, preact=h)   # fused mode: h <- gelu'(pre-activation)
            layers.append(dict(x=x, y1=y1, m1=m1, r1=r1, qkv=qkv, ctx=ctx, lse=lse, x1=x1, y2=y2, m2=m2, r2=r2, h=h, g=g))
            if fuse_res:
                f2 = fb("ffn_o", M, D, 0)
                self._lin_fwd(g, Lk + "mlp.3.weight", Lk + "mlp.3.bias", f2, M, D, Fd)
                pend = (f2, x1)
            else:
                xn = fb("x", M, D, i + 1)
                self._lin_fwd(g, Lk + "mlp.3.weight", Lk + "mlp.3.bias", xn, M, D, Fd, addend=x1)
                x = xn
        feat = self.buf("v_feat", B, D)
        mf, rf = stat("mf"), stat("rf")
        if pend is None:
            self._ln_fwd(x, None, P + "encoder.ln", None, feat, mf, rf, B, D, s.ln_eps, ld_x=Tn * D, ld_y=D)
        else:               # class-token rows only: x = ffn2 + x1 of the top layer at rows b*Tn
            x = fb("x", M, D, s.layers)
            self._ln_fwd(pend[0], pend[1], P + "encoder.ln", x, feat, mf, rf, B, D, s.ln_eps, ld_x=Tn * D, ld_y=D)
        return feat[:B], dict(B=B, patches=patches, xL=x, mf=mf, rf=rf, layers=layers)

    def _vision_backward(self, dfeat, sv):
        s, P = self.vs, "image_model."
        B = sv["B"]
        nP, Tn, D, Fd, H = s.tokens - 1, s.tokens, s.dim, s.ffn, s.heads
        dh, M, Kp = D // H, B * Tn, 3 * s.patch * s.patch
        gb = lambda name, rows, cols: self.buf("vg_" + name, rows, cols, self.gdtype)
        gplanes = self._g_planes and not self.x3f
        dx = gb("dxA", M, D)
        dfe = gb("dfeat", B, D)
        dfe[:B].copy_(dfeat)
        top = P + f"encoder.layers.encoder_layer_{s.layers - 1}."
        tail = bool(sv["layers"][-1].get("tail"))
        if tail:
            dxc = gb("dx_c", B, D)
            self._ln_bwd(dfe, sv["xL"], P + "encoder.ln", sv["mf"], sv["rf"], None, dxc, B, D, dcol=self.G(top + "mlp.3.bias"))
        else:
            dx[:M].zero_()
            # dx is zero except the class-token rows written here, so their column sums are the top layer's mlp.3 bias gradient
            self._ln_bwd(dfe, sv["xL"], P + "encoder.ln", sv["mf"], sv["rf"], None, dx, B, D, ld_dy=D, ld_s=Tn * D, ld_ds=Tn * D,
                         dcol=self.G(top + "mlp.3.bias"))
        self._ready("image_ln")
        for i in reversed(range(s.layers)):
            if tail and i == s.layers - 1:
                self._vision_backward_tail(dxc, sv, dx)
                continue
            Lk, a = P + f"encoder.layers.encoder_layer_{i}.", sv["layers"][i]
            dg = self.planes_buf("vg_dF", M, Fd) if gplanes else gb("dF", M, Fd)
            self._lin_bwd(dx, a["g"], Lk + "mlp.3.weight", Lk + "mlp.3.bias", dg, M, D, Fd, gelu_h=a["h"], bias_done=True,
                          gelu_db=self.G(Lk + "mlp.0.bias"))
            dy2 = gb("dy", M, D)
            self._lin_bwd(dg, a["y2"], Lk + "mlp.0.weight", Lk + "mlp.0.bias", dy2, M, Fd, D, bias_done=True)
            dx1 = gb("dxB", M, D)
            self._ln_bwd(dy2, a["x1"], Lk + "ln_2", a["m2"], a["r2"], dx, dx1, M, D, dcol=self.G(Lk + "self_attention.out_proj.bias"))
            dctx = gb("dctx", M, D)
            self._lin_bwd(dx1, a["ctx"], Lk + "self_attention.out_proj.weight", Lk + "self_attention.out_proj.bias", dctx, M, D, D,
                          bias_done=True)
            dqkv = gb("dqkv", M, 3 * D)
            qkv_b, ctx_b = self._attn_bwd_operands(a, M, D, "vg", M, B)
            L.mha_bwd(qkv_b, None, ctx_b, dctx, a["lse"], dqkv, B, H, Tn, dh, dh ** -0.5, self.bdt, self.attn_impl)
            dy1 = gb("dy", M, D)
            self._lin_bwd(dqkv, a["y1"], Lk + "self_attention.in_proj_weight", Lk + "self_attention.in_proj_bias", dy1, M, 3 * D, D,
                          fuse_db=QKV_BIAS_IN_WGRAD)
            self._wait_first_wgrad()       # mlp.3's weight gradient (side stream) reads dx; this op overwrites it
            below = P + f"encoder.layers.encoder_layer_{i - 1}.mlp.3.bias"
            self._ln_bwd(dy1, a["x"], Lk + "ln_1", a["m1"], a["r1"], dx1, dx, M, D, dcol=(self.G(below) if i > 0 else None))
            self._layer_boundary()
            self._ready(f"image_layer_{i}")
        dproj = gb("dproj", B * nP, D)
        L.vit_assemble_bwd(dx, dproj, self.Gflat(P + "class_token", D), self.Gflat(P + "encoder.pos_embedding", Tn * D), B, nP, D, self.bdt)
        self._lin_bwd(dproj, sv["patches"], P + "conv_proj.weight", P + "conv_proj.bias", None, B * nP, D, Kp, wnumel=D * Kp)
        self._layer_boundary()
        self._ready("image_emb", flush=True)

    def _vision_backward_tail(self, dxc, sv, dx):
        """Backward of the LAST ViT layer when its post-attention part ran on the class-token rows only (CLS_TAIL);
        dxc [B, D] is the gradient at the layer's class-token output, dx [B*Tn, D] receives the gradient at its input."""
        s, P = self.vs, "image_model."
        B = sv["B"]
        Tn, D, Fd, H = s.tokens, s.dim, s.ffn, s.heads
        dh, M, i = D // H, B * Tn, s.layers - 1
        Lk, a = P + f"encoder.layers.encoder_layer_{i}.", sv["layers"][i]
        gb = lambda name, rows, cols: self.buf("vg_" + name, rows, cols, self.gdtype)
        gc = lambda name, cols: self.buf("vg_" + name + "_c", B, cols, self.gdtype)
        dg = self.planes_buf("vg_dF_c", B, Fd) if (self._g_planes and not self.x3f) else gc("dF", Fd)
        self._lin_bwd(dxc, a["g"], Lk + "mlp.3.weight", Lk + "mlp.3.bias", dg, B, D, Fd, gelu_h=a["h"], bias_done=True,
                      gelu_db=self.G(Lk + "mlp.0.bias"))
        dy2 = gc("dy", D)
        self._lin_bwd(dg, a["y2"], Lk + "mlp.0.weight", Lk + "mlp.0.bias", dy2, B, Fd, D, bias_done=True)
        dx1c = gc("dxB", D)
        self._ln_bwd(dy2, a["x1"], Lk + "ln_2", a["m2"], a["r2"], dxc, dx1c, B, D, dcol=self.G(Lk + "self_attention.out_proj.bias"))
        dctx_c = gc("dctx", D)
        self._lin_bwd(dx1c, a["ctx_c"], Lk + "self_attention.out_proj.weight", Lk + "self_attention.out_proj.bias", dctx_c, B, D, D,
                      bias_done=True)
        # back to all rows: the attention mixes the class-token gradient into every key / value row, and the residual
        # stream carries it straight down at the class-token rows
        dx1 = gb("dxB", M, D)
        dx1[:M].zero_()
        dx1[:M].view(B, Tn, D)[:, 0] = dx1c[:B]
        dqkv = gb("dqkv", M, 3 * D)
        qkv_b, ctx_b = self._attn_bwd_operands(a, M, D, "vg", M, B)
        L.mha_cls_bwd(qkv_b, None, ctx_b, dctx_c, a["lse"], dqkv, B, H, Tn, dh, dh ** -0.5, self.bdt)
        dy1 = gb("dy", M, D)
        self._lin_bwd(dqkv, a["y1"], Lk + "self_attention.in_proj_weight", Lk + "self_attention.in_proj_bias", dy1, M, 3 * D, D,
                      fuse_db=QKV_BIAS_IN_WGRAD)
        below = P + f"encoder.layers.encoder_layer_{i - 1}.mlp.3.bias"
        self._ln_bwd(dy1, a["x"], Lk + "ln_1", a["m1"], a["r1"], dx1, dx, M, D, dcol=(self.G(below) if i > 0 else None))
        self._layer_boundary()
        self._ready(f"image_layer_{i}")

    # ------------------------------------------------------------------ whole model
    def _image_forward(self, images, save, bn_train, seed):
        if self.conv is not None:
            feat = self.conv.forward(images, save, bn_train, seed)
            return (feat.to(self.dtype) if feat.dtype != self.dtype else feat), None       # (bf16 conv features under an fp32 head)
        return self._vision_forward(images, save)

    def forward(self, ids, mask, images, drop_p: float = 0.0, seed: int = 0, save: bool = True, enc_drop_p: float = 0.0,
                text_pack: Optional[TextPack] = None, bn_train: bool = False):
        """ids/mask int64 [B,S] and images fp32 [B,3,H,W] in HBM -> logits fp32 [B, n_classes].
        text_pack (``make_text_pack`` of the same batch's host-side mask): run the text encoder on the live tokens only."""
        for t, nm in ((ids, "input ids"), (mask, "attention mask"), (images, "images")):
            if not t.is_cuda:
                raise L.MmrcaError(f"{nm} must be in HBM; the MM-RCA product path has no CPU fallback")
        self.refresh_working_copy()
        self._plane_valid.clear()
        self._plane_fresh.clear()
        B = ids.shape[0]
        main = torch.cuda.current_stream()
        if self._text_stream is not None:
            self._text_stream.wait_stream(main)
            with torch.cuda.stream(self._text_stream):
                cls, tsv = self._text_forward(ids, mask, save, enc_drop_p, int(seed), text_pack)
            feat, vsv = self._image_forward(images, save, bn_train, int(seed))
            main.wait_stream(self._text_stream)
            cls.record_stream(main)
        else:
            cls, tsv = self._text_forward(ids, mask, save, enc_drop_p, int(seed), text_pack)
            feat, vsv = self._image_forward(images, save, bn_train, int(seed))
        logits = torch.empty(B, self.n_classes, dtype=torch.float32, device=self.device)
        # the head's mask seed advances by the same stride per step as the encoders' site seeds: a HIP-graph replay r of this step then
        # draws the masks of step seed + r everywhere (lib.seed_epoch_set)
        head_seed = self._site_seed(int(seed), 255, 0)
        L.head_fwd(feat, cls, self._head_w, logits, B, self.d_img, self.d_txt, self.n_classes, self.reverse, self.mode,
                   float(drop_p), head_seed, self.dt)
        self._saved = dict(B=B, cls=cls, feat=feat, text=tsv, vision=vsv, drop_p=float(drop_p), seed=head_seed, full=save)
        return logits

    def _image_backward(self, dimg, vsv):
        if self.conv is not None:
            self.conv.backward(dimg)
            self._ready("image_emb", flush=True)
        else:
            self._vision_backward(dimg, vsv)

    def backward(self, dlogits, train_text: bool = True, train_image: bool = True):
        sv = self._saved
        if sv is None:
            raise L.MmrcaError("backward() without a forward()")
        if (train_text or train_image) and not sv["full"]:
            raise L.MmrcaError("backward through the encoders needs forward(save=True)")
        B = sv["B"]
        dl = dlogits.to(torch.float32).contiguous()
        dimg = torch.empty(B, self.d_img, dtype=self.dtype, device=self.device) if train_image else None
        dtxt = torch.empty(B, self.d_txt, dtype=self.dtype, device=self.device) if train_text else None
        L.head_bwd(dl, sv["feat"], sv["cls"], self._head_w, self._head_g, dimg, dtxt, B, self.d_img, self.d_txt,
                   self.n_classes, self.reverse, self.mode, sv["drop_p"], sv["seed"], self.dt)
        self._ready("head", flush=not (train_image or train_text))
        main = torch.cuda.current_stream()
        if train_text and self._text_stream is not None:
            # the host queues the text backward first (it is short) on its own stream, then the vision backward on the
            # main stream; the gradient all-reduce hooks see image groups and text groups on their own streams
            self._text_stream.wait_stream(main)
            dtxt.record_stream(self._text_stream)
            if train_image:
                self._side = self._side_v
                self._image_backward(dimg, sv["vision"])
            with torch.cuda.stream(self._text_stream):
                self._side = self._side_t
                self._text_backward(dtxt, sv["text"])
            main.wait_stream(self._text_stream)
        else:
            if train_image:
                self._side = self._side_v
                self._image_backward(dimg, sv["vision"])
            if train_text:
                self._side = self._side_t
                self._text_backward(dtxt, sv["text"])
        self._side = self._side_v
        self.arena.lp_valid = False        # an optimizer step normally follows

    # ------------------------------------------------------------------ flat-arena optimizer steps
    def trainable_spans(self, train_text: bool, train_image: bool):
        """Contiguous [lo, hi) slices of the arenas that an optimizer step / all-reduce must cover."""
        t0, i0, h0 = 0, self.groups["image_emb"][0], self.groups["head"][0]
        spans = []
        if train_text:
            spans.append([t0, i0])
        if train_image:
            if spans and spans[-1][1] == i0:
                spans[-1][1] = h0
            else:
                spans.append([i0, h0])
        if spans and spans[-1][1] == h0:
            spans[-1][1] = self.arena.total
        else:
            spans.append([h0, self.arena.total])
        return [tuple(x) for x in spans]
