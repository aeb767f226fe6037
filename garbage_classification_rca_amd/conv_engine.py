"""Conv image backbones over libmmrca (csrc/conv.hip + the K2 GEMM): EfficientNetV2-M / -L and ShuffleNetV2 x2.0.

What this replaces in the reference: ``eff_net_v2()`` + ``EfficientNetV2MFullFeatureExtractor``
(CVPR_code/multimodal_model.py:11-36, 113-126 -- the image model MM_RCA actually runs; ``main_both.py:259`` forces it) and, for
the generalised ``--image_model`` values of BASELINE.json, torchvision's ``efficientnet_v2_l`` / ``shufflenet_v2_x2_0``
(models.py:261-278).  Parameter / buffer names are those of the reference's ``image_model`` state_dict (the extractor wrapper's
attribute names over torchvision's module tree); the architecture tables are restated from the published torchvision source
(torchvision is not installed here: "torchvision-unpinned", pinned only by the parameter counts the reference quotes at
main_image.py:295,302 and by the CPU restatement in oracle/conv_models.py).

Data layout: NHWC, every activation a row-major [B*H*W, C] matrix (rows padded with zeros to a multiple of 256), so a 1x1
convolution is ``mmrca_gemm`` on the rows and a full 3x3 convolution is ``mmrca_im2row3x3`` + ``mmrca_gemm``.  BatchNorm runs on
batch statistics in training (and updates the running statistics), on running statistics in eval -- the reference calls
``global_model.train()`` in BOTH phases (main_both.py:564, 706), so the frozen phase still normalises with batch statistics.
No torch autograd: forward() saves what backward() needs, backward() accumulates (+=) into the gradient arena.
"""
from __future__ import annotations

import contextlib
from typing import Dict, List, Optional, Tuple

import torch

import os

from . import lib as L

# dense 3x3 convolutions on tap-major patches (see ConvEncoder._tap_major); "0" = torchvision's channel-major order everywhere
TAP_MAJOR = os.environ.get("MMRCA_CONV_TAP_MAJOR", "1") == "1"
PAD_TAP_K = os.environ.get("MMRCA_CONV_PAD_K", "1") == "1"
# ShuffleNetV2 stride-1 units: branch2's first 1x1 convolution reads the second half of the unit's input in place and its input gradient is
# written straight into the unit's input gradient (row pitch = the unit's channel count); "0" restores the split / concat copies
SHUFFLE_INPLACE = os.environ.get("MMRCA_SHUFFLE_INPLACE", "1") == "1"
# dense 3x3 / stride-1 convolutions as implicit GEMMs (csrc/conv_igemm.hip): no patch matrix, BatchNorm moments in the epilogue;
# "0" keeps im2row + GEMM everywhere (the A/B switch of DESIGN 3d)
IGEMM = os.environ.get("MMRCA_CONV_IGEMM", "1") == "1"
# 1x1 convolutions: BatchNorm moments in the GEMM epilogue (mmrca_gemm_bnstats).  OFF by default: measured 604 -> 592 samples/s on
# configs[2] -- the separate moments pass reads z straight after the GEMM wrote it (Infinity-Cache hits, ~5 TB/s) and costs less than the
# extra LDS reduction + barrier per 128x128 tile in the epilogue does.
FUSE_GEMM_BN = os.environ.get("MMRCA_CONV_FUSE_GEMM_BN", "0") == "1"
BN_FLAT = os.environ.get("MMRCA_BN_FLAT", "0") == "1"               # flat BatchNorm reductions (csrc/conv.hip, opt-in): need a 16 MiB workspace
FUSE_SE = os.environ.get("MMRCA_CONV_FUSE_SE", "1") == "1"         # SE backward: dx and the next BatchNorm's backward sums in one pass
# BatchNorm moments + finish in one launch (mmrca_bn_stats_fused): built, parity-tested, measured SLOWER and off -- the device-scope fence every
# workgroup needs in front of its ticket waits for its own atomics to complete (they are fire-and-forget otherwise), and the last workgroup's
# finish is a serial tail: EfficientNetV2-M B = 64 905 -> 692 samples/s, B = 16 482 -> 396, ShuffleNetV2 B = 4 (HIP graph) 655 -> 648
FUSE_BN_FINISH = os.environ.get("MMRCA_CONV_FUSE_BN_FINISH", "0") == "1"
FUSE_SE_MLP = os.environ.get("MMRCA_CONV_FUSE_SE_MLP", "1") == "1"   # squeeze-excitation MLP: one launch forward, two backward
SE_FUSE_MAX = int(os.environ.get("MMRCA_CONV_SE_FUSE_MAX", "160000"))   # c * sq up to which the fused MLP BACKWARD is used (see ConvEncoder._se_fused)
SE_FUSE_MAX_FWD = int(os.environ.get("MMRCA_CONV_SE_FUSE_MAX_FWD", "300000"))   # ... and the forward (sixteen waves per sample)
FUSE_RES = os.environ.get("MMRCA_CONV_FUSE_RES", "1") == "1"       # residual connection inside the block's last BatchNorm pass
IGEMM_DGRAD = os.environ.get("MMRCA_CONV_IGEMM_DGRAD", "1") == "1"
# Weight gradients of the 1x1 and the implicit-GEMM 3x3 convolutions on a SIDE stream, concurrent with the input-gradient chain they are
# not part of (see ConvEncoder._wgrad_stream): "1" always, "0" never, "auto" (default) everywhere but inside a HIP-graph capture --
# measured (EfficientNetV2-M @ 480, bf16x3f, same box): eager B = 64 901.8 -> 927.6 samples/s; the captured B = 16 step 490.9 -> 460-470
# (every fork / join becomes a cross-branch edge of the graph, ~110 per step).  MMRCA_CONV_SIDE_MAXROWS: only layers with at most that
# many output rows (B * Ho * Wo).
SIDE_WGRAD = os.environ.get("MMRCA_CONV_SIDE_WGRAD", "auto")
SIDE_WGRAD = {"0": False, "1": True}.get(SIDE_WGRAD, "auto")
SIDE_MAXROWS = int(os.environ.get("MMRCA_CONV_SIDE_MAXROWS", str(1 << 40)))
# 1x1 weight gradients with at least this many output elements run on the 256x256 split-K slab kernel (mmrca_gemm_splitk on ragged
# output shapes) instead of 128x128 tiles + fp32 atomics.  Isolated (tools/conv_wgrad_bench.py, us, atomics / slabs, B = 64 | B = 16):
# 3072 x 512: 98 / 59 | 60 / 34; 512 x 3072: 89 / 59 | 58 / 34; 176 x 1056: 61 / 60 | 29 / 38; 304 x 1824: 46 / 50 | 24 / 32;
# 160 x 640: 50 / 57 | 26 / 42; 48 x 192 (K = 921,600): 92 / 166 | 33 / 99 -- the slabs pay for whole 256x256 tiles and a second launch,
# which only the widest layers (EfficientNetV2-M stage 7, -L stages 6-7) win back.  "0" = never.
WGRAD_SLAB_MIN_ELEMS = int(os.environ.get("MMRCA_CONV_WGRAD_SLAB_MIN", "1000000"))
# train-mode BatchNorm forward in two launches instead of three: the finish step (sums -> mean / rstd, running statistics) inside the apply
# pass (mmrca_bn_moments + mmrca_bn_act_fwd_fin; bf16, channel counts that are multiples of 8).  OFF by default: parity-green and measured
# at +0.1 .. +2.2 % (EfficientNetV2-M B = 16: 517.3 / 509.5 -> 529.0 / 509.9 samples/s; B = 64 938.0 / 943.6 -> 941.8 / 951.8; ShuffleNetV2 B = 4
# under a HIP graph 654 -> 660; configs[2] 649 -> 647) -- inside the run-to-run noise except for the graph-replayed step: not worth a
# default-on change to the BatchNorm path in this round (tools/bn_fold_ab.sh)
BN_FOLD = os.environ.get("MMRCA_CONV_BN_FOLD", "0") == "1"
SIDE_DW = os.environ.get("MMRCA_CONV_SIDE_DW", "1") == "1"       # ... and the depthwise convolutions' weight gradients with them

ROWPAD = 256

# (fused, expand, kernel, stride, in, out, layers) -- torchvision/models/efficientnet.py, _efficientnet_conf("efficientnet_v2_m" / "_l")
EFFNET_V2 = {
    "eff_v2_medium": dict(sd=0.3, stem=24, cfg=[(1, 1, 3, 1, 24, 24, 3), (1, 4, 3, 2, 24, 48, 5), (1, 4, 3, 2, 48, 80, 5), (0, 4, 3, 2, 80, 160, 7),
                                                (0, 6, 3, 1, 160, 176, 14), (0, 6, 3, 2, 176, 304, 18), (0, 6, 3, 1, 304, 512, 5)]),
    "eff_v2_large": dict(sd=0.5, stem=32, cfg=[(1, 1, 3, 1, 32, 32, 4), (1, 4, 3, 2, 32, 64, 7), (1, 4, 3, 2, 64, 96, 7), (0, 4, 3, 2, 96, 192, 10),
                                               (0, 6, 3, 1, 192, 224, 19), (0, 6, 3, 2, 224, 384, 25), (0, 6, 3, 1, 384, 640, 7)]),
}
SHUFFLE_X2 = dict(repeats=[4, 8, 4], channels=[24, 244, 488, 976, 2048])       # shufflenet_v2_x2_0
CONV_MODELS = {"eff_v2_medium": 1280, "EffNetv2-Medium": 1280, "eff_v2_large": 1280, "shuffle_net": 2048}
_ALIAS = {"EffNetv2-Medium": "eff_v2_medium"}


def _ru(x, m):
    return (x + m - 1) // m * m


class _Unit:
    """conv (1x1 | 3x3 | depthwise 3x3, no bias) -> BatchNorm -> activation.  ``key`` is the torchvision prefix of the
    Conv2dNormActivation (conv = key.0, bn = key.1) or, for ShuffleNetV2's plain Sequentials, (conv key, bn key)."""

    def __init__(self, conv_key, bn_key, cin, cout, k, stride, act, dw=False, eps=1e-3):
        self.conv_key, self.bn_key = conv_key, bn_key
        self.cin, self.cout, self.k, self.stride, self.act, self.dw, self.eps = cin, cout, k, stride, act, dw, eps

    def params(self):
        shp = (self.cout, 1, 3, 3) if self.dw else (self.cout, self.cin, self.k, self.k)
        return [(self.conv_key + ".weight", shp), (self.bn_key + ".weight", (self.cout,)), (self.bn_key + ".bias", (self.cout,))]

    def buffers(self):
        return [(self.bn_key + ".running_mean", (self.cout,)), (self.bn_key + ".running_var", (self.cout,)),
                (self.bn_key + ".num_batches_tracked", ())]


class _SE:
    def __init__(self, key, c, sq):
        self.key, self.c, self.sq = key, c, sq

    def params(self):
        return [(self.key + ".fc1.weight", (self.sq, self.c, 1, 1)), (self.key + ".fc1.bias", (self.sq,)),
                (self.key + ".fc2.weight", (self.c, self.sq, 1, 1)), (self.key + ".fc2.bias", (self.c,))]


class ConvEncoder:
    """Owns the structure of one conv backbone; parameters live in the owner's arenas (``owner.W(key)`` / ``owner.G(key)``
    with the "image_model." prefix), BatchNorm buffers here."""

    def __init__(self, name: str, owner, image_size: int = 224):
        name = _ALIAS.get(name, name)
        if name not in ("eff_v2_medium", "eff_v2_large", "shuffle_net"):
            raise ValueError(f"Wrong image model: {name}")
        self.name, self.o, self.image = name, owner, image_size
        self.dim = CONV_MODELS[name]
        self.blocks: List[dict] = []          # execution plan
        self.sd_probs: List[float] = []
        if name == "shuffle_net":
            self._plan_shuffle()
        else:
            self._plan_effnet()
        self.buffers: Dict[str, torch.Tensor] = {}
        self._nbt_base: Dict[str, int] = {}
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        self._maps: Dict[Tuple, torch.Tensor] = {}
        self.saved = None
        self._sd_p = None
        self._bn_arena, self._bn_off, self._bn_used, self._bn_bwd_seen = None, {}, 0, set()
        self._side, self._side_busy = None, False     # the weight gradients' stream (SIDE_WGRAD), created at the first backward on a GPU
        self._dz_readers: Dict[Tuple, "torch.cuda.Event"] = {}      # dz buffer -> the side-stream launch that read it last
        self.injected_keep = None             # tests: [n_sd_blocks, B] 0/1 keep masks instead of drawing them
        self.n_train_forwards = 0             # = every BatchNorm's num_batches_tracked (written out by sync_buffers())

    # ------------------------------------------------------------------ structure
    def _plan_effnet(self):
        spec = EFFNET_V2[self.name]
        cfg = spec["cfg"]
        names = ["stem.1", "stage1", "stage2", "stage3", "stage4", "stage5", "stage6"]       # extractor attributes (multimodal_model.py:14-21)
        self.stem = _Unit("stem.0.0", "stem.0.1", 3, spec["stem"], 3, 2, L.CONV_SILU)
        total = sum(c[6] for c in cfg)
        bid = 0
        for sname, (fused, expand, k, stride, cin, cout, n) in zip(names, cfg):
            for i in range(n):
                ci, st = (cin if i == 0 else cout), (stride if i == 0 else 1)
                cexp = ci * expand
                P = f"{sname}.{i}.block"
                blk = dict(kind="fused" if fused else "mb", res=(st == 1 and ci == cout), sd=spec["sd"] * bid / total, units=[], se=None, stage=sname)
                if fused:
                    if cexp != ci:
                        blk["units"] = [_Unit(P + ".0.0", P + ".0.1", ci, cexp, 3, st, L.CONV_SILU), _Unit(P + ".1.0", P + ".1.1", cexp, cout, 1, 1, L.CONV_NONE)]
                    else:
                        blk["units"] = [_Unit(P + ".0.0", P + ".0.1", ci, cout, 3, st, L.CONV_SILU)]
                else:
                    j = 0
                    if cexp != ci:
                        blk["units"].append(_Unit(f"{P}.{j}.0", f"{P}.{j}.1", ci, cexp, 1, 1, L.CONV_SILU)); j += 1
                    blk["units"].append(_Unit(f"{P}.{j}.0", f"{P}.{j}.1", cexp, cexp, 3, st, L.CONV_SILU, dw=True)); j += 1
                    blk["se"] = _SE(f"{P}.{j}", cexp, max(1, ci // 4)); j += 1
                    blk["units"].append(_Unit(f"{P}.{j}.0", f"{P}.{j}.1", cexp, cout, 1, 1, L.CONV_NONE))
                self.blocks.append(blk)
                bid += 1
        self.final = _Unit("final_conv.0", "final_conv.1", cfg[-1][5], 1280, 1, 1, L.CONV_SILU)

    def _plan_shuffle(self):
        ch = SHUFFLE_X2["channels"]
        U = lambda ck, bk, ci, co, k, s, act, dw=False: _Unit(ck, bk, ci, co, k, s, act, dw, eps=1e-5)
        self.stem = U("conv1.0", "conv1.1", 3, ch[0], 3, 2, L.CONV_RELU)
        inp = ch[0]
        for sname, rep, oup in zip(("stage2", "stage3", "stage4"), SHUFFLE_X2["repeats"], ch[1:4]):
            bf = oup // 2
            for i in range(rep):
                P = f"{sname}.{i}"
                if i == 0:
                    b1 = [U(P + ".branch1.0", P + ".branch1.1", inp, inp, 3, 2, L.CONV_NONE, dw=True),
                          U(P + ".branch1.2", P + ".branch1.3", inp, bf, 1, 1, L.CONV_RELU)]
                    b2 = [U(P + ".branch2.0", P + ".branch2.1", inp, bf, 1, 1, L.CONV_RELU),
                          U(P + ".branch2.3", P + ".branch2.4", bf, bf, 3, 2, L.CONV_NONE, dw=True),
                          U(P + ".branch2.5", P + ".branch2.6", bf, bf, 1, 1, L.CONV_RELU)]
                    self.blocks.append(dict(kind="shuffle_down", b1=b1, b2=b2, cin=inp, cout=oup, stage=sname))
                else:
                    b2 = [U(P + ".branch2.0", P + ".branch2.1", bf, bf, 1, 1, L.CONV_RELU),
                          U(P + ".branch2.3", P + ".branch2.4", bf, bf, 3, 1, L.CONV_NONE, dw=True),
                          U(P + ".branch2.5", P + ".branch2.6", bf, bf, 1, 1, L.CONV_RELU)]
                    self.blocks.append(dict(kind="shuffle", b2=b2, cin=oup, cout=oup, stage=sname))
            inp = oup
        self.final = U("conv5.0", "conv5.1", inp, ch[4], 1, 1, L.CONV_RELU)

    def _all_units(self):
        yield self.stem
        for b in self.blocks:
            for u in b.get("units", []) + b.get("b1", []) + b.get("b2", []):
                yield u
        yield self.final

    def param_entries(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """(key, shape) of every trainable parameter, in module order (= the oracle's / torchvision's state_dict order)."""
        out = []
        def unit(u):
            out.extend(u.params())
        unit(self.stem)
        for b in self.blocks:
            if b["kind"] in ("fused", "mb"):
                us = list(b["units"])
                if b["se"] is not None:
                    for u in us[:-1]:
                        unit(u)
                    out.extend(b["se"].params())
                    unit(us[-1])
                else:
                    for u in us:
                        unit(u)
            else:
                for u in b.get("b1", []) + b["b2"]:
                    unit(u)
        unit(self.final)
        return out

    def buffer_entries(self):
        return [e for u in self._all_units() for e in u.buffers()]

    def init_buffers(self, device):
        for k, shp in self.buffer_entries():
            if k.endswith("num_batches_tracked"):
                self.buffers[k] = torch.zeros((), dtype=torch.int64, device=device)
            else:
                self.buffers[k] = (torch.ones if k.endswith("running_var") else torch.zeros)(shp, dtype=torch.float32, device=device)

    def sync_buffers(self):
        for k, t in self.buffers.items():
            if k.endswith("num_batches_tracked"):
                t.fill_(self._nbt_base.get(k, 0) + self.n_train_forwards)

    def load_buffers(self, sd: Dict[str, torch.Tensor], prefix: str = ""):
        for k, t in self.buffers.items():
            if prefix + k in sd:
                if k.endswith("num_batches_tracked"):
                    self._nbt_base[k] = int(sd[prefix + k]) - self.n_train_forwards
                t.copy_(torch.as_tensor(sd[prefix + k]).to(t.device, t.dtype))

    # ------------------------------------------------------------------ storage helpers
    # compute dtype of the backbone: the owner's, unless the owner runs its conv encoder in another precision than its transformer
    # encoders (engine.MMRCAEngine in bf16x3f mode: bf16 conv kernels next to the fp32-accurate text encoder)
    @property
    def cdtype(self):
        return getattr(self.o, "conv_dtype", None) or self.o.dtype

    @property
    def cdt(self):
        return L.dtype_code(self.cdtype)

    def buf(self, name, rows, cols, dtype=None):
        dtype = dtype or self.cdtype
        key = (name, rows, cols, dtype)
        t = self._bufs.get(key)
        if t is None:
            t = torch.zeros(_ru(max(rows, 1), ROWPAD), cols, dtype=dtype, device=self.o.device)
            self._bufs[key] = t
        return t

    # BatchNorm statistics ([mean | rstd]) and backward sums ([sum du | sum du xhat]) of EVERY layer live in one fp32 arena that forward()
    # clears with a single fill: the kernels accumulate into them with atomics, and a fill per layer and direction was 2 of the ~18
    # launches a conv -> bn -> act unit costs per step -- the small-batch step is bound by its launch count
    BN_ARENA_FLOATS = 4 << 20

    def _bn_slices(self, u: "_Unit"):
        if self._bn_arena is None:
            self._bn_arena = torch.zeros(self.BN_ARENA_FLOATS, dtype=torch.float32, device=self.o.device)
        off = self._bn_off.get(u.bn_key)
        c = u.cout
        nt = _ru((c + 63) // 64, 4)                          # tickets of the fused statistics launch (mmrca_bn_stats_fused), cleared with the rest
        if off is None:
            off = self._bn_used
            self._bn_used += 7 * _ru(c, 4) + nt             # (16-byte aligned slices; the last three: s1 | s2 | shift of _bn_sums)
            if self._bn_used > self.BN_ARENA_FLOATS:
                raise L.MmrcaError("conv_engine: BatchNorm arena exhausted")
            self._bn_off[u.bn_key] = off
        return self._bn_arena[off: off + 2 * c].view(2, c), self._bn_arena[off + 2 * _ru(c, 4): off + 2 * _ru(c, 4) + 2 * c].view(1, 2 * c)

    def _bn_sums(self, u: "_Unit"):
        """(s1, s2, shift) of the folded forward (BN_FOLD): three fp32 [C] vectors behind the layer's other slices, cleared with them"""
        self._bn_slices(u)
        c4 = _ru(u.cout, 4)
        off = self._bn_off[u.bn_key] + 4 * c4 + _ru((u.cout + 63) // 64, 4)
        return tuple(self._bn_arena[off + k * c4: off + k * c4 + u.cout] for k in range(3))

    def _bn_tickets(self, u: "_Unit"):
        off = self._bn_off[u.bn_key] + 4 * _ru(u.cout, 4)
        return self._bn_arena[off: off + _ru((u.cout + 63) // 64, 4)].view(torch.int32)

    def _bn_scratch(self, u: "_Unit"):
        """the backward sums of u, clear: by forward()'s fill, or -- a second backward over the same forward -- by a fill of its own"""
        scratch = self._bn_slices(u)[1]
        if u.bn_key in self._bn_bwd_seen:
            scratch.zero_()
        self._bn_bwd_seen.add(u.bn_key)
        return scratch

    def _bn_ws(self):
        """16 MiB of fp32 words for the flat BatchNorm reductions (one 64-byte record per streaming thread, csrc/conv.hip)"""
        if not BN_FLAT or self.cdtype != torch.bfloat16:      # the flat reductions are opt-in (MMRCA_BN_FLAT=1): no workspace otherwise
            return None
        return self.buf("tmp.bnws", 4096, 1024, torch.float32)

    def release(self):
        self._bufs.clear()
        self.saved = None

    def W(self, key):
        convW = getattr(self.o, "convW", None)        # (parameter view in the conv encoder's compute dtype when that differs from the owner's)
        return convW("image_model." + key) if convW is not None else self.o.W("image_model." + key)

    def G(self, key):
        return self.o.G("image_model." + key)

    def _cmap(self, name, idx: List[int]):
        t = self._maps.get(name)
        if t is None:
            t = torch.tensor(idx, dtype=torch.int32, device=self.o.device)
            self._maps[name] = t
        return t

    # ------------------------------------------------------------------ conv -> bn -> act
    def _tap_major(self, u: "_Unit") -> bool:
        """dense 3x3 convolutions in bf16 with cin % 8 == 0 run on tap-major patches (k = tap*cin + c: every im2row / col2im
        access is a contiguous 16-byte vector); the 3-channel stem and the fp32 mode keep torchvision's channel-major order"""
        return TAP_MAJOR and self.cdtype == torch.bfloat16 and u.cin % 8 == 0

    def _tap_k(self, u: "_Unit") -> int:
        """Patch width of a tap-major dense 3x3 convolution.  With cout % 128 == 0 the 9*cin columns are padded with zeros to a
        multiple of 128, so that all three GEMMs of the layer qualify for the tiled bf16 MFMA kernels -- forward (contraction
        % 64), input gradient (N = patch width % 128) and weight gradient (N % 128); 9*cin itself is 576 / 864 in
        EfficientNetV2-L's stages 2 / 3, which left them on the general kernel at a fifth of the rate (+11 % / +4 % padded work)."""
        K = 9 * u.cin
        return _ru(K, 128) if (u.cout % 128 == 0 and PAD_TAP_K) else K

    def _igemm(self, u: "_Unit") -> bool:
        """stride-1 dense 3x3 convolutions with cin % 8 == 0 run without a patch matrix (mmrca_conv3x3_fwd / _wgrad); channel counts
        that are not multiples of 32 (EfficientNetV2-M: 24 / 48 / 80) are padded to one per tap in the weight copy only"""
        return IGEMM and self._tap_major(u) and u.k == 3 and not u.dw and u.stride == 1 and u.cin % 8 == 0 and u.cout % 8 == 0

    def _igemm_dgrad(self, u: "_Unit") -> bool:
        """... and their input gradient is the same kernel on dz with flipped, transposed weights when cout % 32 == 0"""
        return IGEMM_DGRAD and self._igemm(u)

    def _tap_weight_igemm(self, u: "_Unit", w, flip=False):
        """weight copy of the implicit-GEMM kernels: [cout, 9, Cp] with Cp = cin rounded up to 32 (zero pad channels); flip: the
        input-gradient form [cin, 9, Cout_p], w'[ci, tap', co] = w[co, ci, 8 - tap']"""
        if flip:
            n, k, src = u.cin, u.cout, w.view(u.cout, u.cin, 9).flip(2).permute(1, 2, 0)
        else:
            n, k, src = u.cout, u.cin, w.view(u.cout, u.cin, 9).transpose(1, 2)
        kp = _ru(k, 32)
        wp = self.buf(f"tmp.wig.{int(flip)}.{n}.{kp}", n, 9 * kp)[:n]           # pad columns stay zero
        wp.view(n, 9, kp)[:, :, :k].copy_(src)
        return wp

    def _tap_weight(self, u: "_Unit", w, pad=True):
        """[cout, cin, 3, 3] -> [cout, 9, cin] (+ zero columns up to _tap_k) copy in the compute dtype (a few KB..1 MB; rebuilt
        at each use: the weights change every optimizer step)"""
        wt = w.view(u.cout, u.cin, 9).transpose(1, 2)
        Kp = self._tap_k(u) if pad else 9 * u.cin
        if Kp == 9 * u.cin:
            return wt.contiguous()
        wp = self.buf(f"tmp.wtap.{u.cout}.{Kp}", u.cout, Kp)[: u.cout]          # pad columns stay zero
        wp[:, : 9 * u.cin].view(u.cout, 9, u.cin).copy_(wt)
        return wp

    def _unit_fwd(self, u: _Unit, x, B, H, Wd, tag, train, save, res=None, rowscale=None, out=None, ld_in=None):
        """x: rows [B*H*W, cin] -> y rows [B*Ho*Wo, cout]; returns (y, Ho, Wo, saved).  With `res` (the block input) the unit is
        the last one of a residual block and writes out = res + rowscale[sample] * y directly (mmrca_bn_act_fwd_res) when the
        fused kernel is built for the dtype; the caller checks `saved["fused_res"]`."""
        dt = self.cdt
        Ho, Wo = (H - 1) // u.stride + 1, (Wd - 1) // u.stride + 1
        rows = B * Ho * Wo
        z = self.buf(tag + ".z", rows, u.cout)
        w = self.W(u.conv_key + ".weight")
        fused_stats = False
        sums = None
        fold = False
        if u.dw:
            L.dwconv3x3_fwd(x, w, z, B, H, Wd, u.cin, u.stride, dt)
        elif u.k == 1:
            if train and FUSE_GEMM_BN and self.o.gemm_impl == L.IMPL_AUTO and L.gemm_bnstats_ok(rows, u.cout, u.cin, dt):
                # BatchNorm moments of z in the GEMM's epilogue: shifted sums per 128-row block (shift = the running mean)
                nsl = (rows + 127) // 128
                sums = (self.buf("tmp.bnsum.s1", nsl, u.cout, torch.float32), self.buf("tmp.bnsum.s2", nsl, u.cout, torch.float32), nsl)
                L.gemm_bnstats(x, w, z, M=rows, N=u.cout, K=u.cin, lda=u.cin, ldb=u.cin, ldc=u.cout, dtype=dt,
                               shift=self.buffers[u.bn_key + ".running_mean"], s1=sums[0], s2=sums[1])
            else:
                # ld_in: x is a column window of a wider row buffer (ShuffleNetV2's second half, read in place instead of through a split copy)
                L.gemm(x, w, z, M=rows, N=u.cout, K=u.cin, lda=(ld_in or u.cin), ldb=u.cin, ldc=u.cout, dtype=dt, impl=self.o.gemm_impl)
        elif self._igemm(u):
            ns = L.conv3x3_stat_slots(B, H, Wd)
            parts = None
            if train:
                parts = (self.buf("tmp.bnpart.mean", ns, u.cout, torch.float32), self.buf("tmp.bnpart.m2", ns, u.cout, torch.float32),
                         self.buf("tmp.bnpart.cnt", 1, ns, torch.float32))
            L.conv3x3_fwd(x, self._tap_weight_igemm(u, w), z, B, H, Wd, u.cin, u.cout, dt, parts)
            fused_stats = parts is not None
        else:
            K = 9 * u.cin
            if self._tap_major(u):
                K = self._tap_k(u)
            col = self.buf("tmp.col", rows, K)
            if self._tap_major(u):
                L.im2row3x3_tap(x, col, B, H, Wd, u.cin, u.stride, K, dt)
                w = self._tap_weight(u, w)
            else:
                L.im2row3x3(x, col, B, H, Wd, u.cin, u.stride, K, dt)
            L.gemm(col, w, z, M=rows, N=u.cout, K=K, lda=K, ldb=K, ldc=u.cout, dtype=dt, impl=self.o.gemm_impl)
        stats, _ = self._bn_slices(u)                                       # [mean | rstd] in the per-step arena (cleared by forward())
        mean, rstd = stats[0], stats[1]
        rm, rv = self.buffers[u.bn_key + ".running_mean"], self.buffers[u.bn_key + ".running_var"]
        if fused_stats:
            L.conv_bn_finish(parts, B, H, Wd, mean, rstd, rm, rv, u.cout, u.eps, 0.1)
        elif sums is not None:
            L.bn_finish_sums(sums[0], sums[1], rm, sums[2], rows, mean, rstd, rm, rv, u.cout, u.eps, 0.1)
        else:
            fold = BN_FOLD and train and not BN_FLAT and not FUSE_BN_FINISH and L.bn_fold_ok(u.cout, u.cout, dt)
            if fold:
                s1, s2, shift = self._bn_sums(u)
                L.bn_moments(z, s1, s2, shift, rows, u.cout, u.cout, dt)
            else:
                fuse = FUSE_BN_FINISH and train and self.cdtype == torch.bfloat16 and u.cout % 8 == 0 and not BN_FLAT
                L.bn_stats(z, mean, rstd, rm, rv, rows, u.cout, u.cout, u.eps, 0.1 if train else 0.0, train, dt, ws=self._bn_ws(), prezeroed=True,
                           tickets=self._bn_tickets(u) if fuse else None)
        fused_res = res is not None and FUSE_RES and self.cdtype == torch.bfloat16 and u.cout % 8 == 0
        if fold:
            y = out if fused_res else self.buf(tag + ".y", rows, u.cout)
            L.bn_act_fwd_fin(z, s1, s2, shift, self.W(u.bn_key + ".weight"), self.W(u.bn_key + ".bias"), res if fused_res else None,
                             rowscale if fused_res else None, y, mean, rstd, rm, rv, rows, u.cout, u.act, Ho * Wo, u.eps, 0.1, dt)
        elif fused_res:
            y = out
            L.bn_act_fwd_res(z, mean, rstd, self.W(u.bn_key + ".weight"), self.W(u.bn_key + ".bias"), res, rowscale, y, rows, u.cout, u.act,
                             Ho * Wo, dt)
        else:
            y = self.buf(tag + ".y", rows, u.cout)
            L.bn_act_fwd(z, mean, rstd, self.W(u.bn_key + ".weight"), self.W(u.bn_key + ".bias"), y, rows, u.cout, u.act, dt)
        return y, Ho, Wo, dict(x=x, z=z, mean=mean, rstd=rstd, H=H, W=Wd, Ho=Ho, Wo=Wo, train=train, fused_res=fused_res, ld_x=(ld_in or u.cin))

    # ------------------------------------------------------------------ weight gradients beside the input-gradient chain
    # A layer's weight gradient reads dz and the saved input and feeds nothing but the optimizer: with SIDE_WGRAD it is launched on a
    # second stream behind the BatchNorm backward that wrote dz, and the chain (input gradient, next layer's BatchNorm backward, ...)
    # goes on beside it.  dz buffers are per (block parity, unit, shape): the next WRITER of one (two blocks further down) waits for the
    # event of its last side-stream reader; the saved inputs live until the next forward.  backward() joins the streams before every
    # hand-over to the gradient exchange and at its end, so nothing outside this class sees the second stream.  Inside a HIP-graph
    # capture the fork / join events become graph edges: the replayed step has the same two branches.
    def _wgrad_slab(self, M: int, N: int, K: int, dt) -> bool:
        """1x1 weight gradient [M = cout, N = cin] over K rows on the split-K slab kernel?  (see WGRAD_SLAB_MIN_ELEMS)"""
        return (WGRAD_SLAB_MIN_ELEMS > 0 and M * N >= WGRAD_SLAB_MIN_ELEMS and self.o.gemm_impl == L.IMPL_AUTO
                and L.gemm_splitk_ragged_ok(M, N, K, dt))

    def _scratch(self, name: str, nbytes: int, per_stream: bool = False):
        """an uninitialised workspace of exactly nbytes (NOT through buf(): its rows are padded to 256, so buf(name, 1, 64 MiB) is a
        16 GiB tensor and, first requested inside a graph capture, a 16 GiB fill node in every replay)"""
        key = ("scratch", name, torch.cuda.current_stream().cuda_stream if per_stream else 0)
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.o.device)
        return t

    def _splitk_ws(self):
        """partial-tile workspace of mmrca_gemm_splitk, one per stream that launches weight gradients (every partial tile is written
        before it is read)"""
        return self._scratch("splitk", L.SPLITK_WS_BYTES, per_stream=True)

    def _dw_ws(self):
        """partial sums of the depthwise weight gradient (csrc/conv.hip: at most 196,608 threads x 288 B, written before they are read)"""
        return self._scratch("dw", 64 << 20, per_stream=True).view(torch.float32)      # (per stream: MMRCA_CONV_SIDE_MAXROWS can leave some layers on the main one)

    def _side_on(self, rows: int) -> bool:
        if not SIDE_WGRAD or rows > SIDE_MAXROWS or torch.device(self.o.device).type != "cuda":
            return False
        if SIDE_WGRAD == "auto" and torch.cuda.is_current_stream_capturing():
            return False
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.o.device)
        return True

    @contextlib.contextmanager
    def _wgrad_stream(self, dz_key, rows: int):
        if not self._side_on(rows):
            yield
            return
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            yield
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._dz_readers[dz_key] = ev
        self._side_busy = True

    def _side_join(self):
        if self._side_busy:
            torch.cuda.current_stream().wait_stream(self._side)
            self._dz_readers.clear()
            self._side_busy = False

    def _unit_bwd(self, u: _Unit, dy, sv, B, need_dx=True, tag="g", sums_ready=False, dx_into=None):
        """dy: gradient at the unit's output rows; returns dx rows (or None).  sums_ready: the BatchNorm-backward sums of this unit are
        already in the shared scratch (the squeeze-excitation backward accumulated them while it wrote dy)."""
        dt = self.cdt
        H, Wd, Ho, Wo = sv["H"], sv["W"], sv["Ho"], sv["Wo"]
        rows = B * Ho * Wo
        dz = self.buf(f"{tag}.dz.{u.cout}", rows, u.cout)
        dz_key = (f"{tag}.dz.{u.cout}", rows, u.cout)
        last_reader = self._dz_readers.pop(dz_key, None)
        if last_reader is not None:                  # a weight gradient on the side stream may still be reading this buffer
            torch.cuda.current_stream().wait_event(last_reader)
        scratch = self._bn_slices(u)[1] if sums_ready else self._bn_scratch(u)      # this layer's backward sums (clear since forward(), or filled by se_dx)
        L.bn_act_bwd(dy, sv["z"], sv["mean"], sv["rstd"], self.W(u.bn_key + ".weight"), self.W(u.bn_key + ".bias"), dz,
                     self.G(u.bn_key + ".weight"), self.G(u.bn_key + ".bias"), scratch, rows, u.cout, u.act, sv["train"], dt,
                     sums_ready=sums_ready, ws=self._bn_ws(), prezeroed=True)
        w, gw = self.W(u.conv_key + ".weight"), self.G(u.conv_key + ".weight")
        rows_in = B * H * Wd
        dx = self.buf(f"{tag}.dx.{u.cin}.{rows_in}", rows_in, u.cin) if need_dx else None
        rows_k = _ru(rows, 64)                       # the contraction of the weight gradient runs over whole 64-row steps (zero pad rows)
        if u.dw and SIDE_DW and self._side_on(rows):
            # (the library runs the input gradient and the weight gradient as separate launches anyway: one call for each)
            with self._wgrad_stream(dz_key, rows):
                L.dwconv3x3_bwd(dz, sv["x"], w, None, gw, B, H, Wd, u.cin, u.stride, dt, ws=self._dw_ws())
            L.dwconv3x3_bwd(dz, sv["x"], w, dx, None, B, H, Wd, u.cin, u.stride, dt)
        elif u.dw:
            L.dwconv3x3_bwd(dz, sv["x"], w, dx, gw, B, H, Wd, u.cin, u.stride, dt, ws=self._dw_ws())
        elif u.k == 1:
            with self._wgrad_stream(dz_key, rows):
                if self._wgrad_slab(u.cout, u.cin, rows_k, dt):
                    # 256x256 tiles, the contraction split over all CUs, fp32 partial tiles through a workspace and a fixed-order
                    # reduction (the ViT's weight-gradient kernel, csrc/gemm256.hip, on ragged output shapes)
                    L.gemm_splitk(dz, sv["x"], gw, self._splitk_ws(), M=u.cout, N=u.cin, K=rows_k, lda=u.cout, ldb=sv.get("ld_x", u.cin), ldc=u.cin)
                else:
                    L.gemm(dz, sv["x"], gw, M=u.cout, N=u.cin, K=rows_k, lda=u.cout, ldb=sv.get("ld_x", u.cin), ldc=u.cin, a_layout=L.KROW,
                           b_layout=L.KROW, accum=True, dtype=dt, impl=self.o.gemm_impl)
            if need_dx:
                ldc = u.cin
                if dx_into is not None:          # (view of a wider row buffer, its row pitch): the input gradient lands in its column window
                    dx, ldc = dx_into
                L.gemm(dz, w, dx, M=rows, N=u.cin, K=u.cout, lda=u.cout, ldb=u.cin, ldc=ldc, a_layout=L.ROWK, b_layout=L.KROW, dtype=dt,
                       impl=self.o.gemm_impl)
        elif self._igemm(u):
            K = 9 * u.cin
            with self._wgrad_stream(dz_key, rows):           # (its scratch is the side stream's own when the stream is in use)
                gwt = self.buf(("gs" if self._side_on(rows) else "g") + f".wtap.{u.cout}.{K}", u.cout, K, torch.float32)[: u.cout]
                gwt.zero_()
                L.conv3x3_wgrad(dz, sv["x"], gwt, B, H, Wd, u.cin, u.cout, dt)
                gw.view(u.cout, u.cin, 9).add_(gwt.view(u.cout, 9, u.cin).transpose(1, 2))
            if need_dx:
                if self._igemm_dgrad(u):
                    # dx = conv3x3(dz, w'), w'[ci, tap', co] = w[co, ci, 8 - tap']
                    L.conv3x3_fwd(dz, self._tap_weight_igemm(u, w, flip=True), dx, B, H, Wd, u.cout, u.cin, dt)
                else:
                    Kp = self._tap_k(u)
                    col = self.buf("tmp.col", rows, Kp)
                    L.gemm(dz, self._tap_weight(u, w), col, M=rows, N=Kp, K=u.cout, lda=u.cout, ldb=Kp, ldc=Kp, a_layout=L.ROWK,
                           b_layout=L.KROW, dtype=dt, impl=self.o.gemm_impl)
                    L.col2im3x3_tap(col, dx, B, H, Wd, u.cin, u.stride, Kp, dt)
        else:
            tap = self._tap_major(u)
            K = self._tap_k(u) if tap else 9 * u.cin
            col = self.buf("tmp.col", rows, K)
            if tap:
                # tap-major patches: the weight gradient comes out as [cout, 9, cin]; it is summed into a zeroed scratch and
                # added to the arena's [cout, cin, 3, 3] gradient through a permuted view
                L.im2row3x3_tap(sv["x"], col, B, H, Wd, u.cin, u.stride, K, dt)
                gwt = self.buf(f"g.wtap.{u.cout}.{K}", u.cout, K, torch.float32)[: u.cout]
                gwt.zero_()
                L.gemm(dz, col, gwt, M=u.cout, N=K, K=rows_k, lda=u.cout, ldb=K, ldc=K, a_layout=L.KROW, b_layout=L.KROW, accum=True,
                       dtype=dt, impl=self.o.gemm_impl)
                gw.view(u.cout, u.cin, 9).add_(gwt[:, : 9 * u.cin].view(u.cout, 9, u.cin).transpose(1, 2))
                w = self._tap_weight(u, w)
            else:
                L.im2row3x3(sv["x"], col, B, H, Wd, u.cin, u.stride, K, dt)
                L.gemm(dz, col, gw, M=u.cout, N=K, K=rows_k, lda=u.cout, ldb=K, ldc=K, a_layout=L.KROW, b_layout=L.KROW, accum=True,
                       dtype=dt, impl=self.o.gemm_impl)
            if need_dx:
                L.gemm(dz, w, col, M=rows, N=K, K=u.cout, lda=u.cout, ldb=K, ldc=K, a_layout=L.ROWK, b_layout=L.KROW, dtype=dt,
                       impl=self.o.gemm_impl)
                (L.col2im3x3_tap if tap else L.col2im3x3)(col, dx, B, H, Wd, u.cin, u.stride, K, dt)
        return dx

    # ------------------------------------------------------------------ squeeze-excitation
    @staticmethod
    def _se_fused(se: "_SE", limit: int) -> bool:
        """the fused squeeze-excitation MLP (one workgroup per sample re-reads both weight matrices, and all B <= 128 workgroups run side
        by side, so a launch lasts as long as one sample) wins while the matrices are small -- and whenever the squeeze width is not a
        multiple of 8 (EfficientNetV2-M: 20 / 44 / 76 -> the general GEMM kernel).  Measured (tools/se_bench.py, us, fused vs GEMM
        sequence): forward 1824 x 76: 23 vs 75, 2304 x 96: 32 vs 44, 3072 x 128: 51 vs 46, 3840 x 160: 79 vs 62; backward 1824 x 76: 48 vs 137,
        2304 x 96: 76 vs 66.  Forward and backward choose independently (both forms save the same tensors)."""
        return FUSE_SE_MLP and se.c % 4 == 0 and se.sq % 4 == 0 and (se.c * se.sq <= limit or se.sq % 8 != 0)

    def _se_fwd(self, se: _SE, x, B, HW, tag, save):
        dt = self.cdt
        n = lambda s, r, c: self.buf(tag + s, r, c)
        pooled, h_pre, h = n(".se.pool", B, se.c), n(".se.hpre", B, se.sq), n(".se.h", B, se.sq)
        s_pre, s = n(".se.spre", B, se.c), n(".se.s", B, se.c)
        L.rowpool_mean(x, pooled, B, HW, se.c, dt)
        if self._se_fused(se, SE_FUSE_MAX_FWD):        # fc1 + SiLU + fc2 + sigmoid in one launch (four M = B GEMM-shaped launches otherwise)
            L.se_mlp_fwd(pooled, self.W(se.key + ".fc1.weight"), self.W(se.key + ".fc1.bias"), self.W(se.key + ".fc2.weight"),
                         self.W(se.key + ".fc2.bias"), h_pre, h, s_pre, s, B, se.c, se.sq, dt)
        else:
            L.gemm(pooled, self.W(se.key + ".fc1.weight"), h_pre, M=B, N=se.sq, K=se.c, lda=se.c, ldb=se.c, ldc=se.sq, dtype=dt, impl=self.o.gemm_impl)
            L.bias_act_fwd(h_pre, self.W(se.key + ".fc1.bias"), h, B, se.sq, L.CONV_SILU, dt)
            L.gemm(h, self.W(se.key + ".fc2.weight"), s_pre, M=B, N=se.c, K=se.sq, lda=se.sq, ldb=se.sq, ldc=se.c, dtype=dt, impl=self.o.gemm_impl)
            L.bias_act_fwd(s_pre, self.W(se.key + ".fc2.bias"), s, B, se.c, L.CONV_SIGMOID, dt)
        y = self.buf(tag + ".se.y", B * HW, se.c)
        L.se_scale_fwd(x, s, y, B, HW, se.c, dt)
        return y, dict(x=x, pooled=pooled, h_pre=h_pre, h=h, s_pre=s_pre, s=s, HW=HW)

    def _se_bwd(self, se: _SE, dy, sv, B, gp="g", bn=None):
        """returns (dx, sums_ready).  bn = (unit, saved) of the BatchNorm + activation that produced the block's SE input: with the
        fused kernels (bf16, c % 8 == 0) its backward sums are accumulated while dx is written (mmrca_se_dx)."""
        dt = self.cdt
        HW = sv["HW"]
        g = lambda s, r, c: self.buf(gp + ".se" + s + f".{c}", r, c)
        dx, ds = self.buf(f"{gp}.se.dx.{se.c}.{B * HW}", B * HW, se.c), g(".ds", B, se.c)
        fused = FUSE_SE and self.cdtype == torch.bfloat16 and se.c % 8 == 0
        L.se_scale_bwd(dy, sv["x"], sv["s"], None if fused else dx, ds, B, HW, se.c, dt)
        ds_pre, dh_pre, dpool = g(".dspre", B, se.c), g(".dhpre", B, se.sq), g(".dpool", B, se.c)
        if self._se_fused(se, SE_FUSE_MAX):        # the whole MLP backward: a per-sample chain + one launch for the batch sums (six launches otherwise)
            L.se_mlp_bwd(ds, sv["pooled"], sv["h_pre"], sv["h"], sv["s_pre"], self.W(se.key + ".fc1.weight"), self.W(se.key + ".fc1.bias"),
                         self.W(se.key + ".fc2.weight"), self.W(se.key + ".fc2.bias"), ds_pre, dh_pre, dpool, self.G(se.key + ".fc1.weight"),
                         self.G(se.key + ".fc1.bias"), self.G(se.key + ".fc2.weight"), self.G(se.key + ".fc2.bias"), B, se.c, se.sq, dt)
        else:
            L.bias_act_bwd(ds, sv["s_pre"], self.W(se.key + ".fc2.bias"), ds_pre, self.G(se.key + ".fc2.bias"), B, se.c, L.CONV_SIGMOID, dt)
            Bk = _ru(B, 64)
            L.gemm(ds_pre, sv["h"], self.G(se.key + ".fc2.weight"), M=se.c, N=se.sq, K=Bk, lda=se.c, ldb=se.sq, ldc=se.sq, a_layout=L.KROW,
                   b_layout=L.KROW, accum=True, dtype=dt, impl=self.o.gemm_impl)
            dh = g(".dh", B, se.sq)
            L.gemm(ds_pre, self.W(se.key + ".fc2.weight"), dh, M=B, N=se.sq, K=se.c, lda=se.c, ldb=se.sq, ldc=se.sq, a_layout=L.ROWK, b_layout=L.KROW,
                   dtype=dt, impl=self.o.gemm_impl)
            L.bias_act_bwd(dh, sv["h_pre"], self.W(se.key + ".fc1.bias"), dh_pre, self.G(se.key + ".fc1.bias"), B, se.sq, L.CONV_SILU, dt)
            L.gemm(dh_pre, sv["pooled"], self.G(se.key + ".fc1.weight"), M=se.sq, N=se.c, K=Bk, lda=se.sq, ldb=se.c, ldc=se.c, a_layout=L.KROW,
                   b_layout=L.KROW, accum=True, dtype=dt, impl=self.o.gemm_impl)
            L.gemm(dh_pre, self.W(se.key + ".fc1.weight"), dpool, M=B, N=se.c, K=se.sq, lda=se.sq, ldb=se.c, ldc=se.c, a_layout=L.ROWK, b_layout=L.KROW,
                   dtype=dt, impl=self.o.gemm_impl)
        if not fused:
            L.rowpool_mean_bwd(dpool, dx, B, HW, se.c, True, dt)
            return dx, False
        bnargs = None
        if bn is not None and bn[0].cout == se.c and bn[1]["train"]:
            u, usv = bn
            scratch = self._bn_scratch(u)                                   # (clear since forward(); _unit_bwd reads the sums from the same slice)
            bnargs = (usv["z"], usv["mean"], usv["rstd"], self.W(u.bn_key + ".weight"), self.W(u.bn_key + ".bias"), u.act, scratch)
        L.se_dx(dy, sv["s"], dpool, dx, B, HW, se.c, dt, bn=bnargs)
        return dx, bnargs is not None

    # ------------------------------------------------------------------ whole network
    def forward(self, images, save: bool, train: bool, seed: int = 0):
        """images [B,3,H,W] fp32 in HBM -> features [B, dim] (the pooled vector MM_RCA consumes, multimodal_model.py:659)."""
        dt = self.cdt
        B, C, H, Wd = images.shape
        if C != 3:
            raise ValueError(f"images must be [B,3,H,W], got {tuple(images.shape)}")
        # the statistics / backward sums of a pending forward(save=True) are views into the arena cleared below: that forward's
        # backward() can no longer run (it fails loudly on saved = None instead of reading zeros)
        self.saved = None
        if self._bn_arena is not None and self._bn_used:
            self._bn_arena[: self._bn_used].zero_()            # (the used prefix only: a few hundred KB of the 16 MiB)
        self._bn_bwd_seen.clear()
        x0 = self.buf("in.rows", B * H * Wd, 3)
        L.nchw_to_rows(images.to(torch.float32).contiguous(), x0, B, 3, H, Wd, dt)
        saved = dict(B=B, blocks=[], train=train)
        x, h, w, saved["stem"] = self._unit_fwd(self.stem, x0, B, H, Wd, "stem", train, save)
        if self.name == "shuffle_net":
            Ho, Wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            c0 = self.stem.cout
            y = self.buf("pool.y", B * Ho * Wo, c0)
            arg = self.buf("pool.arg", B * Ho * Wo, c0, torch.uint8)
            L.maxpool3x3s2_fwd(x, y, arg, B, h, w, c0, dt)
            saved["pool"] = dict(arg=arg, H=h, W=w)
            x, h, w = y, Ho, Wo
        # stochastic depth ("row" mode, train only): one tiny device op draws every block's keep mask
        rowscale = None
        sd_idx = [i for i, b in enumerate(self.blocks) if b.get("res") and b.get("sd", 0.0) > 0.0]
        if train and sd_idx:
            if self._sd_p is None:             # (a constant of the architecture; built once -- a host-to-device copy cannot be captured)
                self._sd_p = torch.tensor([self.blocks[i]["sd"] for i in sd_idx], dtype=torch.float32, device=self.o.device).view(-1, 1)
            p = self._sd_p
            if self.injected_keep is not None:
                rowscale = (self.injected_keep.to(self.o.device, torch.float32) / (1.0 - p)).contiguous()
            else:
                # counter-based draw from (step seed, mask epoch) like the dropout masks: a HIP-graph replay and the eager step of the
                # same index keep the same blocks (--hip_graph does not change which masks a --seed run draws)
                rowscale = self.buf("sd.rowscale", len(sd_idx), B, torch.float32)[: len(sd_idx)]
                L.sd_rowscale(p, rowscale, len(sd_idx), B, self.o._site_seed(int(seed), 254, 0) if hasattr(self.o, "_site_seed") else int(seed))
        sd_pos = {i: j for j, i in enumerate(sd_idx)}
        if train:
            self.n_train_forwards += 1
        for bi, blk in enumerate(self.blocks):
            # without `save` the activations of block bi are only needed by block bi + 1: two alternating buffer pools
            tag = f"b{bi}" if save else f"t{bi % 2}"
            if blk["kind"] in ("fused", "mb"):
                bs = dict(units=[], H=h, W=w, cin=blk["units"][0].cin)
                y, hh, ww = x, h, w
                us = blk["units"]
                rs = rowscale[sd_pos[bi]] if (blk["res"] and rowscale is not None and bi in sd_pos) else None
                for ui, u in enumerate(us):
                    if blk["se"] is not None and ui == len(us) - 1:
                        y, bs["se"] = self._se_fwd(blk["se"], y, B, hh * ww, tag, save)
                    if blk["res"] and ui == len(us) - 1:      # the last unit adds the residual itself when the fused kernel applies
                        y, hh, ww, sv = self._unit_fwd(u, y, B, hh, ww, f"{tag}.u{ui}", train, save, res=x, rowscale=rs,
                                                       out=self.buf(f"{tag}.out", B * hh * ww, u.cout))
                    else:
                        y, hh, ww, sv = self._unit_fwd(u, y, B, hh, ww, f"{tag}.u{ui}", train, save)
                    bs["units"].append(sv)
                if blk["res"]:
                    if not bs["units"][-1]["fused_res"]:
                        out = self.buf(f"{tag}.out", B * hh * ww, us[-1].cout)
                        L.residual_add(x, y, rs, out, B, hh * ww * us[-1].cout, dt)
                        y = out
                    bs["rowscale"] = rs
                x, h, w = y, hh, ww
            elif blk["kind"] == "shuffle_down":
                bs = dict(b1=[], b2=[], H=h, W=w)
                y1, hh, ww = x, h, w
                for ui, u in enumerate(blk["b1"]):
                    y1, hh, ww, sv = self._unit_fwd(u, y1, B, hh, ww, f"{tag}.a{ui}", train, save)
                    bs["b1"].append(sv)
                y2, h2, w2 = x, h, w
                for ui, u in enumerate(blk["b2"]):
                    y2, h2, w2, sv = self._unit_fwd(u, y2, B, h2, w2, f"{tag}.c{ui}", train, save)
                    bs["b2"].append(sv)
                x = self._shuffle_cat(y1, y2, B * hh * ww, blk["cout"], f"{tag}.out")
                h, w = hh, ww
            else:   # stride-1 shuffle unit: x1 passes through, x2 goes through branch2
                Cc = blk["cout"]
                bf = Cc // 2
                rows = B * h * w
                bs = dict(b2=[], H=h, W=w)
                inplace = SHUFFLE_INPLACE and blk["b2"][0].k == 1 and not blk["b2"][0].dw and not (train and FUSE_GEMM_BN)
                if inplace:                      # branch2's first 1x1 convolution reads x[:, bf:] where it lies (row pitch Cc): no split copy
                    y2, h2, w2 = x[:, bf:], h, w
                else:
                    x2 = self.buf(f"{tag}.x2", rows, bf)
                    L.channel_gather(x, self._cmap(("hi", Cc), list(range(bf, Cc))), x2, rows, Cc, bf, bf, 0, dt)
                    y2, h2, w2 = x2, h, w
                for ui, u in enumerate(blk["b2"]):
                    y2, h2, w2, sv = self._unit_fwd(u, y2, B, h2, w2, f"{tag}.c{ui}", train, save, ld_in=(Cc if (inplace and ui == 0) else None))
                    bs["b2"].append(sv)
                bs["inplace"] = inplace
                x = self._shuffle_cat(x, y2, rows, Cc, f"{tag}.out", first_is_full=True)
            saved["blocks"].append(bs)
        y, h, w, saved["final"] = self._unit_fwd(self.final, x, B, h, w, "final", train, save)
        feat = self.buf("feat", B, self.dim)
        L.rowpool_mean(y, feat, B, h * w, self.dim, dt)
        saved["HW"] = h * w
        self.saved = saved if save else None
        return feat[:B]

    def _shuffle_cat(self, a, b, rows, Cc, name, first_is_full=False):
        """channel_shuffle(cat(a', b), groups=2): out[:, 2j] = a'[:, j], out[:, 2j+1] = b[:, j]  (a' = first half of a when
        first_is_full) -- one launch (round 3: two gathers into a concat buffer and a third for the shuffle)."""
        bf = Cc // 2
        out = self.buf(name + f".{Cc}.{rows}", rows, Cc)
        L.channel_interleave2(a, a.shape[1], b, out, rows, bf, self.cdt)
        return out

    def _shuffle_cat_bwd(self, dout, rows, Cc, gp="g", d1_into=None):
        """inverse of _shuffle_cat in one launch: returns (d first half, d second half [rows, bf]).  d1_into: a [rows, Cc] buffer whose
        first bf columns receive the first half (the stride-1 unit: that IS the first half of its input gradient)."""
        bf = Cc // 2
        d2 = self.buf(f"{gp}.sh2.{bf}.{rows}", rows, bf)
        if d1_into is not None:
            L.channel_deinterleave2(dout, d1_into, Cc, d2, rows, bf, self.cdt)
            return None, d2
        d1 = self.buf(f"{gp}.sh1.{bf}.{rows}", rows, bf)
        L.channel_deinterleave2(dout, d1, bf, d2, rows, bf, self.cdt)
        return d1, d2

    def backward(self, dfeat):
        """dfeat [B, dim] -> accumulates every parameter gradient of the backbone (no gradient w.r.t. the images)."""
        sv = self.saved
        if sv is None:
            raise L.MmrcaError("conv backbone: backward() needs forward(save=True)")
        dt = self.cdt
        B = sv["B"]
        HW = sv["HW"]
        dpool = self.buf("g.dfeat", B, self.dim)
        dpool[:B].copy_(dfeat.to(dpool.dtype))
        dy = self.buf("g.dfinal", B * HW, self.dim)
        L.rowpool_mean_bwd(dpool, dy, B, HW, self.dim, False, dt)
        dx = self._unit_bwd(self.final, dy, sv["final"], B)
        # the gradients of a stage are final once its first block's backward is queued: hand them to the data-parallel exchange
        # stage by stage (engine._ready -> GradSync.span_ready), so that it overlaps the backward of the stages below
        ready_ = getattr(self.o, "_ready", None)                             # (a bare owner in the kernel tests has no exchange)

        def ready(group):
            if ready_ is not None:
                self._side_join()           # the stage's weight gradients on the side stream are part of what is handed over
                ready_(group)
        ready("image_stage_" + ("conv5" if self.name == "shuffle_net" else "final_conv"))
        for bi in reversed(range(len(self.blocks))):
            blk, bs = self.blocks[bi], sv["blocks"][bi]
            gp = f"g{bi % 2}"          # gradient buffers alternate between two pools: the incoming gradient (written by block
                                       # bi + 1) must survive until this block's residual sum
            if blk["kind"] in ("fused", "mb"):
                us = blk["units"]
                dres = dx
                rows_out = B * bs["units"][-1]["Ho"] * bs["units"][-1]["Wo"]
                if blk["res"] and bs.get("rowscale") is not None:
                    dbr = self.buf(f"{gp}.dbr.{us[-1].cout}.{rows_out}", rows_out, us[-1].cout)
                    L.residual_add(None, dx, bs["rowscale"], dbr, B, (rows_out // B) * us[-1].cout, dt)
                    d = dbr
                else:
                    d = dx
                sums_ready = False
                for ui in reversed(range(len(us))):
                    first = ui == 0
                    d = self._unit_bwd(us[ui], d, bs["units"][ui], B, need_dx=True, tag=f"{gp}.u{ui}", sums_ready=sums_ready)
                    sums_ready = False
                    if blk["se"] is not None and ui == len(us) - 1:
                        # the unit before the squeeze-excitation (the depthwise conv + BN + SiLU) is next: its BatchNorm sums ride along
                        d, sums_ready = self._se_bwd(blk["se"], d, bs["se"], B, gp, bn=(us[ui - 1], bs["units"][ui - 1]) if ui > 0 else None)
                if blk["res"]:
                    rows_in = B * bs["H"] * bs["W"]
                    out = self.buf(f"{gp}.sum.{bs['cin']}.{rows_in}", rows_in, bs["cin"])
                    L.residual_add(dres, d, None, out, B, (rows_in // B) * bs["cin"], dt)
                    d = out
                dx = d
            elif blk["kind"] == "shuffle_down":
                rows_out = B * bs["b2"][-1]["Ho"] * bs["b2"][-1]["Wo"]
                d1, d2 = self._shuffle_cat_bwd(dx, rows_out, blk["cout"], gp)
                for ui in reversed(range(len(blk["b1"]))):
                    d1 = self._unit_bwd(blk["b1"][ui], d1, bs["b1"][ui], B, tag=f"{gp}.a{ui}")
                for ui in reversed(range(len(blk["b2"]))):
                    d2 = self._unit_bwd(blk["b2"][ui], d2, bs["b2"][ui], B, tag=f"{gp}.c{ui}")
                rows_in = B * bs["H"] * bs["W"]
                out = self.buf(f"{gp}.sum.{blk['cin']}.{rows_in}", rows_in, blk["cin"])
                L.residual_add(d1, d2, None, out, B, (rows_in // B) * blk["cin"], dt)
                dx = out
            else:
                Cc = blk["cout"]
                bf = Cc // 2
                rows = B * bs["H"] * bs["W"]
                out = self.buf(f"{gp}.cat.{Cc}.{rows}", rows, Cc)                 # the unit's input gradient = [d x1 | d x2]
                _, d2 = self._shuffle_cat_bwd(dx, rows, Cc, gp, d1_into=out)           # x1 passed straight through: its gradient lands in place
                for ui in reversed(range(len(blk["b2"]))):
                    into = (out[:, bf:], Cc) if (ui == 0 and bs.get("inplace")) else None      # the first unit's input gradient IS d x2
                    d2 = self._unit_bwd(blk["b2"][ui], d2, bs["b2"][ui], B, tag=f"{gp}.c{ui}", dx_into=into)
                if not bs.get("inplace"):
                    L.channel_gather(d2, self._cmap(("id", bf), list(range(bf))), out, rows, bf, bf, Cc, bf, dt)
                dx = out
            if bi == 0 or self.blocks[bi - 1]["stage"] != blk["stage"]:
                ready("image_stage_" + blk["stage"])
        if self.name == "shuffle_net":
            p = sv["pool"]
            c0 = self.stem.cout
            d = self.buf("g.pool", B * p["H"] * p["W"], c0)
            L.maxpool3x3s2_bwd(dx, p["arg"], d, B, p["H"], p["W"], c0, dt)
            dx = d
        self._unit_bwd(self.stem, dx, sv["stem"], B, need_dx=False)
        self._side_join()
