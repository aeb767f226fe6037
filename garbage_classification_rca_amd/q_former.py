"""BLIP-2 Q-Former classifier path (reference ``q_former_training.py``) on libmmrca.

What the reference's step computes (``q_former_training.py:279-304``):

    outputs = Blip2ForConditionalGeneration(**batch)                     # :289
    x       = outputs['qformer_outputs'].last_hidden_state[:, 0, :]      # :290  (first of the 32 learned queries)
    out     = MultimodalClassifier()(x)                                  # :291  Linear(768, 4), :24-31
    loss    = CrossEntropyLoss()(out, y) / 8 ; loss.backward()           # :293-295

``get_peft_model(model, LoraConfig(target_modules=["q_proj", "k_proj"]))`` (:217-226) freezes every base weight and adds
LoRA factors only to modules named ``q_proj`` / ``k_proj``.  In transformers 5.15.0 those names exist only in the OPT
language model (the vision tower has a fused ``qkv``, the Q-Former ``query`` / ``key`` / ``value``:
modeling_blip_2.py:301, 553-558), and the loss reads ``qformer_outputs`` alone, which the language model does not feed.
So the only parameters that ever receive a gradient are the classifier's: the path from pixels to the loss is

    ViT-g/14 vision tower (frozen, 39 layers, 257 x 1408)  ->  Q-Former (frozen, 12 layers, 32 queries x 768, cross-
    attention to the image tokens in every second layer; train mode: its dropouts are ACTIVE, :276)  ->  Linear(768, 4)

and that is what runs here: encoder + Q-Former forward on the HIP kernels (bf16 MFMA GEMMs, K3x attention, K4
LayerNorm), classifier forward / backward / AdamW(lr 5e-4, eps 1e-5) (:243-244) on the fp32 kernels.  The OPT-2.7B
forward the reference also runs each step (:289) contributes nothing to the loss or the metrics and is not computed.

Reference quirks kept (they change results):
  * ``optimizer.zero_grad()`` runs at the top of EVERY iteration (:283) while ``optimizer.step()`` runs every 8th
    (:299-300, and once more after the epoch when the count is not a multiple of 8, :308-309): the step applies the
    gradient of the LAST micro-batch only, scaled by 1/8;
  * the epoch's reported loss is ``total_loss / step`` with ``step`` the last index (:306), not the count.

Parameter names are those of ``Blip2ForConditionalGeneration.state_dict()`` (vision_model.*, query_tokens, qformer.*)
plus ``classifier.weight / classifier.bias`` (the reference saves the two separately, :33-47).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as L

ROWPAD = 256


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass(frozen=True)
class Blip2Spec:
    """Blip2VisionConfig / Blip2QFormerConfig (configuration_blip_2.py:49-59, 89-102, 155): the defaults ARE
    Salesforce/blip2-opt-2.7b, the checkpoint the reference loads (:203)."""
    v_dim: int = 1408
    v_layers: int = 39
    v_heads: int = 16
    v_mlp: int = 6144
    image_size: int = 224
    patch: int = 14
    v_eps: float = 1e-6
    q_dim: int = 768
    q_layers: int = 12
    q_heads: int = 12
    q_mlp: int = 3072
    cross_freq: int = 2
    n_query: int = 32
    q_eps: float = 1e-12
    hidden_drop: float = 0.1
    attn_drop: float = 0.1
    n_classes: int = 4

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch) ** 2

    @property
    def v_tokens(self) -> int:
        return self.n_patches + 1


BLIP2_OPT_2_7B = Blip2Spec()


def blip2_params(s: Blip2Spec) -> List[Tuple[str, Tuple[int, ...]]]:
    """(state_dict key, shape) of the frozen part, in forward order.  Q/K/V of the Q-Former attentions are listed
    adjacently so that they are ONE [3D, D] (self) / [2D, D_enc] (cross K|V) GEMM operand in the flat store."""
    D, P = s.v_dim, s.patch
    e: List[Tuple[str, Tuple[int, ...]]] = [
        ("vision_model.embeddings.class_embedding", (1, 1, D)),
        ("vision_model.embeddings.position_embedding", (1, s.v_tokens, D)),
        ("vision_model.embeddings.patch_embedding.weight", (D, 3, P, P)),
        ("vision_model.embeddings.patch_embedding.bias", (D,)),
    ]
    for i in range(s.v_layers):
        p = f"vision_model.encoder.layers.{i}."
        e += [(p + "layer_norm1.weight", (D,)), (p + "layer_norm1.bias", (D,)),
              (p + "self_attn.qkv.weight", (3 * D, D)), (p + "self_attn.qkv.bias", (3 * D,)),
              (p + "self_attn.projection.weight", (D, D)), (p + "self_attn.projection.bias", (D,)),
              (p + "layer_norm2.weight", (D,)), (p + "layer_norm2.bias", (D,)),
              (p + "mlp.fc1.weight", (s.v_mlp, D)), (p + "mlp.fc1.bias", (s.v_mlp,)),
              (p + "mlp.fc2.weight", (D, s.v_mlp)), (p + "mlp.fc2.bias", (D,))]
    e += [("vision_model.post_layernorm.weight", (D,)), ("vision_model.post_layernorm.bias", (D,))]
    Q = s.q_dim
    e += [("query_tokens", (1, s.n_query, Q)), ("qformer.layernorm.weight", (Q,)), ("qformer.layernorm.bias", (Q,))]
    for i in range(s.q_layers):
        p = f"qformer.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            e += [(p + f"attention.attention.{n}.weight", (Q, Q))]
        for n in ("query", "key", "value"):
            e += [(p + f"attention.attention.{n}.bias", (Q,))]
        e += [(p + "attention.output.dense.weight", (Q, Q)), (p + "attention.output.dense.bias", (Q,)),
              (p + "attention.output.LayerNorm.weight", (Q,)), (p + "attention.output.LayerNorm.bias", (Q,))]
        if i % s.cross_freq == 0:
            e += [(p + "crossattention.attention.query.weight", (Q, Q)), (p + "crossattention.attention.query.bias", (Q,)),
                  (p + "crossattention.attention.key.weight", (Q, D)), (p + "crossattention.attention.value.weight", (Q, D)),
                  (p + "crossattention.attention.key.bias", (Q,)), (p + "crossattention.attention.value.bias", (Q,)),
                  (p + "crossattention.output.dense.weight", (Q, Q)), (p + "crossattention.output.dense.bias", (Q,)),
                  (p + "crossattention.output.LayerNorm.weight", (Q,)), (p + "crossattention.output.LayerNorm.bias", (Q,))]
        e += [(p + "intermediate_query.dense.weight", (s.q_mlp, Q)), (p + "intermediate_query.dense.bias", (s.q_mlp,)),
              (p + "output_query.dense.weight", (Q, s.q_mlp)), (p + "output_query.dense.bias", (Q,)),
              (p + "output_query.LayerNorm.weight", (Q,)), (p + "output_query.LayerNorm.bias", (Q,))]
    return e


class _FrozenStore:
    """The frozen weights as ONE flat tensor in the compute dtype (no master copy, no gradient, no optimizer state:
    2.2 GB in bf16 for blip2-opt-2.7b's vision tower + Q-Former).  Entries start on 8-element (16-byte) boundaries and the
    store is padded by one 256 x 6144 tile so that whole-tile operand reads stay inside it."""

    def __init__(self, entries, dtype, device):
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...], int]] = {}
        off = 0
        for k, shp in entries:
            n = math.prod(shp)
            # the fused-operand pairs must stay adjacent: only pad when the previous entry ended off a 16-byte boundary
            off = _round_up(off, 8)
            self.offsets[k] = (off, tuple(shp), n)
            off += n
        self.total = _round_up(off, 128) + 256 * 6144
        self.w = torch.zeros(self.total, dtype=dtype, device=device)

    def view(self, key):
        off, shp, n = self.offsets[key]
        return self.w[off:off + n].view(shp)

    def flat(self, key, numel):
        off = self.offsets[key][0]
        return self.w[off:off + numel]


class Blip2QFormerEngine:
    """Frozen ViT-g + Q-Former forward and the trainable 4-class classifier of one replica on one GPU."""

    def __init__(self, spec: Blip2Spec = BLIP2_OPT_2_7B, dtype=torch.bfloat16, device="cuda",
                 gemm_impl: int = L.IMPL_AUTO, attn_impl: int = L.IMPL_AUTO):
        """dtype: torch.bfloat16 (fastest; logits ~2e-2 from the reference's fp32 arithmetic at full depth) | torch.float32 (fp32
        matrix cores / VALU attention: the parity mode) | "bf16x3f" (= "bf16x3" here, the towers are frozen and forward-only): fp32
        residual stream, LayerNorm and softmax statistics, every nn.Linear AND both attention products as three-pass split-bf16
        products on the bf16 matrix cores -- the fp32 arithmetic to ~1e-5 on the logits at a third of the bf16 MFMA rate: the fast
        mode inside the 1e-3 tolerance."""
        L.load()
        name = dtype.lower() if isinstance(dtype, str) else ""
        self.x3 = name in ("bf16x3f", "bf16x3")
        if self.x3:
            dtype = torch.float32
        elif isinstance(dtype, str):
            dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "f32": torch.float32}[name]
        self.mode = "bf16x3f" if self.x3 else ("bf16" if dtype == torch.bfloat16 else "fp32")
        self.s, self.dtype, self.device = spec, dtype, torch.device(device)
        self.dt = L.dtype_code(dtype)
        self.gemm_impl, self.attn_impl = gemm_impl, attn_impl
        if spec.v_dim % spec.v_heads or spec.q_dim % spec.q_heads:
            raise ValueError("hidden sizes must be divisible by their head counts (modeling_blip_2.py:291-295, 540-544)")
        self.entries = blip2_params(spec)
        self.store = _FrozenStore(self.entries, dtype, self.device)
        # bf16x3f: the fp32 store is the master (LayerNorm weights, biases, the K = 588 patch projection read it); the GEMM operands are
        # its two bf16 planes hi + lo, rebuilt after every load (2 x 2.2 GB next to the 4.4 GB master for blip2-opt-2.7b)
        self.w_hi = torch.zeros(self.store.total, dtype=torch.bfloat16, device=self.device) if self.x3 else None
        self.w_lo = torch.zeros(self.store.total, dtype=torch.bfloat16, device=self.device) if self.x3 else None
        for i in range(spec.q_layers):      # fused operands must be adjacent and unpadded
            p = f"qformer.encoder.layer.{i}.attention.attention."
            o = [self.store.offsets[p + n + ".weight"] for n in ("query", "key", "value")]
            assert o[1][0] == o[0][0] + o[0][2] and o[2][0] == o[1][0] + o[1][2], "q/k/v weights not adjacent"
            o = [self.store.offsets[p + n + ".bias"] for n in ("query", "key", "value")]
            assert o[1][0] == o[0][0] + o[0][2] and o[2][0] == o[1][0] + o[1][2], "q/k/v biases not adjacent"
            if i % spec.cross_freq == 0:    # the cross-attention K|V pair is read as one fused [2Q, D] operand with a [2Q] bias as well
                pc = f"qformer.encoder.layer.{i}.crossattention.attention."
                for leaf in (".weight", ".bias"):
                    o = [self.store.offsets[pc + n + leaf] for n in ("key", "value")]
                    assert o[1][0] == o[0][0] + o[0][2], f"cross-attention key/value {leaf[1:]}s not adjacent (q_dim and q_dim * v_dim must be multiples of 8)"
        # the trainable classifier (q_former_training.py:24-31): fp32 parameters, gradients and AdamW moments, one flat
        # tensor each, weight [n_classes, 768] then bias [n_classes]
        C, Q = spec.n_classes, spec.q_dim
        self.n_cls = C * Q + C
        self.cls_p = torch.zeros(self.n_cls, dtype=torch.float32, device=self.device)
        self.cls_g = torch.zeros_like(self.cls_p)
        self.cls_m = torch.zeros_like(self.cls_p)
        self.cls_v = torch.zeros_like(self.cls_p)
        self.training = True
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        self._fwd_count = 0
        self._x_cls = None

    # ------------------------------------------------------------------ parameters
    @property
    def cls_weight(self):
        return self.cls_p[: self.s.n_classes * self.s.q_dim].view(self.s.n_classes, self.s.q_dim)

    @property
    def cls_bias(self):
        return self.cls_p[self.s.n_classes * self.s.q_dim:]

    def param_keys(self) -> List[str]:
        return [k for k, _ in self.entries]

    def load_state_dict(self, sd: Dict[str, "torch.Tensor"], classifier_sd: Optional[Dict[str, "torch.Tensor"]] = None,
                        strict: bool = True):
        """sd: ``Blip2ForConditionalGeneration.state_dict()`` (extra keys -- the language model, LoRA factors -- are
        ignored; a ``base_model.model.`` prefix, which peft adds to the checkpoint the reference saves at :33-40, is
        stripped).  Older checkpoints carry the vision attention bias as ``q_bias`` / ``v_bias`` (the k part is zero)."""
        sd = {k[len("base_model.model."):] if k.startswith("base_model.model.") else k: v for k, v in sd.items()}
        with torch.no_grad():
            for k, shp in self.entries:
                if k in sd:
                    src = torch.as_tensor(sd[k])
                elif k.endswith("self_attn.qkv.bias") and k[:-len("qkv.bias")] + "q_bias" in sd:
                    qb, vb = torch.as_tensor(sd[k[:-len("qkv.bias")] + "q_bias"]), torch.as_tensor(sd[k[:-len("qkv.bias")] + "v_bias"])
                    src = torch.cat([qb, torch.zeros_like(vb), vb])
                elif strict:
                    raise KeyError(k)
                else:
                    continue
                self.store.view(k).copy_(src.to(self.device, self.dtype).reshape(shp))
            if classifier_sd is not None:
                self.cls_weight.copy_(torch.as_tensor(classifier_sd["classifier.weight"]).to(self.device, torch.float32))
                self.cls_bias.copy_(torch.as_tensor(classifier_sd["classifier.bias"]).to(self.device, torch.float32))
        self._refresh_planes()

    def _refresh_planes(self):
        if self.x3:
            L.split_f32(self.store.w, self.w_hi, self.w_lo, self.store.total)

    def classifier_state_dict(self) -> Dict[str, "torch.Tensor"]:
        return {"classifier.weight": self.cls_weight.detach().cpu().clone(), "classifier.bias": self.cls_bias.detach().cpu().clone()}

    def init_parameters(self, seed: int = 0):
        """Random stand-in for the pretrained checkpoint (no network): N(0, 0.02) weights, LayerNorm (1, 0), zero biases
        (Blip2PreTrainedModel._init_weights); classifier with nn.Linear's default U(-1/sqrt(768), 1/sqrt(768))."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.no_grad():
            for k, shp in self.entries:
                v = self.store.view(k)
                if "LayerNorm" in k or "layer_norm" in k or "layernorm" in k:
                    v.fill_(1.0 if k.endswith("weight") else 0.0)
                elif k.endswith("bias"):
                    v.zero_()
                else:
                    v.copy_((torch.randn(shp, generator=g) * 0.02).to(self.dtype))
            bound = 1.0 / math.sqrt(self.s.q_dim)
            self.cls_p.copy_((torch.rand(self.n_cls, generator=g) * 2 - 1) * bound)
        self._refresh_planes()

    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    # ------------------------------------------------------------------ buffers / op helpers
    def buf(self, name, rows, cols, dtype=None):
        dtype = dtype or self.dtype
        key = (name, rows, cols, dtype)
        t = self._bufs.get(key)
        if t is None:
            t = torch.zeros(_round_up(max(rows, 1), ROWPAD), cols, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def release_buffers(self):
        self._bufs.clear()
        self._x_cls = None

    def _lin(self, x, wkey, bkey, out, M, N, K, act=L.ACT_NONE, addend=None, wnumel=None):
        w = self.store.view(wkey) if wnumel is None else self.store.flat(wkey, wnumel)
        b = self.store.view(bkey) if wnumel is None else self.store.flat(bkey, N)
        L.gemm(x, w, out, bias=b, addend=addend, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, a_layout=L.ROWK, b_layout=L.ROWK, act=act,
               dtype=self.dt, impl=self.gemm_impl)

    def pbuf(self, name, rows, cols):
        """an fp32 [rows, cols] matrix as two bf16 planes (the operand form of the bf16x3 GEMMs)"""
        return self.buf(name + ".hi", rows, cols, torch.bfloat16), self.buf(name + ".lo", rows, cols, torch.bfloat16)

    def _lin3(self, xp, wkey, bkey, out, M, N, K, act=L.ACT_NONE, addend=None, wnumel=None):
        """bf16x3 nn.Linear: xp = (hi, lo) planes of the fp32 input; out = an fp32 tensor, or a (hi, lo) pair when only other
        bf16x3 GEMMs read it; bias / addend fp32"""
        off = self.store.offsets[wkey][0]
        n = self.store.offsets[wkey][2] if wnumel is None else wnumel
        w = (self.w_hi[off:off + n], self.w_lo[off:off + n])
        b = self.store.view(bkey) if wnumel is None else self.store.flat(bkey, N)
        planes = isinstance(out, tuple)
        L.gemm_x3(xp, w, out[0] if planes else out, C_lo=out[1] if planes else None, bias=b, addend=addend, M=M, N=N, K=K, lda=K, ldb=K,
                  ldc=N, a_layout=L.ROWK, b_layout=L.ROWK, act=act, impl=L.IMPL_AUTO)

    def _ln3(self, x, res, pfx, y, yp, rows, D, eps, in_drop=(0.0, 0), out_drop=(0.0, 0)):
        """fp32 LayerNorm whose output leaves as two bf16 planes yp (and as fp32 in y when it is also the next residual)"""
        mean, rstd = self.buf("ln_mean", rows, 1, torch.float32), self.buf("ln_rstd", rows, 1, torch.float32)
        L.add_layernorm_fwd(x, res, self.store.view(pfx + ".weight"), self.store.view(pfx + ".bias"), None, y, mean, rstd, rows, D,
                            D, D, eps, L.F32, in_drop=in_drop, out_drop=out_drop, y_planes=yp)

    def _ln(self, x, res, pfx, y, rows, D, eps, in_drop=(0.0, 0), out_drop=(0.0, 0), sum_out=None):
        mean, rstd = self.buf("ln_mean", rows, 1, torch.float32), self.buf("ln_rstd", rows, 1, torch.float32)
        L.add_layernorm_fwd(x, res, self.store.view(pfx + ".weight"), self.store.view(pfx + ".bias"), sum_out, y, mean, rstd, rows, D,
                            D, D, eps, self.dt, in_drop=in_drop, out_drop=out_drop)

    @staticmethod
    def _site_seed(base: int, layer: int, site: int) -> int:
        """one dropout counter space per (forward pass, layer, site)"""
        return (base * 1000003 + layer * 16 + site) & 0x7FFFFFFFFFFFFFFF

    # ------------------------------------------------------------------ forward
    def vision_forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """Blip2VisionModel.forward (modeling_blip_2.py:505-531): embeddings (:243-254), 39 pre-LN layers (:383-403),
        post_layernorm over every token.  Returns image_embeds [B*257 (row-padded), 1408]."""
        s = self.s
        B, D, P, nP, T = pixel_values.shape[0], s.v_dim, s.patch, s.n_patches, s.v_tokens
        if tuple(pixel_values.shape[1:]) != (3, s.image_size, s.image_size):
            raise ValueError(f"pixel_values must be [B, 3, {s.image_size}, {s.image_size}], got {tuple(pixel_values.shape)}")
        M, KP = B * T, 3 * P * P
        images = pixel_values.to(torch.float32).contiguous()
        patches = self.buf("patches", B * nP, KP)
        L.patchify_fwd(images, patches, B, 3, s.image_size, s.image_size, P, self.dt)
        proj = self.buf("proj", B * nP, D)
        self._lin(patches, "vision_model.embeddings.patch_embedding.weight", "vision_model.embeddings.patch_embedding.bias", proj,
                  B * nP, D, KP)
        x = self.buf("vx", M, D)
        L.vit_assemble_fwd(proj, self.store.view("vision_model.embeddings.class_embedding"),
                           self.store.view("vision_model.embeddings.position_embedding"), x, B, nP, D, self.dt)
        dh = D // s.v_heads
        if self.x3:
            return self._vision_layers_x3(x, B)
        y, qkv, ao, x1 = self.buf("vy", M, D), self.buf("vqkv", M, 3 * D), self.buf("vao", M, D), self.buf("vx1", M, D)
        h = self.buf("vh", M, s.v_mlp)
        # GEMMs run over whole 256-row tiles of the (zero-padded) buffers: that is what lets the dispatcher give the
        # persistent 256x256 kernel the shapes it takes (N % 256 == 0: FFN1 and the Q-Former's K|V projection); rows >= M
        # are never read by the attention or the LayerNorms.  The residual adds ride on the GEMM epilogues (addend): moving
        # them onto the next LayerNorm, as engine.LN_RESIDUAL does for ViT-B, measured slower here (N = 1408 cannot use the
        # 256-wide kernel that move pays for; the LayerNorm got 20 us slower per call, the GEMMs 3 us faster).
        Mg = _round_up(M, ROWPAD) if self.dtype == torch.bfloat16 else M
        if Mg > M:
            x[M:Mg].zero_()          # the padding rows of the residual stream would otherwise carry over from the last pass
        for i in range(s.v_layers):
            p = f"vision_model.encoder.layers.{i}."
            self._ln(x, None, p + "layer_norm1", y, M, D, s.v_eps)
            self._lin(y, p + "self_attn.qkv.weight", p + "self_attn.qkv.bias", qkv, Mg, 3 * D, D)
            L.mha_cross_fwd(qkv, 3 * D, qkv[:, D:], 3 * D, qkv[:, 2 * D:], 3 * D, ao, D, B, s.v_heads, T, T, dh, dh ** -0.5,
                            self.dt, self.attn_impl)
            self._lin(ao, p + "self_attn.projection.weight", p + "self_attn.projection.bias", x1, Mg, D, D, addend=x)      # :395
            self._ln(x1, None, p + "layer_norm2", y, M, D, s.v_eps)
            self._lin(y, p + "mlp.fc1.weight", p + "mlp.fc1.bias", h, Mg, s.v_mlp, D, act=L.ACT_GELU)
            self._lin(h, p + "mlp.fc2.weight", p + "mlp.fc2.bias", x, Mg, D, s.v_mlp, addend=x1)                           # :401
        emb = self.buf("image_embeds", M, D)
        self._ln(x, None, "vision_model.post_layernorm", emb, M, D, s.v_eps)                                               # :521
        return emb

    def _vision_layers_x3(self, x, B):
        """the 39 layers + post_layernorm of vision_forward in the bf16x3 form: fp32 residual stream x, LayerNorm outputs / attention
        context / MLP hidden as bf16 plane pairs, q|k|v fp32 (the attention splits them while staging).  Returns image_embeds as a
        plane pair (its only readers are the Q-Former's K|V projections)."""
        s = self.s
        D, T = s.v_dim, s.v_tokens
        M = B * T
        Mg = _round_up(M, ROWPAD)
        dh = D // s.v_heads
        yp, aop, hp = self.pbuf("vy", M, D), self.pbuf("vao", M, D), self.pbuf("vh", M, s.v_mlp)
        qkv, x1 = self.buf("vqkv", M, 3 * D), self.buf("vx1", M, D)
        if Mg > M:
            x[M:Mg].zero_()
        for i in range(s.v_layers):
            p = f"vision_model.encoder.layers.{i}."
            self._ln3(x, None, p + "layer_norm1", None, yp, M, D, s.v_eps)
            self._lin3(yp, p + "self_attn.qkv.weight", p + "self_attn.qkv.bias", qkv, Mg, 3 * D, D)
            L.mha_cross_fwd_x3(qkv, 3 * D, qkv[:, D:], 3 * D, qkv[:, 2 * D:], 3 * D, aop, D, B, s.v_heads, T, T, dh, dh ** -0.5)
            self._lin3(aop, p + "self_attn.projection.weight", p + "self_attn.projection.bias", x1, Mg, D, D, addend=x)
            self._ln3(x1, None, p + "layer_norm2", None, yp, M, D, s.v_eps)
            self._lin3(yp, p + "mlp.fc1.weight", p + "mlp.fc1.bias", hp, Mg, s.v_mlp, D, act=L.ACT_GELU)
            self._lin3(hp, p + "mlp.fc2.weight", p + "mlp.fc2.bias", x, Mg, D, s.v_mlp, addend=x1)
        emb = self.pbuf("image_embeds", M, D)
        self._ln3(x, None, "vision_model.post_layernorm", None, emb, M, D, s.v_eps)
        return emb

    def _qformer_forward_x3(self, emb, B: int, drop_seed: int):
        """qformer_forward in the bf16x3 form (emb: the image tokens as a plane pair)"""
        s = self.s
        Q, NQ, T, D = s.q_dim, s.n_query, s.v_tokens, s.v_dim
        M, Mi = B * NQ, B * T
        hp_, ap = (s.hidden_drop, s.attn_drop) if self.training else (0.0, 0.0)
        H, dh = s.q_heads, Q // s.q_heads
        sd = lambda layer, site: self._site_seed(drop_seed, layer, site)
        qe = self.buf("q_embed", M, Q)
        qe[:M].view(B, NQ, Q).copy_(self.store.view("query_tokens").expand(B, NQ, Q))
        hcur, hnext = self.buf("q_h0", M, Q), self.buf("q_h1", M, Q)
        hcp, hnp = self.pbuf("q_h0p", M, Q), self.pbuf("q_h1p", M, Q)
        self._ln3(qe, None, "qformer.layernorm", hcur, hcp, M, Q, s.q_eps, out_drop=(hp_, sd(0, 0)))
        qkv, o, cq, ckv = self.buf("q_qkv", M, 3 * Q), self.buf("q_o", M, Q), self.buf("q_cq", M, Q), self.buf("q_ckv", Mi, 2 * Q)
        aop, ffp = self.pbuf("q_ao", M, Q), self.pbuf("q_ff", M, s.q_mlp)
        for i in range(s.q_layers):
            p = f"qformer.encoder.layer.{i}."
            self._lin3(hcp, p + "attention.attention.query.weight", p + "attention.attention.query.bias", qkv, M, 3 * Q, Q, wnumel=3 * Q * Q)
            L.mha_cross_fwd_x3(qkv, 3 * Q, qkv[:, Q:], 3 * Q, qkv[:, 2 * Q:], 3 * Q, aop, Q, B, H, NQ, NQ, dh, dh ** -0.5, drop_p=ap,
                               drop_seed=sd(i + 1, 1))
            self._lin3(aop, p + "attention.output.dense.weight", p + "attention.output.dense.bias", o, M, Q, Q)
            self._ln3(o, hcur, p + "attention.output.LayerNorm", hnext, hnp, M, Q, s.q_eps, in_drop=(hp_, sd(i + 1, 2)))
            hcur, hnext, hcp, hnp = hnext, hcur, hnp, hcp
            if i % s.cross_freq == 0:
                self._lin3(hcp, p + "crossattention.attention.query.weight", p + "crossattention.attention.query.bias", cq, M, Q, Q)
                self._lin3(emb, p + "crossattention.attention.key.weight", p + "crossattention.attention.key.bias", ckv, _round_up(Mi, ROWPAD),
                           2 * Q, D, wnumel=2 * Q * D)
                L.mha_cross_fwd_x3(cq, Q, ckv, 2 * Q, ckv[:, Q:], 2 * Q, aop, Q, B, H, NQ, T, dh, dh ** -0.5, drop_p=ap, drop_seed=sd(i + 1, 3))
                self._lin3(aop, p + "crossattention.output.dense.weight", p + "crossattention.output.dense.bias", o, M, Q, Q)
                self._ln3(o, hcur, p + "crossattention.output.LayerNorm", hnext, hnp, M, Q, s.q_eps, in_drop=(hp_, sd(i + 1, 4)))
                hcur, hnext, hcp, hnp = hnext, hcur, hnp, hcp
            self._lin3(hcp, p + "intermediate_query.dense.weight", p + "intermediate_query.dense.bias", ffp, M, s.q_mlp, Q, act=L.ACT_GELU)
            self._lin3(ffp, p + "output_query.dense.weight", p + "output_query.dense.bias", o, M, Q, s.q_mlp)
            self._ln3(o, hcur, p + "output_query.LayerNorm", hnext, hnp, M, Q, s.q_eps, in_drop=(hp_, sd(i + 1, 5)))
            hcur, hnext, hcp, hnp = hnext, hcur, hnp, hcp
        return hcur

    def qformer_forward(self, image_embeds: torch.Tensor, B: int, drop_seed: int = 0) -> torch.Tensor:
        """Blip2QFormerModel.forward with query_embeds = query_tokens.expand(B) and an all-ones image mask
        (modeling_blip_2.py:889-950, called at :1633-1639 region of Blip2ForConditionalGeneration.get_image_features).
        Returns last_hidden_state [B*32 (row-padded), 768].  Dropout (hidden 0.1, attention 0.1) is active in train mode."""
        if self.x3:
            return self._qformer_forward_x3(image_embeds, B, drop_seed)
        s = self.s
        Q, NQ, T, D = s.q_dim, s.n_query, s.v_tokens, s.v_dim
        M, Mi = B * NQ, B * T
        hp = s.hidden_drop if self.training else 0.0
        ap = s.attn_drop if self.training else 0.0
        H, dh = s.q_heads, Q // s.q_heads
        qe = self.buf("q_embed", M, Q)
        qe[:M].view(B, NQ, Q).copy_(self.store.view("query_tokens").expand(B, NQ, Q))
        hcur, hnext = self.buf("q_h0", M, Q), self.buf("q_h1", M, Q)
        self._ln(qe, None, "qformer.layernorm", hcur, M, Q, s.q_eps, out_drop=(hp, self._site_seed(drop_seed, 0, 0)))    # :912-913
        qkv, ao, o = self.buf("q_qkv", M, 3 * Q), self.buf("q_ao", M, Q), self.buf("q_o", M, Q)
        cq, ckv = self.buf("q_cq", M, Q), self.buf("q_ckv", Mi, 2 * Q)
        ff = self.buf("q_ff", M, s.q_mlp)
        for i in range(s.q_layers):
            p = f"qformer.encoder.layer.{i}."
            # self-attention over the 32 queries (:701-712, 561-606, 616-620)
            self._lin(hcur, p + "attention.attention.query.weight", p + "attention.attention.query.bias", qkv, M, 3 * Q, Q, wnumel=3 * Q * Q)
            L.mha_cross_fwd(qkv, 3 * Q, qkv[:, Q:], 3 * Q, qkv[:, 2 * Q:], 3 * Q, ao, Q, B, H, NQ, NQ, dh, dh ** -0.5, self.dt,
                            self.attn_impl, drop_p=ap, drop_seed=self._site_seed(drop_seed, i + 1, 1))
            self._lin(ao, p + "attention.output.dense.weight", p + "attention.output.dense.bias", o, M, Q, Q)
            self._ln(o, hcur, p + "attention.output.LayerNorm", hnext, M, Q, s.q_eps, in_drop=(hp, self._site_seed(drop_seed, i + 1, 2)))
            hcur, hnext = hnext, hcur
            if i % s.cross_freq == 0:
                # cross-attention of the queries to the image tokens (:717-727)
                self._lin(hcur, p + "crossattention.attention.query.weight", p + "crossattention.attention.query.bias", cq, M, Q, Q)
                self._lin(image_embeds, p + "crossattention.attention.key.weight", p + "crossattention.attention.key.bias", ckv,
                          _round_up(Mi, ROWPAD) if self.dtype == torch.bfloat16 else Mi, 2 * Q, D, wnumel=2 * Q * D)
                L.mha_cross_fwd(cq, Q, ckv, 2 * Q, ckv[:, Q:], 2 * Q, ao, Q, B, H, NQ, T, dh, dh ** -0.5, self.dt, self.attn_impl,
                                drop_p=ap, drop_seed=self._site_seed(drop_seed, i + 1, 3))
                self._lin(ao, p + "crossattention.output.dense.weight", p + "crossattention.output.dense.bias", o, M, Q, Q)
                self._ln(o, hcur, p + "crossattention.output.LayerNorm", hnext, M, Q, s.q_eps,
                         in_drop=(hp, self._site_seed(drop_seed, i + 1, 4)))
                hcur, hnext = hnext, hcur
            # feed-forward of the query branch (:729-735, 758-761)
            self._lin(hcur, p + "intermediate_query.dense.weight", p + "intermediate_query.dense.bias", ff, M, s.q_mlp, Q, act=L.ACT_GELU)
            self._lin(ff, p + "output_query.dense.weight", p + "output_query.dense.bias", o, M, Q, s.q_mlp)
            self._ln(o, hcur, p + "output_query.LayerNorm", hnext, M, Q, s.q_eps, in_drop=(hp, self._site_seed(drop_seed, i + 1, 5)))
            hcur, hnext = hnext, hcur
        return hcur

    def features(self, pixel_values: torch.Tensor, drop_seed: Optional[int] = None) -> torch.Tensor:
        """``outputs['qformer_outputs'].last_hidden_state[:, 0, :]`` (q_former_training.py:290) as fp32 [B, 768]."""
        B = pixel_values.shape[0]
        if drop_seed is None:
            self._fwd_count += 1
            drop_seed = self._fwd_count
        emb = self.vision_forward(pixel_values)
        hs = self.qformer_forward(emb, B, drop_seed)
        return hs[: B * self.s.n_query].view(B, self.s.n_query, self.s.q_dim)[:, 0, :].to(torch.float32).contiguous()

    def classify(self, x_cls: torch.Tensor) -> torch.Tensor:
        """MultimodalClassifier.forward (:29-31): logits fp32 [B, 4]."""
        B, C, Q = x_cls.shape[0], self.s.n_classes, self.s.q_dim
        logits = torch.empty(B, C, dtype=torch.float32, device=self.device)
        L.gemm(x_cls, self.cls_weight, logits, bias=self.cls_bias, M=B, N=C, K=Q, lda=Q, ldb=Q, ldc=C, a_layout=L.ROWK,
               b_layout=L.ROWK, dtype=L.F32, impl=self.gemm_impl)
        self._x_cls = x_cls
        return logits

    def forward(self, pixel_values: torch.Tensor, drop_seed: Optional[int] = None) -> torch.Tensor:
        return self.classify(self.features(pixel_values, drop_seed))

    __call__ = forward

    # ------------------------------------------------------------------ classifier backward / optimizer
    def backward(self, dlogits: torch.Tensor):
        """d loss / d classifier.{weight, bias} += (the only parameters with a gradient, see the module docstring)."""
        x = self._x_cls
        if x is None:
            raise RuntimeError("backward() before forward()")
        B, C, Q = x.shape[0], self.s.n_classes, self.s.q_dim
        gw, gb = self.cls_g[: C * Q], self.cls_g[C * Q:]
        L.gemm(dlogits, x, gw, M=C, N=Q, K=B, lda=C, ldb=Q, ldc=Q, a_layout=L.KROW, b_layout=L.KROW, accum=True, dtype=L.F32,
               impl=self.gemm_impl)
        L.colsum_accum(dlogits, gb, B, C, C, L.F32)

    def zero_grad(self):
        self.cls_g.zero_()


class ClassifierAdamW:
    """``torch.optim.AdamW(trainable, lr=5e-4, eps=1e-05)`` (q_former_training.py:243-244; betas (0.9, 0.999) and
    weight_decay 0.01 are torch's defaults) over the classifier's flat fp32 tensors: one kernel launch per step."""

    def __init__(self, engine: Blip2QFormerEngine, lr=5e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=1e-2):
        self.engine, self.lr, self.betas, self.eps, self.weight_decay = engine, lr, betas, eps, weight_decay
        self.t = 0
        self.world = 1          # > 1: data parallel, the classifier gradient is averaged over the ranks before every step

    def zero_grad(self):
        self.engine.zero_grad()

    def step(self):
        e = self.engine
        if self.world > 1:      # the path's only exchange: 3,076 fp32 values (RCCL through torch.distributed)
            import torch.distributed as dist
            dist.all_reduce(e.cls_g)
            e.cls_g.mul_(1.0 / self.world)
        self.t += 1
        L.adamw_step(e.cls_p, e.cls_g, e.cls_m, e.cls_v, None, e.n_cls, float(self.lr), float(self.betas[0]), float(self.betas[1]),
                     float(self.eps), float(self.weight_decay), self.t, 1.0)


ACCUMULATION_STEPS = 8       # q_former_training.py:241


def train_step(engine: Blip2QFormerEngine, optimizer: ClassifierAdamW, pixel_values, labels, step: int,
               accumulation_steps: int = ACCUMULATION_STEPS, world: Optional[int] = None):
    """One iteration of the reference loop (:279-302): zero_grad, forward, CE / accumulation_steps, backward, and an
    optimizer step on every ``accumulation_steps``-th iteration.  Returns the device scalar ``loss / accumulation_steps``
    (what the reference adds to ``total_loss``, :296).
    world (None = leave ``optimizer.world`` alone) > 1 (one process per GPU, torch.distributed initialised): every rank runs its shard of the batch through the
    frozen encoders; the only exchange is the all-reduce (RCCL) of the classifier's 3,076 gradient values inside
    ``optimizer.step()`` (also the remainder step run_one_epoch takes after the loop)."""
    optimizer.zero_grad()                                                        # :283
    logits = engine.forward(pixel_values)                                        # :289-291
    B, C = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    L.xent_fwd_bwd(logits, labels.view(-1).to(torch.int32), None, 0.0, loss, dlogits, B, C, 1.0 / accumulation_steps)   # :293-294
    engine.backward(dlogits)                                                     # :295
    if world is not None:
        optimizer.world = world
    if (step + 1) % accumulation_steps == 0:                                     # :299-300
        optimizer.step()
    return loss / accumulation_steps


def run_one_epoch(engine: Blip2QFormerEngine, optimizer: ClassifierAdamW, loader, device,
                  accumulation_steps: int = ACCUMULATION_STEPS):
    """q_former_training.py:274-309 over batches {'pixel_values': [B,3,224,224], 'labels': [B,1]} (``collate_fn``'s
    output, :94-122; ``input_ids`` / ``attention_mask`` feed only the language model and are ignored).  Returns the
    reference's ``avg_loss`` (:306)."""
    engine.train()
    losses, step = [], -1
    for step, batch in enumerate(loader):
        px = batch["pixel_values"].to(device, non_blocking=True)
        y = batch["labels"].to(device, non_blocking=True)
        losses.append(train_step(engine, optimizer, px, y, step, accumulation_steps))
    if step >= 0 and (step + 1) % accumulation_steps != 0:                        # :308-309
        optimizer.step()
    total = float(torch.stack(losses).sum().item()) if losses else 0.0
    return total / step if step > 0 else total                                   # :306 divides by the last index


def calculate_acc(engine: Blip2QFormerEngine, loader, device) -> float:
    """q_former_training.py:153-186 (eval mode, argmax of the classifier): multiclass accuracy."""
    engine.eval()
    correct, count = 0, 0
    with torch.no_grad():
        for batch in loader:
            px = batch["pixel_values"].to(device)
            y = batch["labels"].to(device).view(-1)
            pred = engine.forward(px).argmax(1)
            correct += int((pred == y).sum().item())
            count += int(y.numel())
    return correct / max(count, 1)
