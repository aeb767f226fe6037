"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain PyTorch fp32) of the MM-RCA hot path.

Not part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package, and only as the checker.  The product path
(``garbage_classification_rca_amd``) never imports it and fails loudly without its HIP library.

Parity status
-------------
* Fusion head (SelfAttention / ReverseCrossAttention / MM_RCA.forward / drop_modalities), the
  cross-entropy loss and the dataset listing are PINNED: ``tests/test_oracle_golden.py`` checks this
  restatement against vectors produced by running the reference's own classes
  (``tests/golden/make_goldens.py``).
* Text encoders (DistilBERT / BERT / RoBERTa) are third-party code (``transformers``, unpinned by the
  reference, 5.15.0 in this image); pinned against that version's outputs.
* ViT-B/16 / ViT-L/16 are torchvision architectures; torchvision is not installed here, so the ViT
  restatement is pinned only against ``transformers.ViTModel`` (an independent implementation of the
  same architecture) -- "parity unpinned" with respect to torchvision itself.

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from oracle import arch as S          # the checker's own tables (independent of the product's spec.py)


class OracleSelfAttention(torch.nn.Module):
    """CVPR_code/multimodal_model.py:39-68."""

    def __init__(self, d_in, d_out_kq, d_out_v, name=""):
        super().__init__()
        self.d_out_kq = d_out_kq
        self.W_query = torch.nn.Linear(d_in, d_out_kq)
        self.W_key = torch.nn.Linear(d_in, d_out_kq)
        self.W_value = torch.nn.Linear(d_in, d_out_v)
        self.norm = torch.nn.LayerNorm(d_out_v)

    def forward(self, x):
        k, q, v = self.W_key(x), self.W_query(x), self.W_value(x)          # :52-54
        a = torch.softmax(q @ k.transpose(-1, -2) / self.d_out_kq ** 0.5, dim=-1)   # :56-60
        return torch.relu(self.norm(a @ v))                               # :62-66


class OracleReverseCrossAttention(torch.nn.Module):
    """CVPR_code/multimodal_model.py:71-108."""

    def __init__(self, d_in_x1, d_in_x2, d_out_kq, d_out_v, reverse):
        super().__init__()
        self.d_out_kq = d_out_kq
        self.W_query = torch.nn.Linear(d_in_x1, d_out_kq)
        self.W_key = torch.nn.Linear(d_in_x2, d_out_kq)
        self.W_value = torch.nn.Linear(d_in_x2, d_out_v)
        self.norm = torch.nn.LayerNorm(d_out_v)
        self.reverse = reverse

    def forward(self, x1, x2):
        q, k, v = self.W_query(x1), self.W_key(x2), self.W_value(x2)      # :83-85
        a = torch.softmax(q @ k.transpose(-1, -2) / self.d_out_kq ** 0.5, dim=-1)   # :87-91
        assert a.shape[1] == a.shape[2]                                   # :93
        if self.reverse:
            a = (1.0 - a) / (a.shape[1] - 1)                              # :95-99
        return torch.relu(self.norm(a @ v))                               # :104-106


class OracleTextEncoder(torch.nn.Module):
    """transformers DistilBertModel / BertModel / RobertaModel (eval-mode math; dropout is applied
    only if ``self.training`` and p>0, as in modeling_distilbert.py:82-259)."""

    def __init__(self, spec: S.TextSpec, dropout: float = 0.1):
        super().__init__()
        self.spec = spec
        self.p = dropout
        self.params = torch.nn.ParameterDict()
        for k, shp in S.text_params(spec):
            self.params[k.replace(".", "/")] = torch.nn.Parameter(torch.zeros(shp))

    def P(self, key):
        return self.params[key.replace(".", "/")]

    def state_dict(self, *a, prefix="", **kw):      # HF key names
        return {prefix + k.replace("/", "."): v.detach() for k, v in self.params.items()}

    def load_flat(self, sd: Dict[str, torch.Tensor], prefix=""):
        with torch.no_grad():
            for k in self.params:
                self.params[k].copy_(torch.as_tensor(sd[prefix + k.replace("/", ".")]))

    def forward(self, input_ids, attention_mask):
        s = self.spec
        B, T = input_ids.shape
        if s.pos_offset:   # roberta: create_position_ids_from_input_ids
            nonpad = (input_ids != s.pad_id).to(torch.int64)
            pos = torch.cumsum(nonpad, dim=1) * nonpad + s.pad_id
        else:
            pos = torch.arange(T).unsqueeze(0).expand(B, T)
        # nn.Embedding(..., padding_idx=pad_token_id) (modeling_distilbert.py Embeddings / BertEmbeddings / RobertaEmbeddings):
        # the pad row never receives a gradient; RoBERTa's position table has the same padding_idx
        x = F.embedding(input_ids, self.P("embeddings.word_embeddings.weight"), padding_idx=s.pad_id) + \
            F.embedding(pos, self.P("embeddings.position_embeddings.weight"), padding_idx=(s.pad_id if s.pos_offset else None))
        if s.type_vocab:
            x = x + self.P("embeddings.token_type_embeddings.weight")[0]
        x = F.layer_norm(x, (s.dim,), self.P("embeddings.LayerNorm.weight"),
                         self.P("embeddings.LayerNorm.bias"), s.ln_eps)
        x = F.dropout(x, self.p, self.training)
        # transformers 5.x resolves to its sdpa attention: boolean key mask, and torch's
        # scaled_dot_product_attention returns 0 for a query row whose keys are ALL masked (a caption
        # zeroed by modality dropout, multimodal_model.py:451-452).  Partially masked rows are
        # identical to the eager additive finfo.min form.
        keymask = attention_mask.to(torch.bool)[:, None, None, :]
        row_valid = keymask.any(dim=-1, keepdim=True)
        dh = s.dim // s.heads
        for i in range(s.layers):
            K = S.text_layer_keys(s, i)
            lin = lambda t, nm: F.linear(t, self.P(K[nm] + ".weight"), self.P(K[nm] + ".bias"))
            q = lin(x, "q").view(B, T, s.heads, dh).transpose(1, 2)
            k = lin(x, "k").view(B, T, s.heads, dh).transpose(1, 2)
            v = lin(x, "v").view(B, T, s.heads, dh).transpose(1, 2)
            sc = (q @ k.transpose(2, 3) * dh ** -0.5).masked_fill(~keymask, float("-inf"))
            a = torch.softmax(torch.where(row_valid, sc, torch.zeros_like(sc)), dim=-1) * row_valid
            a = F.dropout(a, self.p, self.training)
            ctx = (a @ v).transpose(1, 2).reshape(B, T, s.dim)
            att = F.dropout(lin(ctx, "o"), self.p, self.training) if s.name != "distilbert" else lin(ctx, "o")
            x = F.layer_norm(att + x, (s.dim,), self.P(K["ln1"] + ".weight"), self.P(K["ln1"] + ".bias"), s.ln_eps)
            h = F.gelu(lin(x, "f1"))
            h = F.dropout(lin(h, "f2"), self.p, self.training)
            x = F.layer_norm(h + x, (s.dim,), self.P(K["ln2"] + ".weight"), self.P(K["ln2"] + ".bias"), s.ln_eps)
        return x


class OracleViT(torch.nn.Module):
    """torchvision VisionTransformer (vit_b_16 / vit_l_16; reference: models.py:222-258) without the
    classification head: returns the class token after ``encoder.ln`` -> [B, dim]."""

    def __init__(self, spec: S.VisionSpec):
        super().__init__()
        self.spec = spec
        self.params = torch.nn.ParameterDict()
        for k, shp in S.vision_params(spec):
            self.params[k.replace(".", "/")] = torch.nn.Parameter(torch.zeros(shp))

    def P(self, key):
        return self.params[key.replace(".", "/")]

    def state_dict(self, *a, prefix="", **kw):
        return {prefix + k.replace("/", "."): v.detach() for k, v in self.params.items()}

    def load_flat(self, sd, prefix=""):
        with torch.no_grad():
            for k in self.params:
                self.params[k].copy_(torch.as_tensor(sd[prefix + k.replace("/", ".")]))

    def tokens(self, images):
        s = self.spec
        B = images.shape[0]
        x = F.conv2d(images, self.P("conv_proj.weight"), self.P("conv_proj.bias"), stride=s.patch)
        x = x.reshape(B, s.dim, -1).permute(0, 2, 1)
        x = torch.cat([self.P("class_token").expand(B, -1, -1), x], dim=1) + self.P("encoder.pos_embedding")
        dh = s.dim // s.heads
        T = x.shape[1]
        for i in range(s.layers):
            L = f"encoder.layers.encoder_layer_{i}."
            y = F.layer_norm(x, (s.dim,), self.P(L + "ln_1.weight"), self.P(L + "ln_1.bias"), s.ln_eps)
            qkv = F.linear(y, self.P(L + "self_attention.in_proj_weight"), self.P(L + "self_attention.in_proj_bias"))
            q, k, v = (t.view(B, T, s.heads, dh).transpose(1, 2) for t in qkv.split(s.dim, dim=-1))
            a = torch.softmax(q @ k.transpose(2, 3) * dh ** -0.5, dim=-1)
            ctx = (a @ v).transpose(1, 2).reshape(B, T, s.dim)
            x = x + F.linear(ctx, self.P(L + "self_attention.out_proj.weight"), self.P(L + "self_attention.out_proj.bias"))
            y = F.layer_norm(x, (s.dim,), self.P(L + "ln_2.weight"), self.P(L + "ln_2.bias"), s.ln_eps)
            h = F.gelu(F.linear(y, self.P(L + "mlp.0.weight"), self.P(L + "mlp.0.bias")))
            x = x + F.linear(h, self.P(L + "mlp.3.weight"), self.P(L + "mlp.3.bias"))
        return F.layer_norm(x, (s.dim,), self.P("encoder.ln.weight"), self.P("encoder.ln.bias"), s.ln_eps)

    def forward(self, images):
        return self.tokens(images)[:, 0]


class OracleFeatureTable(torch.nn.Module):
    """Stand-in image backbone used with the head goldens (the generator's TableImage)."""

    def __init__(self, table):
        super().__init__()
        self.table = torch.nn.Parameter(torch.as_tensor(table).clone())

    def forward(self, x):
        t = self.table[: x.shape[0]]
        return (t * 0.5 + 0.25) if bool(x.abs().sum() == 0) else t


class OracleMMRCA(torch.nn.Module):
    """MM_RCA (CVPR_code/multimodal_model.py:636-728) on top of the base constructor's layout
    (:158-328), generalised over the image / text feature widths (the reference hard-codes
    1280 / 768 at :257-258)."""

    def __init__(self, n_classes, drop_ratio, image_or_text_dropout_chance, img_prob_dropout,
                 text_model: torch.nn.Module, image_model: torch.nn.Module, d_img: int, d_txt: int,
                 reverse: bool, features_only: bool, cross_attention_only: bool):
        super().__init__()
        self.text_model, self.image_model = text_model, image_model
        self.features_only, self.cross_attention_only = features_only, cross_attention_only
        self.drop = torch.nn.Dropout(p=drop_ratio)
        self.image_or_text_dropout_chance = image_or_text_dropout_chance
        self.img_dropout_prob = img_prob_dropout
        self.num_patches = S.NUM_PATCHES
        self.txt_patch_size, self.img_patch_size = d_txt // 16, d_img // 16
        self.self_attention_image = OracleSelfAttention(self.img_patch_size, S.SA_HID, S.SA_OUT)
        self.self_attention_text = OracleSelfAttention(self.txt_patch_size, S.SA_HID, S.SA_OUT)
        self.cross_attention_1 = OracleReverseCrossAttention(S.SA_OUT, S.SA_OUT, S.CA_HID, S.CA_OUT, reverse)
        self.cross_attention_2 = OracleReverseCrossAttention(S.SA_OUT, S.SA_OUT, S.CA_HID, S.CA_OUT, reverse)
        ca = S.CA_OUT * 16 * 2
        if features_only:
            self.final_features_only_linear = torch.nn.Linear(d_img + d_txt, n_classes)
        if cross_attention_only:
            self.cross_attention_only_linear = torch.nn.Linear(ca, n_classes)
        self.final_with_everything = torch.nn.Linear(ca + d_img + d_txt, n_classes)

    # multimodal_model.py:420-455 (decision(): :110-111 -> one np.random.rand draw each)
    def drop_modalities(self, ids, mask, images, _eval, remove_image, remove_text):
        if _eval:
            if remove_image:
                images = torch.zeros_like(images)
            if remove_text:
                ids, mask = torch.zeros_like(ids), torch.zeros_like(mask)
        else:
            if np.random.rand(1)[0] < self.image_or_text_dropout_chance:
                if np.random.rand(1)[0] < self.img_dropout_prob:
                    images = torch.zeros_like(images)
                else:
                    ids, mask = torch.zeros_like(ids), torch.zeros_like(mask)
        return ids, mask, images

    def head(self, txt, img):
        txt = txt / txt.norm(dim=1, keepdim=True)                         # :662-665
        img = img / img.norm(dim=1, keepdim=True)
        bs = txt.shape[0]
        t_sa = self.self_attention_text(txt.reshape(bs, 16, self.txt_patch_size))    # :668-680
        i_sa = self.self_attention_image(img.reshape(bs, 16, self.img_patch_size))
        t_i = self.cross_attention_1(t_sa, i_sa).flatten(1, 2)           # :683-692
        i_t = self.cross_attention_2(i_sa, t_sa).flatten(1, 2)
        if self.features_only:                                            # :694-716
            cat = torch.cat((img, txt), dim=1)
        elif self.cross_attention_only:
            cat = torch.cat((t_i, i_t), dim=1)
        else:
            cat = torch.cat((t_i, i_t, img, txt), dim=1)
        cat = self.drop(cat)                                              # :719
        if self.features_only:                                            # :721-726
            return self.final_features_only_linear(cat)
        if self.cross_attention_only:
            return self.cross_attention_only_linear(cat)
        return self.final_with_everything(cat)

    def forward(self, _input_ids, _attention_mask, _images, eval=False, remove_image=False, remove_text=False):
        ids, mask, images = self.drop_modalities(_input_ids, _attention_mask, _images, eval, remove_image, remove_text)
        txt = self.text_model(ids, mask)[:, 0]                            # :651-658
        img = self.image_model(images)                                    # :659
        return self.head(txt, img)


# ---- independent numpy (float64) statement of the head, for cross-checking the torch one -------
def head_forward_numpy(sd: Dict[str, np.ndarray], txt: np.ndarray, img: np.ndarray, reverse: bool,
                       features_only: bool = False, cross_attention_only: bool = False) -> np.ndarray:
    f8 = lambda a: np.asarray(a, dtype=np.float64)

    def lin(x, p):
        return x @ f8(sd[p + ".weight"]).T + f8(sd[p + ".bias"])

    def ln(x, p):
        mu = x.mean(-1, keepdims=True)
        var = ((x - mu) ** 2).mean(-1, keepdims=True)
        return (x - mu) / np.sqrt(var + 1e-5) * f8(sd[p + ".weight"]) + f8(sd[p + ".bias"])

    def softmax(z):
        z = z - z.max(-1, keepdims=True)
        e = np.exp(z)
        return e / e.sum(-1, keepdims=True)

    def attn(xq, xkv, p, dkq, rev):
        q, k, v = lin(xq, p + ".W_query"), lin(xkv, p + ".W_key"), lin(xkv, p + ".W_value")
        a = softmax(q @ np.swapaxes(k, -1, -2) / math.sqrt(dkq))
        if rev:
            a = (1.0 - a) / (a.shape[-1] - 1)
        return np.maximum(ln(a @ v, p + ".norm"), 0.0)

    txt, img = f8(txt), f8(img)
    txt = txt / np.linalg.norm(txt, axis=1, keepdims=True)
    img = img / np.linalg.norm(img, axis=1, keepdims=True)
    B = txt.shape[0]
    t3, i3 = txt.reshape(B, 16, -1), img.reshape(B, 16, -1)
    t_sa = attn(t3, t3, "self_attention_text", S.SA_HID, False)
    i_sa = attn(i3, i3, "self_attention_image", S.SA_HID, False)
    t_i = attn(t_sa, i_sa, "cross_attention_1", S.CA_HID, reverse).reshape(B, -1)
    i_t = attn(i_sa, t_sa, "cross_attention_2", S.CA_HID, reverse).reshape(B, -1)
    if features_only:
        return lin(np.concatenate([img, txt], 1), "final_features_only_linear")
    if cross_attention_only:
        return lin(np.concatenate([t_i, i_t], 1), "cross_attention_only_linear")
    return lin(np.concatenate([t_i, i_t, img, txt], 1), "final_with_everything")


def cross_entropy(logits, labels, weight: Optional[torch.Tensor] = None, label_smoothing: float = 0.0):
    """torch.nn.CrossEntropyLoss(weight, label_smoothing) as main_both.py:87-93 builds it, written out:
    loss = [ (1-e) * sum_i w[y_i] * nll_i + (e/C) * sum_i sum_c w[c] * (-logp_ic) ] / sum_i w[y_i]."""
    C = logits.shape[1]
    logp = torch.log_softmax(logits.float(), dim=1)
    w = torch.ones(C) if weight is None else weight.float()
    wy = w[labels]
    nll = -(logp.gather(1, labels[:, None])[:, 0]) * wy
    smooth = -(logp * w[None, :]).sum(1)
    return ((1 - label_smoothing) * nll.sum() + (label_smoothing / C) * smooth.sum()) / wy.sum()


def build_oracle(text_model: str, image_model: str, reverse=True, features_only=False, cross_attention_only=False,
                 n_classes=4, drop_ratio=0.6, image_text_dropout=0.0, image_prob_dropout=0.7, enc_dropout=0.1):
    ts = S.TEXT_SPECS[text_model]
    if image_model in S.VISION_SPECS:
        vs = S.VISION_SPECS[image_model]
        img, d_img = OracleViT(vs), vs.dim
    else:       # conv backbones: oracle/conv_models.py (restated torchvision architectures)
        from oracle import conv_models as CM
        name = {"EffNetv2-Medium": "eff_v2_medium"}.get(image_model, image_model)
        img = CM.build_conv_oracle(name)
        if hasattr(img, "pooled_only"):
            img.pooled_only = True
        d_img = 2048 if name == "shuffle_net" else 1280
    return OracleMMRCA(n_classes, drop_ratio, image_text_dropout, image_prob_dropout,
                       OracleTextEncoder(ts, enc_dropout), img, d_img, ts.dim,
                       reverse, features_only, cross_attention_only)
