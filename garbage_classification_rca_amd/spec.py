"""Architecture specs and parameter inventories (state_dict key names and shapes).

Key names follow the reference's checkpoints so that a ``.pth`` written by either side loads in the
other: head keys from ``CVPR_code/multimodal_model.py:199-328``; text
encoder keys are those of ``transformers`` ``DistilBertModel`` / ``BertModel`` / ``RobertaModel``
(constructed at ``multimodal_model.py:128-153``); vision transformer keys are torchvision's
``vit_b_16`` / ``vit_l_16`` (``models.py:222-258``) minus the classification head.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

Shape = Tuple[int, ...]


@dataclass(frozen=True)
class TextSpec:
    name: str            # distilbert | bert | roberta
    vocab: int
    max_pos: int
    dim: int
    heads: int
    ffn: int
    layers: int
    type_vocab: int      # 0 = no token-type table (distilbert)
    pad_id: int
    pos_offset: int      # roberta: position ids start at pad_id+1 and skip pads
    ln_eps: float
    has_pooler: bool


@dataclass(frozen=True)
class VisionSpec:
    name: str            # transformer_B16 | transformer_L16
    image: int
    patch: int
    dim: int
    heads: int
    ffn: int
    layers: int
    ln_eps: float

    @property
    def tokens(self) -> int:
        return (self.image // self.patch) ** 2 + 1


TEXT_SPECS: Dict[str, TextSpec] = {
    "distilbert": TextSpec("distilbert", 30522, 512, 768, 12, 3072, 6, 0, 0, 0, 1e-12, False),
    "bert": TextSpec("bert", 30522, 512, 768, 12, 3072, 12, 2, 0, 0, 1e-12, True),
    "roberta": TextSpec("roberta", 50265, 514, 768, 12, 3072, 12, 1, 1, 2, 1e-5, False),
}

VISION_SPECS: Dict[str, VisionSpec] = {
    "transformer_B16": VisionSpec("transformer_B16", 224, 16, 768, 12, 3072, 12, 1e-6),
    "transformer_L16": VisionSpec("transformer_L16", 224, 16, 1024, 16, 4096, 24, 1e-6),
}

# feature widths of the conv backbones named by --image_model (SURVEY.md section 8 preamble); built in conv_engine.py
CONV_FEATURE_DIMS = {"eff_v2_medium": 1280, "EffNetv2-Medium": 1280, "eff_v2_large": 1280, "shuffle_net": 2048}


def image_feature_dim(image_model: str) -> int:
    if image_model in VISION_SPECS:
        return VISION_SPECS[image_model].dim
    if image_model in CONV_FEATURE_DIMS:
        return CONV_FEATURE_DIMS[image_model]
    raise ValueError(f"Wrong image model: {image_model}")


# ----------------------------------------------------------------------------------------------
# parameter inventories: ordered (key, shape) lists
# ----------------------------------------------------------------------------------------------

def _lin(prefix: str, out_f: int, in_f: int) -> List[Tuple[str, Shape]]:
    return [(prefix + ".weight", (out_f, in_f)), (prefix + ".bias", (out_f,))]


def _ln(prefix: str, d: int) -> List[Tuple[str, Shape]]:
    return [(prefix + ".weight", (d,)), (prefix + ".bias", (d,))]


def text_params(s: TextSpec) -> List[Tuple[str, Shape]]:
    """Keys of the HF text encoders.  q/k/v are adjacent (q,k,v order) so that a flat arena sees
    them as one fused [3*dim, dim] projection."""
    P: List[Tuple[str, Shape]] = []
    d = s.dim
    P.append(("embeddings.word_embeddings.weight", (s.vocab, d)))
    P.append(("embeddings.position_embeddings.weight", (s.max_pos, d)))
    if s.type_vocab:
        P.append(("embeddings.token_type_embeddings.weight", (s.type_vocab, d)))
    P += _ln("embeddings.LayerNorm", d)
    for i in range(s.layers):
        if s.name == "distilbert":
            L = f"transformer.layer.{i}."
            for n in ("q_lin", "k_lin", "v_lin"):
                P.append((L + f"attention.{n}.weight", (d, d)))
            for n in ("q_lin", "k_lin", "v_lin"):
                P.append((L + f"attention.{n}.bias", (d,)))
            P += _lin(L + "attention.out_lin", d, d)
            P += _ln(L + "sa_layer_norm", d)
            P += _lin(L + "ffn.lin1", s.ffn, d)
            P += _lin(L + "ffn.lin2", d, s.ffn)
            P += _ln(L + "output_layer_norm", d)
        else:
            L = f"encoder.layer.{i}."
            for n in ("query", "key", "value"):
                P.append((L + f"attention.self.{n}.weight", (d, d)))
            for n in ("query", "key", "value"):
                P.append((L + f"attention.self.{n}.bias", (d,)))
            P += _lin(L + "attention.output.dense", d, d)
            P += _ln(L + "attention.output.LayerNorm", d)
            P += _lin(L + "intermediate.dense", s.ffn, d)
            P += _lin(L + "output.dense", d, s.ffn)
            P += _ln(L + "output.LayerNorm", d)
    if s.has_pooler:
        P += _lin("pooler.dense", d, d)     # present in BertModel checkpoints, unused by MM_RCA
    return P


def text_layer_keys(s: TextSpec, i: int) -> Dict[str, str]:
    """Canonical role -> key prefix for layer i."""
    if s.name == "distilbert":
        L = f"transformer.layer.{i}."
        return dict(q=L + "attention.q_lin", k=L + "attention.k_lin", v=L + "attention.v_lin",
                    o=L + "attention.out_lin", ln1=L + "sa_layer_norm", f1=L + "ffn.lin1",
                    f2=L + "ffn.lin2", ln2=L + "output_layer_norm")
    L = f"encoder.layer.{i}."
    return dict(q=L + "attention.self.query", k=L + "attention.self.key", v=L + "attention.self.value",
                o=L + "attention.output.dense", ln1=L + "attention.output.LayerNorm",
                f1=L + "intermediate.dense", f2=L + "output.dense", ln2=L + "output.LayerNorm")


def vision_params(s: VisionSpec) -> List[Tuple[str, Shape]]:
    P: List[Tuple[str, Shape]] = []
    d = s.dim
    P.append(("class_token", (1, 1, d)))
    P.append(("conv_proj.weight", (d, 3, s.patch, s.patch)))
    P.append(("conv_proj.bias", (d,)))
    P.append(("encoder.pos_embedding", (1, s.tokens, d)))
    for i in range(s.layers):
        L = f"encoder.layers.encoder_layer_{i}."
        P += _ln(L + "ln_1", d)
        P.append((L + "self_attention.in_proj_weight", (3 * d, d)))
        P.append((L + "self_attention.in_proj_bias", (3 * d,)))
        P += _lin(L + "self_attention.out_proj", d, d)
        P += _ln(L + "ln_2", d)
        P += _lin(L + "mlp.0", s.ffn, d)
        P += _lin(L + "mlp.3", d, s.ffn)
    P += _ln("encoder.ln", d)
    return P


# ---- fusion head (multimodal_model.py:199-328) ------------------------------------------------
NUM_PATCHES = 16            # :250
SA_HID, SA_OUT = 128, 96    # :252-253
CA_HID, CA_OUT = 64, 48     # :255-256


def head_used_params(d_img: int, d_txt: int, n_classes: int, features_only: bool,
                     cross_attention_only: bool) -> List[Tuple[str, Shape]]:
    """The parameters MM_RCA.forward touches (multimodal_model.py:677-726)."""
    pi, pt = d_img // NUM_PATCHES, d_txt // NUM_PATCHES
    P: List[Tuple[str, Shape]] = []
    for nm, p in (("self_attention_image", pi), ("self_attention_text", pt)):
        P += _lin(nm + ".W_query", SA_HID, p) + _lin(nm + ".W_key", SA_HID, p) + _lin(nm + ".W_value", SA_OUT, p)
        P += _ln(nm + ".norm", SA_OUT)
    for nm in ("cross_attention_1", "cross_attention_2"):
        P += _lin(nm + ".W_query", CA_HID, SA_OUT) + _lin(nm + ".W_key", CA_HID, SA_OUT)
        P += _lin(nm + ".W_value", CA_OUT, SA_OUT) + _ln(nm + ".norm", CA_OUT)
    ca_flat = CA_OUT * NUM_PATCHES * 2
    if features_only:
        P += _lin("final_features_only_linear", n_classes, d_img + d_txt)
    elif cross_attention_only:
        P += _lin("cross_attention_only_linear", n_classes, ca_flat)
    else:
        P += _lin("final_with_everything", n_classes, ca_flat + d_img + d_txt)
    return P


def head_unused_params(d_img: int, d_txt: int, n_classes: int, fc: int, batch_size: int,
                       features_only: bool, cross_attention_only: bool) -> List[Tuple[str, Shape]]:
    """Present-but-unused keys of the base constructor (multimodal_model.py:199-328), kept so that
    checkpoints interchange.  GRU / Hadamard keys follow torch's parameter naming."""
    P: List[Tuple[str, Shape]] = []
    P += _lin("image_to_hidden_size", fc, d_img) + _lin("text_to_hidden_size", fc, d_txt)
    P += _lin("concat_layer", fc, 2 * fc) + _lin("fc_layer", n_classes, fc)
    P += _lin("image_features_hidden_layer", 256, d_img) + _lin("text_features_hidden_layer", 256, d_txt)
    P += _lin("z_layer", 256, 512) + _lin("fc_layer_gated", n_classes, 256)
    P += _lin("clip_fc_layer", n_classes, batch_size)
    P += [("trans_conv.weight", (8, 8, 2)), ("trans_conv.bias", (8,))]
    P += [("logit_scale", ())]
    P += _lin("output_all_features", 4, 640)
    ca_flat = CA_OUT * NUM_PATCHES * 2
    P += _lin("final", n_classes, ca_flat)
    # the mode-specific classifier heads that are NOT the active one still exist when their flag is set;
    # final_with_everything always exists (multimodal_model.py:290-292)
    if features_only or cross_attention_only:
        P += _lin("final_with_everything", n_classes, ca_flat + d_img + d_txt)
    if features_only and cross_attention_only:
        P += _lin("cross_attention_only_linear", n_classes, ca_flat)
    P += _lin("final_hierarchical_image", 512, d_img + 2560 + 2048)
    P += _lin("final_hierarchical_text", 512, d_txt * 3)
    P += _lin("final_hierarchical_all", n_classes, 1024)
    md, hd, pd = 400, 500, 450
    for g, hid in (("gru_text", md), ("gru_audio", md)):
        P += [(f"{g}.weight_ih_l0", (3 * hid, md)), (f"{g}.weight_hh_l0", (3 * hid, hid)),
              (f"{g}.bias_ih_l0", (3 * hid,)), (f"{g}.bias_hh_l0", (3 * hid,))]
    P += [("fusion.kernel1", (md,)), ("fusion.kernel2", (md,)), ("fusion.bias", (md,))]
    P += [("gru_bimodal.weight_ih_l0", (3 * hd, md)), ("gru_bimodal.weight_hh_l0", (3 * hd, hd)),
          ("gru_bimodal.bias_ih_l0", (3 * hd,)), ("gru_bimodal.bias_hh_l0", (3 * hd,))]
    P += _lin("concat_fc", pd, md + hd)
    P += _lin("modality_image_to_dim", md, d_img) + _lin("modality_text_to_dim", md, d_txt)
    P += _lin("classifier", 4, pd)
    return P
