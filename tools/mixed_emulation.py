"""Which roundings can the forward afford?  A CPU emulation of the engine's precision modes on the oracle's architecture:
every tensor the GPU path would store or feed to a matrix core is rounded exactly where the mode rounds it, everything else
runs in float64, and the logits are compared with the float64 evaluation of the same weights (the reference is fp32 end to
end, CVPR_code/multimodal_model.py:651-726; north_star: logits within 1e-3 relative).  This is the measurement behind the
`mixed` mode's choice of fp16 forward operands.  Test infrastructure: imports the oracle.

    python tools/mixed_emulation.py [--batch 8] [--seq 64] [--seeds 0 1]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arch as S            # noqa: E402
from oracle import model as O           # noqa: E402
from garbage_classification_rca_amd.procedural import synth_captions   # noqa: E402


def rnd(t, kind):
    if kind is None:
        return t
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[kind]
    return t.to(dt).to(torch.float64)


class Mode:
    """opnd: rounding of both operands of every nn.Linear product; act_lo: the activation operand additionally carries its bf16
    remainder (two-pass product A_hi B_hi + A_lo B_hi); store: rounding of stored GEMM outputs (q|k|v, branch outputs, gelu);
    stream: rounding of the residual stream / LayerNorm outputs' storage; attn: rounding of the attention operands (q, k, v, P)."""

    def __init__(self, name, opnd=None, store=None, stream=None, attn=None, two_pass=False, ln_out=None, param=None, w_two=False, branch="same"):
        self.name, self.opnd, self.store, self.stream, self.attn, self.two_pass, self.ln_out = name, opnd, store, stream, attn, two_pass, ln_out
        self.w_two = w_two          # the weight operand carries its remainder plane too (with two_pass: the full three-pass product)
        self.branch = store if branch == "same" else branch      # storage of the two GEMM outputs that are added to the residual stream
        self.param = param          # rounding of the parameters that are NOT matrix operands (embedding tables, LayerNorm, biases) and of the features handed to the head


def lin(x, w, b, m: Mode):
    if m.two_pass:       # activation exact to two planes, weight rounded once
        xa = rnd(x, m.opnd) + rnd(x - rnd(x, m.opnd), m.opnd)
    else:
        xa = rnd(x, m.opnd)
    wa = rnd(w, m.opnd) + rnd(w - rnd(w, m.opnd), m.opnd) if m.w_two else rnd(w, m.opnd)
    return rnd(F.linear(xa, wa) + rnd(b, m.param), "fp32")


def attention(q, k, v, dh, m: Mode, keymask=None, row_valid=None):
    q, k, v = rnd(q, m.attn), rnd(k, m.attn), rnd(v, m.attn)
    sc = q @ k.transpose(2, 3) * dh ** -0.5
    if keymask is not None:
        sc = sc.masked_fill(~keymask, float("-inf"))
        sc = torch.where(row_valid, sc, torch.zeros_like(sc))
    mx = sc.max(dim=-1, keepdim=True).values
    p = torch.exp(sc - mx)
    den = p.sum(-1, keepdim=True)
    o = (rnd(p, m.attn) @ v) / den          # the kernels round the unnormalised probabilities and divide the fp32 sum
    if row_valid is not None:
        o = o * row_valid
    return o


def vit(orc, images, m: Mode):
    s = orc.spec
    P = lambda k: rnd(orc.P(k), m.param) if orc.P(k).dim() != 2 or "embedding" in k else orc.P(k)
    B = images.shape[0]
    patches = F.unfold(images, s.patch, stride=s.patch).transpose(1, 2)          # [B, nP, 3*p*p]
    x = lin(patches, orc.P("conv_proj.weight").reshape(s.dim, -1), P("conv_proj.bias"), m)
    x = torch.cat([P("class_token").expand(B, -1, -1), x], dim=1) + P("encoder.pos_embedding")
    x = rnd(x, m.stream)
    dh, T = s.dim // s.heads, x.shape[1]
    for i in range(s.layers):
        L = f"encoder.layers.encoder_layer_{i}."
        y = rnd(F.layer_norm(x, (s.dim,), P(L + "ln_1.weight"), P(L + "ln_1.bias"), s.ln_eps), m.ln_out)
        qkv = rnd(lin(y, P(L + "self_attention.in_proj_weight"), P(L + "self_attention.in_proj_bias"), m), m.store)
        q, k, v = (t.view(B, T, s.heads, dh).transpose(1, 2) for t in qkv.split(s.dim, dim=-1))
        ctx = rnd(attention(q, k, v, dh, m).transpose(1, 2).reshape(B, T, s.dim), m.store)
        x = rnd(x + rnd(lin(ctx, P(L + "self_attention.out_proj.weight"), P(L + "self_attention.out_proj.bias"), m), m.branch), m.stream)
        y = rnd(F.layer_norm(x, (s.dim,), P(L + "ln_2.weight"), P(L + "ln_2.bias"), s.ln_eps), m.ln_out)
        h = rnd(F.gelu(lin(y, P(L + "mlp.0.weight"), P(L + "mlp.0.bias"), m)), m.store)
        x = rnd(x + rnd(lin(h, P(L + "mlp.3.weight"), P(L + "mlp.3.bias"), m), m.branch), m.stream)
    return F.layer_norm(x, (s.dim,), P("encoder.ln.weight"), P("encoder.ln.bias"), s.ln_eps)[:, 0]


def text(orc, ids, mask, m: Mode):
    s = orc.spec
    P = lambda k: rnd(orc.P(k), m.param) if orc.P(k).dim() != 2 or "embedding" in k else orc.P(k)
    B, T = ids.shape
    if s.pos_offset:
        nonpad = (ids != s.pad_id).to(torch.int64)
        pos = torch.cumsum(nonpad, dim=1) * nonpad + s.pad_id
    else:
        pos = torch.arange(T).unsqueeze(0).expand(B, T)
    x = F.embedding(ids, P("embeddings.word_embeddings.weight")) + F.embedding(pos, P("embeddings.position_embeddings.weight"))
    if s.type_vocab:
        x = x + P("embeddings.token_type_embeddings.weight")[0]
    x = rnd(F.layer_norm(x, (s.dim,), P("embeddings.LayerNorm.weight"), P("embeddings.LayerNorm.bias"), s.ln_eps), m.stream)
    keymask = mask.to(torch.bool)[:, None, None, :]
    row_valid = keymask.any(dim=-1, keepdim=True)
    dh = s.dim // s.heads
    for i in range(s.layers):
        K = S.text_layer_keys(s, i)
        l = lambda t, nm: lin(t, P(K[nm] + ".weight"), P(K[nm] + ".bias"), m)
        xin = rnd(x, m.ln_out)              # post-LN encoders: the LayerNorm output is both the stream and the GEMM operand
        q, k, v = (rnd(l(xin, n), m.store).view(B, T, s.heads, dh).transpose(1, 2) for n in ("q", "k", "v"))
        ctx = rnd(attention(q, k, v, dh, m, keymask, row_valid).transpose(1, 2).reshape(B, T, s.dim), m.store)
        x = rnd(F.layer_norm(rnd(l(ctx, "o"), m.branch) + x, (s.dim,), P(K["ln1"] + ".weight"), P(K["ln1"] + ".bias"), s.ln_eps), m.stream)
        h = rnd(F.gelu(l(rnd(x, m.ln_out), "f1")), m.store)
        x = rnd(F.layer_norm(rnd(l(h, "f2"), m.branch) + x, (s.dim,), P(K["ln2"] + ".weight"), P(K["ln2"] + ".bias"), s.ln_eps), m.stream)
    return x[:, 0]


MODES = [
    Mode("fp32 everywhere (reference arithmetic, fp32 rounding of products only)"),
    Mode("bf16 (operands, storage, stream)", opnd="bf16", store="bf16", stream="bf16", attn="bf16", ln_out="bf16"),
    Mode("bf16 (operands, storage, stream, ALL parameters, features)  [the bf16 engine]", opnd="bf16", store="bf16", stream="bf16", attn="bf16", ln_out="bf16", param="bf16"),
    Mode("fp32 arithmetic, bf16 non-matrix parameters + features only", param="bf16"),
    Mode("bf16 operands + attention, fp32 stream and storage", opnd="bf16", attn="bf16"),
    Mode("bf16 operands, two-pass activations, fp32 stream/storage, bf16 attention", opnd="bf16", attn="bf16", two_pass=True),
    Mode("bf16 operands, two-pass activations, fp32 stream/storage, fp32 attention", opnd="bf16", two_pass=True),
    Mode("fp16 operands + fp16 storage + fp16 attention, fp32 stream  [mixed]", opnd="fp16", store="fp16", attn="fp16", ln_out="fp16"),
    Mode("fp16 operands + fp16 storage + fp16 attention + fp16 stream", opnd="fp16", store="fp16", attn="fp16", stream="fp16", ln_out="fp16"),
    Mode("fp16 operands, bf16 storage of branch outputs, fp32 stream", opnd="fp16", store="bf16", attn="fp16", ln_out="fp16"),
    Mode("[mixed] + fp32 branch outputs", opnd="fp16", store="fp16", attn="fp16", ln_out="fp16", branch=None),
    Mode("[mixed] + fp32 branch outputs + exact attention (q|k|v fp32)", opnd="fp16", store=None, attn=None, ln_out="fp16", branch=None),
    Mode("fp16 two-pass activations, single fp16 weights, fp32 storage, exact attention", opnd="fp16", two_pass=True),
    Mode("fp16 two-pass activations, single fp16 weights, fp32 storage, fp16 attention", opnd="fp16", two_pass=True, attn="fp16"),
    Mode("fp16 three-pass, fp32 storage, fp16 attention", opnd="fp16", two_pass=True, w_two=True, attn="fp16"),
    Mode("bf16 three-pass (bf16x3), fp32 storage, exact attention", opnd="bf16", two_pass=True, w_two=True),
    Mode("bf16 three-pass (bf16x3), fp32 storage, fp16 attention", opnd="bf16", two_pass=True, w_two=True, attn="fp16"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq", type=int, default=64)
    ap.add_argument("--seeds", type=int, nargs="+", default=[0])
    ap.add_argument("--text_model", default="distilbert")
    ap.add_argument("--image_model", default="transformer_B16")
    ap.add_argument("--train_steps", type=int, default=0, help="needs an MI355X: first run this many bf16 SGD steps at batch 256 as bench.py does "
                    "and emulate on THOSE weights (bench.py's parity object is measured after its timed steps)")
    ap.add_argument("--weights", default="init", choices=["init", "wide"], help="init: N(0, 0.02) as the bench; wide: N(0, 1/sqrt(fan_in)), "
                    "branches as large as the stream (closer to a trained checkpoint)")
    a = ap.parse_args()
    torch.set_num_threads(8)
    for seed in a.seeds:
        orc = O.build_oracle(a.text_model, a.image_model, True, drop_ratio=0.0, enc_dropout=0.0).eval().double()
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, p in orc.named_parameters():
                if not name.startswith(("text_model.", "image_model.")):
                    continue
                if p.dim() >= 2:
                    std = 0.02 if a.weights == "init" else (1.0 / (p[0].numel() ** 0.5) if "embedding" not in name else 0.05)
                    p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float64) * std)
                    p.copy_(p.float().double())        # the masters are fp32
                elif name.endswith("weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()
        if a.train_steps:
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            from stream_error import trained_weights
            sd = trained_weights(a.train_steps)
            orc = orc.float()
            orc.text_model.load_flat(sd, "text_model.")
            orc.image_model.load_flat(sd, "image_model.")
            orc.load_state_dict({k: v for k, v in sd.items() if not k.startswith(("text_model.", "image_model."))}, strict=False)
            orc = orc.double()
            torch.set_num_threads(16)
        ids, mask = (torch.from_numpy(x) for x in synth_captions(a.batch, a.seq, seed=4321 + seed))
        images = torch.randn(a.batch, 3, 224, 224, generator=torch.Generator().manual_seed(1234 + seed)).double()
        with torch.no_grad():
            ref = orc(ids, mask, images, eval=True)
            print(f"seed {seed}: |logits| max {ref.abs().max():.4f}")
            for m in MODES:
                txt = text(orc.text_model, ids, mask, m)
                img = vit(orc.image_model, images, m)
                out = orc.head(rnd(txt, m.param), rnd(img, m.param))
                err = (out - ref).abs()
                ft = float((txt - orc.text_model(ids, mask)[:, 0]).abs().max() / orc.text_model(ids, mask)[:, 0].abs().max())
                fi = float((img - orc.image_model(images)).abs().max() / orc.image_model(images).abs().max())
                print(f"  {m.name:90s} logits_rel {float(err.max() / ref.abs().max()):.2e}   text feat {ft:.2e}  image feat {fi:.2e}")


if __name__ == "__main__":
    main()
