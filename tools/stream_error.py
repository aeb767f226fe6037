"""Where does a precision mode lose its digits?  Runs the engine (any --dtype) on the bench's initialisation and prints, against
the oracle evaluated in float64 on the same weights: the error of the residual stream entering every encoder layer, of the two
feature vectors and of the logits (all relative to the reference tensor's largest entry).  Needs an MI355X.

    python tools/stream_error.py --dtype bf16 [--batch 8]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd.engine import MMRCAEngine, make_text_pack   # noqa: E402
from garbage_classification_rca_amd.procedural import synth_captions             # noqa: E402
from oracle import model as O                                                    # noqa: E402
from oracle import arch as S                                                     # noqa: E402
import torch.nn.functional as F                                                  # noqa: E402


def oracle_streams(orc, ids, mask, images):
    """float64 residual stream at the input of every layer (same math as oracle/model.py)"""
    tv, tt = [], []
    s, P = orc.image_model.spec, orc.image_model.P
    B = images.shape[0]
    x = F.conv2d(images, P("conv_proj.weight"), P("conv_proj.bias"), stride=s.patch).reshape(B, s.dim, -1).permute(0, 2, 1)
    x = torch.cat([P("class_token").expand(B, -1, -1), x], dim=1) + P("encoder.pos_embedding")
    dh, T = s.dim // s.heads, x.shape[1]
    for i in range(s.layers):
        tv.append(x)
        L = f"encoder.layers.encoder_layer_{i}."
        y = F.layer_norm(x, (s.dim,), P(L + "ln_1.weight"), P(L + "ln_1.bias"), s.ln_eps)
        qkv = F.linear(y, P(L + "self_attention.in_proj_weight"), P(L + "self_attention.in_proj_bias"))
        q, k, v = (t.view(B, T, s.heads, dh).transpose(1, 2) for t in qkv.split(s.dim, dim=-1))
        ctx = (torch.softmax(q @ k.transpose(2, 3) * dh ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(B, T, s.dim)
        x = x + F.linear(ctx, P(L + "self_attention.out_proj.weight"), P(L + "self_attention.out_proj.bias"))
        y = F.layer_norm(x, (s.dim,), P(L + "ln_2.weight"), P(L + "ln_2.bias"), s.ln_eps)
        x = x + F.linear(F.gelu(F.linear(y, P(L + "mlp.0.weight"), P(L + "mlp.0.bias"))), P(L + "mlp.3.weight"), P(L + "mlp.3.bias"))
    return tv


def trained_weights(steps, Bt=256, S_len=64):
    from garbage_classification_rca_amd.multimodal_model import MM_RCA
    from garbage_classification_rca_amd.optim import FlatSGD
    from garbage_classification_rca_amd.training import FusedCrossEntropy, hip_train_step
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = MM_RCA(4, 0.6, 0.0, 0.7, 256, "distilbert", Bt, True, False, False, image_model_name="transformer_B16", dtype=torch.bfloat16,
                       device=torch.device("cuda"), init_seed=0)
    model.train()
    for p in model.parameters():
        p.requires_grad = True
    opt, crit = FlatSGD(model, lr=1e-3, weight_decay=1e-2), FusedCrossEntropy(None, 0.0)
    nb = 2
    ids, mask_host = synth_captions(Bt * nb, S_len, seed=4321)
    ids, mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask_host).cuda()
    images = torch.randn(Bt * nb, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1234))
    labels = (torch.arange(Bt * nb, device="cuda") % 4).to(torch.int32)
    with contextlib.redirect_stdout(io.StringIO()):
        for i in range(steps):
            j = (i % nb) * Bt
            loss = hip_train_step(model, ids[j:j + Bt], mask[j:j + Bt], images[j:j + Bt], labels[j:j + Bt], crit, opt, None,
                                  text_pack=make_text_pack(mask_host[j:j + Bt], "cuda"))
    print(f"trained {steps} steps, loss {float(loss):.4f}")
    e = model.engine
    sd = {k: e.arena.view(k).detach().cpu().clone() for k in e.param_keys}
    e.release_buffers()
    del model, opt
    torch.cuda.empty_cache()
    return sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq", type=int, default=64)
    ap.add_argument("--train_steps", type=int, default=0, help="first run this many bf16 SGD steps at batch 256 the way bench.py does (its parity "
                    "object is measured on the weights AFTER the timed steps)")
    a = ap.parse_args()
    B = a.batch
    dt = {"bf16": torch.bfloat16, "fp32": torch.float32}.get(a.dtype, a.dtype)
    eng = MMRCAEngine("distilbert", "transformer_B16", 4, True, 0, dt)
    eng.init_parameters(seed=0)
    if a.train_steps:
        sd = trained_weights(a.train_steps)
        eng.load_arrays(sd)
    sd = {k: eng.arena.view(k).detach().cpu().clone() for k in eng.param_keys}
    orc = O.build_oracle("distilbert", "transformer_B16", True, drop_ratio=0.0, enc_dropout=0.0).eval()
    orc.text_model.load_flat(sd, "text_model.")
    orc.image_model.load_flat(sd, "image_model.")
    orc.load_state_dict({k: v for k, v in sd.items() if not k.startswith(("text_model.", "image_model."))}, strict=False)
    orc = orc.double()
    ids, mask = (torch.from_numpy(x) for x in synth_captions(B, a.seq, seed=4321))
    images = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(1234))
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = orc(ids, mask, images.double(), eval=True)
        txt_ref = orc.text_model(ids, mask)[:, 0]
        img_ref = orc.image_model(images.double())
        streams = oracle_streams(orc, ids, mask, images.double())
    pack = make_text_pack(mask.numpy(), "cuda")
    logits = eng.forward(ids.cuda(), mask.cuda(), images.cuda(), save=True, text_pack=pack)
    torch.cuda.synchronize()
    rel = lambda got, r: float((got.double().cpu() - r).abs().max() / r.abs().max())
    sv = eng._saved
    Tn, D = eng.vs.tokens, eng.vs.dim
    for i, a_ in enumerate(sv["vision"]["layers"]):
        x = a_["x"]
        if hasattr(x, "hi"):
            x = x.hi.float() + x.lo.float()
        print(f"vision stream into layer {i:2d}: {rel(x[:B * Tn].view(B, Tn, D), streams[i]):.2e}")
    print(f"image feature {rel(sv['feat'][:B], img_ref):.2e}   text feature {rel(sv['cls'][:B], txt_ref):.2e}   logits {rel(logits, ref):.2e}"
          f"   (|logits| max {float(ref.abs().max()):.3f})")


if __name__ == "__main__":
    main()
