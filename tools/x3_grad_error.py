"""bf16x3 mode: worst relative gradient error (per tensor: max |err| / max(|grad|.max, 1e-3 of the largest gradient entry)) of the
engine against the oracle evaluated in float64, on the initialisation the model trains from (the measurement behind the
default MMRCA_X3_WGRAD_PASSES / MMRCA_X3_DGRAD_PASSES).  python tools/x3_grad_error.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd import lib as L   # noqa: E402
from oracle import model as O                            # noqa: E402


def main():
    from tests.test_x3_gpu import _pair_init_weights
    from tests.test_engine_gpu import _inputs, rel
    from garbage_classification_rca_amd.engine import MMRCAEngine
    B, S_len = 3, 24
    eng, orc, sd = _pair_init_weights(0)
    if len(sys.argv) > 1:          # another precision mode on the same weights: python tools/x3_grad_error.py bf16x3f | bf16 | fp32
        dt = {"bf16": torch.bfloat16, "fp32": torch.float32}.get(sys.argv[1], sys.argv[1])
        eng.release_buffers()
        eng = MMRCAEngine("distilbert", "transformer_B16", 4, True, 0, dt)
        eng.load_arrays(sd)
    ids, mask, images = _inputs(B, S_len)
    logits = eng.forward(ids.cuda(), mask.cuda(), images.cuda())
    orc = orc.double()
    for p in orc.parameters():
        p.requires_grad_(True)
    ref = orc(ids, mask, images.double(), eval=True)
    labels = torch.tensor([0, 1, 2][:B])
    cw = torch.tensor([0.7, 1.3, 0.9, 1.1])
    O.cross_entropy(ref, labels, cw.double(), 0.1).backward()
    loss, dl = torch.empty(1, device="cuda"), torch.empty(B, 4, device="cuda")
    L.xent_fwd_bwd(logits, labels.int().cuda(), cw.cuda(), 0.1, loss, dl, B, 4)
    eng.arena.g.zero_()
    eng.backward(dl)
    torch.cuda.synchronize()
    named = {"text_model." + k.replace("/", "."): p for k, p in orc.text_model.params.items()}
    named.update({"image_model." + k.replace("/", "."): p for k, p in orc.image_model.params.items()})
    named.update({k: p for k, p in orc.named_parameters() if not k.startswith(("text_model.", "image_model."))})
    gmax = max(float(p.grad.abs().max()) for p in named.values() if p.grad is not None)
    errs = []
    for k in eng.param_keys:
        gr = named[k].grad
        if gr is None:
            continue
        got = eng.arena.view(k, "g").cpu().double()
        err = (got - gr.view_as(got)).abs().max().item()
        errs.append((err / max(gr.abs().max().item(), 1e-3 * gmax), k))
    errs.sort(reverse=True)
    print(f"logits {rel(logits, ref.detach()):.2e}  worst grad {errs[0][0]:.2e} ({errs[0][1]})  median {errs[len(errs) // 2][0]:.2e}  "
          f"tensors over 2e-3: {sum(e > 2e-3 for e, _ in errs)} of {len(errs)}")


if __name__ == "__main__":
    main()
