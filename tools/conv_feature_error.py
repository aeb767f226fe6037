"""Which encoder carries the bf16 error of a conv-backbone MM-RCA?  Engine (bf16) features / logits against the oracle (fp32 CPU) on
the same weights, plus the logits when only ONE of the two feature vectors is the bf16 one (the other taken from the oracle).
    python tools/conv_feature_error.py [eff_v2_medium] [480] [8]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garbage_classification_rca_amd.engine import MMRCAEngine
from garbage_classification_rca_amd.procedural import synth_captions
from oracle import model as O

name = sys.argv[1] if len(sys.argv) > 1 else "eff_v2_medium"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 480
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
torch.set_num_threads(16)
eng = MMRCAEngine("distilbert", name, 4, True, 0, torch.bfloat16, image_size=size)
eng.init_parameters(0)
sd = {k: eng.arena.view(k).detach().cpu().clone() for k in eng.param_keys}
orc = O.build_oracle("distilbert", name, True, False, False, drop_ratio=0.0, enc_dropout=0.0).eval()
orc.text_model.load_flat(sd, "text_model.")
orc.image_model.load_state_dict({k[len("image_model."):]: v for k, v in sd.items() if k.startswith("image_model.")}, strict=False)
orc.load_state_dict({k: v for k, v in sd.items() if not k.startswith(("text_model.", "image_model."))}, strict=False)
ids, mask = (torch.from_numpy(a) for a in synth_captions(B, 64, seed=4321))
images = torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(1234))
logits = eng.forward(ids.cuda(), mask.cuda(), images.cuda(), save=True, bn_train=False).cpu()
feat, cls = eng._saved["feat"][:B].float().cpu(), eng._saved["cls"][:B].float().cpu()
with torch.no_grad():
    t_ref = orc.text_model(ids, mask)[:, 0]
    i_ref = orc.image_model(images)
    ref = orc.head(t_ref, i_ref)
    only_img = orc.head(t_ref, feat)
    only_txt = orc.head(cls, i_ref)
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
print(f"{name} {size}^2 B={B}: image feature {rel(feat, i_ref):.2e}  text feature {rel(cls, t_ref):.2e}  logits {rel(logits, ref):.2e}")
print(f"   logits with only the bf16 IMAGE feature {rel(only_img, ref):.2e};  with only the bf16 TEXT feature {rel(only_txt, ref):.2e}")
