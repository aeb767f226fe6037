"""Where does the bf16 conv backbone leave the fp32 one?  Per-block output error (relative to the block's largest entry) of the bf16
ConvEncoder against the fp32 ConvEncoder on the same weights and input, in train (batch statistics) and eval (running statistics)
mode, plus the final features against the oracle.   python tools/conv_error_probe.py [name] [B] [size]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_conv_gpu import _conv_pair, _sd_blocks    # noqa: E402
from oracle import conv_models as CM                       # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "eff_v2_medium"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
size = int(sys.argv[3]) if len(sys.argv) > 3 else 160
torch.set_num_threads(16)
rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
for train in (False, True):
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        enc, own, orc = _conv_pair(name, dt)
        g = torch.Generator().manual_seed(9)
        images = torch.randn(B, 3, size, size, generator=g)
        blocks = _sd_blocks(orc)
        keep = (torch.rand(len(blocks), B, generator=g) > 0.2).float()
        enc.injected_keep = keep
        feat = enc.forward(images.cuda(), save=True, train=train)
        torch.cuda.synchronize()
        per = []
        for bi, bs in enumerate(enc.saved["blocks"]):
            # the input of block bi+1 = output of block bi: take the x saved by the first unit of the next block
            pass
        xs = [bs["units"][0]["x"] if "units" in bs else None for bs in enc.saved["blocks"]]
        outs[dt] = (feat.float().cpu(), [x.float().cpu().clone() if x is not None else None for x in xs],
                    [(bs["units"][-1]["mean"].cpu().clone(), bs["units"][-1]["rstd"].cpu().clone()) for bs in enc.saved["blocks"] if "units" in bs])
        if dt == torch.float32:
            orc.train(train)
            for m, k in zip(blocks, keep):
                m.stochastic_depth.keep = k
            with torch.no_grad():
                ref = CM.conv_features(orc, images)
            print(f"train={train}: fp32 engine features vs oracle {rel(feat.cpu(), ref):.2e}")
        enc.release()
    f32, x32, st32 = outs[torch.float32]
    f16, x16, st16 = outs[torch.bfloat16]
    print(f"train={train}: bf16 features vs fp32 engine {rel(f16, f32):.2e}")
    for bi, (a, b) in enumerate(zip(x16, x32)):
        if a is None:
            continue
        n = b.shape[0]
        l2 = float((a.double() - b.double()).norm() / b.double().norm())
        print(f"   input of block {bi:2d} [{tuple(b.shape)}]: max-rel {rel(a, b):.2e}  L2-rel {l2:.2e}   rstd of its last BN: rel {rel(st16[bi][1], st32[bi][1]):.2e}")
