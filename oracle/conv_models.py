"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain torch.nn, fp32) of the conv image backbones.

Not part of the product (see oracle/model.py for the rules).  This file restates the PUBLISHED torchvision
architectures that the reference instantiates:

* ``efficientnet_v2_m`` -- ``CVPR_code/multimodal_model.py:113-126`` (``eff_net_v2()``), wrapped by
  ``EfficientNetV2MFullFeatureExtractor`` (``:11-36``: stem = features[:2], stage1..6 = features[2..7],
  final_conv = features[8], avgpool, classifier truncated to its Dropout); MM_RCA consumes the pooled [B,1280] (``:659``);
* ``efficientnet_v2_l`` -- ``models.py`` EffNetV2_L / ``main_image.py:296-302`` (BASELINE configs[2]);
* ``shufflenet_v2_x2_0`` -- ``models.py:261-278`` (BASELINE configs[0]); features = conv5 output averaged over H, W ([B,2048]).

PARITY UNPINNED with respect to torchvision: torchvision is not installed in this image and the reference holds no
test or fixture for these networks, so the module structure, parameter names, BatchNorm epsilons (1e-3 for
EfficientNetV2, 1e-5 for ShuffleNetV2), stochastic-depth schedule (0.3 / 0.5, "row" mode) and channel tables below are
restated from the published torchvision source (torchvision/models/efficientnet.py, shufflenetv2.py, ops/misc.py) and
checked only for self-consistency (parameter counts: 52,863,480 / 117,239,396 with the 4-class head, as quoted at
``main_image.py:295,302``; see tests).  The product's HIP kernels are tested against THIS restatement.
"""
from __future__ import annotations

from functools import partial
from typing import List, Tuple

import torch
from torch import nn


# ----------------------------------------------------------------------------------------------------------------------
# EfficientNetV2 (torchvision/models/efficientnet.py)
# ----------------------------------------------------------------------------------------------------------------------
class ConvNormAct(nn.Sequential):
    """torchvision.ops.misc.Conv2dNormActivation: conv (no bias) -> norm -> activation; children named 0, 1, 2."""

    def __init__(self, cin, cout, k=3, stride=1, groups=1, norm=nn.BatchNorm2d, act=nn.SiLU):
        layers: List[nn.Module] = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), norm(cout)]
        if act is not None:
            layers.append(act())
        super().__init__(*layers)


class SqueezeExcitation(nn.Module):
    def __init__(self, channels, squeeze):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(channels, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, channels, 1)
        self.activation = nn.SiLU()
        self.scale_activation = nn.Sigmoid()

    def forward(self, x):
        s = self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))
        return x * s


class StochasticDepth(nn.Module):
    """torchvision.ops.stochastic_depth, mode "row": a per-sample Bernoulli(1-p) keep mask scaled by 1/(1-p), train only.
    ``keep`` can be injected ([B] 0/1 tensor) so that a test drives the product with the same mask."""

    def __init__(self, p):
        super().__init__()
        self.p, self.keep = float(p), None

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        keep = self.keep if self.keep is not None else torch.bernoulli(torch.full((x.shape[0],), 1.0 - self.p))
        return x * (keep.to(x.dtype) / (1.0 - self.p)).view(-1, 1, 1, 1)


class MBConv(nn.Module):
    def __init__(self, expand, k, stride, cin, cout, sd_prob, norm, fused):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        cexp = cin * expand
        layers: List[nn.Module] = []
        if fused:
            if cexp != cin:
                layers.append(ConvNormAct(cin, cexp, k, stride, norm=norm))
                layers.append(ConvNormAct(cexp, cout, 1, norm=norm, act=None))
            else:
                layers.append(ConvNormAct(cin, cout, k, stride, norm=norm))
        else:
            if cexp != cin:
                layers.append(ConvNormAct(cin, cexp, 1, norm=norm))
            layers.append(ConvNormAct(cexp, cexp, k, stride, groups=cexp, norm=norm))
            layers.append(SqueezeExcitation(cexp, max(1, cin // 4)))
            layers.append(ConvNormAct(cexp, cout, 1, norm=norm, act=None))
        self.block = nn.Sequential(*layers)
        self.stochastic_depth = StochasticDepth(sd_prob)

    def forward(self, x):
        y = self.block(x)
        if self.use_res_connect:
            y = self.stochastic_depth(y) + x
        return y


# (fused, expand, kernel, stride, in, out, layers)
EFFNET_V2 = {
    "eff_v2_medium": dict(sd=0.3, cfg=[(1, 1, 3, 1, 24, 24, 3), (1, 4, 3, 2, 24, 48, 5), (1, 4, 3, 2, 48, 80, 5), (0, 4, 3, 2, 80, 160, 7),
                                       (0, 6, 3, 1, 160, 176, 14), (0, 6, 3, 2, 176, 304, 18), (0, 6, 3, 1, 304, 512, 5)]),
    "eff_v2_large": dict(sd=0.5, cfg=[(1, 1, 3, 1, 32, 32, 4), (1, 4, 3, 2, 32, 64, 7), (1, 4, 3, 2, 64, 96, 7), (0, 4, 3, 2, 96, 192, 10),
                                      (0, 6, 3, 1, 192, 224, 19), (0, 6, 3, 2, 224, 384, 25), (0, 6, 3, 1, 384, 640, 7)]),
}


def efficientnet_v2_features(name: str) -> nn.Sequential:
    """``torchvision.models.efficientnet_v2_{m,l}().features`` (children 0..8)."""
    spec = EFFNET_V2[name]
    norm = partial(nn.BatchNorm2d, eps=1e-3)
    cfg = spec["cfg"]
    layers: List[nn.Module] = [ConvNormAct(3, cfg[0][4], 3, 2, norm=norm)]
    total = sum(c[6] for c in cfg)
    bid = 0
    for fused, expand, k, stride, cin, cout, n in cfg:
        stage = []
        for i in range(n):
            stage.append(MBConv(expand, k, stride if i == 0 else 1, cin if i == 0 else cout, cout, spec["sd"] * bid / total, norm, bool(fused)))
            bid += 1
        layers.append(nn.Sequential(*stage))
    layers.append(ConvNormAct(cfg[-1][5], 1280, 1, norm=norm))
    feats = nn.Sequential(*layers)
    for m in feats.modules():                       # torchvision's init
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out")
            if m.bias is not None:
                nn.init.zeros_(m.bias)
    return feats


class OracleEffNetV2Extractor(nn.Module):
    """``EfficientNetV2MFullFeatureExtractor`` (multimodal_model.py:11-36) over the restated features: same attribute
    names, hence the same state_dict keys (stem.0.0.weight, stem.1.0.block.0.0.weight, stage1..., final_conv.0.weight)."""

    def __init__(self, name="eff_v2_medium"):
        super().__init__()
        f = efficientnet_v2_features(name)
        self.stem = f[:2]
        self.stage1, self.stage2, self.stage3, self.stage4, self.stage5, self.stage6 = f[2], f[3], f[4], f[5], f[6], f[7]
        self.final_conv = f[8]
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential(nn.Dropout(0.3 if name == "eff_v2_medium" else 0.4))     # truncated (:120-121)
        self.pooled_only = False        # True: forward returns only the pooled features (what MM_RCA.forward uses, :659)

    def forward(self, x):
        x = self.stem(x)
        x = self.stage1(x)
        x = self.stage2(x)
        out_stage3 = self.stage3(x)
        x = self.stage4(out_stage3)
        x = self.stage5(x)
        out_stage6 = self.stage6(x)
        x = self.final_conv(out_stage6)
        pooled = torch.flatten(self.avgpool(x), 1)
        return pooled if self.pooled_only else (out_stage3, out_stage6, pooled)


# ----------------------------------------------------------------------------------------------------------------------
# ShuffleNetV2 x2.0 (torchvision/models/shufflenetv2.py)
# ----------------------------------------------------------------------------------------------------------------------
def channel_shuffle(x, groups):
    b, c, h, w = x.shape
    return x.view(b, groups, c // groups, h, w).transpose(1, 2).contiguous().view(b, c, h, w)


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride):
        super().__init__()
        self.stride = stride
        bf = oup // 2
        dw = lambda c, s: nn.Conv2d(c, c, 3, s, 1, groups=c, bias=False)
        if stride > 1:
            self.branch1 = nn.Sequential(dw(inp, stride), nn.BatchNorm2d(inp), nn.Conv2d(inp, bf, 1, 1, 0, bias=False),
                                         nn.BatchNorm2d(bf), nn.ReLU())
        else:
            self.branch1 = nn.Sequential()
        self.branch2 = nn.Sequential(nn.Conv2d(inp if stride > 1 else bf, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf), nn.ReLU(),
                                     dw(bf, stride), nn.BatchNorm2d(bf), nn.Conv2d(bf, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf), nn.ReLU())

    def forward(self, x):
        if self.stride == 1:
            x1, x2 = x.chunk(2, dim=1)
            out = torch.cat((x1, self.branch2(x2)), dim=1)
        else:
            out = torch.cat((self.branch1(x), self.branch2(x)), dim=1)
        return channel_shuffle(out, 2)


SHUFFLE_X2 = dict(repeats=[4, 8, 4], channels=[24, 244, 488, 976, 2048])


class OracleShuffleNetV2(nn.Module):
    """``shufflenet_v2_x2_0`` without its ``fc``: forward returns the [B,2048] mean-pooled conv5 features."""

    def __init__(self):
        super().__init__()
        ch = SHUFFLE_X2["channels"]
        self.conv1 = nn.Sequential(nn.Conv2d(3, ch[0], 3, 2, 1, bias=False), nn.BatchNorm2d(ch[0]), nn.ReLU())
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inp = ch[0]
        for name, rep, oup in zip(("stage2", "stage3", "stage4"), SHUFFLE_X2["repeats"], ch[1:4]):
            seq = [InvertedResidual(inp, oup, 2)] + [InvertedResidual(oup, oup, 1) for _ in range(rep - 1)]
            setattr(self, name, nn.Sequential(*seq))
            inp = oup
        self.conv5 = nn.Sequential(nn.Conv2d(inp, ch[4], 1, 1, 0, bias=False), nn.BatchNorm2d(ch[4]), nn.ReLU())

    def forward(self, x):
        x = self.maxpool(self.conv1(x))
        x = self.conv5(self.stage4(self.stage3(self.stage2(x))))
        return x.mean([2, 3])


def build_conv_oracle(name: str) -> nn.Module:
    if name in ("eff_v2_medium", "eff_v2_large"):
        return OracleEffNetV2Extractor(name)
    if name == "shuffle_net":
        return OracleShuffleNetV2()
    raise ValueError(name)


def conv_features(m: nn.Module, images: torch.Tensor) -> torch.Tensor:
    out = m(images)
    return out[2] if isinstance(out, tuple) else out
