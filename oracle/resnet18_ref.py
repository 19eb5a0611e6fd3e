"""Oracle: fp32 CPU restatement of the ResNet18 forward the reference runs.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED by reference
fixtures: the graph itself is torchvision.models.resnet18 (third party, pinned
torchvision==0.16.0+cu121 in /root/reference/src/requirements.txt:111, absent
here).  What is restated is the published torchvision ResNet-18 definition
(BasicBlock x [2,2,2,2], widths 64/128/256/512, 7x7/2 stem, 3x3/2 max-pool,
1x1/2 projection shortcut on the first block of stages 2-4, adaptive avg-pool,
fc), evaluated the way the reference's wrappers call it:

  * ResNet18FeatureExtractor.forward  src/models/resnet.py:36-40
        nn.Sequential(*children[:-1]) then view(B,-1)          -> [B,512]
  * UnifiedResNet.forward             src/models/resnet.py:54-55
        fc = Identity (or Linear(512,2) when classifier=True)  -> [B,512]|[B,2]
  * ResNet18Classifier.forward        src/models/resnet.py:69-77
        fc = Linear(512,2)                                     -> [B,2]
  * argmax labels                     src/main.py:510, :521, :1010

All arithmetic is torch.nn.functional on CPU in fp32, eval-mode batch-norm
with eps=1e-5 (torchvision default).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
STAGES = (("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2), ("layer4", 512, 2))

# The four key layouts the reference writes / expects.
#   classifier : "model.<tv>"   (optionally "module.model.<tv>" under DataParallel)
#   unified    : "model.<tv>"
#   simclr     : "encoder.<tv>" + "projector.{0,2}.*"
#   extractor  : "features.{0,1,4,5,6,7}.<...>"  (children()[:-1] of resnet18)
_FEATURES_INDEX = {"conv1": "0", "bn1": "1", "layer1": "4", "layer2": "5", "layer3": "6", "layer4": "7"}


def canonical_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Map any of the reference's key layouts to bare torchvision names."""
    out: Dict[str, torch.Tensor] = {}
    inv_features = {v: k for k, v in _FEATURES_INDEX.items()}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        if k.startswith("model."):
            k = k[len("model."):]
        elif k.startswith("encoder."):
            k = k[len("encoder."):]
        elif k.startswith("features."):
            rest = k[len("features."):]
            idx, _, tail = rest.partition(".")
            if idx not in inv_features:
                continue
            k = inv_features[idx] + "." + tail
        elif k.startswith("projector."):
            out[k] = v
            continue
        out[k] = v
    return out


def _bn(x, sd, name):
    return F.batch_norm(
        x,
        sd[name + ".running_mean"],
        sd[name + ".running_var"],
        sd[name + ".weight"],
        sd[name + ".bias"],
        training=False,
        momentum=0.0,
        eps=BN_EPS,
    )


def basic_block(x, sd, prefix: str, stride: int, taps: Optional[dict] = None):
    """torchvision BasicBlock: conv3x3-bn-relu, conv3x3-bn, (+proj), add, relu."""
    identity = x
    y = F.conv2d(x, sd[prefix + ".conv1.weight"], None, stride=stride, padding=1)
    y = F.relu(_bn(y, sd, prefix + ".bn1"))
    if taps is not None:
        taps[prefix + ".conv1"] = y
    y = F.conv2d(y, sd[prefix + ".conv2.weight"], None, stride=1, padding=1)
    y = _bn(y, sd, prefix + ".bn2")
    if (prefix + ".downsample.0.weight") in sd:
        identity = F.conv2d(x, sd[prefix + ".downsample.0.weight"], None, stride=stride, padding=0)
        identity = _bn(identity, sd, prefix + ".downsample.1")
        if taps is not None:
            taps[prefix + ".downsample"] = identity
    y = F.relu(y + identity)
    if taps is not None:
        taps[prefix] = y
    return y


@torch.no_grad()
def resnet18_forward(
    x: torch.Tensor, sd: Dict[str, torch.Tensor], taps: Optional[dict] = None
) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """x: float32[B,3,224,224] NCHW, ImageNet-normalised.

    Returns (features float32[B,512], logits float32[B,C] or None when the
    state dict has no fc).  ``taps`` (optional dict) receives every
    intermediate activation, NCHW fp32, for layer-wise parity tests.
    """
    sd = canonical_state_dict(sd)
    x = x.to(torch.float32)
    y = F.conv2d(x, sd["conv1.weight"], None, stride=2, padding=3)
    y = F.relu(_bn(y, sd, "bn1"))
    if taps is not None:
        taps["stem"] = y
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    if taps is not None:
        taps["maxpool"] = y
    for name, _, stride in STAGES:
        y = basic_block(y, sd, name + ".0", stride, taps)
        y = basic_block(y, sd, name + ".1", 1, taps)
    feats = torch.flatten(F.adaptive_avg_pool2d(y, 1), 1)
    logits = None
    if "fc.weight" in sd:
        logits = F.linear(feats, sd["fc.weight"], sd["fc.bias"])
    return feats, logits


def predict_labels(logits: torch.Tensor) -> torch.Tensor:
    """src/main.py:510 / :521 / :1010 -- ``outputs.argmax(dim=1)`` (int64)."""
    return logits.argmax(dim=1)


def simclr_projector(feats: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """src/models/simclr.py:20-24 -- Linear(512,512) -> ReLU -> Linear(512,out_dim)."""
    sd = canonical_state_dict(sd)
    h = F.relu(F.linear(feats, sd["projector.0.weight"], sd["projector.0.bias"]))
    return F.linear(h, sd["projector.2.weight"], sd["projector.2.bias"])


# ----------------------------------------------------------------------------
# Known answers of the published architecture (the only pin available).
# ----------------------------------------------------------------------------
N_CONV_BN_PARAMS = 11_176_512
N_FC_PARAMS_2CLASS = 1_026


def expected_param_shapes(num_classes: Optional[int] = 2) -> List[Tuple[str, Tuple[int, ...]]]:
    shapes: List[Tuple[str, Tuple[int, ...]]] = [("conv1.weight", (64, 3, 7, 7))]
    shapes += [("bn1." + s, (64,)) for s in ("weight", "bias", "running_mean", "running_var")]
    cin = 64
    for name, cout, stride in STAGES:
        for blk in (0, 1):
            p = f"{name}.{blk}"
            bc_in = cin if blk == 0 else cout
            shapes.append((p + ".conv1.weight", (cout, bc_in, 3, 3)))
            shapes += [(p + ".bn1." + s, (cout,)) for s in ("weight", "bias", "running_mean", "running_var")]
            shapes.append((p + ".conv2.weight", (cout, cout, 3, 3)))
            shapes += [(p + ".bn2." + s, (cout,)) for s in ("weight", "bias", "running_mean", "running_var")]
            if blk == 0 and (stride != 1 or cin != cout):
                shapes.append((p + ".downsample.0.weight", (cout, cin, 1, 1)))
                shapes += [
                    (p + ".downsample.1." + s, (cout,)) for s in ("weight", "bias", "running_mean", "running_var")
                ]
        cin = cout
    if num_classes:
        shapes.append(("fc.weight", (num_classes, 512)))
        shapes.append(("fc.bias", (num_classes,)))
    return shapes


class ModuleResNet18(torch.nn.Module):
    """Independent nn.Module restatement (same published graph) used only to
    cross-check the functional form above and its key naming."""

    class _Block(torch.nn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            nn = torch.nn
            self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(cout)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(cout)
            self.downsample = None
            if stride != 1 or cin != cout:
                self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            y = self.relu(self.bn1(self.conv1(x)))
            y = self.bn2(self.conv2(y))
            return self.relu(y + idt)

    def __init__(self, num_classes: Optional[int] = 2):
        super().__init__()
        nn = torch.nn
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for name, cout, stride in STAGES:
            setattr(self, name, nn.Sequential(self._Block(cin, cout, stride), self._Block(cout, cout, 1)))
            cin = cout
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, num_classes) if num_classes else nn.Identity()

    def forward(self, x):
        y = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        y = self.layer4(self.layer3(self.layer2(self.layer1(y))))
        return self.fc(torch.flatten(self.avgpool(y), 1))
