"""state_dict key layouts of the reference, and conversion between them.

The reference writes / expects four layouts:

  classifier : ``model.<tv>``      ResNet18Classifier   src/models/resnet.py:57-77
               (``module.model.<tv>`` when saved from nn.DataParallel, src/main.py:481-482, :530)
  unified    : ``model.<tv>``      UnifiedResNet        src/models/resnet.py:42-55  (no fc keys when Identity)
  extractor  : ``features.N.<..>`` ResNet18FeatureExtractor src/models/resnet.py:36
               (children()[:-1] of resnet18: 0=conv1 1=bn1 4..7=layer1..4)
  simclr     : ``encoder.<tv>`` + ``projector.{0,2}.{weight,bias}``  src/models/simclr.py:17-24

where ``<tv>`` are torchvision's resnet18 names (conv1.weight, bn1.*, layerK.B.convJ.weight,
layerK.B.bnJ.*, layerK.0.downsample.{0,1}.*, fc.*).

The reference's own loaders use ``strict=False`` with mismatched prefixes and therefore
silently load nothing (resnet.py:28-31, :48-50; main.py:852-859).  ``load_into`` below
maps prefixes correctly; ``reference_quirks=True`` reproduces the silent no-op instead.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch

FEATURES_INDEX = {"conv1": "0", "bn1": "1", "layer1": "4", "layer2": "5", "layer3": "6", "layer4": "7"}
_FEATURES_INV = {v: k for k, v in FEATURES_INDEX.items()}
LAYOUTS = ("bare", "classifier", "unified", "extractor", "simclr")


def detect_layout(keys: Iterable[str]) -> str:
    ks = [k[7:] if k.startswith("module.") else k for k in keys]
    if any(k.startswith("features.") for k in ks):
        return "extractor"
    if any(k.startswith("encoder.") for k in ks):
        return "simclr"
    if any(k.startswith("model.") for k in ks):
        return "classifier"
    return "bare"


def canonical_state_dict(sd: Dict[str, torch.Tensor], keep_projector: bool = False) -> Dict[str, torch.Tensor]:
    """Any reference layout -> bare torchvision names."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[7:]
        if k.startswith("projector."):
            if keep_projector:
                out[k] = v
            continue
        if k.startswith("model."):
            k = k[6:]
        elif k.startswith("encoder."):
            k = k[8:]
        elif k.startswith("features."):
            idx, _, tail = k[9:].partition(".")
            if idx not in _FEATURES_INV:
                continue
            k = _FEATURES_INV[idx] + "." + tail
        out[k] = v
    return out


def to_layout(bare: Dict[str, torch.Tensor], layout: str, data_parallel: bool = False) -> Dict[str, torch.Tensor]:
    """Bare torchvision names -> one of the reference layouts (for saving)."""
    if layout not in LAYOUTS:
        raise ValueError(f"layout must be one of {LAYOUTS}")
    out: Dict[str, torch.Tensor] = {}
    for k, v in bare.items():
        if k.startswith("projector."):
            if layout == "simclr":
                out[k] = v
            continue
        if layout == "bare":
            nk = k
        elif layout in ("classifier", "unified"):
            nk = "model." + k
        elif layout == "simclr":
            nk = "encoder." + k
        else:  # extractor: children()[:-1] has no fc
            head, _, tail = k.partition(".")
            if head not in FEATURES_INDEX:
                continue
            nk = f"features.{FEATURES_INDEX[head]}.{tail}"
        out[("module." + nk) if data_parallel else nk] = v
    return out


def load_into(module: torch.nn.Module, sd: Dict[str, torch.Tensor], drop_fc: bool = False,
              reference_quirks: bool = False) -> Dict[str, list]:
    """Load a checkpoint written in ANY reference layout into ``module`` (whose own
    keys may be in any layout too).  Returns {"loaded": [...], "skipped": [...]}.

    ``reference_quirks=True`` mimics the reference: keys are matched verbatim with
    ``strict=False`` so a prefix mismatch loads nothing (and says nothing)."""
    own = module.state_dict()
    if reference_quirks:
        hit = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape and not (drop_fc and "fc" in k)}
        module.load_state_dict(hit, strict=False)
        return {"loaded": sorted(hit), "skipped": sorted(set(sd) - set(hit))}
    src = canonical_state_dict(sd, keep_projector=True)
    own_layout = detect_layout(own.keys())
    mapped = to_layout(src, own_layout)
    hit = {}
    for k, v in mapped.items():
        if drop_fc and (k.endswith("fc.weight") or k.endswith("fc.bias")):
            continue
        if k in own and own[k].shape == v.shape:
            hit[k] = v
    module.load_state_dict(hit, strict=False)
    return {"loaded": sorted(hit), "skipped": sorted(set(mapped) - set(hit))}
