"""Host-side mirror of the reference's model classes (src/models/resnet.py).

Same class names, constructor arguments, ``forward`` shapes and ``state_dict`` key
layouts as the reference, so a script written against ``src/models/resnet.py`` keeps
working.  What changes is what runs: in ``eval()`` mode on a ROCm device ``forward``
hands the batch to the hand-written HIP path (libhipac_hip.so) -- BN folded, MFMA
implicit-GEMM convolutions -- and there is NO CPU fallback: a CPU tensor in eval mode
raises.  TRAINING is native as well, but not through ``forward``: the step (train-mode forward,
backward, loss, Adam) is one chain of launches behind ``train_native.NativeClassifierTrainer`` /
``NativeSimCLRTrainer``, driven by ``train.train_resnet_classifier`` and ``simclr.pretrain_simclr``.
A ``train()``-mode ``forward`` on a ROCm tensor therefore RAISES with that pointer instead of silently
running a torch-autograd graph on the GPU; on a CPU tensor it runs the module's own parameter tree
under torch autograd -- the graph tests/test_oracle_train.py pins the training oracle on.

torchvision is not a dependency: ``ResNet18Graph`` declares the same parameter tree
(and therefore the same key names) as ``torchvision.models.resnet18``.  Nothing is
ever downloaded: ``pretrained=True`` in the reference (resnet.py:63-65) becomes
"weights come from a local file or from seeded init".
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import capi
from .weights import canonical_state_dict, load_into


class BasicBlock(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
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


class ResNet18Graph(nn.Module):
    """Parameter tree with torchvision.models.resnet18's attribute names."""

    def __init__(self, num_classes: Optional[int] = 1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (cout, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), start=1):
            setattr(self, f"layer{i}", nn.Sequential(BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1)))
            cin = cout
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes) if num_classes else nn.Identity()
        for m in self.modules():  # torchvision's init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def features(self, x):
        y = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        y = self.layer4(self.layer3(self.layer2(self.layer1(y))))
        return torch.flatten(self.avgpool(y), 1)

    def forward(self, x):
        return self.fc(self.features(x))


class _HipBacked(nn.Module):
    """Shared machinery: pack-on-demand and dispatch of eval-mode forwards to HIP."""

    precision: str = "bf16"

    def _bare_state(self) -> Dict[str, torch.Tensor]:
        return canonical_state_dict(self.state_dict())

    def _state_version(self) -> int:
        return sum(int(p._version) for p in list(self.parameters()) + list(self.buffers()))

    def packed(self) -> capi.PackedResNet18:
        key = (self._state_version(), self.precision)
        cached = self.__dict__.get("_packed")
        if cached is None or cached[0] != key:
            net = capi.PackedResNet18(self._bare_state(), precision=self.precision)
            self.__dict__["_packed"] = (key, net)
        return self.__dict__["_packed"][1]

    def set_precision(self, precision: str):
        if precision not in capi.PRECISIONS:
            raise capi.HipacError(f"precision must be one of {sorted(capi.PRECISIONS)}")
        self.precision = precision
        return self

    def _train_guard(self, x: torch.Tensor) -> None:
        """train()-mode forward: CPU tensors run the torch graph (the oracle's pin); a ROCm tensor would be a second,
        non-native GPU path a caller could land on without noticing -- refuse it and name the native one."""
        if x.is_cuda and os.environ.get("HIPAC_ALLOW_TORCH_AUTOGRAD", "0") != "1":
            raise capi.HipacError(
                f"{type(self).__name__}.forward in train() mode on a ROCm tensor: training runs on the native step, not "
                "through forward -- use train.train_resnet_classifier / simclr.pretrain_simclr (or "
                "train_native.NativeClassifierTrainer / NativeSimCLRTrainer with this module's state_dict()); call .eval() "
                "for inference.  (HIPAC_ALLOW_TORCH_AUTOGRAD=1 runs the torch-autograd graph anyway.)")

    def _hip_forward(self, x: torch.Tensor, want: str) -> torch.Tensor:
        if not x.is_cuda:
            raise capi.HipacError(
                "eval-mode forward got a CPU tensor: the HIP path has no CPU fallback "
                "(move the batch to a ROCm device)"
            )
        native = x.dim() == 4 and tuple(x.shape[1:]) == (capi.PAD_H, capi.PAD_W, 4)
        if not native:
            x = x.contiguous().to(torch.float32)
        f, l, _ = self.packed().forward(x, want_feats=want == "feats", want_logits=want == "logits",
                                        native_layout=native)
        return f if want == "feats" else l


class ResNet18FeatureExtractor(_HipBacked):
    """src/models/resnet.py:22-40.  ``features`` = children()[:-1] of resnet18 =>
    keys ``features.{0,1,4,5,6,7}.*``; forward -> [B,512]."""

    def __init__(self, weight_path: str = "resnet18_patch_classifier.pth", reference_quirks: bool = False):
        super().__init__()
        resnet = ResNet18Graph(num_classes=1000)
        self.features = nn.Sequential(*list(resnet.children())[:-1])
        path = os.path.join(os.getcwd(), "src", "models", weight_path) if weight_path else None
        if path and os.path.exists(path):
            sd = torch.load(path, map_location="cpu", weights_only=True)
            load_into(self, sd, drop_fc=True, reference_quirks=reference_quirks)

    def forward(self, x):
        if self.training:
            self._train_guard(x)
            y = self.features(x)
            return y.view(y.size(0), -1)
        return self._hip_forward(x, "feats")


class UnifiedResNet(_HipBacked):
    """src/models/resnet.py:42-55.  keys ``model.*``; fc = Identity -> [B,512], or
    Linear(512,2) when ``classifier=True`` -> [B,2]."""

    def __init__(self, pretrained_weights_path: Optional[str] = None, classifier: bool = False,
                 reference_quirks: bool = False):
        super().__init__()
        self.model = ResNet18Graph(num_classes=None)
        if pretrained_weights_path and os.path.exists(pretrained_weights_path):
            sd = torch.load(pretrained_weights_path, map_location="cpu", weights_only=True)
            load_into(self, sd, drop_fc=True, reference_quirks=reference_quirks)
        self.classifier = classifier
        if classifier:
            self.model.fc = nn.Linear(512, 2)

    def forward(self, x):
        if self.training:
            self._train_guard(x)
            return self.model(x)
        return self._hip_forward(x, "logits" if self.classifier else "feats")


class ResNet18Classifier(_HipBacked):
    """src/models/resnet.py:57-77.  keys ``model.*`` with ``model.fc`` = Linear(512,2);
    forward -> [B,2].  The reference starts from downloaded ImageNet weights
    (``pretrained=True``); here the start is seeded init unless ``weights_path`` names a
    local file -- nothing is fetched."""

    def __init__(self, weights_path: Optional[str] = None):
        super().__init__()
        self.model = ResNet18Graph(num_classes=1000)
        num_ftrs = self.model.fc.in_features
        self.model.fc = nn.Linear(num_ftrs, 2)
        if weights_path and os.path.exists(weights_path):
            load_into(self, torch.load(weights_path, map_location="cpu", weights_only=True))

    def forward(self, x):
        if self.training:
            self._train_guard(x)
            return self.model(x)
        return self._hip_forward(x, "logits")

    @torch.no_grad()
    def predict(self, x):
        """argmax labels as the reference takes them (src/main.py:510, :521, :1010)."""
        if not x.is_cuda:
            raise capi.HipacError("predict needs a ROCm tensor (no CPU fallback)")
        native = tuple(x.shape[1:]) == (capi.PAD_H, capi.PAD_W, 4)
        _, l, lab = self.packed().forward(x if native else x.contiguous().float(), want_feats=False,
                                          want_logits=True, want_labels=True, native_layout=native)
        return lab, l


class ResNet18ClassifierSIMCLR(_HipBacked):
    """src/models/resnet.py:80-91.  keys ``encoder.*`` (fc = Linear(512,num_classes))."""

    def __init__(self, pretrained_weights_path: Optional[str] = None, num_classes: int = 2):
        super().__init__()
        self.encoder = ResNet18Graph(num_classes=1000)
        if pretrained_weights_path:
            sd = torch.load(pretrained_weights_path, map_location="cpu", weights_only=True)
            load_into(self, sd, drop_fc=True)
        self.encoder.fc = nn.Linear(512, num_classes)

    def forward(self, x):
        if self.training:
            self._train_guard(x)
            return self.encoder(x)
        return self._hip_forward(x, "logits")
