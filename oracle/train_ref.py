"""Oracle: the training steps of the reference, as torch autograd on the CPU computes them.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED by reference fixtures (the reference holds
none for training, and its model files need torchvision); pinned on torch itself: the functional graph below
is checked against an nn.Module graph with the same parameter names under autograd
(tests/test_oracle_train.py).

Follows
  * SimCLRModel.forward            src/models/simclr.py:26-29   encoder (fc = Identity, TRAIN-mode BN) -> projector
  * the step of pretrain_simclr    src/models/simclr.py:88-94   z_i = model(x_i); z_j = model(x_j); nt_xent; backward
  * the classifier step            src/main.py:499-506          CrossEntropyLoss(weight=class_weights); backward
in fp32 (pretrain_simclr runs without autocast; the classifier loops' fp16 autocast is not imitated).
Batch-norm: training statistics of the call's own batch, running statistics updated with momentum 0.1
and the unbiased variance (torch.nn.BatchNorm2d defaults).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .ntxent_ref import nt_xent_loss_ref
from .resnet18_ref import BN_EPS, STAGES, canonical_state_dict

BN_MOMENTUM = 0.1


class _ReluGiven(torch.autograd.Function):
    """relu(x) whose backward uses a GIVEN activation pattern instead of (x > 0).  A ReLU's derivative is
    discontinuous at 0: two fp32 implementations whose pre-activations differ by 1e-7 disagree on a handful of
    the ~10 M units of a step, and each disagreement moves the (heavily cancelling) gradient sums by percents.
    Tests therefore hand the oracle the pattern of the implementation under test: the values stay the oracle's
    own, and with equal patterns the gradients must agree to rounding."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x.clamp_min(0)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask.to(g.dtype), None


class _MaxPoolGiven(torch.autograd.Function):
    """max_pool2d(3, 2, 1) whose backward routes the gradient to a GIVEN winner per window (uint8 dy*3+dx),
    for the same reason as _ReluGiven (ties and near-ties inside a window)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, C_, H, W = ctx.shape
        HO, WO = g.shape[2], g.shape[3]
        oy = torch.arange(HO).view(1, 1, HO, 1)
        ox = torch.arange(WO).view(1, 1, 1, WO)
        iy = 2 * oy - 1 + (idx.long() // 3)
        ix = 2 * ox - 1 + (idx.long() % 3)
        flat = (iy * W + ix).view(B, C_, -1)
        out = torch.zeros((B, C_, H * W), dtype=g.dtype)
        out.scatter_add_(2, flat, g.reshape(B, C_, -1))
        return out.view(B, C_, H, W), None


def _relu(x, masks, name):
    if masks is not None and name in masks:
        return _ReluGiven.apply(x, masks[name])
    return F.relu(x)


def _bn_train(x, p, stats, name):
    return F.batch_norm(x, stats[name + ".running_mean"], stats[name + ".running_var"], p[name + ".weight"], p[name + ".bias"],
                        training=True, momentum=BN_MOMENTUM, eps=BN_EPS)


def encoder_train_forward(x: torch.Tensor, p: Dict[str, torch.Tensor], stats: Dict[str, torch.Tensor],
                          taps: Optional[dict] = None, masks: Optional[dict] = None) -> torch.Tensor:
    """x [B,3,224,224] -> features [B,512]; ``p`` bare torchvision-named parameters, ``stats`` running statistics
    (updated in place, as nn.BatchNorm2d does in train mode).  ``taps`` (optional dict) receives, per conv name,
    the conv output before BN ("<conv>.pre") and the map after BN / residual / ReLU ("<conv>.post").
    ``masks`` (optional): activation patterns for the backward -- "<conv>.post" -> bool tensor for every ReLU,
    "pool_idx" -> uint8 winners of the max-pool (see _ReluGiven)."""
    def tap(k, v):
        if taps is not None:
            taps[k] = v.detach()
        return v

    y = tap("conv1.pre", F.conv2d(x, p["conv1.weight"], None, stride=2, padding=3))
    y = tap("conv1.post", _relu(_bn_train(y, p, stats, "bn1"), masks, "conv1.post"))
    if masks is not None and "pool_idx" in masks:
        y = tap("pool", _MaxPoolGiven.apply(y, masks["pool_idx"]))
    else:
        y = tap("pool", F.max_pool2d(y, kernel_size=3, stride=2, padding=1))
    for name, _, stride in STAGES:
        for blk in (0, 1):
            pre = f"{name}.{blk}"
            st = stride if blk == 0 else 1
            idt = y
            t = tap(pre + ".conv1.pre", F.conv2d(y, p[pre + ".conv1.weight"], None, stride=st, padding=1))
            t = tap(pre + ".conv1.post", _relu(_bn_train(t, p, stats, pre + ".bn1"), masks, pre + ".conv1.post"))
            t = tap(pre + ".conv2.pre", F.conv2d(t, p[pre + ".conv2.weight"], None, stride=1, padding=1))
            t = _bn_train(t, p, stats, pre + ".bn2")
            if (pre + ".downsample.0.weight") in p:
                idt = tap(pre + ".downsample.0.pre", F.conv2d(y, p[pre + ".downsample.0.weight"], None, stride=st))
                idt = tap(pre + ".downsample.0.post", _bn_train(idt, p, stats, pre + ".downsample.1"))
            y = tap(pre + ".conv2.post", _relu(t + idt, masks, pre + ".conv2.post"))
    return torch.flatten(F.adaptive_avg_pool2d(y, 1), 1)


def _split(sd: Dict[str, torch.Tensor], dtype=torch.float32):
    """bare state dict -> (leaf parameters requiring grad, running statistics (clones))."""
    p, stats = {}, {}
    for k, v in sd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            stats[k] = v.detach().clone().to(dtype)
        elif k.endswith("num_batches_tracked"):
            continue
        else:
            p[k] = v.detach().clone().to(dtype).requires_grad_(True)
    return p, stats


def simclr_step_ref(sd: Dict[str, torch.Tensor], x_i: torch.Tensor, x_j: torch.Tensor, temperature: float = 0.5,
                    dtype=torch.float32, masks_i: Optional[dict] = None, masks_j: Optional[dict] = None):
    """``sd``: SimCLRModel state_dict (encoder.*, projector.{0,2}.*).  Returns (loss, gradients keyed like ``sd``,
    running statistics after the two forwards keyed like ``sd``).  ``dtype=torch.float64`` gives the reference
    the fp32 results of BOTH sides are measured against (the fp32 autograd result itself carries summation error:
    batch-norm bias gradients are sums with heavy cancellation)."""
    enc = canonical_state_dict({k: v for k, v in sd.items() if not k.startswith("projector.")})
    p, stats = _split(enc, dtype)
    proj = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items() if k.startswith("projector.")}
    x_i, x_j = x_i.to(dtype), x_j.to(dtype)

    def model(x, masks):
        f = encoder_train_forward(x, p, stats, masks=masks)
        h = _relu(F.linear(f, proj["projector.0.weight"], proj["projector.0.bias"]), masks, "projector.hidden")
        return F.linear(h, proj["projector.2.weight"], proj["projector.2.bias"])

    z_i = model(x_i, masks_i)  # two separate passes: each normalises with its own batch statistics (simclr.py:88-91)
    z_j = model(x_j, masks_j)
    loss = nt_xent_loss_ref(z_i, z_j, temperature)
    loss.backward()
    grads = {"encoder." + k: v.grad for k, v in p.items()}
    grads.update({k: v.grad for k, v in proj.items()})
    return loss.detach(), grads, {"encoder." + k: v for k, v in stats.items()}


def classifier_step_ref(sd: Dict[str, torch.Tensor], x: torch.Tensor, labels: torch.Tensor,
                        class_weights: Optional[torch.Tensor] = None, dtype=torch.float32, masks: Optional[dict] = None):
    """``sd``: ResNet18Classifier state_dict in any reference layout.  Returns (loss, logits, gradients (bare names),
    running statistics (bare names))."""
    bare = canonical_state_dict(sd)
    p, stats = _split(bare, dtype)
    x = x.to(dtype)
    class_weights = None if class_weights is None else class_weights.to(dtype)
    f = encoder_train_forward(x, p, stats, masks=masks)
    logits = F.linear(f, p["fc.weight"], p["fc.bias"])
    loss = F.cross_entropy(logits, labels, weight=class_weights)
    loss.backward()
    return loss.detach(), logits.detach(), {k: v.grad for k, v in p.items()}, stats


def adam_ref(param: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
             betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8):
    """One torch.optim.Adam update (no weight decay), returns (param, m, v)."""
    m = betas[0] * m + (1 - betas[0]) * grad
    v = betas[1] * v + (1 - betas[1]) * grad * grad
    bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
    return param - (lr / bc1) * m / (v.sqrt() / (bc2 ** 0.5) + eps), m, v
