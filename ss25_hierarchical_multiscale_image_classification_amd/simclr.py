"""Host-side mirror of src/models/simclr.py (SimCLRModel, nt_xent_loss).

Scope note (SURVEY.md section 8, rows a-12/a-13 are "next"): the SimCLR training step
is not yet a native HIP path.  ``SimCLRModel`` keeps the reference's parameter tree and
key layout (``encoder.*``, ``projector.{0,2}.*``) and, in eval mode, runs the encoder on
the HIP path; in train mode it is an ordinary autograd graph.  ``nt_xent_loss`` is the
loss exactly as the reference defines it, written with torch ops, with an optional
``gather`` hook so a process-per-GPU run sees the GLOBAL batch of negatives as
nn.DataParallel does in the reference (loss computed on the gathered outputs,
src/models/simclr.py:88-95).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .resnet import ResNet18Graph, _HipBacked


class SimCLRModel(_HipBacked):
    """src/models/simclr.py:14-29."""

    def __init__(self, base_model: str = "resnet18", out_dim: int = 128):
        super().__init__()
        if base_model != "resnet18":
            raise ValueError("only resnet18 is on the hot path (src/models/simclr.py:15 default)")
        self.encoder = ResNet18Graph(num_classes=1000)
        dim_mlp = self.encoder.fc.in_features
        self.encoder.fc = nn.Identity()
        self.projector = nn.Sequential(nn.Linear(dim_mlp, dim_mlp), nn.ReLU(), nn.Linear(dim_mlp, out_dim))

    def forward(self, x):
        if self.training:
            return self.projector(self.encoder(x))
        feats = self._hip_forward(x, "feats")
        return self.projector(feats)


class _NTXentHip(torch.autograd.Function):
    """Value and gradient from one native call (hipac_ntxent_fwd_bwd); the gradient is kept for backward."""

    @staticmethod
    def forward(ctx, z, temperature):
        from . import capi

        loss, dz = capi.ntxent_fwd_bwd(z.contiguous(), temperature, want_grad=z.requires_grad)
        ctx.save_for_backward(dz) if dz is not None else None
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        (dz,) = ctx.saved_tensors
        return dz * grad_out, None


def nt_xent_loss(z_i: torch.Tensor, z_j: torch.Tensor, temperature: float = 0.5,
                 gather: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                 backend: Optional[str] = None) -> torch.Tensor:
    """src/models/simclr.py:31-54.  ``gather`` (optional) maps the local [n,D] block to
    the global [N,D] one, differentiably, before the loss.  ``backend``: "hip" (the native
    kernel; default for float32 ROCm tensors with D <= 256), "torch" (the reference's own
    formula on any device)."""
    if gather is not None:
        z_i, z_j = gather(z_i), gather(z_j)
    if backend is None:
        backend = "hip" if (z_i.is_cuda and z_i.dtype == torch.float32 and z_i.shape[1] <= 256) else "torch"
    if backend == "hip":
        return _NTXentHip.apply(torch.cat([z_i, z_j], dim=0), float(temperature))
    n = z_i.size(0)
    z = F.normalize(torch.cat([z_i, z_j], dim=0), dim=1)
    sim = torch.matmul(z, z.T) / temperature
    sim = sim.masked_fill(torch.eye(2 * n, dtype=torch.bool, device=z.device), float("-inf"))
    positives = torch.cat([torch.diag(sim, n), torch.diag(sim, -n)]).unsqueeze(1)
    denominator = torch.logsumexp(sim, dim=1, keepdim=True)
    return (-positives + denominator).mean()
