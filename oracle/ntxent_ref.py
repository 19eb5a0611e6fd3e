"""Oracle: NT-Xent loss as the reference computes it.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PINNED on the reference itself: src/models/simclr.py cannot be
imported as a module here (its top-level ``import torchvision.models`` fails), but ``nt_xent_loss`` is pure torch;
tests/golden/make_golden_ntxent.py compiles that one function definition from the reference file and stores its
values and autograd gradients on seeded inputs (tests/golden/ntxent_golden.npz); tests/test_oracle_resnet.py holds
this restatement to them (1e-6), tests/test_gpu_ntxent.py the HIP kernel (2e-5).

Follows src/models/simclr.py:31-54 step for step:
  z = cat(z_i, z_j)            [2N, D]
  z = F.normalize(z, dim=1)
  S = z @ z.T / temperature    [2N, 2N]
  S[diag] = -inf
  pos = cat(diag(S, N), diag(S, -N))
  loss = mean(-pos + logsumexp(S, dim=1))
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def nt_xent_loss_ref(z_i: torch.Tensor, z_j: torch.Tensor, temperature: float = 0.5) -> torch.Tensor:
    n = z_i.size(0)
    z = torch.cat([z_i, z_j], dim=0)
    z = F.normalize(z, dim=1)
    sim = torch.matmul(z, z.T) / temperature
    eye = torch.eye(2 * n, dtype=torch.bool, device=z.device)
    sim = sim.masked_fill(eye, float("-inf"))
    positives = torch.cat([torch.diag(sim, n), torch.diag(sim, -n)]).unsqueeze(1)
    denominator = torch.logsumexp(sim, dim=1, keepdim=True)
    return (-positives + denominator).mean()
