"""Host-side mirror of src/models/simclr.py (SimCLRModel, nt_xent_loss, get_simclr_transform,
pretrain_simclr).

``SimCLRModel`` keeps the reference's parameter tree and key layout (``encoder.*``,
``projector.{0,2}.*``); in eval mode its encoder runs on the HIP inference path.  The TRAINING step
is native too: ``pretrain_simclr`` drives ``train_native.NativeSimCLRTrainer`` (train-mode encoder
forward / backward, projector, NT-Xent, Adam: csrc/train.hip, csrc/ntxent.hip) and uses this module
only for the initial weights and the checkpoint format.  ``nt_xent_loss`` is the loss exactly as the
reference defines it, with an optional ``gather`` hook so a process-per-GPU run sees the GLOBAL batch
of negatives as nn.DataParallel does in the reference (src/models/simclr.py:88-95); the
module's own ``forward`` in train mode runs the torch graph on CPU tensors only (the graph the training oracle is
pinned on) and raises on a ROCm tensor with a pointer to the native step (resnet._HipBacked._train_guard).
"""
from __future__ import annotations

import os
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
            self._train_guard(x)
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
    kernel; default for float32 ROCm tensors with D <= 4096), "torch" (the reference's own
    formula on any device)."""
    if gather is not None:
        z_i, z_j = gather(z_i), gather(z_j)
    if backend is None:
        backend = "hip" if (z_i.is_cuda and z_i.dtype == torch.float32 and z_i.shape[1] <= 4096) else "torch"
    if backend == "hip":
        return _NTXentHip.apply(torch.cat([z_i, z_j], dim=0), float(temperature))
    n = z_i.size(0)
    z = F.normalize(torch.cat([z_i, z_j], dim=0), dim=1)
    sim = torch.matmul(z, z.T) / temperature
    sim = sim.masked_fill(torch.eye(2 * n, dtype=torch.bool, device=z.device), float("-inf"))
    positives = torch.cat([torch.diag(sim, n), torch.diag(sim, -n)]).unsqueeze(1)
    denominator = torch.logsumexp(sim, dim=1, keepdim=True)
    return (-positives + denominator).mean()


def get_simclr_transform():
    """src/models/simclr.py:57-66 (host-side PIL pipeline, see transforms.py)."""
    from .transforms import simclr_transform

    return simclr_transform()


def pretrain_simclr(patch_dir: str, epochs: int = 200, batch_size: int = 512, lr: float = 1e-3, device: str = "cuda",
                    num_workers: int = 8, out_dir: str = ".", max_steps: Optional[int] = None, verbose: bool = True,
                    final_path: Optional[str] = None, precision: str = "fp32", device_aug: bool = False, pool=None):
    """src/models/simclr.py:68-124 with the step on the native kernels: PatchDataset(transform=None) ->
    SimCLRDataset(two augmented views) -> DataLoader(batch_size, shuffle) -> per batch
    ``z_i = model(x_i); z_j = model(x_j); loss = nt_xent_loss(z_i, z_j); backward; Adam(lr).step()``.
    Same bookkeeping: best-loss checkpoint ``simclr_encoder_best.pth``, early-stop check every 20 epochs
    (patience 20), ``simclr_encoder_epoch{N}.pth`` every 50 epochs, final ``simclr_encoder.pth`` -- all
    SimCLRModel state_dicts (``encoder.*``, ``projector.*``).  ``max_steps`` (additive) bounds the run for tests;
    ``final_path`` (additive) names the final checkpoint instead of ``out_dir/simclr_encoder.pth``.
    Under ``main.py --world_size N`` (one process per GPU, the reference's nn.DataParallel at :77-78) every rank takes its
    share of each global batch of ``batch_size`` view pairs, batch-norm statistics stay per replica, NT-Xent sees the
    all-gathered projections and the gradients are all-reduced; rank 0 writes the checkpoints.
    ``device_aug`` (additive): the decoded patches stay in HBM and the two views of every step are made by
    ``hipac_augment_views`` (augment.py: Pillow's arithmetic, host-drawn parameters) instead of DataLoader workers -- the
    host pipeline delivers ~1 k view pairs/s, the native step consumes 3 k (fp32) to 14 k (fp16).
    Returns (SimCLRModel with the trained weights, list of per-epoch mean losses)."""
    from torch.utils.data import DataLoader

    from .dist import RankBatchSampler, rank_world
    from .patch_dataset import PatchDataset
    from .simclr_dataset import SimCLRDataset
    from .train_native import NativeSimCLRTrainer

    rank, world = rank_world()
    if pool is not None:  # (additive) an ``augment.DevicePatchPool`` made elsewhere, e.g. straight from the slides: no PNG tree
        from .augment import DeviceSimCLRLoader

        loader = DeviceSimCLRLoader(pool, batch_size, shuffle=True, seed=0, rank=rank, world=world)
        base = ds = None
    else:
        base = PatchDataset(patch_dir, transform=None)
        ds = SimCLRDataset(base, transform=get_simclr_transform())
    if pool is not None:
        pass
    elif device_aug:
        from .augment import DevicePatchPool, DeviceSimCLRLoader

        loader = DeviceSimCLRLoader(DevicePatchPool.from_patch_dataset(base, device=device, workers=max(1, num_workers)), batch_size,
                                    shuffle=True, seed=0, rank=rank, world=world)
    elif world > 1:
        loader = DataLoader(ds, batch_sampler=RankBatchSampler(len(ds), batch_size, rank, world, True, 0), num_workers=num_workers)
    else:
        loader = DataLoader(ds, batch_size=batch_size, shuffle=True, num_workers=num_workers)
    dev = torch.device(device)
    model = SimCLRModel()
    trainer = NativeSimCLRTrainer(model.state_dict(), device=dev, lr=lr, precision=precision)  # fp32 = the reference's loop (:85-96)
    trainer.sync_from_rank0()
    verbose = verbose and rank == 0

    def save(name):
        if rank == 0:
            torch.save(trainer.state_dict(), name if os.path.isabs(name) or os.path.dirname(name) else os.path.join(out_dir, name))

    best_loss, no_improve, best_epoch, history, steps = float("inf"), 0, -1, [], 0
    for epoch in range(epochs):
        total, n_batches = 0.0, 0
        for x_i, x_j in loader:
            loss = trainer.step(x_i.to(dev, torch.float32).contiguous(), x_j.to(dev, torch.float32).contiguous())
            total += float(loss)
            n_batches += 1
            steps += 1
            if max_steps is not None and steps >= max_steps:
                break
        avg = total / max(1, n_batches)
        history.append(avg)
        if verbose:
            print(f"Epoch {epoch+1}, Loss: {avg:.4f}")
        if avg < best_loss:
            best_loss, no_improve, best_epoch = avg, 0, epoch + 1
            save("simclr_encoder_best.pth")
        else:
            no_improve += 1
        if (epoch + 1) % 20 == 0 and no_improve >= 20:
            if verbose:
                print(f"[INFO] Early stopping triggered at epoch {epoch+1}. Best epoch: {best_epoch} with loss {best_loss:.4f}")
            break
        if (epoch + 1) % 50 == 0:
            save(f"simclr_encoder_epoch{epoch+1}.pth")
        if max_steps is not None and steps >= max_steps:
            break
    save(final_path if final_path else "simclr_encoder.pth")
    model.load_state_dict(trainer.state_dict())
    if verbose:
        print("[INFO] SimCLR pretraining complete.")
    return model, history
