"""Classifier fine-tune loops (mirror of src/main.py:412-606) behind the ``--train`` /
``--train_strategy`` flags.  The training step (train-mode ResNet18 forward + backward, weighted
cross-entropy, Adam) runs on the native kernels (train_native.py, csrc/train.hip); the per-epoch
validation scoring uses the HIP inference path.  Data loading and augmentation stay on the host.
"""
from __future__ import annotations

import os
import random
from typing import Optional

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset

from .patch_dataset import PatchDataset
from .resnet import ResNet18Classifier
from .transforms import eval_transform, train_transform

SAMPLES_PER_CLASS = 7480  # src/main.py:50


def get_dataloaders(patch_dir: str, test_ratio: float = 0.2, batch_size: int = 512, balanced: bool = False,
                    rank: int = 0, world: int = 1, seed: int = 0):
    """src/main.py:412-470: slide-level split (random_state=42), tumour patches
    augmented, normal patches not, validation set balanced with default_rng(42).
    ``world > 1`` (one process per GPU): every rank builds the same datasets (the caller seeds ``random`` identically on
    all ranks) and draws its share of every global batch (``dist.RankBatchSampler``: the chunk DataParallel's scatter
    gives replica ``rank``); the validation batches are shared out the same way, unshuffled."""
    from sklearn.model_selection import train_test_split

    from .dist import RankBatchSampler

    slide_dirs = sorted(d for d in os.listdir(patch_dir) if os.path.isdir(os.path.join(patch_dir, d)))
    if len(slide_dirs) > 1:
        train_slides, val_slides = train_test_split(slide_dirs, test_size=test_ratio, random_state=42)
    else:
        train_slides = val_slides = slide_dirs
    train_ds = PatchDataset(patch_dir, slide_names=train_slides, tumor_transform=train_transform(),
                            normal_transform=eval_transform(), balanced=balanced,
                            max_samples=SAMPLES_PER_CLASS if balanced else None)
    val_ds = PatchDataset(patch_dir, slide_names=val_slides, tumor_transform=eval_transform(),
                          normal_transform=eval_transform())
    labels = np.array(val_ds.labels)
    tum, nor = np.where(labels == 1)[0], np.where(labels == 0)[0]
    if len(tum) and len(nor):
        n_min = min(len(tum), len(nor))
        rng = np.random.default_rng(42)
        sel = np.concatenate([rng.choice(tum, n_min, replace=False), rng.choice(nor, n_min, replace=False)])
        val_ds = Subset(val_ds, sel)
    if world > 1:
        return (DataLoader(train_ds, batch_sampler=RankBatchSampler(len(train_ds), batch_size, rank, world, True, seed)),
                DataLoader(val_ds, batch_sampler=_val_share(len(val_ds), batch_size, rank, world)), train_ds, val_ds)
    return (DataLoader(train_ds, batch_size=batch_size, shuffle=True),
            DataLoader(val_ds, batch_size=batch_size, shuffle=False), train_ds, val_ds)


class _PoolView:
    """The slice of a DevicePatchPool a loop trains / validates on, with the two things the loops ask a dataset."""

    def __init__(self, pool, indices):
        self.pool, self.indices = pool, list(indices)

    def __len__(self):
        return len(self.indices)

    def get_class_counts(self):
        from collections import Counter

        return dict(Counter(self.pool.labels[i] for i in self.indices))


def get_device_loaders(slides, level: int = 3, test_ratio: float = 0.2, batch_size: int = 512, balanced: bool = False,
                       rank: int = 0, world: int = 1, stride: Optional[int] = None):
    """``get_dataloaders`` without the PNG tree: the kept windows of ``level`` go from the slides in HBM straight into one
    ``DevicePatchPool`` (``--patch``'s PNGs are lossless: the same pixels and labels), the split is the same slide-level
    split (sorted slide names, random_state=42), ``balanced`` draws the same number of patches per class
    (src/main.py:432-438), the validation set is balanced with default_rng(42) (:446-450)."""
    from sklearn.model_selection import train_test_split

    from .augment import DeviceClassifierLoader, DevicePatchPool

    pool = DevicePatchPool.from_slides(slides, level=level, stride=stride)
    names = sorted(set(pool.slide_names))
    if len(names) > 1:
        train_slides, val_slides = train_test_split(names, test_size=test_ratio, random_state=42)
    else:
        train_slides = val_slides = names
    tr_set, va_set = set(train_slides), set(val_slides)
    by_class = {0: [], 1: []}
    for i, (n, lab) in enumerate(zip(pool.slide_names, pool.labels)):
        if n in tr_set:
            by_class[lab].append(i)
    train_idx = []
    if balanced:
        floor = min(len(v) for v in by_class.values())
        for lab, idx in by_class.items():
            train_idx += random.sample(idx, min(floor, SAMPLES_PER_CLASS, len(idx)))
    else:
        train_idx = by_class[0] + by_class[1]
    val_idx = np.array([i for i, n in enumerate(pool.slide_names) if n in va_set], np.int64)
    labels = np.array([pool.labels[i] for i in val_idx])
    tum, nor = np.where(labels == 1)[0], np.where(labels == 0)[0]
    if len(tum) and len(nor):
        n_min = min(len(tum), len(nor))
        rng = np.random.default_rng(42)
        val_idx = val_idx[np.concatenate([rng.choice(tum, n_min, replace=False), rng.choice(nor, n_min, replace=False)])]
    train_ds, val_ds = _PoolView(pool, train_idx), _PoolView(pool, val_idx.tolist())
    return (DeviceClassifierLoader(pool, batch_size, shuffle=True, augment=True, seed=0, rank=rank, world=world, indices=train_idx),
            DeviceClassifierLoader(pool, batch_size, shuffle=False, augment=False, rank=rank, world=world, indices=val_idx.tolist()),
            train_ds, val_ds, pool)


def _val_share(n: int, batch_size: int, rank: int, world: int):
    """Validation indices of this rank (a contiguous range: every sample is scored exactly once), in batches."""
    from .dist import shard_columns

    i0, i1 = shard_columns(n, rank, world)
    return [list(range(b, min(b + batch_size, i1))) for b in range(i0, i1, batch_size)]


def class_weights(train_ds: PatchDataset, strategy: Optional[str]) -> Optional[torch.Tensor]:
    """The loss weights of the two loops: plain ``--train`` uses 1/count normalised by the smaller weight
    (src/main.py:485-490); ``--train_strategy`` computes total/count for every strategy (:549-552) and applies it
    for ``weighted_loss`` and ``self_supervised`` (:558, :572) -- ``balanced`` trains unweighted (:566)."""
    counts = train_ds.get_class_counts()
    c0, c1 = max(counts.get(0, 0), 1), max(counts.get(1, 0), 1)
    if strategy is None:
        w = torch.tensor([1.0 / c0, 1.0 / c1])
        return w / w.min()
    if strategy in ("weighted_loss", "self_supervised"):
        tot = c0 + c1
        return torch.tensor([tot / c0, tot / c1])
    return None


def train_resnet_classifier(patch_dir: str, strategy: Optional[str] = None, epochs: Optional[int] = None,
                            batch_size: int = 512, precision: str = "bf16", lr: float = 1e-4,
                            save_path: Optional[str] = None, device: str = "cuda", simclr_epochs: int = 200,
                            simclr_path: str = "simclr_encoder.pth", max_steps: Optional[int] = None,
                            train_precision: str = "fp16", simclr_precision: str = "fp32", device_aug: bool = False,
                            slides=None, level: int = 3):
    """``train_resnet_classifier`` (strategy None, 30 epochs, src/main.py:472-534) and
    ``train_resnet_classifier_strategic`` (5 epochs, :536-606).  The training step runs on the native kernels
    (``train_native.NativeClassifierTrainer``) in ``train_precision``: "fp16" (default) = the reference's
    ``autocast()`` + ``GradScaler`` arithmetic (:499-508, :578-587: fp16 operands, fp32 accumulation and master weights,
    dynamic loss scale), "fp32" = the exact f32 MFMA; validation scores with the HIP inference path in ``precision``.  ``self_supervised``: pre-trains SimCLR when
    ``simclr_path`` does not exist (:556-557), then starts the classifier from that encoder -- what the reference
    means to do (its constructor call raises a TypeError there).  ``device_aug`` (additive): both sets'
    decoded patches stay in HBM and every batch is transformed on the device (``augment.DeviceClassifierLoader``; SimCLR
    pre-training: ``augment.DeviceSimCLRLoader``) instead of in DataLoader workers.  ``slides`` (additive; a list of
    ``extract.DeviceSlide``): no PNG tree at all -- the kept windows of ``level`` go from the pyramids in HBM into the pool
    (``get_device_loaders``), ``patch_dir`` is not read."""
    from .dist import all_reduce_sum_scalars, rank_world
    from .train_native import NativeClassifierTrainer
    from .weights import canonical_state_dict

    rank, world = rank_world()  # > 1 under ``main.py --world_size N``: one process per GPU, global batch = batch_size
    epochs = epochs if epochs is not None else (30 if strategy is None else 5)  # :494 / :575
    pool_all = None
    if slides is not None:
        train_loader, val_loader, train_ds, val_ds, pool_all = get_device_loaders(slides, level, 0.2, batch_size,
                                                                                   balanced=strategy == "balanced", rank=rank, world=world)
        device_aug = False  # the loaders are device loaders already
    else:
        train_loader, val_loader, train_ds, val_ds = get_dataloaders(patch_dir, 0.2, batch_size,
                                                                     balanced=strategy == "balanced", rank=rank, world=world)
    dev = torch.device(device)
    model = ResNet18Classifier().set_precision(precision)
    if strategy == "self_supervised":
        if not os.path.exists(simclr_path):
            from .simclr import pretrain_simclr

            # the final checkpoint goes to simclr_path itself, whatever its basename
            pretrain_simclr(patch_dir, epochs=simclr_epochs, batch_size=batch_size, device=device,
                            out_dir=os.path.dirname(simclr_path) or ".", max_steps=max_steps, final_path=simclr_path,
                            precision=simclr_precision, device_aug=device_aug, pool=pool_all)
            if world > 1:
                torch.distributed.barrier()  # rank 0 wrote the file
        if not os.path.exists(simclr_path):
            raise FileNotFoundError(f"SimCLR encoder checkpoint {simclr_path} is missing after pre-training")
        enc = canonical_state_dict(torch.load(simclr_path, map_location="cpu", weights_only=True))
        sd = model.state_dict()
        for k, v in enc.items():  # encoder.* -> model.*; the projector is dropped, model.fc keeps its init
            if not k.startswith("projector.") and ("model." + k) in sd:
                sd["model." + k] = v
        model.load_state_dict(sd)
    if device_aug:
        # the decoded patches of both sets stay in HBM; flips / rotation / jitter of the tumour patches and the eval transform
        # of the others are made per batch on the device (augment.py); 224-pixel patches only (the level `--train` reads)
        from .augment import OUT, DeviceClassifierLoader, DevicePatchPool

        tr_pool = DevicePatchPool.from_patch_dataset(train_ds, device=device)
        if tr_pool.P == OUT:
            train_loader = DeviceClassifierLoader(tr_pool, batch_size, shuffle=True, augment=True, seed=0, rank=rank, world=world)
            val_loader = DeviceClassifierLoader(DevicePatchPool.from_patch_dataset(val_ds, device=device), batch_size, shuffle=False,
                                                augment=False, rank=rank, world=world)
        else:  # the rotation / jitter of train_transform act at the source resolution before the resize: host transforms there
            if rank == 0:
                print(f"[INFO] --device_aug: {tr_pool.P}-pixel patches keep the host transforms (the device pipeline of the "
                      f"classifier loops takes 224-pixel patches)")
            del tr_pool
    w = class_weights(train_ds, strategy)
    trainer = NativeClassifierTrainer(model.state_dict(), device=dev, lr=lr, class_weights=w, precision=train_precision)
    trainer.sync_from_rank0()
    history, steps = [], 0
    for epoch in range(epochs):
        total, correct, seen = 0.0, 0, 0
        # a fresh GradScaler per epoch, as both reference loops construct theirs INSIDE the epoch loop (src/main.py:498, :577):
        # scale and growth tracker restart at 65536 / 0, so the first steps of an epoch skip on overflow as the reference's do
        trainer.scaler = type(trainer.scaler)(enabled=trainer.scaler.enabled)
        for imgs, labels, _ in train_loader:
            loss, logits = trainer.step(imgs.to(dev, torch.float32).contiguous(), labels)
            total += float(loss)  # N > 1: already the loss of the global batch
            correct += int((logits.argmax(1).cpu() == labels).sum())
            seen += int(labels.numel())
            steps += 1
            if max_steps is not None and steps >= max_steps:
                break
        trainer.sync_from_rank0()  # N > 1: replica 0's running statistics are the module's (the others' updates are dropped)
        model.load_state_dict(trainer.state_dict())
        model.to(dev).eval()  # validation scoring on the HIP inference path (BN folded from the running statistics)
        v_correct, v_seen = 0, 0
        with torch.no_grad():
            for imgs, labels, _ in val_loader:
                pred, _ = model.predict(imgs.to(dev))
                v_correct += int((pred.cpu() == labels).sum())
                v_seen += int(labels.numel())
        correct, v_correct, v_seen = all_reduce_sum_scalars([correct, v_correct, v_seen], dev)  # counters over all ranks
        # the reference divides the training hits by len(train_dataset) (src/main.py:514); that equals the samples
        # seen unless max_steps cut the epoch short
        n_train = len(train_ds) if (max_steps is None or steps < max_steps) else all_reduce_sum_scalars([seen], dev)[0]
        history.append((total, correct / max(n_train, 1), v_correct / max(v_seen, 1)))
        if rank == 0:
            print(f"Epoch {epoch+1}, Train Loss: {total:.4f}, Train Acc: {history[-1][1]:.4f}, Val Acc: {history[-1][2]:.4f}")
        if rank == 0 and strategy is None and (epoch + 1) % 10 == 0 and save_path is None:  # :528-531
            os.makedirs(os.path.join("src", "models"), exist_ok=True)
            torch.save(trainer.state_dict(), os.path.join("src", "models", f"resnet18_patch_classifier_epoch{epoch+1}.pth"))
        if max_steps is not None and steps >= max_steps:
            break
    if save_path is None:
        name = "resnet18_patch_classifier.pth" if strategy is None else f"resnet18_patch_classifier_{strategy}.pth"
        save_path = os.path.join("src", "models", name)  # src/main.py:533 / :605
    if rank == 0:  # replica 0's parameters and running statistics are the module's (nn.DataParallel)
        os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
        torch.save(trainer.state_dict(), save_path)
    model.load_state_dict(trainer.state_dict())
    return model, history
