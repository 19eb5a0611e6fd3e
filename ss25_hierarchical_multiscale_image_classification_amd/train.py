"""Classifier fine-tune loops (mirror of src/main.py:412-606), kept functional so the
``--train`` / ``--train_strategy`` flags work.

Scope (SURVEY.md section 8, a-13 = "next"): the TRAINING forward/backward runs on the
module's ordinary autograd graph; only the per-epoch VALIDATION scoring uses the HIP
inference path.  A native fwd+bwd is a later row of the scope table.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, Subset

from .patch_dataset import PatchDataset
from .resnet import ResNet18Classifier
from .transforms import eval_transform, train_transform

SAMPLES_PER_CLASS = 7480  # src/main.py:50


def get_dataloaders(patch_dir: str, test_ratio: float = 0.2, batch_size: int = 512, balanced: bool = False):
    """src/main.py:412-470: slide-level split (random_state=42), tumour patches
    augmented, normal patches not, validation set balanced with default_rng(42)."""
    from sklearn.model_selection import train_test_split

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
    return (DataLoader(train_ds, batch_size=batch_size, shuffle=True),
            DataLoader(val_ds, batch_size=batch_size, shuffle=False), train_ds, val_ds)


def class_weights(train_ds: PatchDataset, strategy: Optional[str]) -> Optional[torch.Tensor]:
    counts = train_ds.get_class_counts()
    c0, c1 = max(counts.get(0, 0), 1), max(counts.get(1, 0), 1)
    if strategy is None:  # src/main.py:485-490: 1/count normalised by the smaller weight
        w = torch.tensor([1.0 / c0, 1.0 / c1])
        return w / w.min()
    if strategy == "weighted_loss":  # src/main.py:549-552: total/count
        tot = c0 + c1
        return torch.tensor([tot / c0, tot / c1])
    return None


def train_resnet_classifier(patch_dir: str, strategy: Optional[str] = None, epochs: Optional[int] = None,
                            batch_size: int = 512, precision: str = "bf16", lr: float = 1e-4,
                            save_path: Optional[str] = None, device: str = "cuda"):
    epochs = epochs if epochs is not None else (30 if strategy is None else 5)  # :494 / :575
    train_loader, val_loader, train_ds, val_ds = get_dataloaders(patch_dir, 0.2, batch_size,
                                                                 balanced=strategy == "balanced")
    dev = torch.device(device)
    model = ResNet18Classifier().to(dev).set_precision(precision)
    w = class_weights(train_ds, strategy)
    criterion = nn.CrossEntropyLoss(weight=w.to(dev) if w is not None else None)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    history = []
    for epoch in range(epochs):
        model.train()
        total, correct, seen = 0.0, 0, 0
        for imgs, labels, _ in train_loader:
            imgs, labels = imgs.to(dev), labels.to(dev)
            opt.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dev.type == "cuda"):
                out = model(imgs)
                loss = criterion(out.float(), labels)
            loss.backward()
            opt.step()
            total += float(loss)
            correct += int((out.argmax(1) == labels).sum())
            seen += int(labels.numel())
        model.eval()  # validation scoring on the HIP path
        v_correct, v_seen = 0, 0
        with torch.no_grad():
            for imgs, labels, _ in val_loader:
                pred, _ = model.predict(imgs.to(dev))
                v_correct += int((pred.cpu() == labels).sum())
                v_seen += int(labels.numel())
        history.append((total, correct / max(seen, 1), v_correct / max(v_seen, 1)))
        print(f"Epoch {epoch+1}, Train Loss: {total:.4f}, Train Acc: {history[-1][1]:.4f}, Val Acc: {history[-1][2]:.4f}")
    if save_path is None:
        name = "resnet18_patch_classifier.pth" if strategy is None else f"resnet18_patch_classifier_{strategy}.pth"
        save_path = os.path.join("src", "models", name)  # src/main.py:533 / :605
    os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
    torch.save(model.state_dict(), save_path)
    return model, history
