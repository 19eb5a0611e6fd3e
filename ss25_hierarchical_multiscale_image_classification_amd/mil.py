"""Host-side mirror of the reference's MIL pieces.

* ``MILAttentionPooling`` / ``MILClassifier`` -- src/models/mil_classifier.py:5-45: same
  constructor arguments, ``forward(bag) -> (logits, attention)`` shapes and state_dict keys
  (``aggregator.attn_V.*``, ``aggregator.attn_U.*``, ``classifier.0.*``, ``classifier.2.*``).
  In ``eval()`` mode ``forward`` runs ``hipac_mil_forward`` (HIP) -- a CPU tensor raises, there is
  no CPU fallback; in ``train()`` mode it runs the ordinary autograd graph.
  ``forward_bags`` scores MANY bags in one launch pair (the reference loops over bags).
* ``group_patches_by_wsi`` / ``WSIMILDDataset`` -- src/datasets/mildataset.py:6-47.  By default the
  bag key is the reference's as written: ``'_'.join(basename.split('_')[:-2])``, which for the patch
  names ``{slide}_x{x}_y{y}_{label}.png`` keeps the ``_x{x}`` field (one bag per slide COLUMN);
  ``by_slide=True`` drops it, which is what the reference's comment describes.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import capi


def group_patches_by_wsi(paths: Sequence[str], labels: Sequence[int], by_slide: bool = False
                         ) -> Tuple[np.ndarray, np.ndarray, List[str], np.ndarray]:
    """-> (order int64[n]: row indices sorted by bag, first-appearance bag order, original row order
    inside a bag; offsets int64[n_bags+1]; bag names; wsi_labels int64[n_bags] = any member label == 1)."""
    drop = 3 if by_slide else 2
    index, rows, wsi = {}, [], []
    for i, p in enumerate(paths):
        key = "_".join(os.path.basename(p).split("_")[:-drop])
        b = index.setdefault(key, len(index))
        if b == len(rows):
            rows.append([]), wsi.append(0)
        rows[b].append(i)
        if int(labels[i]) == 1:
            wsi[b] = 1
    order = np.array([i for r in rows for i in r], np.int64)
    offsets = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    return order, offsets, list(index.keys()), np.array(wsi, np.int64)


class WSIMILDDataset(torch.utils.data.Dataset):
    """src/datasets/mildataset.py:6-47: bags from the (features .npy, labels .npy, paths .txt) triple that
    ``--extract_features`` writes.  ``__getitem__ -> (features float32[n_i,F], wsi_label int64 scalar)``.
    The files are this repository's own output or the user's: ``allow_pickle`` stays off."""

    def __init__(self, features_path, labels_path, paths_path, by_slide: bool = False):
        self.features = np.load(features_path)
        self.labels = np.load(labels_path)
        with open(paths_path, "r") as f:
            self.paths = [line.strip() for line in f]
        self.order, self.offsets, self.names, wsi = group_patches_by_wsi(self.paths, self.labels, by_slide)
        self.wsi_data = [{"features": torch.tensor(self.features[self.order[a:b]], dtype=torch.float32),
                          "patch_labels": torch.tensor(self.labels[self.order[a:b]], dtype=torch.long),
                          "wsi_label": torch.tensor(int(w), dtype=torch.long)}
                         for a, b, w in zip(self.offsets[:-1], self.offsets[1:], wsi)]

    def __len__(self):
        return len(self.wsi_data)

    def __getitem__(self, idx):
        return self.wsi_data[idx]["features"], self.wsi_data[idx]["wsi_label"]


class MILAttentionPooling(nn.Module):
    """mil_classifier.py:5-18 (ABMIL, Ilse et al.)."""

    def __init__(self, in_dim, attn_dim=128):
        super().__init__()
        self.attn_V = nn.Linear(in_dim, attn_dim)
        self.attn_U = nn.Linear(attn_dim, 1)

    def forward(self, x):
        a = torch.softmax(self.attn_U(torch.tanh(self.attn_V(x))), dim=0)
        return torch.sum(a * x, dim=0), a


class MILClassifier(nn.Module):
    """mil_classifier.py:20-45."""

    def __init__(self, feature_dim, num_classes=2, pooling="attention"):
        super().__init__()
        if pooling not in ("attention", "mean", "max"):
            raise ValueError("Unknown pooling: choose from 'attention', 'mean', 'max'")
        self.pooling = pooling
        if pooling == "attention":
            self.aggregator = MILAttentionPooling(feature_dim)
        self.classifier = nn.Sequential(nn.Linear(feature_dim, 128), nn.ReLU(), nn.Linear(128, num_classes))

    def _aggregate(self, bag):
        if self.pooling == "attention":
            return self.aggregator(bag)
        return (bag.mean(dim=0), None) if self.pooling == "mean" else (bag.max(dim=0)[0], None)

    def forward_bags(self, feats: torch.Tensor, bag_offsets, want_pooled: bool = False):
        """HIP path for many bags: feats float32[n,F] (bag rows contiguous, on a ROCm device),
        bag_offsets int[n_bags+1] -> (logits[n_bags,C], attn[n] or None[, pooled[n_bags,F]])."""
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        offs = torch.as_tensor(np.asarray(bag_offsets)) if not torch.is_tensor(bag_offsets) else bag_offsets
        logits, attn, pooled = capi.mil_forward(sd, self.pooling, feats.contiguous(), offs, want_pooled=want_pooled)
        return (logits, attn, pooled) if want_pooled else (logits, attn)

    def forward(self, bag):
        """bag: (num_patches, feature_dim) -> (logits (num_classes), attention (num_patches, 1) or None)."""
        if self.training:
            pooled, attn = self._aggregate(bag)
            return self.classifier(pooled), attn
        logits, attn = self.forward_bags(bag, torch.tensor([0, bag.shape[0]]))
        return logits[0], (None if attn is None else attn.unsqueeze(1))

    def predict(self, bag):
        """Class probabilities of one bag (the reference's ``predict`` does not run as written: it applies a
        numpy softmax to the (logits, attn) tuple, mil_classifier.py:47-50; this is its evident intent)."""
        logits, _ = self.forward(bag)
        return torch.softmax(logits.float(), dim=-1)
