"""Oracle: the MIL bag builder and ABMIL head as the reference computes them.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PINNED: tests/golden/mil_golden.npz and
mil_dataset_ref.json were produced by the reference's own modules imported by file
location (tests/golden/make_golden_mil.py); tests/test_oracle_mil.py checks this
restatement against them.

Follows, step for step:
  src/models/mil_classifier.py:12-18  MILAttentionPooling.forward
      A = tanh(attn_V(x)); A = attn_U(A); A = softmax(A, dim=0); M = sum(A * x, dim=0)
  src/models/mil_classifier.py:21-45  MILClassifier: aggregator in {attention, mean, max},
      classifier = Linear(F,128) -> ReLU -> Linear(128, num_classes); forward -> (logits, attn)
  src/datasets/mildataset.py:15-38    WSIMILDDataset._group_patches_by_wsi:
      key = '_'.join(basename(path).split('_')[:-2])   (first-appearance order of keys,
      original order of rows inside a bag); wsi_label = 1 iff any member label == 1.
      NOTE the key keeps the `_x{x}` field of `{slide}_x{x}_y{y}_{label}.png` -- the
      comment in the reference promises the slide name, the code groups by slide AND x.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence, Tuple

import numpy as np


def mil_forward_ref(sd: Dict[str, np.ndarray], bag: np.ndarray, pooling: str = "attention"):
    """float64 accumulation inside numpy's float32 matmuls is not requested: plain float32 like torch."""
    x = np.asarray(bag, np.float32)
    attn = None
    if pooling == "attention":
        a = np.tanh(x @ sd["aggregator.attn_V.weight"].T + sd["aggregator.attn_V.bias"])
        a = a @ sd["aggregator.attn_U.weight"].T + sd["aggregator.attn_U.bias"]  # (N,1)
        a = a - a.max(axis=0, keepdims=True)
        e = np.exp(a)
        attn = (e / e.sum(axis=0, keepdims=True)).astype(np.float32)
        pooled = (attn * x).sum(axis=0)
    elif pooling == "mean":
        pooled = x.mean(axis=0)
    elif pooling == "max":
        pooled = x.max(axis=0)
    else:
        raise ValueError("Unknown pooling: choose from 'attention', 'mean', 'max'")  # mil_classifier.py:31
    h = np.maximum(pooled @ sd["classifier.0.weight"].T + sd["classifier.0.bias"], 0)
    logits = h @ sd["classifier.2.weight"].T + sd["classifier.2.bias"]
    return logits.astype(np.float32), attn, pooled.astype(np.float32)


def wsi_key(path: str, keep_x_field: bool = True) -> str:
    """mildataset.py:24.  keep_x_field=True is the reference as written ([:-2])."""
    parts = os.path.basename(path).split("_")
    return "_".join(parts[:-2] if keep_x_field else parts[:-3])


def group_patches_ref(paths: Sequence[str], labels: Sequence[int], keep_x_field: bool = True
                      ) -> List[Tuple[str, List[int], int]]:
    bags: Dict[str, Tuple[List[int], List[int]]] = {}
    for i, p in enumerate(paths):
        k = wsi_key(p, keep_x_field)
        if k not in bags:
            bags[k] = ([], [0])
        bags[k][0].append(i)
        if labels[i] == 1:
            bags[k][1][0] = 1
    return [(k, rows, lab[0]) for k, (rows, lab) in bags.items()]
