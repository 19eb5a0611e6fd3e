"""Mirror of src/datasets/patch_dataset.py:8-85 (the loader surface the reference's
pipelines are written against).

Same constructor signature, attributes (``image_paths``, ``labels``, ``label_map``),
label rule (``_tumor`` / ``_normal`` substring of the file name), slide filter (first
path component under ``root_dir``), balancing / sub-sampling via the global ``random``
module, final global ``random.shuffle`` and ``(image, label, path)`` items.

Additions (do not change the reference behaviour): ``raw=True`` returns the decoded
patch as a uint8 HWC tensor so that Resize/ToTensor/Normalize run on the GPU
(``capi.tile_preprocess``) instead of in PIL workers.
"""
from __future__ import annotations

import glob
import os
import random
from collections import Counter, defaultdict

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


class PatchDataset(Dataset):
    def __init__(self, root_dir, transform=None, tumor_transform=None, normal_transform=None, balanced=False,
                 max_samples=None, slide_names=None, raw=False, verbose=True):
        self.tumor_transform = tumor_transform if tumor_transform is not None else transform
        self.normal_transform = normal_transform if normal_transform is not None else transform
        self.transform = transform
        self.raw = raw
        self.image_paths = []
        self.labels = []
        self.label_map = {"_normal": 0, "_tumor": 1}

        by_class = defaultdict(list)
        for path in glob.glob(os.path.join(root_dir, "**", "*.png"), recursive=True):
            if slide_names is not None:
                slide_dir = os.path.relpath(path, root_dir).split(os.sep)[0]
                if slide_dir not in slide_names:
                    continue
            name = os.path.basename(path)
            if "_tumor" in name:
                by_class[1].append(path)
            elif "_normal" in name:
                by_class[0].append(path)
            elif verbose:
                print(f"[WARNING] Could not determine label from filename: {name}")

        if balanced:
            floor = min(len(v) for v in by_class.values())
            for label, paths in by_class.items():
                count = min(floor, max_samples) if max_samples else floor
                chosen = random.sample(paths, min(count, len(paths)))
                self.image_paths.extend(chosen)
                self.labels.extend([label] * len(chosen))
        else:
            for label, paths in by_class.items():
                if max_samples:
                    paths = random.sample(paths, min(len(paths), max_samples))
                self.image_paths.extend(paths)
                self.labels.extend([label] * len(paths))

        if self.image_paths:
            pairs = list(zip(self.image_paths, self.labels))
            random.shuffle(pairs)
            self.image_paths, self.labels = (list(t) for t in zip(*pairs))

        if verbose:
            counts = Counter(self.labels)
            print(f"[INFO] PatchDataset initialized: {len(self.labels)} total patches.")
            print(f"[INFO] Tumor patches: {counts.get(1, 0)} | Normal patches: {counts.get(0, 0)}")

    def __len__(self):
        return len(self.image_paths)

    def __getitem__(self, idx):
        path = self.image_paths[idx]
        label = self.labels[idx]
        image = Image.open(path).convert("RGB")
        if self.raw:
            return torch.from_numpy(np.array(image)), label, path
        if label == 1 and self.tumor_transform:
            image = self.tumor_transform(image)
        elif label == 0 and self.normal_transform:
            image = self.normal_transform(image)
        elif self.transform:
            image = self.transform(image)
        return image, label, path

    def get_class_counts(self):
        return dict(Counter(self.labels))
