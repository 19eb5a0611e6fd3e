"""Minimal host-side stand-ins for the torchvision.transforms the reference composes
(src/main.py:417-430, :812-818): torchvision is not installed in this image.  Only used
on the PNG-tree / training data-loader path; the inference hot path does
Resize/ToTensor/Normalize on the GPU (capi.tile_preprocess)."""
from __future__ import annotations

import random

import numpy as np
import torch
from PIL import Image

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


class Resize:
    """PIL backend of torchvision.transforms.Resize((h, w)): Image.resize(BILINEAR)."""

    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, img: Image.Image):
        return img.resize(self.size[::-1], Image.BILINEAR)


class ToTensor:
    def __call__(self, img: Image.Image):
        a = torch.from_numpy(np.array(img, dtype=np.uint8, copy=True))
        return a.permute(2, 0, 1).contiguous().to(torch.float32).div(255)


class Normalize:
    def __init__(self, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, t: torch.Tensor):
        return (t - self.mean) / self.std


class RandomHorizontalFlip:
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img):
        return img.transpose(Image.FLIP_LEFT_RIGHT) if random.random() < self.p else img


class RandomVerticalFlip:
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img):
        return img.transpose(Image.FLIP_TOP_BOTTOM) if random.random() < self.p else img


def eval_transform():
    """src/main.py:812-818."""
    return Compose([Resize((224, 224)), ToTensor(), Normalize()])


def train_transform():
    """The flip part of src/main.py:417-425 (rotation / colour jitter are omitted:
    augmentation stays host-side and is out of the hot path's scope)."""
    return Compose([RandomHorizontalFlip(), RandomVerticalFlip(), Resize((224, 224)), ToTensor(), Normalize()])
