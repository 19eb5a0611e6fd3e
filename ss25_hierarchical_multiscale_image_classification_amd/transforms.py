"""Host-side stand-ins for the torchvision.transforms the reference composes
(src/main.py:417-430, :812-818; src/models/simclr.py:57-66): torchvision is not installed in
this image, so the PIL-backend behaviour of each transform is restated here (parameter
distributions, op order, PIL calls).  PARITY UNPINNED against torchvision itself: only the
deterministic cores (hue shift, grayscale, crop arithmetic) are checked, in tests/test_host_logic.py.
Only used on the PNG-tree / training data-loader path; augmentation stays on the host (out of the
hot path's scope); the inference path does Resize/ToTensor/Normalize on the GPU."""
from __future__ import annotations

import math
import random

import numpy as np
import torch
from PIL import Image, ImageEnhance

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


class RandomRotation:
    """torchvision.transforms.RandomRotation(degrees): angle ~ U(-degrees, degrees), PIL rotate with NEAREST
    resampling, no expansion, fill 0 (the defaults the reference uses, src/main.py:420)."""

    def __init__(self, degrees):
        self.degrees = (-float(degrees), float(degrees)) if not isinstance(degrees, (tuple, list)) else tuple(degrees)

    def __call__(self, img: Image.Image):
        angle = float(torch.empty(1).uniform_(self.degrees[0], self.degrees[1]).item())
        return img.rotate(angle, resample=Image.NEAREST, expand=False, fillcolor=0)


def adjust_hue(img: Image.Image, hue_factor: float) -> Image.Image:
    """torchvision's PIL adjust_hue: shift the H channel of the HSV image by hue_factor * 255 with uint8 wrap."""
    if not -0.5 <= hue_factor <= 0.5:
        raise ValueError("hue_factor is not in [-0.5, 0.5]")
    if img.mode in ("L", "1", "I", "F"):
        return img
    h, s_, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h += np.uint8(int(hue_factor * 255) & 0xFF)
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s_, v)).convert(img.mode)


class ColorJitter:
    """torchvision.transforms.ColorJitter(brightness, contrast, saturation, hue) on PIL images: factors
    ~ U(max(0, 1 - x), 1 + x), hue ~ U(-hue, hue), the four adjustments applied in a random order
    (src/main.py:421, src/models/simclr.py:61)."""

    def __init__(self, brightness=0.0, contrast=0.0, saturation=0.0, hue=0.0):
        rng = lambda x: None if not x else (max(0.0, 1.0 - x), 1.0 + x)
        self.brightness, self.contrast, self.saturation = rng(brightness), rng(contrast), rng(saturation)
        self.hue = None if not hue else (-float(hue), float(hue))

    def __call__(self, img: Image.Image):
        order = torch.randperm(4).tolist()
        draw = lambda r: None if r is None else float(torch.empty(1).uniform_(r[0], r[1]).item())
        b, c, s_, h = draw(self.brightness), draw(self.contrast), draw(self.saturation), draw(self.hue)
        for fn in order:
            if fn == 0 and b is not None:
                img = ImageEnhance.Brightness(img).enhance(b)
            elif fn == 1 and c is not None:
                img = ImageEnhance.Contrast(img).enhance(c)
            elif fn == 2 and s_ is not None:
                img = ImageEnhance.Color(img).enhance(s_)
            elif fn == 3 and h is not None:
                img = adjust_hue(img, h)
        return img


class RandomResizedCrop:
    """torchvision.transforms.RandomResizedCrop(size): 10 attempts at area ~ U(scale) * A, log-ratio ~ U(log ratio),
    else the centre crop clamped to the ratio range; then Resize((size, size)) bilinear (src/models/simclr.py:59)."""

    def __init__(self, size, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
        self.size = (size, size) if isinstance(size, int) else tuple(size)
        self.scale, self.ratio = scale, ratio

    @staticmethod
    def get_params(width: int, height: int, scale, ratio):
        area = height * width
        log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
        for _ in range(10):
            target = area * float(torch.empty(1).uniform_(scale[0], scale[1]).item())
            ar = math.exp(float(torch.empty(1).uniform_(log_ratio[0], log_ratio[1]).item()))
            w, h = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
            if 0 < w <= width and 0 < h <= height:
                i = int(torch.randint(0, height - h + 1, (1,)).item())
                j = int(torch.randint(0, width - w + 1, (1,)).item())
                return i, j, h, w
        in_ratio = width / height
        if in_ratio < min(ratio):
            w, h = width, int(round(width / min(ratio)))
        elif in_ratio > max(ratio):
            h, w = height, int(round(height * max(ratio)))
        else:
            w, h = width, height
        return (height - h) // 2, (width - w) // 2, h, w

    def __call__(self, img: Image.Image):
        i, j, h, w = self.get_params(img.width, img.height, self.scale, self.ratio)
        return img.crop((j, i, j + w, i + h)).resize(self.size[::-1], Image.BILINEAR)


class RandomApply:
    def __init__(self, transforms, p=0.5):
        self.transforms, self.p = list(transforms), p

    def __call__(self, img):
        if self.p < float(torch.rand(1).item()):
            return img
        for t in self.transforms:
            img = t(img)
        return img


class RandomGrayscale:
    """torchvision.transforms.RandomGrayscale(p): with probability p the L-mode image replicated on 3 channels."""

    def __init__(self, p=0.1):
        self.p = p

    def __call__(self, img: Image.Image):
        if float(torch.rand(1).item()) < self.p:
            g = np.array(img.convert("L"), dtype=np.uint8)
            return Image.fromarray(np.dstack([g, g, g]), "RGB")
        return img


def eval_transform():
    """src/main.py:812-818."""
    return Compose([Resize((224, 224)), ToTensor(), Normalize()])


def train_transform():
    """src/main.py:417-425: flips, RandomRotation(90), ColorJitter(0.2, 0.2, 0.2, 0.1), Resize, ToTensor, Normalize."""
    return Compose([RandomHorizontalFlip(), RandomVerticalFlip(), RandomRotation(90),
                    ColorJitter(brightness=0.2, contrast=0.2, saturation=0.2, hue=0.1),
                    Resize((224, 224)), ToTensor(), Normalize()])


def simclr_transform():
    """src/models/simclr.py:57-66 (get_simclr_transform)."""
    return Compose([RandomResizedCrop(224), RandomHorizontalFlip(),
                    RandomApply([ColorJitter(0.4, 0.4, 0.4, 0.1)], p=0.8), RandomGrayscale(p=0.2),
                    ToTensor(), Normalize()])
