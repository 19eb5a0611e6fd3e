"""Oracle: the eval transform  Resize((224,224)) -> ToTensor -> Normalize.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference call sites: src/main.py:812-818 (same composition at :426-430,
:898-902, :986-990).  The arithmetic lives in third-party code: torchvision's
PIL backend forwards to ``PIL.Image.resize(size, BILINEAR)`` (Pillow, present
in this image: the real dependency, used here directly as the pin), then
``ToTensor`` (uint8 HWC -> float32 CHW, ``/255``) and ``Normalize``
(``(x - mean) / std`` in fp32).

Two forms are provided:
  * ``pillow_resize``      -- Pillow itself.  This is the pin.
  * ``resample_restated``  -- a numpy restatement of Pillow's two-pass 8bpc
    resampler (libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
    ImagingResampleHorizontal_8bpc / Vertical_8bpc; PRECISION_BITS = 32-8-2),
    checked bit-for-bit against ``pillow_resize`` in the CPU test-suite.  The
    HIP kernel implements exactly this integer recipe, and the coefficient
    tables it consumes are produced by ``precompute_coeffs`` below.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Pillow: libImaging/Resample.c
OUT_SIZE = 224
IMAGENET_MEAN = (0.485, 0.456, 0.406)  # src/main.py:816
IMAGENET_STD = (0.229, 0.224, 0.225)


def pillow_resize(patch_u8: np.ndarray, size: int = OUT_SIZE) -> np.ndarray:
    """uint8[P,P,3] -> uint8[size,size,3] through Pillow (the real dependency)."""
    from PIL import Image

    img = Image.fromarray(np.ascontiguousarray(patch_u8), "RGB")
    return np.asarray(img.resize((size, size), Image.BILINEAR))


def _triangle(x: float) -> float:
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def precompute_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the
    bilinear (triangle, support 1.0) filter over the full box [0, in_size).

    Returns (bounds int32[out,2] = (xmin, count), kk int32[out,ksize], ksize).
    Every step is IEEE double exactly as in the C source.
    """
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_triangle((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            if v < 0:
                kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS))
            else:
                kk[xx, x] = int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx, 0] = xmin
        bounds[xx, 1] = xmax
    return bounds, kk, ksize


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_restated(patch_u8: np.ndarray, size: int = OUT_SIZE) -> np.ndarray:
    """numpy restatement of Pillow's 8bpc resize: horizontal pass, uint8
    rounding, vertical pass, uint8 rounding.  Identity when P == size
    (Pillow returns a copy when neither pass is needed)."""
    p = patch_u8.shape[0]
    assert patch_u8.shape == (p, p, 3) and patch_u8.dtype == np.uint8
    if p == size:
        return patch_u8.copy()
    bounds, kk, _ = precompute_coeffs(p, size)
    src = patch_u8.astype(np.int64)
    # horizontal: [P, size, 3]
    hor = np.empty((p, size, 3), np.uint8)
    for xx in range(size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full((p, 3), 1 << (PRECISION_BITS - 1), np.int64)
        acc += np.einsum("ykc,k->yc", src[:, x0 : x0 + n, :], kk[xx, :n].astype(np.int64))
        hor[:, xx, :] = _clip8(acc)
    # vertical: [size, size, 3]
    out = np.empty((size, size, 3), np.uint8)
    hor64 = hor.astype(np.int64)
    for yy in range(size):
        y0, n = int(bounds[yy, 0]), int(bounds[yy, 1])
        acc = np.full((size, 3), 1 << (PRECISION_BITS - 1), np.int64)
        acc += np.einsum("kxc,k->xc", hor64[y0 : y0 + n], kk[yy, :n].astype(np.int64))
        out[yy] = _clip8(acc)
    return out


def normalize_lut() -> np.ndarray:
    """float32[3,256]: ``(v/255 - mean_c) / std_c`` evaluated in fp32 with
    torchvision's op order (ToTensor: ``.div(255)``; Normalize: ``sub_(mean)
    .div_(std)`` on fp32 tensors).  A table is exactly equivalent because the
    input takes only 256 values per channel."""
    import torch

    v = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)
    lut = torch.empty(3, 256, dtype=torch.float32)
    for c in range(3):
        m = torch.tensor(IMAGENET_MEAN[c], dtype=torch.float32)
        s = torch.tensor(IMAGENET_STD[c], dtype=torch.float32)
        lut[c] = (v - m) / s
    return lut.numpy()


def to_tensor_normalize(img_u8: np.ndarray) -> np.ndarray:
    """uint8[224,224,3] -> float32[3,224,224], ToTensor + Normalize done with
    torch ops the way torchvision does them (not via the LUT)."""
    import torch

    t = torch.from_numpy(np.ascontiguousarray(img_u8)).permute(2, 0, 1).contiguous()
    t = t.to(torch.float32).div(255)
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(3, 1, 1)
    return ((t - mean) / std).numpy()


def eval_transform(patch_u8: np.ndarray) -> np.ndarray:
    """The whole src/main.py:812-818 pipeline on one RGB patch (Pillow pin)."""
    return to_tensor_normalize(pillow_resize(patch_u8))
