"""Synthetic inputs: whole-slide pyramids and seeded ResNet18 state dicts.

There is no network for CAMELYON16 or checkpoints (and the reference's own
trained checkpoint is absent, /root/reference/.MISSING_LARGE_BLOBS), so every
test and benchmark runs on data made here.  Nothing in this file is part of
the reference's algorithm; it only fixes the *inputs* both the HIP path and
the oracle are fed (SURVEY.md section 7 "What level k means for a synthetic
slide").
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# ----------------------------------------------------------------------------
# slides
# ----------------------------------------------------------------------------


def synth_level0(
    width: int,
    height: int,
    seed: int = 2,
    n_blobs: int = 6,
    device: str | torch.device = "cpu",
    band_rows: int = 2048,
) -> torch.Tensor:
    """uint8[H,W,3] level-0 image: near-white background (250 +- 3) with
    ``n_blobs`` seeded elliptical "tissue" regions of textured pink/purple.
    Generated in row bands so a 50k x 50k slide (7.5 GB) never needs more than
    a band of temporaries.  Deterministic per (seed, device type)."""
    dev = torch.device(device)
    rng = np.random.RandomState(seed)
    blobs = []
    for _ in range(n_blobs):
        cx, cy = rng.uniform(0.1, 0.9) * width, rng.uniform(0.1, 0.9) * height
        rx, ry = rng.uniform(0.08, 0.28) * width, rng.uniform(0.08, 0.28) * height
        th = rng.uniform(0, math.pi)
        col = np.array([rng.uniform(150, 215), rng.uniform(70, 150), rng.uniform(140, 200)], np.float32)
        blobs.append((cx, cy, rx, ry, th, col))
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    out = torch.empty((height, width, 3), dtype=torch.uint8, device=dev)
    xs = torch.arange(width, device=dev, dtype=torch.float32)[None, :]
    for y0 in range(0, height, band_rows):
        y1 = min(height, y0 + band_rows)
        ys = torch.arange(y0, y1, device=dev, dtype=torch.float32)[:, None]
        band = 247.0 + torch.randint(0, 7, (y1 - y0, width, 3), generator=gen, device=dev, dtype=torch.int16).float()
        for cx, cy, rx, ry, th, col in blobs:
            dx, dy = xs - cx, ys - cy
            u = (dx * math.cos(th) + dy * math.sin(th)) / rx
            v = (-dx * math.sin(th) + dy * math.cos(th)) / ry
            inside = (u * u + v * v) < 1.0
            tex = torch.randint(-40, 41, (y1 - y0, width, 3), generator=gen, device=dev, dtype=torch.int16).float()
            tissue = (torch.tensor(col, device=dev)[None, None, :] + tex).clamp_(0, 255)
            band = torch.where(inside[:, :, None], tissue, band)
        out[y0:y1] = band.to(torch.uint8)
    return out


def downsample2x(img: torch.Tensor) -> torch.Tensor:
    """uint8[H,W,3] -> uint8[H//2,W//2,3]: 2x2 box mean, round half up.
    (Ours: the reference reads pre-built TIFF levels and never downsamples.)"""
    h2, w2 = img.shape[0] // 2, img.shape[1] // 2
    out = torch.empty((h2, w2, 3), dtype=torch.uint8, device=img.device)
    band = 4096  # output rows per pass: bounds the int32 temporaries for 100k-wide slides
    for r0 in range(0, h2, band):
        r1 = min(h2, r0 + band)
        a = img[2 * r0: 2 * r1, : w2 * 2].reshape(r1 - r0, 2, w2, 2, 3).to(torch.int32)
        out[r0:r1] = ((a.sum(dim=(1, 3)) + 2) >> 2).to(torch.uint8)
    return out


def build_pyramid(level0: torch.Tensor, n_levels: int = 4) -> List[torch.Tensor]:
    levels = [level0]
    for _ in range(1, n_levels):
        levels.append(downsample2x(levels[-1]))
    return levels


def synth_polygons(width: int, height: int, seed: int = 2, n: int = 2) -> List[List[Tuple[float, float]]]:
    """A few convex "tumour" annotation polygons in level-0 float coordinates
    (the reference reads them from ASAP XML, src/main.py:395-409)."""
    rng = np.random.RandomState(seed + 1000)
    polys = []
    for _ in range(n):
        cx, cy = rng.uniform(0.2, 0.8) * width, rng.uniform(0.2, 0.8) * height
        r = rng.uniform(0.03, 0.12) * min(width, height)
        k = rng.randint(5, 9)
        ang = np.sort(rng.uniform(0, 2 * math.pi, k))
        polys.append([(float(cx + r * math.cos(a)), float(cy + r * math.sin(a))) for a in ang])
    return polys


# ----------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------

_STAGES = (("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2), ("layer4", 512, 2))


def seeded_resnet18_state_dict(
    seed: int = 0, num_classes: Optional[int] = 2, prefix: str = ""
) -> Dict[str, torch.Tensor]:
    """A full torchvision-named ResNet18 state dict with random weights and
    NON-TRIVIAL batch-norm statistics (so BN folding is really exercised).
    Conv weights: kaiming-normal fan_out (torchvision's init); BN gamma in
    [0.5,1.5], beta ~ N(0,0.1), running_mean ~ N(0,0.1), running_var in
    [0.5,1.5]; the last BN of each block gets gamma scaled by 0.5 so the
    residual stream keeps O(1) magnitude through 8 blocks."""
    g = torch.Generator().manual_seed(seed)

    def conv(cout, cin, k):
        std = math.sqrt(2.0 / (cout * k * k))
        return torch.randn(cout, cin, k, k, generator=g) * std

    def bn(name, c, sd, gamma_scale=1.0):
        sd[name + ".weight"] = (0.5 + torch.rand(c, generator=g)) * gamma_scale
        sd[name + ".bias"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_mean"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)
        sd[name + ".num_batches_tracked"] = torch.tensor(100, dtype=torch.int64)

    sd: Dict[str, torch.Tensor] = {}
    sd["conv1.weight"] = conv(64, 3, 7)
    bn("bn1", 64, sd)
    cin = 64
    for name, cout, stride in _STAGES:
        for blk in (0, 1):
            p = f"{name}.{blk}"
            sd[p + ".conv1.weight"] = conv(cout, cin if blk == 0 else cout, 3)
            bn(p + ".bn1", cout, sd)
            sd[p + ".conv2.weight"] = conv(cout, cout, 3)
            bn(p + ".bn2", cout, sd, gamma_scale=0.5)
            if blk == 0 and (stride != 1 or cin != cout):
                sd[p + ".downsample.0.weight"] = conv(cout, cin, 1)
                bn(p + ".downsample.1", cout, sd)
        cin = cout
    if num_classes:
        bound = 1.0 / math.sqrt(512)
        sd["fc.weight"] = (torch.rand(num_classes, 512, generator=g) * 2 - 1) * bound
        sd["fc.bias"] = (torch.rand(num_classes, generator=g) * 2 - 1) * bound
    if prefix:
        sd = {prefix + k: v for k, v in sd.items()}
    return sd


def synth_patches_u8(n: int, seed: int = 1, size: int = 224, device: str | torch.device = "cpu") -> torch.Tensor:
    """uint8[n,size,size,3] uniform random patches (BASELINE config 1/2 input)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randint(0, 256, (n, size, size, 3), generator=g, device=dev, dtype=torch.uint8)
