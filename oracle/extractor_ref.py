"""Oracle: the sliding-window patch extractor, followed line by line.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates ``extract_patches`` (src/main.py:609-732) and ``parse_xml_mask``
(src/main.py:372-410) with the slide source replaced by an in-memory pyramid
that honours the openslide calls the reference makes (``level_dimensions``,
``level_downsamples``, ``read_region(location_l0, level, size)`` -> RGBA PIL
image).  PNG write/read is the identity on RGB8 and is skipped
(src/main.py:722-726).  Grid arithmetic is pinned on the reference notebooks'
known answers; pixels are PARITY UNPINNED (real slides absent).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
from PIL import Image, ImageDraw, ImageOps

PATCH_SIZES = {0: 1792, 1: 896, 2: 448, 3: 224}  # src/main.py:614
WHITE_MEAN_THRESHOLD = 240  # src/main.py:719
LABEL_NORMAL, LABEL_TUMOR = 0, 1  # src/datasets/patch_dataset.py:15


class ArraySlide:
    """Minimal openslide.OpenSlide stand-in over uint8[H,W,3] level arrays.

    ``read_region`` follows openslide: ``location`` is in level-0 pixels, the
    result is RGBA, anything outside the level is transparent black (which
    ``.convert("RGB")`` turns into (0,0,0)).  The reference never reads outside
    (it clips ``patch_w/h`` first, src/main.py:688-689)."""

    def __init__(self, levels: Sequence[np.ndarray]):
        self.levels = [np.ascontiguousarray(a) for a in levels]
        self.level_dimensions = tuple((a.shape[1], a.shape[0]) for a in self.levels)
        w0 = self.level_dimensions[0][0]
        self.level_downsamples = tuple(float(2 ** i) for i in range(len(self.levels)))
        assert all(a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3 for a in self.levels)
        del w0

    def read_region(self, location: Tuple[int, int], level: int, size: Tuple[int, int]) -> Image.Image:
        ds = self.level_downsamples[level]
        x = int(location[0] / ds)
        y = int(location[1] / ds)
        w, h = size
        lvl = self.levels[level]
        rgba = np.zeros((h, w, 4), np.uint8)
        x0, y0 = max(x, 0), max(y, 0)
        x1, y1 = min(x + w, lvl.shape[1]), min(y + h, lvl.shape[0])
        if x1 > x0 and y1 > y0:
            rgba[y0 - y : y1 - y, x0 - x : x1 - x, :3] = lvl[y0:y1, x0:x1]
            rgba[y0 - y : y1 - y, x0 - x : x1 - x, 3] = 255
        return Image.fromarray(rgba, "RGBA")


def parse_mask(
    polygons_l0: Iterable[Sequence[Tuple[float, float]]],
    level_dims: Tuple[int, int],
    base_dims: Tuple[int, int],
) -> Image.Image:
    """src/main.py:387-410 with the XML walk replaced by a polygon list
    (level-0 float coordinates): scale by level/base dims, ``int()``
    truncation, ``ImageDraw.polygon(coords, outline=255, fill=255)`` on an
    'L' image of the level's size."""
    scale_x = level_dims[0] / base_dims[0]
    scale_y = level_dims[1] / base_dims[1]
    mask = Image.new("L", level_dims, 0)
    draw = ImageDraw.Draw(mask)
    for poly in polygons_l0:
        coords = [(int(float(x) * scale_x), int(float(y) * scale_y)) for x, y in poly]
        if coords:
            draw.polygon(coords, outline=255, fill=255)
    return mask


@dataclass
class Window:
    level: int
    x: int  # level-L pixel coordinates, as in the reference's file names (:722)
    y: int
    pw: int
    ph: int
    pixel_sum: int  # sum over the padded PxPx3 uint8 window
    keep: bool  # not (mean > 240)
    label: int  # 0 normal / 1 tumor


def window_grid(width: int, height: int, level: int, stride: Optional[int] = None, pad: bool = True):
    """The loop bounds of src/main.py:611-615, :658-665, :682-691, in the
    reference's order (x outer, y inner).  ``stride=None`` reproduces the
    reference exactly: ``stride = stride or patch_size`` is evaluated with the
    default ``patch_size=224`` *before* the per-level override, so the stride
    is 224 at every level."""
    patch_size = 224
    stride = stride or patch_size  # :611
    patch_size = PATCH_SIZES.get(level, 224)  # :614-615
    if pad:
        pad_w = (patch_size - width % patch_size) % patch_size
        pad_h = (patch_size - height % patch_size) % patch_size
    else:
        pad_w = pad_h = 0
    out = []
    for x in range(0, width + pad_w, stride):
        for y in range(0, height + pad_h, stride):
            if x >= width or y >= height:
                continue
            pw = min(patch_size, width - x)
            ph = min(patch_size, height - y)
            if pw <= 0 or ph <= 0:
                continue
            out.append((x, y, pw, ph))
    return patch_size, stride, (pad_w, pad_h), out


def extract_patches_ref(
    slide: ArraySlide,
    level: int,
    polygons_l0: Optional[Iterable[Sequence[Tuple[float, float]]]] = None,
    stride: Optional[int] = None,
    pad: bool = True,
    return_pixels: bool = True,
) -> Tuple[List[Window], List[np.ndarray]]:
    """One slide, one level of src/main.py:609-732.  Returns every window the
    loop visits (kept or not, in loop order) and, for the kept ones, the
    PxPx3 uint8 patch the reference would have written to PNG."""
    downsample = slide.level_downsamples[level]
    width, height = slide.level_dimensions[level]
    patch_size, stride, (pad_w, pad_h), grid = window_grid(width, height, level, stride, pad)

    mask = None
    if polygons_l0 is not None:
        mask = parse_mask(polygons_l0, (width, height), slide.level_dimensions[0])
        if pad and (pad_w > 0 or pad_h > 0):
            mask = ImageOps.expand(mask, (0, 0, pad_w, pad_h), fill=0)  # :673

    windows: List[Window] = []
    pixels: List[np.ndarray] = []
    for x, y, pw, ph in grid:
        region = slide.read_region((int(x * downsample), int(y * downsample)), level, (pw, ph)).convert("RGB")
        if pw < patch_size or ph < patch_size:  # :700-703
            canvas = Image.new("RGB", (patch_size, patch_size), (255, 255, 255))
            canvas.paste(region, (0, 0))
            region = canvas
        if mask:  # :707-716 (PIL images are truthy when non-empty)
            mask_patch = mask.crop((x, y, x + patch_size, y + patch_size))
            label = LABEL_TUMOR if np.any(np.array(mask_patch) > 0) else LABEL_NORMAL
        else:
            label = LABEL_NORMAL
        arr = np.array(region)
        keep = not (np.mean(arr) > WHITE_MEAN_THRESHOLD)  # :719
        windows.append(Window(level, x, y, pw, ph, int(arr.sum(dtype=np.int64)), bool(keep), label))
        if keep and return_pixels:
            pixels.append(arr)
    return windows, pixels


def whiteness_keep_integer(pixel_sum: int, patch_size: int) -> bool:
    """Exact integer form of ``not (np.mean(arr) > 240)``:
    ``sum <= 240 * 3 * P * P``.  float64 represents both sides exactly
    (sum < 2^32), so the comparison is identical."""
    return pixel_sum <= WHITE_MEAN_THRESHOLD * 3 * patch_size * patch_size


def patch_file_name(prefix: str, x: int, y: int, label: int) -> str:
    """src/main.py:722."""
    return f"{prefix}_x{x}_y{y}_{'tumor' if label == LABEL_TUMOR else 'normal'}.png"
