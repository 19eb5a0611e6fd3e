"""SimCLR's input pipeline with the patches resident in HBM and the augmentation on the device.

The reference (src/models/simclr.py:57-83) reads PNG patches through ``PatchDataset(transform=None)``, wraps them in
``SimCLRDataset(base, get_simclr_transform())`` and lets DataLoader workers run the torchvision / Pillow transforms per
sample, twice per patch.  Here the decoded patches live once in HBM (``DevicePatchPool``: uint8 [N, P, P, 3]) and a step's
two views are produced by ``hipac_augment_views`` (csrc/augment.hip: Pillow's arithmetic, bit-exact).  What stays on the
host is the random DRAWING: ``draw_simclr_view`` takes the draws of ``transforms.simclr_transform()`` in the same order from
the same generators (torch's global generator for crop / jitter / grayscale, ``random`` for the flip), so that with equal
seeds the host pipeline and this one see the same parameters -- which is how tests/test_gpu_augment.py compares them.
"""
from __future__ import annotations

import math
import random
from concurrent.futures import ThreadPoolExecutor
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import capi
from .transforms import RandomResizedCrop

PARAMS = 24  # int32 per view (include/hipac.h, hipac_augment_views)
IDENTITY_FIX = (65536, 0, 0, 0, 65536, 0)
OUT = 224


def _f32_bits(x: float) -> int:
    return int(np.float32(x).view(np.int32))


def draw_simclr_view(index: int, width: int, height: int, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0),
                     jitter=(0.4, 0.4, 0.4, 0.1), p_jitter: float = 0.8, p_gray: float = 0.2, p_flip: float = 0.5) -> List[int]:
    """The 16 parameters of one view: the draws of RandomResizedCrop.get_params, RandomHorizontalFlip, RandomApply,
    ColorJitter (randperm(4), then brightness / contrast / saturation / hue) and RandomGrayscale, in that order
    (src/models/simclr.py:58-62; transforms.py restates the torchvision classes)."""
    top, left, h, w = RandomResizedCrop.get_params(width, height, scale, ratio)
    flip = 1 if random.random() < p_flip else 0
    ops = [-1, -1, -1, -1]
    b = c = s = 1.0
    hue = 0
    if not (p_jitter < float(torch.rand(1).item())):
        ops = torch.randperm(4).tolist()
        draw = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi).item())
        b = draw(max(0.0, 1.0 - jitter[0]), 1.0 + jitter[0])
        c = draw(max(0.0, 1.0 - jitter[1]), 1.0 + jitter[1])
        s = draw(max(0.0, 1.0 - jitter[2]), 1.0 + jitter[2])
        hue = int(draw(-jitter[3], jitter[3]) * 255) & 0xFF  # adjust_hue: np.uint8(int(hue_factor * 255))
    gray = 1 if float(torch.rand(1).item()) < p_gray else 0
    return [index, top, left, h, w, flip, *ops, gray, _f32_bits(b), _f32_bits(c), _f32_bits(s), hue, 0, 0, *IDENTITY_FIX, 0]


def pil_rotate_fixed(angle: float, w: int, h: int) -> Tuple[int, int, int, int, int, int]:
    """The six 16.16 integers with which Pillow resamples ``img.rotate(angle, NEAREST, expand=False)``: Image.rotate's matrix
    (cos / sin rounded to 15 decimals, rotation about (w/2, h/2)), then Geometry.c's affine_fixed (FIX(v) = FLOOR(v 65536 + .5),
    half-pixel centre folded into a2 / a5); the transpose fast paths of 0 / 90 / 180 / 270 degrees as exact integer maps."""
    angle = angle % 360.0
    if angle == 0:
        return IDENTITY_FIX
    if angle == 180:
        return (-65536, 0, (w - 1) << 16, 0, -65536, (h - 1) << 16)
    if angle in (90, 270) and w == h:
        return (0, -65536, (w - 1) << 16, 65536, 0, 0) if angle == 90 else (0, 65536, 0, -65536, 0, (h - 1) << 16)
    rad = -math.radians(angle)
    m = [round(math.cos(rad), 15), round(math.sin(rad), 15), 0.0, round(-math.sin(rad), 15), round(math.cos(rad), 15), 0.0]
    cx, cy = w / 2, h / 2
    m[2], m[5] = m[0] * -cx + m[1] * -cy + m[2], m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    floor_c = lambda x: int(math.floor(x)) if x < 0.0 else int(x)
    fix = lambda v: floor_c(v * 65536.0 + 0.5)
    return (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


def identity_view(index: int) -> List[int]:
    """Geometry 1 with nothing drawn: the eval transform of a 224-pixel patch (normal patches, validation)."""
    return [index, 0, 0, OUT, OUT, 0, -1, -1, -1, -1, 0, _f32_bits(1.0), _f32_bits(1.0), _f32_bits(1.0), 0, 0, 0, *IDENTITY_FIX, 0]


def draw_train_view(index: int, jitter=(0.2, 0.2, 0.2, 0.1), degrees: float = 90.0) -> List[int]:
    """The draws of ``transforms.train_transform()`` (src/main.py:417-425) in its order: RandomHorizontalFlip and
    RandomVerticalFlip (``random``), RandomRotation's angle, ColorJitter's randperm(4) and four factors (torch)."""
    hflip = 1 if random.random() < 0.5 else 0
    vflip = 1 if random.random() < 0.5 else 0
    angle = float(torch.empty(1).uniform_(-degrees, degrees).item())
    ops = torch.randperm(4).tolist()
    draw = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi).item())
    b = draw(max(0.0, 1.0 - jitter[0]), 1.0 + jitter[0])
    c = draw(max(0.0, 1.0 - jitter[1]), 1.0 + jitter[1])
    s = draw(max(0.0, 1.0 - jitter[2]), 1.0 + jitter[2])
    hue = int(draw(-jitter[3], jitter[3]) * 255) & 0xFF
    return [index, 0, 0, OUT, OUT, hflip, *ops, 0, _f32_bits(b), _f32_bits(c), _f32_bits(s), hue, 0, vflip,
            *pil_rotate_fixed(angle, OUT, OUT), 0]


def draw_train_batch(indices: Sequence[int], augment_mask: Sequence[bool], rng: np.random.Generator, jitter=(0.2, 0.2, 0.2, 0.1),
                     degrees: float = 90.0) -> np.ndarray:
    """``draw_train_view``'s distributions for a batch (numpy generator); rows where ``augment_mask`` is False are the
    identity (normal patches take the eval transform, src/main.py:440-444)."""
    n = len(indices)
    out = np.array([identity_view(int(i)) for i in indices], np.int32).reshape(n, PARAMS)
    m = np.asarray(augment_mask, bool)
    k = int(m.sum())
    if k == 0:
        return out
    rows = np.where(m)[0]
    out[rows, 5] = rng.random(k) < 0.5
    out[rows, 16] = rng.random(k) < 0.5
    out[rows, 6:10] = np.argsort(rng.random((k, 4)), axis=1)
    for col, j in ((11, 0), (12, 1), (13, 2)):
        out[rows, col] = rng.uniform(max(0.0, 1.0 - jitter[j]), 1.0 + jitter[j], k).astype(np.float32).view(np.int32)
    out[rows, 14] = (rng.uniform(-jitter[3], jitter[3], k) * 255).astype(np.int64) & 0xFF
    for r, a in zip(rows, rng.uniform(-degrees, degrees, k)):
        out[r, 17:23] = pil_rotate_fixed(float(a), OUT, OUT)
    return out


def draw_simclr_batch(indices: Sequence[int], P: int, rng: np.random.Generator, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0),
                      jitter=(0.4, 0.4, 0.4, 0.1), p_jitter: float = 0.8, p_gray: float = 0.2, p_flip: float = 0.5) -> np.ndarray:
    """The same distributions as ``draw_simclr_view`` for a whole batch at once (numpy generator; per view the Python
    version costs ~0.1 ms of host time, i.e. more than the training step at 2 x 1024 views).  int32 [len(indices), 16]."""
    n = len(indices)
    out = np.zeros((n, PARAMS), np.int32)
    out[:, 0] = np.asarray(indices, np.int32)
    area = float(P) * P
    # RandomResizedCrop.get_params: the first of 10 attempts that fits, else the whole (square) patch
    target = area * rng.uniform(scale[0], scale[1], (n, 10))
    ar = np.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1]), (n, 10)))
    w = np.rint(np.sqrt(target * ar)).astype(np.int64)
    h = np.rint(np.sqrt(target / ar)).astype(np.int64)
    ok = (w > 0) & (w <= P) & (h > 0) & (h <= P)
    first = np.where(ok.any(1), ok.argmax(1), 0)
    rows = np.arange(n)
    hh = np.where(ok.any(1), h[rows, first], P)
    ww = np.where(ok.any(1), w[rows, first], P)
    top = np.floor(rng.random(n) * (P - hh + 1)).astype(np.int64)
    left = np.floor(rng.random(n) * (P - ww + 1)).astype(np.int64)
    top, left = np.where(ok.any(1), top, 0), np.where(ok.any(1), left, 0)
    out[:, 1], out[:, 2], out[:, 3], out[:, 4] = top, left, hh, ww
    out[:, 5] = rng.random(n) < p_flip
    apply = rng.random(n) <= p_jitter
    perm = np.argsort(rng.random((n, 4)), axis=1).astype(np.int32)
    out[:, 6:10] = np.where(apply[:, None], perm, -1)
    fac = lambda j: np.where(apply, rng.uniform(max(0.0, 1.0 - jitter[j]), 1.0 + jitter[j], n), 1.0).astype(np.float32)
    out[:, 11], out[:, 12], out[:, 13] = fac(0).view(np.int32), fac(1).view(np.int32), fac(2).view(np.int32)
    hue = (rng.uniform(-jitter[3], jitter[3], n) * 255).astype(np.int64) & 0xFF  # int() truncates toward zero, as astype does
    out[:, 14] = np.where(apply, hue, 0)
    out[:, 10] = rng.random(n) < p_gray
    out[:, 17:23] = IDENTITY_FIX
    return out


class DevicePatchPool:
    """uint8 [N, P, P, 3] patches in HBM + the resampling tables of every crop size 1..P -> 224."""

    def __init__(self, patches: torch.Tensor, labels: Optional[Sequence[int]] = None):
        if patches.dtype != torch.uint8 or patches.dim() != 4 or patches.shape[3] != 3 or patches.shape[1] != patches.shape[2]:
            raise capi.HipacError("DevicePatchPool: patches must be uint8 [N, P, P, 3]")
        if not patches.is_cuda:
            raise capi.HipacError("DevicePatchPool: the pool lives in HBM (no CPU fallback)")
        self.patches = patches.contiguous()
        self.n, self.P = int(patches.shape[0]), int(patches.shape[1])
        self.labels = None if labels is None else [int(v) for v in labels]
        self.device = patches.device
        per = [capi.resample_coeffs(size, OUT) for size in range(1, self.P + 1)]
        self.ksize = max(k for _, _, k in per)
        tb = np.zeros((self.P, OUT, 2), np.int32)
        tk = np.zeros((self.P, OUT, self.ksize), np.int32)
        for i, (bounds, kk, k) in enumerate(per):
            tb[i], tk[i, :, :k] = bounds, kk
        self.tab_bounds = torch.from_numpy(tb).to(self.device)
        self.tab_kk = torch.from_numpy(tk).to(self.device)
        self.lut = torch.from_numpy(capi.normalize_lut()).to(self.device)
        self._scratch = {}

    def __len__(self):
        return self.n

    @classmethod
    def from_patch_dataset(cls, base, device="cuda", workers: int = 16) -> "DevicePatchPool":
        """Decode every PNG of a ``PatchDataset(transform=None)`` once (host threads) and keep the pixels in HBM."""
        from PIL import Image

        if hasattr(base, "indices") and hasattr(base, "dataset"):  # torch.utils.data.Subset (the balanced validation set)
            paths = [base.dataset.image_paths[i] for i in base.indices]
            labels = [base.dataset.labels[i] for i in base.indices]
        else:
            paths, labels = list(base.image_paths), list(getattr(base, "labels", []))

        def load(p):
            with Image.open(p) as im:
                return np.asarray(im.convert("RGB"), dtype=np.uint8)

        with ThreadPoolExecutor(max(1, workers)) as ex:
            arrs = list(ex.map(load, paths))
        if not arrs:
            raise capi.HipacError("DevicePatchPool: the dataset holds no patches")
        shp = arrs[0].shape
        if shp[0] != shp[1] or any(a.shape != shp for a in arrs):
            raise capi.HipacError("DevicePatchPool: patches of one pool must share one square size (one level directory)")
        dev = torch.device(device)
        pool = torch.empty((len(arrs),) + shp, dtype=torch.uint8, device=dev)
        step = max(1, (256 << 20) // (shp[0] * shp[1] * 3))
        for i in range(0, len(arrs), step):
            pool[i:i + step] = torch.from_numpy(np.stack(arrs[i:i + step])).to(dev)
        out = cls(pool, labels=labels or None)
        out.paths = paths
        return out

    @classmethod
    def from_slides(cls, slides, level: int = 3, stride: Optional[int] = None) -> "DevicePatchPool":
        """The kept windows of ``level`` of every slide, straight from the pyramids in HBM: what ``--patch`` would write as
        PNGs (src/main.py:722-726) and ``PatchDataset`` read back -- PNG is lossless, so at level 3 (224-pixel windows) the
        pool holds exactly those pixels, without the disk round trip; at other levels it holds the windows ALREADY resized
        to 224 (the PNG tree keeps them at P x P).  ``labels``: 1 tumour / 0 normal from the slide's mask; ``meta``: int32
        [N, 4] (level, x, y, label); ``slide_names``: per patch."""
        from . import extract

        parts, labels, names, metas = [], [], [], []
        for sl in slides:
            lw = extract.LevelWindows(sl, level, stride)
            k = lw.kept_index()
            if k.numel() == 0:
                continue
            parts.append(lw.patches(k))
            m = lw.meta(k)
            metas.append(m.cpu())
            labels += m[:, 3].cpu().tolist()
            names += [sl.name] * int(k.numel())
        if not parts:
            raise capi.HipacError("DevicePatchPool.from_slides: no window passed the whiteness filter")
        out = cls(torch.cat(parts), labels=labels)
        out.slide_names, out.meta = names, torch.cat(metas)
        return out

    def _buffers(self, n_views: int):
        key = n_views
        if key not in self._scratch:
            dev = self.device
            self._scratch = {key: (torch.empty((n_views, PARAMS), dtype=torch.int32, device=dev),
                                   torch.empty((n_views, self.P, OUT, 3), dtype=torch.uint8, device=dev),
                                   torch.empty((n_views, OUT, OUT, 3), dtype=torch.uint8, device=dev))}
        return self._scratch[key]

    def augment(self, params: np.ndarray, want_u8: bool = False, want_float: bool = True, geometry: int = 0):
        """params int32 [n_views, 24] (host) -> float32 [n_views, 3, 224, 224] on the device (and / or the uint8
        [n_views, 224, 224, 3] image before ToTensor).  ``geometry``: 0 = resized crop (SimCLR), 1 = flips + rotation
        (the classifier loops' transform; 224-pixel patches)."""
        params = np.ascontiguousarray(params, dtype=np.int32)
        if params.ndim != 2 or params.shape[1] != PARAMS:
            raise capi.HipacError("augment: params must be int32 [n_views, 24]")
        n = params.shape[0]
        pdev, tmp, crops = self._buffers(n)
        # the parameters travel through a ring of PINNED host buffers: an asynchronous copy from pageable memory would make the
        # host wait for everything queued before it (the previous training step), and the pinned buffer must stay untouched
        # until its copy has run -- an event per slot says when
        slot = self._pin_next = (getattr(self, "_pin_next", -1) + 1) % 4
        ring = self.__dict__.setdefault("_pin", [None] * 4)
        evs = self.__dict__.setdefault("_pin_ev", [None] * 4)
        if evs[slot] is not None:
            evs[slot].synchronize()  # the copy that last read this slot has run (before the slot is rewritten or replaced)
        if ring[slot] is None or ring[slot].shape[0] < n:
            ring[slot] = torch.empty((n, PARAMS), dtype=torch.int32).pin_memory()
        host = ring[slot][:n]
        host.numpy()[:] = params
        out = torch.empty((n, 3, OUT, OUT), dtype=torch.float32, device=self.device) if want_float else None
        out_u8 = torch.empty((n, OUT, OUT, 3), dtype=torch.uint8, device=self.device) if want_u8 else None
        lib = capi.load_library()
        with torch.cuda.device(self.device):
            capi._check(lib.hipac_augment_views(self.patches.data_ptr(), self.n, self.P, int(geometry), host.data_ptr(), pdev.data_ptr(), n,
                                                self.tab_bounds.data_ptr(), self.tab_kk.data_ptr(), self.ksize, self.lut.data_ptr(),
                                                tmp.data_ptr(), crops.data_ptr(), out.data_ptr() if out is not None else None,
                                                out_u8.data_ptr() if out_u8 is not None else None, capi._stream()),
                        "hipac_augment_views")
            evs[slot] = torch.cuda.Event()
            evs[slot].record(torch.cuda.current_stream(self.device))
        return (out, out_u8) if want_u8 else out


class DeviceSimCLRLoader:
    """Batches of (x_i, x_j) view pairs, float32 [B, 3, 224, 224] each, made on the device: what
    ``DataLoader(SimCLRDataset(base, get_simclr_transform()), batch_size, shuffle=True)`` yields
    (src/models/simclr.py:70-73).  ``indices_of(epoch)`` may be overridden by a sampler (rank shares)."""

    def __init__(self, pool: DevicePatchPool, batch_size: int, shuffle: bool = True, seed: int = 0, rank: int = 0, world: int = 1,
                 drop_last: bool = False, mirror_host_draws: bool = False):
        """``mirror_host_draws``: take every view's parameters with ``draw_simclr_view`` from torch's / random's global
        generators (the host transforms' own draw order; tests), instead of the batched numpy draws."""
        self.pool, self.batch_size, self.shuffle, self.seed = pool, int(batch_size), shuffle, seed
        self.rank, self.world, self.drop_last, self.epoch = rank, world, drop_last, 0
        self.mirror_host_draws = mirror_host_draws
        self._rng = np.random.default_rng([seed, rank])

    def __len__(self):
        n = len(self.pool)
        return n // self.batch_size if self.drop_last else math.ceil(n / self.batch_size)

    def _batches(self) -> Iterator[List[int]]:
        n = len(self.pool)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = list(range(n))
        for i in range(0, n, self.batch_size):
            b = order[i:i + self.batch_size]
            if len(b) < self.batch_size and self.drop_last:
                return
            if self.world > 1:  # this rank's contiguous share of the global batch, equal on all ranks (dist.RankBatchSampler)
                per = len(b) // self.world
                b = b[self.rank * per:(self.rank + 1) * per]
            if b:
                yield b

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        P = self.pool.P
        for b in self._batches():
            # view i then view j of every sample, as SimCLRDataset.__getitem__ draws them (src/datasets/simclr_dataset.py:10-11)
            rows = np.empty((2, len(b), PARAMS), np.int32)
            if self.mirror_host_draws:
                for k, idx in enumerate(b):
                    rows[0, k] = draw_simclr_view(idx, P, P)
                    rows[1, k] = draw_simclr_view(idx, P, P)
            else:
                rows[0], rows[1] = draw_simclr_batch(b, P, self._rng), draw_simclr_batch(b, P, self._rng)
            x = self.pool.augment(rows.reshape(-1, PARAMS))
            yield x[:len(b)], x[len(b):]
        self.epoch += 1


class DeviceClassifierLoader:
    """Batches of (images float32 [B, 3, 224, 224] on the device, labels int64 [B] on the host, indices) from a pool of
    224-pixel patches: what the classifier loops' DataLoader yields (src/main.py:440-452) -- tumour patches through
    ``train_transform``, normal patches through the eval transform; ``augment=False``: every patch through the eval
    transform (validation)."""

    def __init__(self, pool: DevicePatchPool, batch_size: int, shuffle: bool = True, augment: bool = True, seed: int = 0, rank: int = 0,
                 world: int = 1, indices: Optional[Sequence[int]] = None):
        if pool.P != OUT:
            raise capi.HipacError("DeviceClassifierLoader: the flips / rotation / jitter act on 224-pixel patches (level 3); "
                                  "other levels keep the host transforms")
        if pool.labels is None:
            raise capi.HipacError("DeviceClassifierLoader: the pool has no labels")
        self.pool, self.batch_size, self.shuffle, self.augment = pool, int(batch_size), shuffle, augment
        self.seed, self.rank, self.world, self.epoch = seed, rank, world, 0
        self.indices = list(range(len(pool))) if indices is None else [int(i) for i in indices]
        self._rng = np.random.default_rng([seed, rank, 7])

    def __len__(self):
        return math.ceil(len(self.indices) / self.batch_size)

    def __iter__(self):
        n = len(self.indices)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = [self.indices[i] for i in torch.randperm(n, generator=g).tolist()]
        else:
            order = list(self.indices)
        for i in range(0, n, self.batch_size):
            b = order[i:i + self.batch_size]
            if self.world > 1:
                if self.shuffle:  # training: equal contiguous shares of the global batch (dist.RankBatchSampler)
                    per = len(b) // self.world
                    b = b[self.rank * per:(self.rank + 1) * per]
                else:  # validation: every sample once
                    from .dist import shard_columns

                    i0, i1 = shard_columns(len(b), self.rank, self.world)
                    b = b[i0:i1]
            if not b:
                continue
            labels = [self.pool.labels[j] for j in b]
            rows = draw_train_batch(b, [self.augment and lab == 1 for lab in labels], self._rng)
            yield self.pool.augment(rows, geometry=1), torch.tensor(labels, dtype=torch.int64), b
        self.epoch += 1


def bench_input_pipeline(n_pairs: int = 256, P: int = 224, steps: int = 5, host_views: int = 512, device="cuda") -> dict:
    """bench.py's ``simclr.input_pipeline``: view pairs per second of the device pipeline (batched draws + hipac_augment_views,
    wall clock around ``steps`` batches) next to the host transforms (Pillow, one thread, ``host_views`` views)."""
    import time

    from PIL import Image

    from . import transforms

    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(5)
    pool = DevicePatchPool(torch.randint(0, 256, (max(n_pairs, 8), P, P, 3), generator=g, device=dev, dtype=torch.uint8))
    loader = DeviceSimCLRLoader(pool, n_pairs, seed=1)
    for _ in loader:  # warm-up epoch: tables, scratch, pinned ring
        break
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        for x_i, x_j in loader:
            pass
    torch.cuda.synchronize(dev)
    n_batches = steps * len(loader)
    dt = time.perf_counter() - t0
    rng = np.random.default_rng(0)
    t1 = time.perf_counter()
    for _ in range(20):
        draw_simclr_batch(list(range(n_pairs)), P, rng)
    t_draw = (time.perf_counter() - t1) / 20
    T = transforms.simclr_transform()
    # >= 512 views over 16 different patches (round 3 timed 48 views of ONE patch: the random crop sizes of so few draws
    # made the figure swing 20x between runs), with a short warm-up so that Pillow's tables and the allocator are hot
    imgs = [Image.fromarray(pool.patches[i % len(pool)].cpu().numpy(), "RGB") for i in range(16)]
    for k in range(16):
        T(imgs[k])
    t2 = time.perf_counter()
    for k in range(host_views):
        T(imgs[k % 16])
    t_host = (time.perf_counter() - t2) / host_views
    return {"patch": P, "pairs_per_batch": n_pairs, "device_pairs_per_s": n_batches * min(n_pairs, len(pool)) / dt,
            "host_draws_ms_per_batch": t_draw * 1e3,
            "host_pillow_pairs_per_s_one_thread": 0.5 / t_host, "host_views_timed": host_views, "host_patches": 16,
            "note": "device: batched numpy draws + hipac_augment_views (crop/resize, flip, ColorJitter, grayscale, normalise; "
                    "bit-exact vs Pillow); host: transforms.simclr_transform() = the reference's torchvision pipeline on Pillow, "
                    "PNG decoding not included on either side"}
