"""Sliding-window patch extraction over a device-resident slide pyramid.

Mirrors ``extract_patches`` (src/main.py:609-732) and ``parse_xml_mask``
(src/main.py:372-410): same level -> window-size table, the same (quirky) stride,
x-outer / y-inner order, border padding with white, whiteness filter, tumour label
from the annotation mask, level-L pixel coordinates in the names.  The per-window
pixel work (crop, pad, whiteness sum, resize to 224, normalise) runs in the HIP
kernel ``hipac_tile_preprocess`` on the slide as it sits in HBM; only the loop
bounds and the polygon rasterisation (Pillow) stay on the host.
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import capi, synth, trace

PATCH_SIZES = {0: 1792, 1: 896, 2: 448, 3: 224}  # src/main.py:614
LABEL_NAMES = {0: "normal", 1: "tumor"}  # src/datasets/patch_dataset.py:15


def window_grid(width: int, height: int, level: int, stride: Optional[int] = None, pad: bool = True):
    """Loop bounds of src/main.py:611-615, :658-665, :682-691.  Returns
    (patch_size, stride, xy int32[n,2]) with windows in the reference's visiting order
    (x outer, y inner).  ``stride=None`` keeps the reference's behaviour: the default
    ``patch_size=224`` is what ``stride = stride or patch_size`` sees (:611), so the
    stride is 224 at every level; pass ``stride=PATCH_SIZES[level]`` for the
    non-overlapping grid the README describes."""
    patch_size = 224
    stride = stride or patch_size
    patch_size = PATCH_SIZES.get(level, 224)
    pad_w = (patch_size - width % patch_size) % patch_size if pad else 0
    pad_h = (patch_size - height % patch_size) % patch_size if pad else 0
    xs = np.arange(0, width + pad_w, stride, dtype=np.int64)
    ys = np.arange(0, height + pad_h, stride, dtype=np.int64)
    xs, ys = xs[xs < width], ys[ys < height]
    xy = np.stack([np.repeat(xs, len(ys)), np.tile(ys, len(xs))], axis=1).astype(np.int32)
    return patch_size, stride, xy


def parse_annotation_xml(xml_path: str) -> List[List[Tuple[float, float]]]:
    """The XML walk of src/main.py:395-407 (ASAP format: every
    ``Annotation/Coordinates`` node, ``Coordinate`` children with X / Y attributes in
    level-0 pixels) with the standard library instead of lxml."""
    tree = ET.parse(xml_path)
    polys = []
    for ann in tree.getroot().iter("Annotation"):
        node = ann.find("Coordinates")
        if node is None:
            continue
        pts = []
        for c in node.findall("Coordinate"):
            try:
                pts.append((float(c.get("X")), float(c.get("Y"))))
            except (TypeError, ValueError):
                continue
        if pts:
            polys.append(pts)
    return polys


def rasterize_mask(polygons_l0, level_dims: Tuple[int, int], base_dims: Tuple[int, int]) -> np.ndarray:
    """src/main.py:387-410: scale level-0 polygon vertices by level/base dims, truncate
    with int(), fill + outline with 255 on an 'L' image of the level's size.
    Returns uint8[H,W]."""
    from PIL import Image, ImageDraw

    sx, sy = level_dims[0] / base_dims[0], level_dims[1] / base_dims[1]
    mask = Image.new("L", level_dims, 0)
    draw = ImageDraw.Draw(mask)
    for poly in polygons_l0:
        pts = [(int(float(x) * sx), int(float(y) * sy)) for x, y in poly]
        if pts:
            draw.polygon(pts, outline=255, fill=255)
    return np.array(mask)


def rasterize_mask_bands(polygons_l0, level_dims: Tuple[int, int], base_dims: Tuple[int, int], band: int = 2048, margin: int = 2):
    """The same mask as ``rasterize_mask``, produced ONLY where a polygon reaches: yields (first row, first column,
    uint8[rows, cols]) pieces -- every other byte of the mask is zero.  The rows that polygons touch are cut into bands of at
    most ``band`` rows; a band is drawn on an image of its own height (+ ``margin`` rows above and below, discarded) with the
    polygons translated in y: Pillow's scanline fill computes its crossings as (y - y0) * dx + x0, which a translation in
    y leaves bit for bit, and the margin keeps the band's own clipping away from the rows that are kept.  In x nothing is
    translated (x0 enters a float32 sum): the canvas starts at column 0 and merely ends behind the band's rightmost vertex;
    only the columns between the leftmost and the rightmost vertex are returned.  tests/test_host_logic.py compares with
    the full-size raster (vertices on band boundaries included).  At level 0 of a 100 000^2 slide the full raster is a
    10 GB host image (allocate, fill, convert, upload: ~11 s)."""
    from PIL import Image, ImageDraw

    W, H = level_dims
    sx, sy = level_dims[0] / base_dims[0], level_dims[1] / base_dims[1]
    polys = []
    for poly in polygons_l0:
        pts = [(int(float(x) * sx), int(float(y) * sy)) for x, y in poly]
        if pts:
            xs, ys = [x for x, _ in pts], [y for _, y in pts]
            polys.append((pts, min(ys), max(ys), min(xs), max(xs)))
    # rows any polygon touches, merged into runs, runs cut into bands
    spans = sorted((max(0, lo), min(H, hi + 1)) for _, lo, hi, _, _ in polys if hi >= 0 and lo < H)
    runs = []
    for lo, hi in spans:
        if runs and lo <= runs[-1][1]:
            runs[-1][1] = max(runs[-1][1], hi)
        else:
            runs.append([lo, hi])
    for r0, r1 in runs:
        for b0 in range(r0, r1, band):
            b1 = min(r1, b0 + band)
            t0, t1 = b0 - margin, b1 + margin
            sel = [(pts, x0, x1) for pts, lo, hi, x0, x1 in polys if hi >= t0 and lo < t1]
            if not sel:
                continue
            c0 = max(0, min(x0 for _, x0, _ in sel))
            c1 = min(W, max(x1 for _, _, x1 in sel) + 1)
            if c1 <= c0:
                continue  # entirely left or right of the level
            img = Image.new("L", (min(W, c1 + 1), t1 - t0), 0)
            draw = ImageDraw.Draw(img)
            for pts, _, _ in sel:
                draw.polygon([(x, y - t0) for x, y in pts], outline=255, fill=255)
            yield b0, c0, np.asarray(img)[margin:margin + (b1 - b0), c0:c1]


class DeviceSlide:
    """A slide pyramid resident in HBM: per level a uint8[H, Wpad, 3] tensor whose row
    pitch is a multiple of 48 bytes (16 pixels) so the preprocess kernel can use
    16-byte loads.  Exposes the two openslide attributes the reference reads
    (``level_dimensions``, ``level_downsamples``, src/main.py:654-655)."""

    def __init__(self, levels: Sequence[torch.Tensor], device: torch.device | str = "cuda", name: str = "slide"):
        dev = torch.device(device)
        self.name = name
        self.level_dimensions = tuple((int(l.shape[1]), int(l.shape[0])) for l in levels)
        self.level_downsamples = tuple(self.level_dimensions[0][0] / d[0] for d in self.level_dimensions)
        self.levels: List[torch.Tensor] = []
        for l in levels:
            if l.dtype != torch.uint8 or l.dim() != 3 or l.shape[2] != 3:
                raise capi.HipacError("levels must be uint8[H,W,3]")
            h, w = int(l.shape[0]), int(l.shape[1])
            wp = (w + 15) // 16 * 16
            buf = torch.empty((h, wp, 3), dtype=torch.uint8, device=dev)
            buf[:, :w] = l.to(dev)
            if wp > w:
                buf[:, w:] = 0
            self.levels.append(buf)
        self.device = self.levels[0].device if self.levels else dev  # the indexed device ("cuda" alone != "cuda:0")
        self.polygons: Optional[list] = None
        self._masks: Dict[int, torch.Tensor] = {}

    @classmethod
    def synthetic(cls, width: int, height: int, seed: int = 2, n_levels: int = 4, device="cuda",
                  with_polygons: bool = True, name: Optional[str] = None):
        l0 = synth.synth_level0(width, height, seed=seed, device=device)
        s = cls(synth.build_pyramid(l0, n_levels), device=device, name=name or f"synthetic_{seed}")
        if with_polygons:
            s.polygons = synth.synth_polygons(width, height, seed=seed)
        return s

    @classmethod
    def from_tiff(cls, path: str, device="cuda", n_levels: int = 4, name: Optional[str] = None, workers: int = 16):
        """Open a tiled pyramidal TIFF / BigTIFF (the CAMELYON16 container; ``openslide.OpenSlide(path)`` in
        the reference, src/main.py:650) and bring its first ``n_levels`` levels into HBM
        (``tiff_pyramid.TiffPyramid``: tiles decoded on host threads, copied band by band)."""
        from .tiff_pyramid import TiffPyramid

        tp = TiffPyramid(path)
        dev = torch.device(device)
        s = cls.__new__(cls)
        s.name = name or os.path.splitext(os.path.basename(path))[0]
        use = list(range(min(n_levels, tp.level_count)))
        s.level_dimensions = tuple(tp.level_dimensions[i] for i in use)
        s.level_downsamples = tuple(tp.level_downsamples[i] for i in use)
        s.levels = [t for t, _ in tp.to_device_levels(dev, use, workers)]
        s.device = dev
        s.polygons = None
        s._masks = {}
        return s

    def mask(self, level: int) -> Optional[torch.Tensor]:
        if self.polygons is None:
            return None
        if level not in self._masks:
            w, h = self.level_dimensions[level]
            wp = (w + 15) // 16 * 16  # zero-padded row pitch: 16-byte loads in hipac_mask_cells
            buf = torch.zeros((h, wp), dtype=torch.uint8, device=self.device)
            # Pillow's raster (the reference's, src/main.py:387-410) band by band, only where a polygon reaches
            for y0, x0, piece in rasterize_mask_bands(self.polygons, (w, h), self.level_dimensions[0]):
                buf[y0:y0 + piece.shape[0], x0:x0 + piece.shape[1]] = torch.from_numpy(np.ascontiguousarray(piece)).to(self.device)
            self._masks[level] = buf
        return self._masks[level]


@dataclass
class LevelScan:
    """Everything the extractor decides for one level, on device, in visiting order."""
    level: int
    patch_size: int
    xy: torch.Tensor      # int32[n,2]
    sums: torch.Tensor    # int32[n] (uint32 bit pattern)
    keep: torch.Tensor    # uint8[n]
    labels: torch.Tensor  # uint8[n]

    def names(self, prefix: str) -> List[str]:
        """File names the reference would have written (src/main.py:722), kept windows only."""
        xy = self.xy.cpu().numpy()
        keep = self.keep.cpu().numpy().astype(bool)
        lab = self.labels.cpu().numpy()
        return [f"{prefix}_x{x}_y{y}_{LABEL_NAMES[int(l)]}.png" for (x, y), k, l in zip(xy, keep, lab) if k]


def iter_level(slide: DeviceSlide, level: int, out_format: str = "bf16", batch_windows: int = 512,
               stride: Optional[int] = None, pad: bool = True, kept_only: bool = True,
               use_planes: Optional[bool] = None) -> Iterator[dict]:
    """Stream one level: for each batch of windows run the preprocess kernel and yield
    {"x": network input (kept windows only when ``kept_only``), "xy", "sums", "keep",
    "labels"} -- all device tensors.

    ``use_planes`` (default: whenever it applies, i.e. uint8 output, P > 224 and a stride that is
    a multiple of 224) resamples the level once (``capi.LevelPlanes``) and gathers windows from
    it instead of resampling every overlapping window separately; results are identical."""
    width, height = slide.level_dimensions[level]
    P, eff_stride, xy_np = window_grid(width, height, level, stride, pad)
    img = slide.levels[level]
    mask = slide.mask(level)
    can_planes = out_format == "u8" and P in (448, 896, 1792) and eff_stride % 224 == 0
    if use_planes is None:
        use_planes = can_planes
    if use_planes and not can_planes:
        raise capi.HipacError("planes path needs uint8 output, P in (448, 896, 1792) and a stride multiple of 224")
    if use_planes:
        planes = capi.LevelPlanes(img, P, width=width)
        for i0 in range(0, len(xy_np), batch_windows):
            xy = torch.from_numpy(xy_np[i0 : i0 + batch_windows]).to(slide.device)
            sums, keep = planes.stats(xy)
            labels = (capi.window_labels(mask, xy, P) if mask is not None
                      else torch.zeros((xy.shape[0],), dtype=torch.uint8, device=slide.device))
            sel = xy.index_select(0, torch.nonzero(keep, as_tuple=False).flatten()) if kept_only else xy
            yield {"x": planes.gather(sel), "xy": xy, "sums": sums, "keep": keep, "labels": labels}
        return
    for i0 in range(0, len(xy_np), batch_windows):
        xy = torch.from_numpy(xy_np[i0 : i0 + batch_windows]).to(slide.device)
        out, sums, keep = capi.tile_preprocess(img, xy, P, out_format, width=width)
        if mask is not None:
            labels = capi.window_labels(mask, xy, P)
        else:
            labels = torch.zeros((xy.shape[0],), dtype=torch.uint8, device=slide.device)
        if kept_only:
            idx = torch.nonzero(keep, as_tuple=False).flatten()
            out = out.index_select(0, idx)
        yield {"x": out, "xy": xy, "sums": sums, "keep": keep, "labels": labels}


def scan_level(slide: DeviceSlide, level: int, stride: Optional[int] = None, pad: bool = True, **_ignored) -> LevelScan:
    """Decisions only (no pixels kept): the device-side equivalent of one
    ``extract_patches(level=...)`` pass over one slide."""
    lw = LevelWindows(slide, level, stride=stride, pad=pad)
    return LevelScan(level, lw.P, lw.xy, lw.sums, lw.keep, lw.labels)


class LevelWindows:
    """All extractor decisions of one level at once (device tensors, reference visiting
    order): window origins, whiteness sums, keep flags, tumour labels -- plus the resized
    uint8 pixels of any subset of windows on demand."""

    CHUNK = 8192  # windows per launch of the per-window kernel (non-lattice grids)

    def __init__(self, slide: DeviceSlide, level: int, stride: Optional[int] = None, pad: bool = True,
                 use_planes: Optional[bool] = None):
        self.level = level
        width, height = slide.level_dimensions[level]
        self.P, eff_stride, xy_np = window_grid(width, height, level, stride, pad)
        self.width, self.height = width, height
        self.img = slide.levels[level]
        self.xy = torch.from_numpy(xy_np).to(slide.device)
        on_lattice = eff_stride % 224 == 0
        can_planes = self.P in (448, 896, 1792) and on_lattice
        self.planes = capi.LevelPlanes(self.img, self.P, width=width) if (can_planes and use_planes is not False) else None
        if use_planes and not can_planes:
            raise capi.HipacError("planes path needs P in (448, 896, 1792) and a stride multiple of 224")
        self._kept_u8, self._kept_pos = None, None
        if self.planes is not None:
            self.sums, self.keep = self.planes.stats(self.xy)
        else:
            # any other window list: per-window kernel, in chunks, keeping only the pixels of kept windows (a level
            # scanned with a small stride has hundreds of thousands of windows: 150 KB each for all of them, dropped
            # ones included, would be tens of GB)
            n = self.xy.shape[0]
            sums, keeps, pix = [], [], []
            for i0 in range(0, n, self.CHUNK):
                o, sm, kp = capi.tile_preprocess(self.img, self.xy[i0:i0 + self.CHUNK].contiguous(), self.P, "u8", width=width)
                sums.append(sm), keeps.append(kp)
                pix.append(o.index_select(0, torch.nonzero(kp, as_tuple=False).flatten()))
            if n:
                self.sums, self.keep = torch.cat(sums), torch.cat(keeps)
                self._kept_u8 = torch.cat(pix)
            else:
                self.sums = torch.empty((0,), dtype=torch.int32, device=slide.device)
                self.keep = torch.empty((0,), dtype=torch.uint8, device=slide.device)
                self._kept_u8 = torch.empty((0, 224, 224, 3), dtype=torch.uint8, device=slide.device)
            self._kept_pos = torch.cumsum(self.keep.to(torch.int64), 0) - 1  # window index -> row of _kept_u8
        mask = slide.mask(level)
        if mask is None:
            self.labels = torch.zeros((self.xy.shape[0],), dtype=torch.uint8, device=slide.device)
        elif on_lattice:
            self.labels = capi.window_labels_cells(capi.mask_cells(mask, width=width), width, height, self.xy, self.P)
        else:
            self.labels = capi.window_labels(mask, self.xy, self.P)

    def kept_index(self) -> torch.Tensor:
        return torch.nonzero(self.keep, as_tuple=False).flatten()

    def patches(self, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """uint8[len(idx),224,224,3] resized pixels of the selected KEPT windows (into ``out`` if given)."""
        if self._kept_u8 is not None:
            rows = self._kept_pos.index_select(0, idx)
            if out is None:
                return self._kept_u8.index_select(0, rows)
            torch.index_select(self._kept_u8, 0, rows, out=out)
            return out
        return self.planes.gather(self.xy.index_select(0, idx), out=out)

    def meta(self, idx: torch.Tensor) -> torch.Tensor:
        xy = self.xy.index_select(0, idx)
        lab = self.labels.index_select(0, idx).to(torch.int32)
        return torch.cat([torch.full_like(lab, self.level)[:, None], xy, lab[:, None]], dim=1)


class WSIPatchStream:
    """Iterable over (x, meta) device batches of KEPT windows for the fused path:
    x = the resized uint8[B,224,224,3] pixels (the network's stem applies
    ToTensor/Normalize itself), meta int32[B,4] = (level, x, y, label)."""

    def __init__(self, slide: DeviceSlide, levels: Sequence[int] = (0, 1, 2, 3), batch_windows: int = 4096,
                 precision: str = "bf16", stride: Optional[int] = None):
        self.slide, self.levels, self.batch_windows = slide, tuple(levels), batch_windows
        self.precision, self.stride = precision, stride

    def __iter__(self):
        for level in self.levels:
            stride = self.stride(level) if callable(self.stride) else self.stride
            lw = LevelWindows(self.slide, level, stride)
            kept = lw.kept_index()
            for i0 in range(0, kept.shape[0], self.batch_windows):
                idx = kept[i0 : i0 + self.batch_windows]
                yield lw.patches(idx), lw.meta(idx)


@torch.no_grad()
def score_slide(slide: DeviceSlide, net: capi.PackedResNet18, levels: Sequence[int] = (0, 1, 2, 3),
                batch_windows: int = 4096, stride=None, want_logits: bool = True, fwd_batch: int = 8192):
    """Whole-slide hierarchical scan: windows -> whiteness/labels -> resize -> ResNet18 ->
    per-patch features / logits / labels.  ``stride``: None (reference: 224), an int, or a callable level -> stride.
    Returns device tensors (feats[n,512], logits[n,C] or None, pred int64[n] or None,
    meta int32[n,4] = (level, x, y, label)) in level-major, reference visiting order.

    Three phases on the caller's stream, ONE wait of the host per slide:
      1. the extractor's decisions for every requested level (whole-level kernels: resample planes, whiteness sums,
         keep flags, labels) -- queued back to back, then the host reads the four kept counts;
      2. the kept windows of all levels gathered, level-major, into ONE uint8 batch buffer (cached on ``net``, grow-only:
         no multi-GB allocation per slide);
      3. the ResNet over that buffer in slices of ``fwd_batch`` patches (the late layers need thousands of patches per
         launch to fill the chip), results already in output order.
    Measured on a 50 000^2 slide (tools/wsi_phases.py): 5 + 2.4 + 136 ms -- the scan IS the ResNet; running the level-0
    decisions on a side stream underneath the first forwards (rounds 1-2) bought nothing (143 vs 144 ms: concurrent
    kernels slow the persistent-grid convolutions down by what they hide) and cost a cross-stream lifetime race."""
    has_fc = net.num_classes > 0 and want_logits
    dev = slide.device
    levels = list(levels)
    FWD = max(1, int(fwd_batch))

    def stride_of(level):
        return stride(level) if callable(stride) else stride

    dbg = os.environ.get("HIPAC_SCAN_TIMES") == "1"  # developer aid: synchronise after every phase and print its wall clock
    import time as _time

    def _mark(label, t_prev):
        if not dbg:
            return t_prev
        torch.cuda.synchronize(dev)
        t = _time.perf_counter()
        st = torch.cuda.memory_stats(dev)
        print(f"[score_slide] {label}: {(t - t_prev) * 1e3:.1f} ms  (reserved {st['reserved_bytes.all.current'] / 2**30:.1f} GiB, "
              f"allocated {st['allocated_bytes.all.current'] / 2**30:.1f} GiB, alloc retries {st['num_alloc_retries']}, "
              f"segments {st['segment.all.current']})", flush=True)
        return t

    t_dbg = _mark("start", _time.perf_counter()) if dbg else 0.0
    with trace.span("window decisions, all levels"):
        lws = [LevelWindows(slide, lv, stride_of(lv)) for lv in levels]
        # the host's one wait: the kept counts of all levels in one small copy (the index lists themselves follow at once)
        counts = torch.stack([lw.keep.sum(dtype=torch.int64) for lw in lws]).cpu().tolist() if lws else []
    n_total = int(sum(counts))
    if n_total == 0:
        return (torch.empty((0, 512), device=dev), None, None, torch.empty((0, 4), dtype=torch.int32, device=dev))
    t_dbg = _mark("decisions + counts", t_dbg)
    kepts = [lw.kept_index() for lw in lws]
    metas = torch.cat([lw.meta(k) for lw, k in zip(lws, kepts)])
    buf = net.batch_buffer(n_total, dev)
    t_dbg = _mark("index lists, meta, batch buffer", t_dbg)
    with trace.span("gather kept windows"):
        done = 0
        for lw, k in zip(lws, kepts):
            for i0 in range(0, k.shape[0], max(1, int(batch_windows))):
                idx = k[i0:i0 + batch_windows]
                lw.patches(idx, out=buf[done:done + idx.shape[0]])
                done += idx.shape[0]
    del lws  # the resampled planes (GBs at level 0) go back to the allocator before the forwards allocate
    t_dbg = _mark("gather", t_dbg)
    out_chunks = []
    for lo in range(0, n_total, FWD):
        end = min(n_total, lo + FWD)
        with trace.span(f"resnet18 forward [{lo}:{end})"):
            out_chunks.append(net.forward(buf[lo:end], want_feats=True, want_logits=has_fc, want_labels=has_fc))
    t_dbg = _mark("forwards", t_dbg)
    one = len(out_chunks) == 1
    feats = out_chunks[0][0] if one else torch.cat([c[0] for c in out_chunks])
    logits = (out_chunks[0][1] if one else torch.cat([c[1] for c in out_chunks])) if has_fc else None
    preds = (out_chunks[0][2] if one else torch.cat([c[2] for c in out_chunks])) if has_fc else None
    return feats, logits, preds, metas


def save_patch_pngs(slide: DeviceSlide, level: int, out_dir: str, stride: Optional[int] = None) -> int:
    """Optional reference-compatible output: write every kept window as
    ``<out_dir>/<slide>/<slide>_x{x}_y{y}_{label}.png`` (src/main.py:722-726).  This is
    the slow, host-bound leg (D2H copy + PNG encode) kept for downstream tools; the fused
    path never touches disk (PNG is lossless, so skipping it changes nothing)."""
    from PIL import Image

    scan = scan_level(slide, level, stride=stride)
    P = scan.patch_size
    width, height = slide.level_dimensions[level]
    img = slide.levels[level]
    save_dir = os.path.join(out_dir, slide.name)
    os.makedirs(save_dir, exist_ok=True)
    xy = scan.xy.cpu().numpy()
    keep = scan.keep.cpu().numpy().astype(bool)
    lab = scan.labels.cpu().numpy()
    n = 0
    for (x, y), k, l in zip(xy, keep, lab):
        if not k:
            continue
        path = os.path.join(save_dir, f"{slide.name}_x{x}_y{y}_{LABEL_NAMES[int(l)]}.png")
        if not os.path.exists(path):
            canvas = np.full((P, P, 3), 255, np.uint8)
            pw, ph = min(P, width - x), min(P, height - y)
            canvas[:ph, :pw] = img[y : y + ph, x : x + pw].cpu().numpy()
            Image.fromarray(canvas, "RGB").save(path)
        n += 1
    return n
