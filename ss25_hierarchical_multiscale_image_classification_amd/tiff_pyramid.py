"""Tiled pyramidal TIFF / BigTIFF reader feeding the device pyramid.

The reference opens CAMELYON16 slides with openslide (src/main.py:650-655:
``OpenSlide(path)``, ``level_dimensions``, ``level_downsamples``) and pulls every window
through ``read_region(location, level, size)`` (:693-697).  openslide is not available in
this image, so this module reads the container itself: classic TIFF and BigTIFF, tiled
IFDs, 8-bit RGB, compression none / deflate / LZW-free JPEG (old-style excluded), one
pyramid level per full-resolution or reduced-resolution tiled IFD, largest first
(the "generic tiled TIFF" layout of the CAMELYON16 files).

Tiles are decoded on host threads (Pillow's JPEG / zlib decoders release the GIL) in row
bands and copied band by band into the level's HBM tensor, so the host never holds more
than one band of one level.  Semantics kept from openslide:
  * ``level_dimensions`` / ``level_downsamples`` (downsample = level-0 width / level width),
  * pixels of missing tiles (byte count 0) are transparent black, which the reference's
    ``.convert("RGB")`` turns into (0,0,0) -- here they are written as 0,
  * ``read_region`` (host, RGBA uint8, out-of-bounds = transparent black) for small
    regions and for tests.
"""
from __future__ import annotations

import io
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 7: ("B", 1), 8: ("h", 2),
          9: ("i", 4), 10: ("ii", 8), 11: ("f", 4), 12: ("d", 8), 16: ("Q", 8), 17: ("q", 8), 18: ("Q", 8)}


class TiffError(ValueError):
    pass


@dataclass
class TiffLevel:
    width: int
    height: int
    tile_w: int
    tile_h: int
    compression: int
    photometric: int
    samples: int
    offsets: Sequence[int]
    counts: Sequence[int]
    jpeg_tables: Optional[bytes]
    subfile_type: int

    @property
    def tiles_across(self) -> int:
        return (self.width + self.tile_w - 1) // self.tile_w

    @property
    def tiles_down(self) -> int:
        return (self.height + self.tile_h - 1) // self.tile_h


def _parse_ifds(buf) -> List[dict]:
    bo = {b"II": "<", b"MM": ">"}.get(bytes(buf[:2]))
    if bo is None:
        raise TiffError("not a TIFF file")
    magic = struct.unpack(bo + "H", buf[2:4])[0]
    if magic == 42:
        big, off = False, struct.unpack(bo + "I", buf[4:8])[0]
    elif magic == 43:
        big, off = True, struct.unpack(bo + "Q", buf[8:16])[0]
    else:
        raise TiffError(f"bad TIFF magic {magic}")
    ifds = []
    seen = set()
    while off and off not in seen and len(ifds) < 64:
        seen.add(off)
        if big:
            (n,) = struct.unpack(bo + "Q", buf[off:off + 8])
            pos, esz, cw = off + 8, 20, 8
        else:
            (n,) = struct.unpack(bo + "H", buf[off:off + 2])
            pos, esz, cw = off + 2, 12, 4
        tags = {}
        for i in range(n):
            e = buf[pos + i * esz: pos + (i + 1) * esz]
            tag, typ = struct.unpack(bo + "HH", e[:4])
            (cnt,) = struct.unpack(bo + ("Q" if big else "I"), e[4:4 + cw])
            if typ not in _TYPES:
                continue
            fmt, sz = _TYPES[typ]
            nbytes = cnt * sz
            if nbytes <= cw:
                raw = bytes(e[4 + cw:4 + cw + nbytes])
            else:
                (voff,) = struct.unpack(bo + ("Q" if big else "I"), e[4 + cw:4 + 2 * cw])
                raw = bytes(buf[voff:voff + nbytes])
            if typ in (2, 7):
                tags[tag] = raw
            elif typ in (5, 10):
                vals = struct.unpack(bo + fmt[0] * (2 * cnt), raw)
                tags[tag] = [vals[2 * j] / max(1, vals[2 * j + 1]) for j in range(cnt)]
            else:
                tags[tag] = list(struct.unpack(bo + fmt * cnt, raw))
        ifds.append(tags)
        nxt = buf[pos + n * esz: pos + n * esz + cw]
        (off,) = struct.unpack(bo + ("Q" if big else "I"), nxt)
    return ifds


def _adobe_rgb_marker() -> bytes:
    # APP14 "Adobe" with transform = 0: the three components are RGB, not YCbCr
    return b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x00"


class TiffPyramid:
    """Tiled TIFF pyramid.  ``level_dimensions`` / ``level_downsamples`` as in openslide."""

    def __init__(self, path: str):
        self.path = path
        self._mm = np.memmap(path, dtype=np.uint8, mode="r")
        self._buf = memoryview(self._mm)
        levels = []
        for t in _parse_ifds(self._buf):
            if 322 not in t or 324 not in t:  # not tiled (label / macro / thumbnail strips): not a pyramid level
                continue
            bits = t.get(258, [8])
            spp = t.get(277, [1])[0]
            if any(b != 8 for b in bits) or spp not in (3, 4) or t.get(284, [1])[0] != 1:
                continue
            lv = TiffLevel(width=t[256][0], height=t[257][0], tile_w=t[322][0], tile_h=t[323][0],
                           compression=t.get(259, [1])[0], photometric=t.get(262, [2])[0], samples=spp,
                           offsets=t[324], counts=t[325], jpeg_tables=t.get(347), subfile_type=t.get(254, [0])[0])
            if lv.compression not in (1, 7, 8, 32946):
                raise TiffError(f"unsupported tile compression {lv.compression} (none, JPEG and deflate are read)")
            levels.append(lv)
        if not levels:
            raise TiffError("no tiled 8-bit RGB image directory found")
        levels.sort(key=lambda l: -l.width * l.height)
        self.levels: List[TiffLevel] = levels
        self.level_count = len(levels)
        self.level_dimensions = tuple((l.width, l.height) for l in levels)
        self.dimensions = self.level_dimensions[0]
        self.level_downsamples = tuple(self.dimensions[0] / l.width for l in levels)

    # ---- tiles -------------------------------------------------------------------------
    def _decode_tile(self, lv: TiffLevel, index: int) -> Optional[np.ndarray]:
        """uint8[tile_h, tile_w, 3] or None for a missing tile."""
        off, cnt = lv.offsets[index], lv.counts[index]
        if cnt == 0:
            return None
        raw = bytes(self._buf[off:off + cnt])
        if lv.compression == 1:
            a = np.frombuffer(raw, np.uint8)
        elif lv.compression in (8, 32946):
            a = np.frombuffer(zlib.decompress(raw), np.uint8)
        else:
            from PIL import Image

            data = raw
            if lv.jpeg_tables:
                tb = lv.jpeg_tables
                data = tb[:-2] + raw[2:] if tb[-2:] == b"\xff\xd9" and raw[:2] == b"\xff\xd8" else raw
            if lv.photometric == 2:  # RGB stored in the JPEG: keep Pillow from applying YCbCr -> RGB
                data = data[:2] + _adobe_rgb_marker() + data[2:]
            im = Image.open(io.BytesIO(data))
            im.load()
            a = np.asarray(im.convert("RGB"))
            if a.shape[0] != lv.tile_h or a.shape[1] != lv.tile_w:
                raise TiffError("JPEG tile size does not match the directory")
            return a
        a = a[: lv.tile_h * lv.tile_w * lv.samples].reshape(lv.tile_h, lv.tile_w, lv.samples)
        return a[:, :, :3]

    def read_band(self, level: int, tile_row: int, pool: Optional[ThreadPoolExecutor] = None) -> np.ndarray:
        """uint8[rows, width, 3] of one row of tiles (clipped to the level), missing tiles = 0."""
        lv = self.levels[level]
        y0 = tile_row * lv.tile_h
        rows = min(lv.tile_h, lv.height - y0)
        band = np.zeros((rows, lv.width, 3), np.uint8)
        idx = [tile_row * lv.tiles_across + tx for tx in range(lv.tiles_across)]
        tiles = list(pool.map(lambda i: self._decode_tile(lv, i), idx)) if pool else [self._decode_tile(lv, i) for i in idx]
        for tx, t in enumerate(tiles):
            if t is None:
                continue
            x0 = tx * lv.tile_w
            cols = min(lv.tile_w, lv.width - x0)
            band[:, x0:x0 + cols] = t[:rows, :cols]
        return band

    def read_region(self, location: Tuple[int, int], level: int, size: Tuple[int, int]) -> np.ndarray:
        """openslide semantics: ``location`` in level-0 coordinates, ``size`` in level pixels;
        returns uint8[h, w, 4] RGBA, transparent black outside the level and in missing tiles."""
        lv = self.levels[level]
        ds = self.level_downsamples[level]
        x0, y0 = int(location[0] / ds), int(location[1] / ds)
        w, h = size
        out = np.zeros((h, w, 4), np.uint8)
        for ty in range(max(0, y0 // lv.tile_h), min(lv.tiles_down, (y0 + h - 1) // lv.tile_h + 1)):
            for tx in range(max(0, x0 // lv.tile_w), min(lv.tiles_across, (x0 + w - 1) // lv.tile_w + 1)):
                t = self._decode_tile(lv, ty * lv.tiles_across + tx)
                if t is None:
                    continue
                gx0, gy0 = tx * lv.tile_w, ty * lv.tile_h
                ax0, ay0 = max(gx0, x0), max(gy0, y0)
                ax1 = min(gx0 + lv.tile_w, lv.width, x0 + w)
                ay1 = min(gy0 + lv.tile_h, lv.height, y0 + h)
                if ax1 <= ax0 or ay1 <= ay0:
                    continue
                out[ay0 - y0:ay1 - y0, ax0 - x0:ax1 - x0, :3] = t[ay0 - gy0:ay1 - gy0, ax0 - gx0:ax1 - gx0]
                out[ay0 - y0:ay1 - y0, ax0 - x0:ax1 - x0, 3] = 255
        return out

    # ---- device pyramid ----------------------------------------------------------------
    def _device_jpeg_levels(self, lvs, devs, chunk_bytes: Optional[int] = None):
        """JPEG tiles of the given levels decoded on the device (csrc/jpeg_decode.hip: Huffman one lane per tile, libjpeg's
        integer IDCT, fancy upsampling, YCbCr -> RGB), written into ``devs`` (uint8[H, Wpad, 3] each).  The tiles of ALL levels
        go into the same calls (a call lasts as long as its slowest tile).  Returns, per level, the indices of the tiles the
        device decoder did not take (another sampling, progressive, ...): the caller decodes those on the host."""
        import ctypes as C

        import torch

        from . import capi

        lib = capi.load_library()
        device = devs[0].device
        import warnings

        file_dev = torch.empty(int(self._mm.shape[0]) + 64, dtype=torch.uint8, device=device)  # + slack behind the end
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # "the given NumPy array is not writable": it is only read
            file_dev[:self._mm.shape[0]].copy_(torch.from_numpy(self._mm))
        if chunk_bytes is None:  # scratch for as many tiles per call as HBM allows: a call lasts as long as its slowest tile
            free, _ = torch.cuda.mem_get_info(device)
            chunk_bytes = int(max(1 << 30, min(96 << 30, free // 2)))
        tabs = [np.frombuffer(bytes(lv.jpeg_tables), np.uint8) if lv.jpeg_tables else None for lv in lvs]
        arr = (capi.JpegLevel * len(lvs))()
        for i, (lv, dv) in enumerate(zip(lvs, devs)):
            arr[i] = capi.JpegLevel(dv.data_ptr(), int(dv.stride(0)), lv.width, lv.height, lv.tile_w, lv.tile_h, int(lv.photometric), 0,
                                    tabs[i].ctypes.data if tabs[i] is not None else None, int(tabs[i].shape[0]) if tabs[i] is not None else 0)
        off, cnt, xyl, lvl_of, idx_of = [], [], [], [], []
        for i, lv in enumerate(lvs):
            n = lv.tiles_across * lv.tiles_down
            k = np.arange(n)
            off.append(np.asarray(lv.offsets[:n], np.int64)), cnt.append(np.asarray(lv.counts[:n], np.int64))
            xyl.append(np.stack([(k % lv.tiles_across) * lv.tile_w, (k // lv.tiles_across) * lv.tile_h, np.full(n, i)], 1).astype(np.int32))
            lvl_of.append(np.full(n, i)), idx_of.append(k)
        off, cnt, xyl = np.concatenate(off), np.concatenate(cnt), np.concatenate(xyl)
        lvl_of, idx_of = np.concatenate(lvl_of), np.concatenate(idx_of)
        n = off.shape[0]
        tw, th = max(lv.tile_w for lv in lvs), max(lv.tile_h for lv in lvs)
        per_tile = lib.hipac_jpeg_workspace_bytes(tw, th, 1)
        step = int(max(1, min(32768, chunk_bytes // max(per_tile, 1))))
        ws = torch.empty(lib.hipac_jpeg_workspace_bytes(tw, th, min(step, n)), dtype=torch.uint8, device=device)
        left = [[] for _ in lvs]
        with torch.cuda.device(device):
            for i0 in range(0, n, step):
                m = min(step, n - i0)
                status = np.ones(m, np.uint8)
                o, c, q = np.ascontiguousarray(off[i0:i0 + m]), np.ascontiguousarray(cnt[i0:i0 + m]), np.ascontiguousarray(xyl[i0:i0 + m])
                capi._check(lib.hipac_jpeg_decode_tiles(self._mm.ctypes.data, file_dev.data_ptr(), int(self._mm.shape[0]), C.addressof(arr),
                                                        len(lvs), o.ctypes.data, c.ctypes.data, q.ctypes.data, m, ws.data_ptr(),
                                                        int(ws.numel()), status.ctypes.data, capi._stream()), "hipac_jpeg_decode_tiles")
                for k in np.nonzero(status == 1)[0]:
                    left[int(lvl_of[i0 + k])].append(int(idx_of[i0 + k]))
                self.device_decoded = getattr(self, "device_decoded", 0) + int((status == 0).sum())
        return left

    def to_device_levels(self, device="cuda", levels: Optional[Sequence[int]] = None, workers: int = 16,
                         device_jpeg: Optional[bool] = None):
        """Into uint8[H, Wpad, 3] HBM tensors (row pitch a multiple of 16 pixels, as ``DeviceSlide`` lays levels out).
        JPEG levels on a ROCm device: the compressed file goes to HBM once and the tiles are decoded there
        (``_device_jpeg_levels``; ``device_jpeg=False`` or ``HIPAC_DEVICE_JPEG=0`` keeps the host decoder); tiles the device
        decoder does not take, and the other compressions, are decoded on host threads and copied band by band.
        Returns a list of (tensor, width)."""
        import os

        import torch

        out = []
        use_dev = (device_jpeg if device_jpeg is not None else os.environ.get("HIPAC_DEVICE_JPEG", "1") != "0") and \
            torch.device(device).type == "cuda"
        use = list(range(self.level_count) if levels is None else levels)
        bufs = {}
        for li in use:
            lv = self.levels[li]
            bufs[li] = torch.zeros((lv.height, (lv.width + 15) // 16 * 16, 3), dtype=torch.uint8, device=device)
        on_dev = [li for li in use if use_dev and self.levels[li].compression == 7 and self.levels[li].samples == 3]
        left = dict(zip(on_dev, self._device_jpeg_levels([self.levels[li] for li in on_dev], [bufs[li] for li in on_dev]))) if on_dev else {}
        with ThreadPoolExecutor(max_workers=workers) as pool:
            for li in use:
                lv = self.levels[li]
                dev = bufs[li]
                if li in left:
                    todo = left[li]
                    for idx, t in zip(todo, pool.map(lambda i: self._decode_tile(lv, i), todo)):
                        if t is None:
                            continue
                        ty, tx = divmod(idx, lv.tiles_across)
                        y0, x0 = ty * lv.tile_h, tx * lv.tile_w
                        rows, cols = min(lv.tile_h, lv.height - y0), min(lv.tile_w, lv.width - x0)
                        dev[y0:y0 + rows, x0:x0 + cols] = torch.from_numpy(np.ascontiguousarray(t[:rows, :cols])).to(device)
                    out.append((dev, lv.width))
                    continue
                for tr in range(lv.tiles_down):
                    band = torch.from_numpy(self.read_band(li, tr, pool))
                    if dev.is_cuda:
                        band = band.pin_memory()
                    y0 = tr * lv.tile_h
                    dev[y0:y0 + band.shape[0], :lv.width].copy_(band, non_blocking=False)
                out.append((dev, lv.width))
        return out


def _split_jpeg_tables(data: bytes) -> Tuple[bytes, bytes]:
    """A complete baseline JPEG -> (tables-only stream SOI DQT.. DHT.. EOI, abbreviated image stream without
    DQT / DHT): the two halves of TIFF's JPEGTables (tag 347) scheme, the form CAMELYON16's files use."""
    if data[:2] != b"\xff\xd8":
        raise TiffError("not a JPEG stream")
    pos, tables, rest = 2, b"", b""
    while True:
        if data[pos] != 0xFF:
            raise TiffError("JPEG marker expected")
        m = data[pos + 1]
        if m == 0xDA:  # start of scan: the entropy-coded data and EOI follow
            rest += data[pos:]
            break
        length = int.from_bytes(data[pos + 2:pos + 4], "big")
        seg = data[pos:pos + 2 + length]
        if m in (0xDB, 0xC4):
            tables += seg
        else:
            rest += seg
        pos += 2 + length
    return b"\xff\xd8" + tables + b"\xff\xd9", b"\xff\xd8" + rest


def write_tiled_tiff(path: str, levels: Sequence[np.ndarray], tile: int = 256, compression: str = "jpeg",
                     quality: int = 90, bigtiff: bool = False, missing: Sequence[Tuple[int, int, int]] = (),
                     jpeg_tables: bool = False, subsampling: int = -1, jpeg_options: Optional[dict] = None):
    """Minimal writer of a tiled pyramid (tests and synthetic data only): ``levels`` are uint8[H,W,3]
    arrays, largest first.  compression: "none" | "deflate" | "jpeg" (YCbCr; every tile a complete JPEG, or with
    ``jpeg_tables=True`` abbreviated streams plus one JPEGTables tag per directory, as real slide files have
    them).  ``missing``: (level, ty, tx) tiles written with byte count 0."""
    from PIL import Image

    comp = {"none": 1, "deflate": 8, "jpeg": 7}[compression]
    bo = "<"
    blobs, ifd_specs = [], []
    pos = 16 if bigtiff else 8
    tables_of_level = []
    for li, img in enumerate(levels):
        h, w = img.shape[:2]
        ta, td = (w + tile - 1) // tile, (h + tile - 1) // tile
        offs, cnts = [], []
        level_tables = None
        for ty in range(td):
            for tx in range(ta):
                if (li, ty, tx) in missing:
                    offs.append(0), cnts.append(0)
                    continue
                t = np.zeros((tile, tile, 3), np.uint8)
                part = img[ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile]
                t[:part.shape[0], :part.shape[1]] = part
                if comp == 1:
                    data = t.tobytes()
                elif comp == 8:
                    data = zlib.compress(t.tobytes(), 6)
                else:
                    bio = io.BytesIO()
                    Image.fromarray(t, "RGB").save(bio, "JPEG", quality=quality, subsampling=subsampling, **(jpeg_options or {}))
                    data = bio.getvalue()
                    if jpeg_tables:  # fixed quality, default Huffman tables: every tile shares one set
                        tb, data = _split_jpeg_tables(data)
                        if level_tables is not None and tb != level_tables:
                            raise TiffError("tiles of one level do not share their JPEG tables")
                        level_tables = tb
                offs.append(pos), cnts.append(len(data))
                blobs.append(data)
                pos += len(data)
        ifd_specs.append((w, h, ta * td, offs, cnts))
        tables_of_level.append(level_tables)
    out = bytearray()
    # data area first, then IFDs (offsets known up front)
    body = b"".join(blobs)
    ifd_pos = (16 if bigtiff else 8) + len(body)
    chunks = []
    cur = ifd_pos
    for li, (w, h, nt, offs, cnts) in enumerate(ifd_specs):
        photometric = 6 if comp == 7 else 2
        entries = [(254, 4, [1 if li else 0]), (256, 4, [w]), (257, 4, [h]), (258, 3, [8, 8, 8]), (259, 3, [comp]),
                   (262, 3, [photometric]), (277, 3, [3]), (284, 3, [1]), (322, 4, [tile]), (323, 4, [tile]),
                   (324, 16 if bigtiff else 4, offs), (325, 16 if bigtiff else 4, cnts)]
        if tables_of_level[li] is not None:
            entries.append((347, 7, list(tables_of_level[li])))  # JPEGTables (UNDEFINED bytes); tags stay sorted
        n = len(entries)
        esz, cw = (20, 8) if bigtiff else (12, 4)
        head = 8 if bigtiff else 2
        ifd_len = head + n * esz + cw
        extra = bytearray()
        ent_bytes = bytearray()
        for tag, typ, vals in entries:
            fmt, sz = _TYPES[typ]
            raw = struct.pack(bo + fmt * len(vals), *vals)
            ent_bytes += struct.pack(bo + "HH", tag, typ) + struct.pack(bo + ("Q" if bigtiff else "I"), len(vals))
            if len(raw) <= cw:
                ent_bytes += raw.ljust(cw, b"\0")
            else:
                ent_bytes += struct.pack(bo + ("Q" if bigtiff else "I"), cur + ifd_len + len(extra))
                extra += raw
                if len(extra) % 2:
                    extra += b"\0"
        nxt = cur + ifd_len + len(extra) if li + 1 < len(ifd_specs) else 0
        blob = (struct.pack(bo + "Q", n) if bigtiff else struct.pack(bo + "H", n)) + bytes(ent_bytes) + \
            struct.pack(bo + ("Q" if bigtiff else "I"), nxt) + bytes(extra)
        chunks.append(blob)
        cur += len(blob)
    if bigtiff:
        out += b"II" + struct.pack(bo + "HHH", 43, 8, 0) + struct.pack(bo + "Q", ifd_pos)
    else:
        out += b"II" + struct.pack(bo + "H", 42) + struct.pack(bo + "I", ifd_pos)
    out += body
    for c in chunks:
        out += c
    with open(path, "wb") as f:
        f.write(out)
