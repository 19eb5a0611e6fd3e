"""JPEG tiles of a tiled pyramidal TIFF decoded on the device (csrc/jpeg_decode.hip) against the host decoder -- Pillow's
libjpeg, the arithmetic openslide reaches for the reference (src/main.py:650, :693): bit for bit, for every tile layout the
device decoder takes, and with the host fallback for the ones it does not."""
import numpy as np
import pytest
import torch

from ss25_hierarchical_multiscale_image_classification_amd import extract, synth, tiff_pyramid

pytestmark = pytest.mark.gpu


def _pyramid(w, h, seed, noise=False):
    if noise:
        l0 = torch.randint(0, 256, (h, w, 3), generator=torch.Generator().manual_seed(seed), dtype=torch.uint8)
    else:
        l0 = synth.synth_level0(w, h, seed=seed, device="cpu")
    return [t.numpy() for t in synth.build_pyramid(l0, 3)]


def _both(path, workers=4):
    tp = tiff_pyramid.TiffPyramid(path)
    dev = [t.cpu() for t, _ in tp.to_device_levels("cuda", workers=workers)]
    host = [t.cpu() for t, _ in tiff_pyramid.TiffPyramid(path).to_device_levels("cuda", workers=workers, device_jpeg=False)]
    return tp, dev, host


@pytest.mark.parametrize("tile,quality,tables,sub,bigtiff,noise", [
    (256, 90, False, -1, False, False), (512, 75, True, -1, True, False), (256, 30, True, -1, False, False),
    (256, 100, False, -1, False, True), (128, 95, True, 0, False, False), (256, 100, False, 0, False, True),
    (64, 85, False, -1, False, False), (48, 85, False, -1, False, False)])
def test_device_jpeg_equals_the_host_decoder(tmp_path, tile, quality, tables, sub, bigtiff, noise):
    levels = _pyramid(1500, 1100, 3 + tile + quality, noise)
    path = str(tmp_path / "s.tif")
    tiff_pyramid.write_tiled_tiff(path, levels, tile=tile, compression="jpeg", quality=quality, jpeg_tables=tables, subsampling=sub,
                                  bigtiff=bigtiff)
    tp, dev, host = _both(path)
    n_tiles = sum(l.tiles_across * l.tiles_down for l in tp.levels)
    assert tp.device_decoded == n_tiles  # every tile went through the device decoder
    for a, b in zip(dev, host):
        assert torch.equal(a, b), int((a != b).sum())


def test_restart_markers_optimised_tables_and_fallbacks(tmp_path):
    levels = _pyramid(900, 700, 8)
    cases = [({"restart_marker_rows": 1}, -1, True), ({"restart_marker_blocks": 5}, 0, True), ({"optimize": True}, -1, True),
             ({"progressive": True}, -1, False), ({}, 1, False)]  # progressive and 4:2:2 tiles: the host decodes them
    for k, (opts, sub, on_device) in enumerate(cases):
        path = str(tmp_path / f"s{k}.tif")
        tiff_pyramid.write_tiled_tiff(path, levels, tile=256, compression="jpeg", quality=88, subsampling=sub, jpeg_options=opts)
        tp, dev, host = _both(path)
        n_tiles = sum(l.tiles_across * l.tiles_down for l in tp.levels)
        if on_device:
            assert tp.device_decoded > 0 and (opts.get("optimize") or tp.device_decoded == n_tiles), (opts, tp.device_decoded)
        else:
            assert getattr(tp, "device_decoded", 0) == 0, opts
        for a, b in zip(dev, host):
            assert torch.equal(a, b), (opts, sub, int((a != b).sum()))


def test_missing_tiles_and_the_slide_object(tmp_path):
    levels = _pyramid(1300, 1000, 12)
    path = str(tmp_path / "s.tif")
    tiff_pyramid.write_tiled_tiff(path, levels, tile=256, compression="jpeg", quality=90, jpeg_tables=True, missing=[(0, 1, 2), (1, 0, 0)])
    tp, dev, host = _both(path)
    assert all(torch.equal(a, b) for a, b in zip(dev, host))
    assert not dev[0][256:512, 512:768].any()  # the missing tile stays zero
    slide = extract.DeviceSlide.from_tiff(path, n_levels=3)
    for lv, ref in zip(slide.levels, host):
        assert torch.equal(lv.cpu(), ref)
