"""Developer tool (GPU box): how fast a tiled pyramidal TIFF gets into HBM (tiff_pyramid.TiffPyramid.to_device_levels:
tiles decoded on host threads, copied band by band) next to the scan of the same slide.
usage: python tools/tiffbench.py [side] [compression: jpeg|deflate|none] [workers]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ss25_hierarchical_multiscale_image_classification_amd import capi, extract, synth, tiff_pyramid  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
comp = sys.argv[2] if len(sys.argv) > 2 else "jpeg"
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 16
l0 = synth.synth_level0(side, side, seed=2, device="cuda")
levels = [t.cpu().numpy() for t in synth.build_pyramid(l0, 4)]
path = os.path.join(tempfile.mkdtemp(prefix="hipac_tiff_"), "slide.tif")
t = time.perf_counter()
tiff_pyramid.write_tiled_tiff(path, levels, tile=512, compression=comp)
print(f"wrote {path}: {os.path.getsize(path) / 1e6:.0f} MB in {time.perf_counter() - t:.1f} s")
del l0, levels
for rep in range(2):
    t = time.perf_counter()
    slide = extract.DeviceSlide.from_tiff(path, workers=workers)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    px = sum(w * h for w, h in slide.level_dimensions)
    print(f"from_tiff ({comp}, {workers} threads): {dt:.2f} s = {px * 3 / dt / 1e9:.2f} GB/s of decoded pixels, {px / 1e6 / dt:.0f} Mpx/s")
net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision="bf16")
for rep in range(2):
    torch.cuda.synchronize()
    t = time.perf_counter()
    f, l, p, m = extract.score_slide(slide, net, levels=(0, 1, 2, 3))
    torch.cuda.synchronize()
    print(f"score_slide: {time.perf_counter() - t:.3f} s, {f.shape[0]} kept windows")
