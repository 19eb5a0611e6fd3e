"""Developer tool (GPU box): everything in one process, over and over -- slide files written, loaded (device JPEG decoder), scanned
in two precisions, pools cut from them, both training loops fed by the device pipelines -- watching the allocator for growth.
usage: python tools/soak.py [rounds]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ss25_hierarchical_multiscale_image_classification_amd import augment, capi, extract, simclr as S, synth, tiff_pyramid, train_native as TN  # noqa: E402
from ss25_hierarchical_multiscale_image_classification_amd.resnet import ResNet18Classifier  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nets = {p: capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision=p) for p in ("bf16", "fp16x3", "fp16q8")}
torch.manual_seed(0)
sim = TN.NativeSimCLRTrainer({k: v.clone() for k, v in S.SimCLRModel().state_dict().items()}, device="cuda", precision="fp16")
clf = TN.NativeClassifierTrainer(ResNet18Classifier().state_dict(), device="cuda", lr=1e-4, class_weights=torch.tensor([1.0, 2.0]), precision="fp16")
tmp = tempfile.mkdtemp(prefix="hipac_soak_")
peak0 = None
t0 = time.perf_counter()
for r in range(rounds):
    side = int(np.random.default_rng(r).integers(6000, 16000))
    l0 = synth.synth_level0(side, side - 517, seed=100 + r, device="cuda")
    levels = [t.cpu().numpy() for t in synth.build_pyramid(l0, 4)]
    path = os.path.join(tmp, f"s{r % 3}.tif")
    tiff_pyramid.write_tiled_tiff(path, levels, tile=(256, 512)[r % 2], compression="jpeg", jpeg_tables=bool(r % 3), quality=80 + r % 15)
    slide = extract.DeviceSlide.from_tiff(path)
    ref = extract.DeviceSlide.synthetic(side, side - 517, seed=100 + r, with_polygons=True)
    slide.polygons = ref.polygons
    n = 0
    for p, net in nets.items():
        f, l, pr, meta = extract.score_slide(slide, net)
        n = f.shape[0]
        assert torch.isfinite(f).all() and torch.isfinite(l).all()
    pool = augment.DevicePatchPool.from_slides([slide], level=3)
    for x_i, x_j in list(augment.DeviceSimCLRLoader(pool, 32, seed=r))[:3]:
        if x_i.shape[0] >= 2:
            assert np.isfinite(float(sim.step(x_i, x_j)))
    for x, y, _ in list(augment.DeviceClassifierLoader(pool, 32, seed=r))[:3]:
        loss, _ = clf.step(x, y)
        assert np.isfinite(float(loss))
    del slide, ref, pool, l0
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    alloc = torch.cuda.memory_allocated() / 2**20
    if r == 2:
        peak0 = alloc
    print(f"round {r}: side {side}, {n} kept windows, allocated after the round {alloc:.0f} MiB, {time.perf_counter() - t0:.0f} s", flush=True)
    if peak0 is not None:
        assert alloc <= peak0 * 1.05 + 64, ("allocator growth", alloc, peak0)
print("soak ok")
