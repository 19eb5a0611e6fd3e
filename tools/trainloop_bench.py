"""Developer tool (GPU box): the SimCLR loop end to end on a PNG tree -- DataLoader workers running the Pillow transforms
(the reference's arrangement, src/models/simclr.py:68-96) against patches resident in HBM + device augmentation -- view pairs
per second over whole epochs, native step in fp16.  usage: python tools/trainloop_bench.py [n_patches] [batch] [workers]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

from ss25_hierarchical_multiscale_image_classification_amd import augment, simclr as S, synth, train_native as TN  # noqa: E402
from ss25_hierarchical_multiscale_image_classification_amd.patch_dataset import PatchDataset  # noqa: E402
from ss25_hierarchical_multiscale_image_classification_amd.simclr_dataset import SimCLRDataset  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 14
root = tempfile.mkdtemp(prefix="hipac_png_")
l0 = synth.synth_level0(224 * 48, 224 * 48, seed=3, device="cpu").numpy()
d = os.path.join(root, "level_3", "tumor_001")
os.makedirs(d)
for i in range(n):
    y, x = (i // 48) % 48, i % 48
    Image.fromarray(l0[224 * y:224 * y + 224, 224 * x:224 * x + 224], "RGB").save(os.path.join(d, f"tumor_001_x{224 * x}_y{224 * y}_normal.png"))
base = PatchDataset(os.path.join(root, "level_3"), transform=None)
torch.manual_seed(0)
tr = TN.NativeSimCLRTrainer({k: v.clone() for k, v in S.SimCLRModel().state_dict().items()}, device="cuda", precision="fp16")


def run(loader, epochs):
    pairs, t0 = 0, None
    for ep in range(epochs + 1):  # the first epoch warms up (workers, tables, workspaces)
        if ep == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        for x_i, x_j in loader:
            tr.step(x_i.to("cuda", torch.float32).contiguous(), x_j.to("cuda", torch.float32).contiguous())
            pairs += x_i.shape[0] if ep >= 1 else 0
    torch.cuda.synchronize()
    return pairs / (time.perf_counter() - t0)


t = time.perf_counter()
pool = augment.DevicePatchPool.from_patch_dataset(base)
t_pool = time.perf_counter() - t
dev_rate = run(augment.DeviceSimCLRLoader(pool, batch, seed=1), 3)
host_rate = run(DataLoader(SimCLRDataset(base, transform=S.get_simclr_transform()), batch_size=batch, shuffle=True, num_workers=workers), 1)
print(f"{n} PNG patches, batch {batch}: device pipeline {dev_rate:.0f} view pairs/s (pool built in {t_pool:.2f} s, once); "
      f"DataLoader with {workers} workers on Pillow {host_rate:.0f} view pairs/s; ratio {dev_rate / host_rate:.1f}x")
