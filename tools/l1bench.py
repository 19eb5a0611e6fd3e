"""Developer tool (GPU box): the layer1 block op alone, ns per patch, for the library selected by HIPAC_LIB_NAME."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ss25_hierarchical_multiscale_image_classification_amd import capi, synth  # noqa: E402

dev = torch.device("cuda:0")
net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision=sys.argv[1] if len(sys.argv) > 1 else "bf16")
u8 = synth.synth_patches_u8(512, seed=1, device=dev)
net.forward(u8)
i = [n for n, _ in bench.OPS].index("l1b0")
for _ in range(5):
    net.run_ops(u8, i, i)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    net.run_ops(u8, i, i)
e1.record()
torch.cuda.synchronize()
print(f"{os.environ.get('HIPAC_LIB_NAME', 'libhipac_hip.so')}: l1 block op {e0.elapsed_time(e1) / 50 / 512 * 1e6:.0f} ns per patch")
