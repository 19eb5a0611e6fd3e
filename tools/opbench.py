"""Developer tool (GPU box): per-op timing table of the trunk for the library named by
HIPAC_LIB_NAME (A/B builds).  usage: python tools/opbench.py [precision] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ss25_hierarchical_multiscale_image_classification_amd import capi, synth  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision=prec)
ops = bench.per_op_times(net, dev, reps=reps)
tot = sum(o["us_per_image"] for o in ops)  # us per image over the whole trunk
print(f"lib={os.environ.get('HIPAC_LIB_NAME','libhipac_hip.so')} trunk {tot:.3f} us/img -> {1e6/tot:.0f} img/s "
      f"({1e6/tot*3.627e9/1e12:.0f} TF)")
print(" ".join(f"{o['op']}:{o['us_per_image']*1e3:.0f}ns/{(o['tflops'] or 0):.0f}" for o in ops))
