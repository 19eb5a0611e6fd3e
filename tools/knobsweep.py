"""Developer tool (GPU box): whole-trunk patches/s for combinations of the per-call environment knobs.
usage: python tools/knobsweep.py [--precision fp16x3] "HIPAC_SUBBATCH=256,512" "HIPAC_LANES=1,2" ..."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ss25_hierarchical_multiscale_image_classification_amd import capi, synth  # noqa: E402

dev = torch.device("cuda:0")
prec = "bf16"
if len(sys.argv) > 2 and sys.argv[1] == "--precision":
    prec = sys.argv[2]
    del sys.argv[1:3]
net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision=prec)
u8 = synth.synth_patches_u8(8192, seed=1, device=dev)
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[1:]]
for combo in itertools.product(*[v for _, v in axes]):
    for (k, _), v in zip(axes, combo):
        os.environ[k] = v
    for _ in range(2):
        net.forward(u8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        net.forward(u8)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 6
    print(" ".join(f"{k}={v}" for (k, _), v in zip(axes, combo)), f": {ms:.2f} ms  {8192 / ms * 1e3:.0f} patches/s", flush=True)
