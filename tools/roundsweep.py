"""Developer tool (GPU box): tile-round quantisation.  Persistent kernels run ceil(tiles / 512) rounds; a sub-batch of 512
images gives layer2's 3x3 convs 1568 tiles = 3.06 rounds (4 are paid), a group of 4096 gives layer4's 3136 = 6.125 (7).
usage: python tools/roundsweep.py  -> patches/s for (batch, sub-batch, group) combinations"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ss25_hierarchical_multiscale_image_classification_amd import capi, synth  # noqa: E402

dev = torch.device("cuda:0")
net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision="bf16")
big = synth.synth_patches_u8(8192, seed=1, device=dev)
for batch, bc, gc in ((8192, 512, 4096), (8016, 501, 4008), (8016, 1002, 4008), (8192, 1024, 4096), (8024, 1003, 4012), (8016, 501, 8016)):
    os.environ["HIPAC_SUBBATCH"], os.environ["HIPAC_GROUP"] = str(bc), str(gc)
    x = big[:batch].contiguous()
    for _ in range(2):
        net.forward(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        net.forward(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 6
    print(f"batch {batch} sub-batch {bc} group {gc}: {ms:.2f} ms  {batch / ms * 1e3:.0f} patches/s", flush=True)
