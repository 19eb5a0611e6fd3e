import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ss25_hierarchical_multiscale_image_classification_amd import capi, synth
dev = torch.device("cuda:0")
net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision="bf16")
x = synth.synth_patches_u8(512, seed=3, device=dev)
net.forward(x); torch.cuda.synchronize()
for _ in range(3): net.run_ops(x, 0, 0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): net.run_ops(x, 0, 0)
e1.record(); e1.synchronize()
print(os.environ.get("HIPAC_LIB_NAME"), f"{e0.elapsed_time(e1)/20/512*1e6:.0f} ns/img")
