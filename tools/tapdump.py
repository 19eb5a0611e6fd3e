"""Developer tool: dump block outputs (taps) of a small batch for the library named by HIPAC_LIB_NAME."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ss25_hierarchical_multiscale_image_classification_amd import capi, synth
prec = sys.argv[1]; out = sys.argv[2]
net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision=prec)
u8 = synth.synth_patches_u8(6, seed=5, device="cuda")
f, l, _ = net.forward(u8, want_logits=True)
taps = {f"tap{t}": net.tap(6, t).cpu() for t in range(1, 10)}
taps["feats"] = f.cpu()
torch.save(taps, out)
