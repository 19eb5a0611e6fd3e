"""Developer tool (GPU box): precision fp16q8 against the fp32 oracle -- features, logits and every tap on random patches, float and
uint8 input -- beside fp16x3, and both modes' throughput on 8192 patches.  usage: python tests/tools/q8check.py [n_patches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import resnet18_ref as R, transform_ref as T
from ss25_hierarchical_multiscale_image_classification_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
TAPS = ["stem", "maxpool"] + [f"layer{s}.{k}" for s in (1, 2, 3, 4) for k in (0, 1)]
sd = synth.seeded_resnet18_state_dict(7, num_classes=2)
u8 = synth.synth_patches_u8(n, seed=31)
x = torch.stack([torch.from_numpy(T.to_tensor_normalize(p.numpy())) for p in u8])
taps = {}
ref_f, ref_l = R.resnet18_forward(x, sd, taps)
rel = lambda a, b: float((a.cpu().float() - b).abs().max() / b.abs().max())
for prec in ("fp16q8", "fp16x3"):
    net = capi.PackedResNet18(sd, precision=prec)
    f, l, lab = net.forward(x.cuda(), want_feats=True, want_logits=True, want_labels=True)
    print(f"{prec} float input: features {rel(f, ref_f):.2e}  logits {rel(l, ref_l):.2e}  labels equal {int((lab.cpu() == ref_l.argmax(1)).sum())}/{n}", flush=True)
    print("   taps:", "  ".join(f"{name} {rel(net.tap(n, i), taps[name]):.1e}" for i, name in enumerate(TAPS)), flush=True)
    f, l, lab = net.forward(u8.cuda(), want_feats=True, want_logits=True, want_labels=True)
    print(f"{prec} uint8 input: features {rel(f, ref_f):.2e}  logits {rel(l, ref_l):.2e}  labels equal {int((lab.cpu() == ref_l.argmax(1)).sum())}/{n}", flush=True)
xb = torch.randint(0, 256, (8192, 224, 224, 3), dtype=torch.uint8, device="cuda")
for prec in ("fp16q8", "fp16x3"):
    net = capi.PackedResNet18(sd, precision=prec)
    net.forward(xb); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): net.forward(xb)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    print(f"{prec}: {8192 / dt:.0f} patches/s", flush=True)
