"""Developer tool (GPU box): a whole-slide scan with its phases run one after another and timed apart (window
decisions of all levels, gather of every kept window into one uint8 batch, the ResNet forwards), next to
extract.score_slide's own wall clock.  usage: python tools/wsi_phases.py [side] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ss25_hierarchical_multiscale_image_classification_amd import capi, extract, synth

side = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision=prec)
slide = extract.DeviceSlide.synthetic(side, side, seed=10, with_polygons=True)
sync = torch.cuda.synchronize
for rep in range(3):
    sync(); t0 = time.perf_counter()
    out = extract.score_slide(slide, net)
    sync(); t_all = time.perf_counter() - t0
    sync(); t0 = time.perf_counter()
    lws = [extract.LevelWindows(slide, lv) for lv in (0, 1, 2, 3)]
    kept = [lw.kept_index() for lw in lws]
    sync(); t_dec = time.perf_counter() - t0
    n = sum(int(k.shape[0]) for k in kept)
    buf = torch.empty((n, 224, 224, 3), dtype=torch.uint8, device="cuda")
    sync(); t0 = time.perf_counter()
    o = 0
    for lw, k in zip(lws, kept):
        for i0 in range(0, k.shape[0], 4096):
            idx = k[i0:i0 + 4096]
            lw.patches(idx, out=buf[o:o + idx.shape[0]])
            o += idx.shape[0]
    sync(); t_gather = time.perf_counter() - t0
    t0 = time.perf_counter()
    for i in range(0, n, 8192):
        net.forward(buf[i:i + 8192], want_logits=True, want_labels=True)
    sync(); t_net = time.perf_counter() - t0
    print(f"rep {rep}: score_slide {t_all*1e3:.1f} ms | serial: decisions {t_dec*1e3:.1f} + gather {t_gather*1e3:.1f} + "
          f"forwards {t_net*1e3:.1f} = {(t_dec+t_gather+t_net)*1e3:.1f} ms ({n} kept)")
    del buf, lws, kept
