"""Developer tool (GPU box): small-batch latency of hipac_resnet18_forward, plain launches vs a captured HIP graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ss25_hierarchical_multiscale_image_classification_amd import capi, synth

net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision="bf16")
for n in (1, 16, 64, 256):
    u8 = synth.synth_patches_u8(n, seed=n, device="cuda")
    f0, l0, p0 = net.forward(u8, want_logits=True, want_labels=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        net.forward(u8, want_logits=True, want_labels=True)
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / 50
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        net.forward(u8, want_logits=True, want_labels=True)  # warm-up on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fg, lg, pg = net.forward(u8, want_logits=True, want_labels=True)
    g.replay()
    torch.cuda.synchronize()
    ok = torch.equal(fg, f0) and torch.equal(lg, l0) and torch.equal(pg, p0)
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 50
    print(f"batch {n}: plain {plain*1e6:.0f} us, graph replay {graph*1e6:.0f} us, identical {ok}")
