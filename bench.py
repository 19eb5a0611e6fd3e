#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload resnet|wsi]

Default workload (BASELINE.json configs[1]): batched ResNet18 bf16 inference over
synthetic 224x224 patches that are already resident in HBM as uint8 HWC.  One STEP =
one pass of the hot path over one batch of 8192 patches: ToTensor/Normalize (fused into
the stem kernel) + the MFMA ResNet18 forward producing features [B,512],
logits [B,2] and argmax labels.  8 steps = the 64k patches of the config.  With N > 1
every rank scores its own patches (slides shard, no data-path collective) and the
per-patch logits/labels are all-gathered over RCCL once per step (weak scaling).

Prints ONE JSON line (rank 0) with the driver's fields plus
  "roofline"     -- dominant kernel: algorithmic FLOPs per launch / its average launch
                    duration measured here with HIP events on the launch stream
  "cpu_baseline" -- the oracle (torch fp32 functional ResNet18 on the host cores) timed
                    on a bounded sample of the same workload (rank 0, N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ss25_hierarchical_multiscale_image_classification_amd import capi, dist as hdist, synth  # noqa: E402

FLOP_PER_PATCH = 2 * 1_813_562_368  # convs + fc, BN folded
PEAK_BF16_DENSE_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16/fp16 MFMA
PEAK_HBM_GBS = 8000.0

# trunk ops (hipac_resnet18_run_ops): name, MACs per image
def _conv_macs(cin, cout, k, ho):
    return cin * cout * k * k * ho * ho


OPS = [("stem7x7+pool", _conv_macs(3, 64, 7, 112)), ("(fused)", 0)]
for _s, (_ci, _co, _ho) in enumerate(((64, 64, 56), (64, 128, 28), (128, 256, 14), (256, 512, 7))):
    if _s in (1, 2):  # layers 2-3: the 1x1/2 projection shortcut rides inside the 3x3/2 launch (empty op slot)
        OPS.append((f"l{_s+1}b0c1+proj", _conv_macs(_ci, _co, 3, _ho) + _conv_macs(_ci, _co, 1, _ho)))
        OPS.append(("(fused)", 0))
    else:
        OPS.append((f"l{_s+1}b0c1", _conv_macs(_ci, _co, 3, _ho)))
        if _s == 3:       # layer4 keeps a separate projection launch
            OPS.append(("l4proj", _conv_macs(_ci, _co, 1, _ho)))
    OPS.append((f"l{_s+1}b0c2", _conv_macs(_co, _co, 3, _ho)))
    OPS.append((f"l{_s+1}b1c1", _conv_macs(_co, _co, 3, _ho)))
    OPS.append((f"l{_s+1}b1c2", _conv_macs(_co, _co, 3, _ho)))
assert len(OPS) == 21 and sum(m for _, m in OPS) + 1024 == 1_813_562_368


def per_op_times(net: capi.PackedResNet18, dev, reps: int = 20):
    """Average launch duration (ms) of every trunk kernel, HIP events on torch's current
    stream == the stream the library launches on.  Early ops (stem..layer2) run on one
    sub-batch (512 images), late ops (layer3, layer4) on one group (4096 images), exactly
    as inside hipac_resnet18_forward."""
    bc = int(os.environ.get("HIPAC_SUBBATCH", "512"))
    gc = max(int(os.environ.get("HIPAC_GROUP", "4096")), bc)
    x = synth.synth_patches_u8(gc, seed=99, device=dev)  # the bench's own input form: uint8 HWC
    net.forward(x, want_feats=True)  # fills the workspace with real activations
    torch.cuda.synchronize()
    out = []
    for i, (name, macs) in enumerate(OPS):
        n_img = bc if i < 11 else gc
        for _ in range(3):
            net.run_ops(x, i, i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            net.run_ops(x, i, i)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out.append({"op": name, "ms": ms, "images": n_img, "us_per_image": ms * 1e3 / n_img,
                    "tflops": (2 * macs * n_img / (ms * 1e-3) / 1e12) if macs else None,
                    "flops_per_launch": 2 * macs * n_img})
    return out


def cpu_baseline(n_sample: int = 256):
    """Oracle ResNet18 forward on the host cores (config 1: 256 random patches, fp32)."""
    from oracle import resnet18_ref, transform_ref  # the checker, timed as the reported baseline only

    sd = synth.seeded_resnet18_state_dict(0, num_classes=2)
    u8 = synth.synth_patches_u8(n_sample, seed=1)
    lut = torch.from_numpy(transform_ref.normalize_lut())
    x = torch.stack([lut[c][u8[..., c].long()] for c in range(3)], dim=1)  # [n,3,224,224] fp32
    threads = torch.get_num_threads()
    resnet18_ref.resnet18_forward(x[:32], sd)  # warm-up
    t0 = time.time()
    done = 0
    while done < n_sample:
        resnet18_ref.resnet18_forward(x[done : done + 64], sd)
        done += 64
    dt = time.time() - t0
    return {"value": n_sample / dt, "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"{n_sample} random 224x224x3 patches, fp32 torch-functional ResNet18 (oracle), batch 64, "
                      f"{dt:.1f} s of CPU work"}


def cpu_baseline_wsi(side: int = 4000):
    """Oracle extractor + Pillow resize + normalise (no PNG I/O, no network) on a bounded
    crop of the same kind of synthetic slide, single process like the reference's loop."""
    from oracle import extractor_ref, transform_ref

    levels = synth.build_pyramid(synth.synth_level0(side, side, seed=10, n_blobs=6), 4)
    slide = extractor_ref.ArraySlide([l.numpy() for l in levels])
    t0 = time.time()
    n_win = n_kept = 0
    for level in (0, 1, 2, 3):
        wins, pix = extractor_ref.extract_patches_ref(slide, level)
        n_win += len(wins)
        for p in pix:
            transform_ref.eval_transform(p)
            n_kept += 1
    dt = time.time() - t0
    return {"value": n_win / dt, "unit": "windows/s", "cores": 1, "kind": "port",
            "sample": f"{side}x{side} level-0 crop, levels 0-3, reference stride 224: {n_win} windows, {n_kept} kept, "
                      f"extractor + Pillow resize + normalise only (no ResNet), {dt:.1f} s of CPU work"}


def run_resnet(args, rank, world, dev):
    B = args.batch
    sd = synth.seeded_resnet18_state_dict(0, num_classes=2)
    net = capi.PackedResNet18(sd, precision=args.precision)
    # synthetic uint8 patches resident in HBM: a pool of `pool` distinct batches cycled over the steps
    pool = max(1, min(args.steps, 8))
    data = [synth.synth_patches_u8(B, seed=1 + rank * 1000 + i, device=dev) for i in range(pool)]

    def step(i):
        u8 = data[i % pool]  # uint8[B,224,224,3] in HBM; ToTensor/Normalize is fused into the stem kernel
        f, l, lab = net.forward(u8, want_feats=True, want_logits=True, want_labels=True)
        if world > 1:
            l, lab = hdist.all_gather_equal(l), hdist.all_gather_equal(lab)  # equal shards: no count exchange
        return f, l, lab

    for i in range(args.warmup):
        step(i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert out[1].shape[0] == B * world
    value = world * B * args.steps / dt
    rec = {
        "metric": "224x224 patches/sec (ResNet18 fwd)", "value": value, "unit": "patches/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"resnet18_fwd_{B * args.steps // 1024}k_patches_{args.precision}",
                   "patches_per_step_per_gpu": B, "input": "uint8 HWC patches resident in HBM",
                   "outputs": "features[B,512] + logits[B,2] + labels", "weights": "seeded random init",
                   "parallelism": f"patch shards x{world}, all-gather of logits/labels" if world > 1 else "single GPU"},
        "tflops_whole_path": value * FLOP_PER_PATCH / 1e12 / world,
    }
    if rank == 0:
        ops = per_op_times(net, dev)
        convs = [o for o in ops if o["tflops"]]
        dom = max(convs, key=lambda o: o["us_per_image"])  # the kernel that costs most per patch
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(dom["op"])
        kname = {"stem7x7+pool": "stem_pool_kernel", "l1": "conv3x3_c64_kernel"}.get(
            dom["op"] if dom["op"].startswith("stem") else dom["op"][:2], "conv3x3_halo_kernel / conv_glds_kernel")
        rec["roofline"] = {"bound": "mfma", "kernel": f"{kname}[{dom['op']}]", "achieved": dom["tflops"],
                           "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": dom["tflops"] / PEAK_BF16_DENSE_TFLOPS,
                           "traffic": traffic, "launch_ms": dom["ms"], "flops_per_launch": dom["flops_per_launch"],
                           "images_per_launch": dom["images"]}
        rec["per_op"] = [{"op": o["op"], "images": o["images"], "ms": round(o["ms"], 4),
                          "tflops": None if o["tflops"] is None else round(o["tflops"], 1)} for o in ops]
        if traffic is not None:
            rec["roofline"]["traffic_source"] = "profiles/roofline_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
        if world == 1 and args.precision == "bf16":
            # same inputs, same kernels instantiated for fp16 operands: the precision that meets the
            # north star's 1e-3 relative tolerance (tests/test_gpu_resnet.py) at the same rate
            alt = capi.PackedResNet18(sd, precision="fp16")
            for i in range(max(1, args.warmup)):
                alt.forward(data[i % pool], want_feats=True, want_logits=True, want_labels=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                alt.forward(data[i % pool], want_feats=True, want_logits=True, want_labels=True)
            torch.cuda.synchronize()
            rec["alt_precision"] = {"dtype": "fp16", "value": B * args.steps / (time.perf_counter() - t0),
                                    "unit": "patches/s"}
        if world == 1:
            rec["pcie_inclusive"] = pcie_inclusive(net, data[0], steps=min(4, args.steps))
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline()
    return rec


def pcie_inclusive(net, u8_dev, steps):
    """The same step when the boundary hands over HOST buffers: uint8 patches in pinned host memory,
    copied to HBM on a side stream one batch ahead of the forward that consumes them (never `value`)."""
    host = torch.empty(u8_dev.shape, dtype=torch.uint8, pin_memory=True)
    host.copy_(u8_dev)
    bufs = [torch.empty_like(u8_dev) for _ in range(2)]
    copy_stream = torch.cuda.Stream()
    ready = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    main = torch.cuda.current_stream()

    def upload(i):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(done[i & 1])  # the forward that last read this buffer has finished
            bufs[i & 1].copy_(host, non_blocking=True)
            ready[i & 1].record(copy_stream)

    for ev in done:
        ev.record(main)
    upload(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if i + 1 < steps:
            upload(i + 1)
        main.wait_event(ready[i & 1])
        net.forward(bufs[i & 1], want_feats=True, want_logits=True, want_labels=True)
        done[i & 1].record(main)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = u8_dev.shape[0]
    # the first batch's upload is outside the timed region; steps-1 uploads overlap the forwards
    return {"value": n * steps / dt, "unit": "patches/s", "steps": steps,
            "note": "uint8 patches from pinned host memory, H2D copy double-buffered against the forward"}


def run_wsi(args, rank, world, dev):
    """BASELINE.json configs[2]: hierarchical scan of one synthetic level-0 slide across
    levels 0-3 with on-GPU whiteness filter + resize; one STEP = one whole slide."""
    from ss25_hierarchical_multiscale_image_classification_amd import extract

    side = args.slide_side
    net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision=args.precision)
    slide = extract.DeviceSlide.synthetic(side, side, seed=10 + rank, with_polygons=True)
    stride_of = (lambda lvl: None) if args.grid == "reference" else (lambda lvl: extract.PATCH_SIZES[lvl])

    def step():
        f, l, p, meta = extract.score_slide(slide, net, levels=(0, 1, 2, 3), batch_windows=args.batch_windows,
                                            stride=stride_of)
        n_all = 0
        for lvl in (0, 1, 2, 3):
            w, h = slide.level_dimensions[lvl]
            n_all += len(extract.window_grid(w, h, lvl, stride_of(lvl))[2])
        if world > 1:
            hdist.gather_results(f, l, meta)
        return f.shape[0], n_all

    for _ in range(args.warmup):
        step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_kept, n_all = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    unique_bytes = sum(w * h * 3 for (w, h) in slide.level_dimensions)
    extra = {}
    if rank == 0:
        # dominant HBM-bound kernel of this workload: the level-0 resampler (hpass + vpass).
        # algorithmic bytes = every level-0 source byte once (unique bytes)
        w0, h0 = slide.level_dimensions[0]
        capi.LevelPlanes(slide.levels[0], 1792, width=w0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            capi.LevelPlanes(slide.levels[0], 1792, width=w0)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 3
        gbs = w0 * h0 * 3 / (ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath) and side == 50000:  # the PMC passes were taken on the default 50 000^2 slide
            traffic = json.load(open(tpath)).get("wsi_level0_planes")
        extra["roofline"] = {"bound": "hbm", "kernel": "hipac_level_build_planes[level 0: hpass_kernel<8> + vpass_kernel<8>]",
                             "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                             "traffic": traffic, "launch_ms": ms, "bytes_per_launch": w0 * h0 * 3}
        if world == 1 and not args.no_cpu_baseline:
            extra["cpu_baseline"] = cpu_baseline_wsi()
    return {**extra,
        "metric": "whole-WSI wall-clock (levels 0-3)", "value": dt / args.steps, "unit": "s/slide", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": False,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8+" + args.precision, "data": "synthetic",
        "config": {"workload": f"wsi_scan_{side}x{side}_levels0-3_{args.grid}_grid", "windows": n_all,
                   "kept": n_kept, "parallelism": f"one slide per rank x{world}"},
        "windows_per_s": world * n_all * args.steps / dt, "kept_patches_per_s": world * n_kept * args.steps / dt,
        "unique_source_GBps": world * unique_bytes * args.steps / dt / 1e9,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["resnet", "wsi"], default="resnet")
    ap.add_argument("--precision", choices=["bf16", "fp16"], default="bf16")
    ap.add_argument("--batch", type=int, default=8192, help="patches per step per GPU")
    ap.add_argument("--slide_side", type=int, default=50000)
    ap.add_argument("--grid", choices=["reference", "nonoverlap"], default="reference")
    ap.add_argument("--batch_windows", type=int, default=4096)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="process-group backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--one_device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo; RCCL wants one GPU per rank)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the HIP path has no CPU fallback")
    if args.one_device:
        os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
    rank, world, local = hdist.init_from_env(args.backend if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device("cuda", 0 if args.one_device else local)
    torch.cuda.set_device(dev)
    rec = run_resnet(args, rank, world, dev) if args.workload == "resnet" else run_wsi(args, rank, world, dev)
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
