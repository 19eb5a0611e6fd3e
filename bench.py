#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload resnet|wsi|simclr]

One command covers BASELINE.json's whole metric.  Default workload (configs[1]): batched ResNet18 bf16
inference over synthetic 224x224 patches already resident in HBM as uint8 HWC; one STEP = one pass of the hot
path over one batch of 8192 patches (ToTensor/Normalize folded into the stem kernel + the MFMA ResNet18
forward -> features [B,512], logits [B,2], argmax labels); 8 steps = the 64k patches of the config.  The same
JSON line also carries
  "wsi"          -- configs[2] and north_star's 100k slide: whole-slide hierarchical scan (levels 0-3, reference
                    grid), s/slide, kept patches/s, the level-0 resampler's HBM roofline, CPU extractor baseline
  "parity"       -- max norm-relative error of features / logits and label mismatches of every precision against
                    the fp32 oracle on the first 256 patches of the workload (configs[0]), outside the timed region
  "roofline"     -- dominant kernel: algorithmic FLOPs per launch / average launch duration (HIP events, launch stream)
  "cpu_baseline" -- configs[0]: the oracle on the host cores, batch 256 (and 512), warm-up + timed iterations, median
`--workload wsi` makes the whole-slide scan the timed step (with --gpus N: configs[3], one slide per rank and
one ragged all-gather of features + logits + meta); `--workload simclr` times the native SimCLR training step
(configs[4]).

--gpus N > 1: when not already under torch.distributed.run (no WORLD_SIZE in the environment) this process
only LAUNCHES: it starts N fresh child processes, one rank per GPU (RCCL), before anything touches the GPU,
relays rank 0's JSON line and exits non-zero if a child fails.  Every rank scores its own units (slides /
patch shards: no data-path collective), results are all-gathered once per step (weak scaling).
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

from ss25_hierarchical_multiscale_image_classification_amd import capi, dist as hdist, launch as hlaunch, synth  # noqa: E402

FLOP_PER_PATCH = 2 * 1_813_562_368  # convs + fc, BN folded
PEAK_BF16_DENSE_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16/fp16 MFMA
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


# trunk ops (hipac_resnet18_run_ops): name, MACs per image
def _conv_macs(cin, cout, k, ho):
    return cin * cout * k * k * ho * ho


L1_FUSED = os.environ.get("HIPAC_L1_FUSED", "1") != "0"  # a layer1 BasicBlock is one launch (block16_c64_kernel)
PROJK = os.environ.get("HIPAC_PROJK", "1") != "0"  # layers 2-4: the projection shortcut is folded into the block's second conv
OPS = [("stem7x7+pool", _conv_macs(3, 64, 7, 112)), ("(fused)", 0)]
for _s, (_ci, _co, _ho) in enumerate(((64, 64, 56), (64, 128, 28), (128, 256, 14), (256, 512, 7))):
    if _s == 0 and L1_FUSED:  # op slots of conv2 stay empty
        OPS += [("l1b0", 2 * _conv_macs(64, 64, 3, 56)), ("(fused)", 0), ("l1b1", 2 * _conv_macs(64, 64, 3, 56)), ("(fused)", 0)]
        continue
    _pm = _conv_macs(_ci, _co, 1, _ho)  # the 1x1/2 projection shortcut
    if _s >= 1 and PROJK:  # its K steps ride in block0.conv2 (halo16 kernel, PCIN): the entry conv is plain (band16 kernel), the slot empty
        OPS += [(f"l{_s+1}b0c1", _conv_macs(_ci, _co, 3, _ho)), ("(fused)", 0), (f"l{_s+1}b0c2+proj", _conv_macs(_co, _co, 3, _ho) + _pm)]
    elif _s in (1, 2):  # it rides inside the 3x3/2 launch (second accumulator set, empty op slot)
        OPS += [(f"l{_s+1}b0c1+proj", _conv_macs(_ci, _co, 3, _ho) + _pm), ("(fused)", 0), (f"l{_s+1}b0c2", _conv_macs(_co, _co, 3, _ho))]
    else:  # HIPAC_PROJK=0: layer4 keeps a separate projection launch
        OPS += [(f"l{_s+1}b0c1", _conv_macs(_ci, _co, 3, _ho)), ("l4proj", _pm), (f"l{_s+1}b0c2", _conv_macs(_co, _co, 3, _ho))]
    OPS.append((f"l{_s+1}b1c1", _conv_macs(_co, _co, 3, _ho)))
    OPS.append((f"l{_s+1}b1c2", _conv_macs(_co, _co, 3, _ho)))
assert len(OPS) == 21 and sum(m for _, m in OPS) + 1024 == 1_813_562_368

def op_bytes(op: str) -> int:
    """ALGORITHMIC HBM bytes of one trunk op per patch in the 2-byte precisions (every input element read once, every output
    element written once; weights -- shared by the whole launch -- and halo re-reads not counted): what the op's HBM roofline is
    priced with in `per_op` (hbm_frac).  The early layers move 0.6-0.8 MB per patch and op: at the measured rates that is
    2.5-3.8 TB/s, i.e. HBM is as close a bound for them as the MFMA pipe (profiles/r04/pmc_per_kernel.csv has the counters)."""
    if op.startswith("stem"):
        return 224 * 224 * 3 + 56 * 56 * 64 * 2
    st = int(op[1]) - 1
    ci, co, ho = ((64, 64, 56), (64, 128, 28), (128, 256, 14), (256, 512, 7))[st]
    out = ho * ho * co * 2
    if op in ("l1b0", "l1b1"):
        return 2 * out  # the fused block: its input (also the shortcut) and its output
    if op.endswith("b0c1"):
        return (2 * ho) * (2 * ho) * ci * 2 + out
    if op.endswith("b0c1+proj"):
        return (2 * ho) * (2 * ho) * ci * 2 + 2 * out
    if op.endswith("b0c2+proj"):
        return out + ho * ho * ci * 2 + out  # conv input, the block input's pixels (2y, 2x), output
    if op.endswith("b0c2") or op.endswith("b1c2"):
        return (2 * out if op != "l4b1c2" else out) + out  # input, shortcut, output (the last conv leaves pooled sums, not a map)
    return 2 * out  # b1c1, l4proj


KERNEL_OF_OP = {"stem7x7+pool": "stem_pool_strip2_kernel", "l1": "block16_c64_kernel" if L1_FUSED else "conv3x3_c64_kernel", "l4proj": "conv_glds_kernel"}


def kernel_of(op: str) -> str:
    # every 3x3 conv of layers 2-4 (stride 1, and the stride-2 entry convs as four plane bands): the halo kernel on
    # v_mfma_f32_16x16x32 (csrc/halo16.h) for bf16 / fp16
    return KERNEL_OF_OP.get(op) or KERNEL_OF_OP.get(op[:2]) or "conv3x3_halo16_kernel"


def host_threads() -> int:
    """CPU threads this process may really use: the smaller of the affinity mask and the cgroup CPU quota
    (a GPU box hands one GPU's share of a big host: os.cpu_count() there is the HOST's core count and
    running torch on that many threads oversubscribes the share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


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


def cpu_baseline(u8_256: torch.Tensor, budget_s: float = 45.0):
    """BASELINE configs[0] as SURVEY.md 8(d) states it: the oracle (torch fp32 functional ResNet18) on the host
    cores, batch 256 (then 512), 3 warm-up + up to 10 timed iterations, median, threads stated.  Bounded: the
    timed loops stop once `budget_s` of CPU work is spent (the iteration counts actually run are reported).
    Also returns the oracle's outputs on the 256 patches (the parity reference)."""
    from oracle import resnet18_ref, transform_ref  # the checker, timed as the reported baseline only

    threads = host_threads()
    torch.set_num_threads(threads)
    sd = synth.seeded_resnet18_state_dict(0, num_classes=2)
    lut = torch.from_numpy(transform_ref.normalize_lut())
    x = torch.stack([lut[c][u8_256[..., c].long()] for c in range(3)], dim=1)  # [256,3,224,224] fp32
    t_start = time.time()
    ref = resnet18_ref.resnet18_forward(x, sd)  # warm-up 1 (and the parity reference)
    runs = {}
    for batch, warm, timed in ((256, 2, 10), (512, 1, 5)):
        xb = x if batch == 256 else torch.cat([x, x])
        for _ in range(warm):
            resnet18_ref.resnet18_forward(xb, sd)
        ts = []
        for _ in range(timed):
            t0 = time.time()
            resnet18_ref.resnet18_forward(xb, sd)
            ts.append(time.time() - t0)
            if time.time() - t_start > budget_s and len(ts) >= 3:
                break
        ts.sort()
        runs[batch] = {"patches_per_s": batch / ts[len(ts) // 2], "timed_iterations": len(ts), "warmup": warm + (batch == 256)}
        if time.time() - t_start > budget_s:
            break
    dt = time.time() - t_start
    rec = {"value": runs[256]["patches_per_s"], "unit": "patches/s", "cores": threads, "kind": "port",
           "sample": f"configs[0]: 256 random 224x224x3 patches, fp32 torch-functional ResNet18 (oracle), batch 256, "
                     f"{runs[256]['warmup']} warm-up + {runs[256]['timed_iterations']} timed iterations, median; "
                     f"{dt:.1f} s of CPU work in all", "batches": runs, "host_cpu_count": os.cpu_count()}
    return rec, ref


def parity_block(nets: dict, u8_256_dev: torch.Tensor, ref):
    """Per precision: max norm-relative error (max|a-b| / max|b|) of features and logits against the fp32 oracle
    on the 256 patches of configs[0], label mismatches, and how many of those sit on oracle near-ties."""
    ref_f, ref_l = ref
    ref_lab = ref_l.argmax(1)
    margin = (ref_l[:, 0] - ref_l[:, 1]).abs()
    out = {"patches": int(u8_256_dev.shape[0]), "reference": "oracle/resnet18_ref.py (torch fp32, CPU)",
           "metric": "max|a-b| / max|b| per tensor"}
    for name, net in nets.items():
        f, l, lab = net.forward(u8_256_dev, want_feats=True, want_logits=True, want_labels=True)
        f, l, lab = f.cpu(), l.cpu(), lab.cpu()
        ef = float((f - ref_f).abs().max() / ref_f.abs().max())
        el = float((l - ref_l).abs().max() / ref_l.abs().max())
        bad = lab != ref_lab
        bound = 2 * float((l - ref_l).abs().max())
        out[name] = {"features": ef, "logits": el, "label_mismatches": int(bad.sum()),
                     "label_mismatches_outside_near_ties": int((bad & (margin > bound)).sum()),
                     "meets_1e-3": bool(ef <= 1e-3 and el <= 1e-3)}
    return out


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


def _max_over_ranks(dt: float, world: int, dev) -> float:
    if world == 1:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def planes_roofline(slide, side):
    """The HBM-bound kernel of the slide scan: the level-0 resampler (hipac_level_build_planes).  Algorithmic bytes
    = every level-0 source byte once (unique bytes); HIP events on the launch stream."""
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
    return {"bound": "hbm", "kernel": "hipac_level_build_planes[level 0]", "achieved": gbs, "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": traffic, "launch_ms": ms,
            "bytes_per_launch": w0 * h0 * 3}


def scan_slide_timed(net, slide, args, world, steps, warmup, dev):
    """Time `steps` whole-slide scans (levels 0-3).  Returns (seconds per slide, windows, kept)."""
    from ss25_hierarchical_multiscale_image_classification_amd import extract

    stride_of = (lambda lvl: None) if args.grid == "reference" else (lambda lvl: extract.PATCH_SIZES[lvl])
    n_all = 0
    for lvl in (0, 1, 2, 3):
        w, h = slide.level_dimensions[lvl]
        n_all += len(extract.window_grid(w, h, lvl, stride_of(lvl))[2])

    def step():
        f, l, p, meta = extract.score_slide(slide, net, levels=(0, 1, 2, 3), batch_windows=args.batch_windows,
                                            stride=stride_of)
        n_local = f.shape[0]
        if world > 1:  # configs[3]: the one exchange of the path -- ragged all-gather, rank-major order
            f, l, meta = hdist.gather_results(f, l, meta)
        return n_local

    torch.cuda.synchronize()
    t_cold = time.perf_counter()
    for i in range(warmup):
        step()
        if i == 0:  # the first scan of a slide: + the annotation mask raster (host, Pillow), the window tables, first allocations
            torch.cuda.synchronize()
            scan_slide_timed.cold_s = time.perf_counter() - t_cold
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for _ in range(steps):
        n_kept = step()
        if world == 1:  # per-scan wall clock as well (diagnostic: a one-off cost shows as one slow scan)
            torch.cuda.synchronize()
            marks.append(time.perf_counter())
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0  # this rank's own scans + exchanges, before it waits for the others
    scan_slide_timed.rank_stats = None
    if world > 1:
        torch.distributed.barrier()
    dt = _max_over_ranks(time.perf_counter() - t0, world, dev)
    if world > 1:
        # load balance of one-slide-per-rank: every rank's kept windows and own time (the only thing between this
        # partitioning and linear scaling is the spread of these)
        t = torch.tensor([dt_local / steps, float(n_kept)], dtype=torch.float64,
                         device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        lst = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(lst, t)
        secs = [float(x[0]) for x in lst]
        scan_slide_timed.rank_stats = {"kept_per_rank": [int(x[1]) for x in lst], "rank_s_per_slide": [round(v, 5) for v in secs],
                                       "rank_s_max": max(secs), "rank_s_min": min(secs)}
    scan_slide_timed.last_scans = [b - a for a, b in zip([t0] + marks[:-1], marks)]
    return dt / steps, n_all, n_kept


def wsi_object(net, args, rank, world, dev, sides):
    """The whole-WSI half of the metric inside the default line: per slide side a record of the scan."""
    from ss25_hierarchical_multiscale_image_classification_amd import extract

    out = {"levels": "0-3", "grid": args.grid, "dtype": "u8+" + args.precision,
           "step": "one whole synthetic slide per rank: window decisions + Pillow-exact resize + ResNet18"}
    for side in sides:
        key = f"{side}x{side}"
        try:
            slide = extract.DeviceSlide.synthetic(side, side, seed=10 + rank, with_polygons=True)
            s_per, n_all, n_kept = scan_slide_timed(net, slide, args, world, steps=2, warmup=1, dev=dev)
            rec = {"s_per_slide": s_per, "scans_s": [round(t, 4) for t in getattr(scan_slide_timed, "last_scans", [])],
                   "first_scan_s": round(getattr(scan_slide_timed, "cold_s", float("nan")), 4),
                   "windows": n_all, "kept": n_kept, "n_gpus": world, "slides": world,
                   "kept_patches_per_s": world * n_kept / s_per if world == 1 else None,
                   "unique_source_GBps": world * sum(w * h * 3 for (w, h) in slide.level_dimensions) / s_per / 1e9}
            if getattr(scan_slide_timed, "rank_stats", None):
                rec.update(scan_slide_timed.rank_stats)
                rec["kept_all_ranks"] = sum(rec["kept_per_rank"])
                rec["kept_patches_per_s"] = rec["kept_all_ranks"] / s_per
            if rank == 0:
                rec["roofline"] = planes_roofline(slide, side)
            out[key] = rec
            del slide
            torch.cuda.empty_cache()
        except (RuntimeError, capi.HipacError) as e:  # e.g. out of memory on a smaller card: report, do not die
            out[key] = {"error": str(e)[:300]}
            torch.cuda.empty_cache()
    return out


def run_resnet(args, rank, world, dev):
    B = args.batch
    sd = synth.seeded_resnet18_state_dict(0, num_classes=2)
    net = capi.PackedResNet18(sd, precision=args.precision)
    # synthetic uint8 patches resident in HBM: a pool of `pool` distinct batches cycled over the steps
    pool = max(1, min(args.steps, 8))
    data = [synth.synth_patches_u8(B, seed=1 + rank * 1000 + i, device=dev) for i in range(pool)]

    def step(i):
        u8 = data[i % pool]  # uint8[B,224,224,3] in HBM; ToTensor/Normalize is folded into the stem kernel
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
    dt = _max_over_ranks(time.perf_counter() - t0, world, dev)
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
        "frac_of_peak_whole_path": value * FLOP_PER_PATCH / 1e12 / world / PEAK_BF16_DENSE_TFLOPS,
    }
    if rank == 0:
        ops = per_op_times(net, dev)
        convs = [o for o in ops if o["tflops"]]
        dom = max(convs, key=lambda o: o["us_per_image"])  # the kernel that costs most per patch
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(dom["op"])
        rec["roofline"] = {"bound": "mfma", "kernel": f"{kernel_of(dom['op'])}[{dom['op']}]", "achieved": dom["tflops"],
                           "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": dom["tflops"] / PEAK_BF16_DENSE_TFLOPS,
                           "traffic": traffic, "launch_ms": dom["ms"], "flops_per_launch": dom["flops_per_launch"],
                           "images_per_launch": dom["images"]}
        # (the op slots a fused kernel leaves empty are not work: their "time" is the event pair's own overhead -- left out)
        # frac = of the dense MFMA peak; hbm_gbs / hbm_frac = the op's algorithmic bytes (op_bytes) over its time, of the 8 TB/s peak
        rec["per_op"] = [{"op": o["op"], "kernel": kernel_of(o["op"]), "images": o["images"],
                          "ms": round(o["ms"], 4), "tflops": round(o["tflops"], 1),
                          "frac": round(o["tflops"] / PEAK_BF16_DENSE_TFLOPS, 3),
                          "hbm_gbs": round(op_bytes(o["op"]) * o["images"] / (o["ms"] * 1e-3) / 1e9, 0),
                          "hbm_frac": round(op_bytes(o["op"]) * o["images"] / (o["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 3)} for o in ops if o["tflops"]]
        if traffic is not None:
            tj = json.load(open(tpath))
            rec["roofline"]["traffic_source"] = ("profiles/roofline_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes), "
                                                 f"collected at commit {tj.get('_commit', 'unrecorded')}: stale if that kernel changed since")
    nets = {args.precision: net}
    if rank == 0 and world == 1:
        # the same kernels instantiated for the other operand type, and the exact f32 MFMA parity mode
        for alt_name in ("fp16", "bf16"):
            if alt_name == args.precision:
                continue
            alt = capi.PackedResNet18(sd, precision=alt_name)
            for i in range(max(1, args.warmup)):
                alt.forward(data[i % pool], want_feats=True, want_logits=True, want_labels=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                alt.forward(data[i % pool], want_feats=True, want_logits=True, want_labels=True)
            torch.cuda.synchronize()
            rec["alt_precision"] = {"dtype": alt_name, "value": B * args.steps / (time.perf_counter() - t0),
                                    "unit": "patches/s"}
            nets[alt_name] = alt
        # the parity modes (fp16 (hi, lo) pairs; per-patch results within 1e-3 of the reference's fp32 path) -- the same workload, the
        # same number of steps.  `parity_mode` is the faster one: hi*hi on the fp16 MFMA, the two cross products of every 3x3 / stride 1
        # conv on the e4m3 MX MFMA (fp16q8); `parity_mode_tight` spends three fp16 products per term (fp16x3)
        def timed_mode(prec):
            netp = capi.PackedResNet18(sd, precision=prec)
            for i in range(max(1, min(2, args.warmup))):
                netp.forward(data[i % pool], want_feats=True, want_logits=True, want_labels=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                netp.forward(data[i % pool], want_feats=True, want_logits=True, want_labels=True)
            torch.cuda.synchronize()
            nets[prec] = netp
            return B * args.steps / (time.perf_counter() - t0)
        vq8 = timed_mode("fp16q8")
        rec["parity_mode"] = {"dtype": "fp16q8", "value": vq8, "unit": "patches/s", "patches": B * args.steps,
                              "tflops_network_arithmetic": vq8 * FLOP_PER_PATCH / 1e12,
                              "note": "every product of the 13 3x3 / stride 1 convs = hi*hi on v_mfma_f32_16x16x32_f16 + (hi*lo + lo*hi) on "
                                      "v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3, constant scales): 2 MFMA time units per term; stem and the "
                                      "stride-2 entry convs as in fp16x3; meets_1e-3 is measured in `parity`"}
        vx3 = timed_mode("fp16x3")
        rec["parity_mode_tight"] = {"dtype": "fp16x3", "value": vx3, "unit": "patches/s", "patches": B * args.steps,
                                    "tflops_network_arithmetic": vx3 * FLOP_PER_PATCH / 1e12,
                                    "tflops_mfma_issued": 3 * vx3 * FLOP_PER_PATCH / 1e12,
                                    "frac_of_f16_mfma_peak": 3 * vx3 * FLOP_PER_PATCH / 1e12 / PEAK_BF16_DENSE_TFLOPS,
                                    "note": "every product = hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 (halo16x2.h, X3 form; 2 products in the "
                                            "stem: bytes are exact); meets_1e-3 is measured in `parity`"}
        # the debugging reference: fp32 storage, exact f32 MFMA
        net32 = capi.PackedResNet18(sd, precision="fp32")
        n32 = min(B, 2048)
        net32.forward(data[0][:n32].contiguous(), want_feats=True, want_logits=True, want_labels=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net32.forward(data[0][:n32].contiguous(), want_feats=True, want_logits=True, want_labels=True)
        torch.cuda.synchronize()
        v32 = n32 / (time.perf_counter() - t0)
        rec["debug_reference_mode"] = {"dtype": "fp32", "value": v32, "unit": "patches/s", "patches": n32,
                                       "tflops": v32 * FLOP_PER_PATCH / 1e12, "frac_of_f32_mfma_peak": v32 * FLOP_PER_PATCH / 1e12 / PEAK_F32_MFMA_TFLOPS}
        rec["parity_mode"]["vs_fp32_mode"] = vq8 / v32
        rec["parity_mode_tight"]["vs_fp32_mode"] = vx3 / v32
        nets["fp32"] = net32
        rec["pcie_inclusive"] = pcie_inclusive(net, data[0], steps=min(4, args.steps))
        if not args.no_cpu_baseline:
            u8_256 = data[0][:256].contiguous()
            rec["cpu_baseline"], ref = cpu_baseline(u8_256.cpu())
            rec["parity"] = parity_block(nets, u8_256, ref)
    del data
    torch.cuda.empty_cache()
    if not args.no_wsi:
        sides = [int(s) for s in args.wsi_sides.split(",") if s]
        if world > 1:
            sides = sides[:1]  # configs[3]: one 50k slide per rank, gathered
        w = wsi_object(net, args, rank, world, dev, sides)
        if rank == 0 and world == 1 and "fp16q8" in nets and sides:
            # the same scan in the parity mode (fp16q8: per-patch results within 1e-3 of the reference's fp32 path), first side only
            from ss25_hierarchical_multiscale_image_classification_amd import extract
            try:
                slide = extract.DeviceSlide.synthetic(sides[0], sides[0], seed=10, with_polygons=True)
                s_per, n_all, n_kept = scan_slide_timed(nets["fp16q8"], slide, args, 1, steps=1, warmup=1, dev=dev)
                w["parity_mode"] = {"dtype": "u8+fp16q8", "side": sides[0], "s_per_slide": s_per, "kept": n_kept,
                                    "kept_patches_per_s": n_kept / s_per}
                del slide
            except (RuntimeError, capi.HipacError) as e:
                w["parity_mode"] = {"error": str(e)[:300]}
            torch.cuda.empty_cache()
        if rank == 0 and world == 1 and not args.no_tiff:
            w["tiff_ingest"] = tiff_ingest_object(dev, net)
        if rank == 0:
            if world == 1 and not args.no_cpu_baseline:
                w["cpu_baseline"] = cpu_baseline_wsi()
            rec["wsi"] = w
    if not args.no_simclr:
        # configs[4], bounded: the native SimCLR step on 2 x 256 views over all ranks, a few steps (the full 2 x 1024 is
        # `--workload simclr`); same record as that workload, with its roofline.  N > 1: data parallel -- all-gather of z,
        # all-reduce of the gradients (RCCL) -- under a watchdog, so that a stuck collective costs this object, not the line
        import argparse as _ap

        from ss25_hierarchical_multiscale_image_classification_amd import train_native

        def simclr_object():
            torch.cuda.set_device(dev)
            sargs = _ap.Namespace(simclr_views=256, warmup=1, steps=3, train_precision="fp32")
            out = train_native.bench_simclr_step(sargs, rank, world, dev)  # the reference's arithmetic for this loop
            sargs.train_precision = "fp16"  # mixed precision (the reference's autocast arithmetic of the fine-tune loops)
            out["mixed_precision"] = train_native.bench_simclr_step(sargs, rank, world, dev)
            if world == 1:  # a-13: the classifier loops' fine-tune step at the reference's batch size, its autocast arithmetic
                out["classifier_step"] = train_native.bench_classifier_step(512, 3, 1, dev, "fp16")
            if world == 1:  # where the step's input comes from: views made on the device vs the host's Pillow transforms
                from ss25_hierarchical_multiscale_image_classification_amd import augment
                out["input_pipeline"] = augment.bench_input_pipeline(n_pairs=256, P=224, steps=5, device=dev)
            return out

        try:
            if world == 1:
                obj = simclr_object()
            else:
                import concurrent.futures as _cf

                fut = _cf.ThreadPoolExecutor(1).submit(simclr_object)
                try:
                    obj = fut.result(timeout=args.simclr_timeout)
                except _cf.TimeoutError:
                    obj = {"error": f"no result within {args.simclr_timeout:.0f} s (a collective did not return)"}
                    rec["_abandon_process_group"] = True  # main(): print the line and leave without the group's teardown
        except (RuntimeError, capi.HipacError) as e:
            obj = {"error": str(e)[:300]}
        if rank == 0:
            rec["simclr"] = obj
        torch.cuda.empty_cache()
    return rec


def tiff_ingest_object(dev, net=None, side=20000):
    """How a slide FILE gets into HBM: a synthetic `side`^2 pyramid written as a tiled TIFF with JPEG tiles (512 x 512, 4:2:0, one
    JPEGTables set per level -- the CAMELYON16 layout), loaded with the tiles decoded on the device (hipac_jpeg_decode_tiles) and,
    for comparison, on the host's threads (Pillow / libjpeg-turbo).  Both give the same bytes (tests/test_gpu_jpeg.py)."""
    import tempfile

    from ss25_hierarchical_multiscale_image_classification_amd import extract, tiff_pyramid

    try:
        l0 = synth.synth_level0(side, side, seed=2, device=dev)
        levels = [t.cpu().numpy() for t in synth.build_pyramid(l0, 4)]
        del l0
        with tempfile.TemporaryDirectory(prefix="hipac_tiff_") as d:
            path = os.path.join(d, "slide.tif")
            tiff_pyramid.write_tiled_tiff(path, levels, tile=512, compression="jpeg", jpeg_tables=True)
            px = sum(a.shape[0] * a.shape[1] for a in levels)
            del levels
            out = {"side": side, "file_MB": os.path.getsize(path) / 1e6, "tile": 512, "pixels_all_levels": px}
            for key, flag in (("device_decode_s", True), ("host_decode_s", False)):
                ts = []
                for _ in range(2 if flag else 1):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    tp = tiff_pyramid.TiffPyramid(path)
                    lv = tp.to_device_levels(dev, list(range(min(4, tp.level_count))), workers=host_threads(), device_jpeg=flag)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                    del lv
                out[key] = min(ts)
            if net is not None:  # the whole way: file on disk -> levels in HBM -> every window of levels 0-3 scored (a first scan)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                slide = extract.DeviceSlide.from_tiff(path, device=dev, workers=host_threads())
                f, _, _, _ = extract.score_slide(slide, net, levels=(0, 1, 2, 3))
                torch.cuda.synchronize()
                out["file_to_scores_s"], out["kept_windows"] = time.perf_counter() - t0, int(f.shape[0])
                del slide, f
            out["device_Gpx_per_s"] = px / out["device_decode_s"] / 1e9
            out["host_Gpx_per_s"] = px / out["host_decode_s"] / 1e9
            out["host_threads"] = host_threads()
        torch.cuda.empty_cache()
        return out
    except (RuntimeError, capi.HipacError, OSError) as e:
        torch.cuda.empty_cache()
        return {"error": str(e)[:300]}


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
    """BASELINE.json configs[2] (N = 1) / configs[3] (N > 1): hierarchical scan of one synthetic level-0 slide per
    rank across levels 0-3 with on-GPU whiteness filter + resize; one STEP = one whole slide per rank; with N > 1
    features + logits + meta of every rank are all-gathered (rank-major order = DataParallel's gather order)."""
    from ss25_hierarchical_multiscale_image_classification_amd import extract

    side = args.slide_side
    net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision=args.precision)
    slide = extract.DeviceSlide.synthetic(side, side, seed=10 + rank, with_polygons=True)
    s_per, n_all, n_kept = scan_slide_timed(net, slide, args, world, args.steps, args.warmup, dev)
    unique_bytes = sum(w * h * 3 for (w, h) in slide.level_dimensions)
    extra = dict(getattr(scan_slide_timed, "rank_stats", None) or {})
    if rank == 0:
        extra["roofline"] = planes_roofline(slide, side)
        if world == 1 and not args.no_cpu_baseline:
            extra["cpu_baseline"] = cpu_baseline_wsi()
    return {**extra,
        "metric": "whole-WSI wall-clock (levels 0-3)", "value": s_per, "unit": "s/slide", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": s_per * 1e3, "higher_is_better": False,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8+" + args.precision, "data": "synthetic",
        "config": {"workload": f"wsi_scan_{side}x{side}_levels0-3_{args.grid}_grid", "windows": n_all,
                   "kept_rank0": n_kept, "parallelism": f"one slide per rank x{world}, ragged all-gather of features/logits/meta"
                   if world > 1 else "single GPU"},
        "slides_per_s": world / s_per, "windows_per_s": world * n_all / s_per,
        "unique_source_GBps": world * unique_bytes / s_per / 1e9,
    }


def run_simclr(args, rank, world, dev):
    """BASELINE.json configs[4]: the native SimCLR training step (src/models/simclr.py:85-96): two train-mode
    encoder + projector forwards on augmented-pair batches, NT-Xent over the global batch, backward, Adam.
    One STEP = 2 x (1024 / N) views per rank; gradients are all-reduced over RCCL."""
    from ss25_hierarchical_multiscale_image_classification_amd import train_native

    rec = train_native.bench_simclr_step(args, rank, world, dev)
    if world == 1:  # what feeds the step: both views made on the device from patches resident in HBM, next to the host's Pillow path
        from ss25_hierarchical_multiscale_image_classification_amd import augment
        rec["input_pipeline"] = augment.bench_input_pipeline(n_pairs=min(args.simclr_views, 1024), P=224, steps=5, device=dev)
    return rec


# ---------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` from a bare shell
# ---------------------------------------------------------------------------------------------------------

def _free_port() -> int:
    return hlaunch.free_port()


def child_commands(argv, n: int, port: int, python: str = sys.executable, base_env=None):
    """(argv, env) of every rank's process: the same command line plus --_child, torchrun's environment
    variables, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    return hlaunch.child_commands([os.path.join(ROOT, "bench.py")], argv, n, port, python=python, base_env=base_env)


def launch_ranks(argv, n: int, rank_timeout=None) -> int:
    """Start one fresh process per rank BEFORE this process touches the GPU, relay rank 0's stdout; every child is
    polled, and when one exits non-zero the others are terminated and its code is returned (launch.py)."""
    return hlaunch.launch_ranks(child_commands(argv, n, _free_port()), rank_timeout=rank_timeout, name="bench.py")


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["resnet", "wsi", "simclr"], default="resnet")
    ap.add_argument("--precision", choices=["bf16", "fp16"], default="bf16")
    ap.add_argument("--batch", type=int, default=8192, help="patches per step per GPU")
    ap.add_argument("--slide_side", type=int, default=50000)
    ap.add_argument("--wsi_sides", default="50000,100000", help="slide sides of the `wsi` object of the default line")
    ap.add_argument("--no_wsi", action="store_true", help="default workload: skip the `wsi` object")
    ap.add_argument("--no_simclr", action="store_true", help="default workload: skip the bounded `simclr` object (configs[4])")
    ap.add_argument("--grid", choices=["reference", "nonoverlap"], default="reference")
    ap.add_argument("--batch_windows", type=int, default=4096)
    ap.add_argument("--simclr_views", type=int, default=1024, help="simclr: images per view per step, over all ranks")
    ap.add_argument("--train_precision", choices=["fp32", "fp16"], default="fp32",
                    help="simclr workload: fp32 (the reference's loop) or fp16 mixed precision (autocast-style)")
    ap.add_argument("--simclr_timeout", type=float, default=240.0, help="N > 1: seconds the bounded `simclr` object may take")
    ap.add_argument("--no_tiff", action="store_true", help="default workload: skip the `wsi.tiff_ingest` object")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="process-group backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--one_device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo; RCCL wants one GPU per rank)")
    ap.add_argument("--rank_timeout", type=float, default=1500.0,
                    help="--gpus > 1 from a bare shell: seconds after which the launcher gives up and terminates every rank")
    ap.add_argument("--_child", action="store_true", help=argparse.SUPPRESS)
    return ap


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not under_launcher:
        # parent: nothing here may initialise the GPU (torch.cuda.is_available() would)
        return launch_ranks(argv, args.gpus, rank_timeout=args.rank_timeout)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the HIP path has no CPU fallback")
    if args.one_device:
        os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
    rank, world, local = hdist.init_from_env(args.backend if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", 0 if args.one_device else local)
    torch.cuda.set_device(dev)
    run = {"resnet": run_resnet, "wsi": run_wsi, "simclr": run_simclr}[args.workload]
    rec = run(args, rank, world, dev)
    abandon = bool(rec.pop("_abandon_process_group", False))
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if abandon:
        # a collective is still stuck in a worker thread: the line is out (with its "error" object), the group is not waited
        # for -- and the process reports the fault: a hung collective is a failure the launcher and the driver must see
        # (launch_ranks names the rank and terminates its siblings), never a silent rc 0
        sys.stdout.flush()
        sys.stderr.write(f"[bench] rank {rank}: a collective did not return within --simclr_timeout; exiting with code 3\n")
        sys.stderr.flush()
        os._exit(3)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
