"""Command line mirroring the hot-path flags of the reference's ``src/main.py``
(:1073-1166): ``--patch``, ``--patch_level {0,1,2,3,all}``, ``--extract_features``,
``--train``, ``--train_strategy``, ``--strategy {balanced,weighted_loss,self_supervised}``.

Layout under ``--data_root`` (default ``./data/camelyon16``, README.md:142-164):
    train/img/<slide>.{npz,tif}      slides   (.tif: tiled pyramidal TIFF / BigTIFF, levels 0-3 are used;
                                              .npz: keys level0..levelN uint8[H,W,3])
    train/mask/annotations/<slide>.xml
    patches/level_L/<slide>/...      extractor output
Outputs of ``--extract_features`` are written to the current directory exactly like
the reference (patch_features_L.npy, patch_labels_L.npy, patch_paths_L.txt).

Additive flags: ``--data_root``, ``--synthetic W,H,SEED[,NAME]`` (repeatable; a slide
made on the GPU instead of a file), ``--write_png`` (also emit the reference's PNG
tree), ``--precision {bf16,fp16,fp16x3,fp16q8,fp32}``, ``--weights PATH``, ``--stride N``,
``--world_size N`` (one process per GPU over RCCL where the reference wraps its model in
nn.DataParallel, src/main.py:481-482, :841-842: ``--patch`` / ``--extract_features`` shard the
slides, ``--train*`` the batches; started by this program itself before it touches a GPU).

Everything outside the hot path (download, FROC, plots, MIL) is out of scope and the
corresponding reference flags are accepted but answered with a clear message.
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import List, Optional

import numpy as np
import torch

OUT_OF_SCOPE = ("download", "remote", "prepare", "validation", "validate", "evaluate", "run_evaluation",
                "balance_dataset", "count_tumor_patches", "patch_one_slide", "slide", "move_files",
                "check_good_downloaded_files")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="HiPAC hot path on MI355X (patch extraction + ResNet18 scoring)")
    # --- the reference's surface (src/main.py:1075-1093) ---
    p.add_argument("-p", "--patch", action="store_true", help="Extract patches")
    p.add_argument("--patch_level", type=str, default="3", help="WSI level (0, 1, 2, 3, or 'all')")
    p.add_argument("--extract_features", action="store_true", help="Extract features from patches")
    p.add_argument("-train", "--train", action="store_true", help="Train ResNet18 patch classifier")
    p.add_argument("--train_strategy", action="store_true", help="Train with a specific strategy")
    p.add_argument("--strategy", type=str, default="self_supervised",
                   choices=["balanced", "weighted_loss", "self_supervised"])
    for name in OUT_OF_SCOPE:
        if name in ("patch_one_slide", "slide"):
            p.add_argument(f"--{name}", type=str, default=None, help="(reference flag; out of scope here)")
        else:
            p.add_argument(f"--{name}", action="store_true", help="(reference flag; out of scope here)")
    # --- additive ---
    p.add_argument("--data_root", type=str, default=None)
    p.add_argument("--synthetic", action="append", default=[], metavar="W,H,SEED[,NAME]")
    p.add_argument("--write_png", action="store_true")
    p.add_argument("--precision", choices=["bf16", "fp16", "fp16x3", "fp16q8", "fp32"], default="bf16",
                   help="MFMA operand type; fp16x3 = parity mode (fp16 pairs, three products per term: the reference's fp32 "
                        "results to 1e-3 at ~1/3 of the bf16 throughput); fp16q8 = the faster parity mode (the same pairs, cross "
                        "products on the e4m3 MX MFMA: logits within ~1e-5); fp32 = debugging reference (exact f32 MFMA, ~1/9)")
    p.add_argument("--weights", type=str, default=None, help="state_dict (.pth, any reference key layout)")
    p.add_argument("--stride", type=int, default=None, help="window stride (default: the reference's 224)")
    p.add_argument("--epochs", type=int, default=None)
    p.add_argument("--simclr_epochs", type=int, default=200, help="pre-training epochs of --strategy self_supervised (:557)")
    p.add_argument("--simclr_encoder", type=str, default=None,
                   help="--extract_features with a SimCLR encoder checkpoint (extract_features_with_simclr, src/main.py:897-932)")
    p.add_argument("--batch_size", type=int, default=512)  # BATCH_SIZE, src/main.py:46
    p.add_argument("--world_size", type=int, default=1,
                   help="number of GPUs = processes (the reference uses every visible GPU through nn.DataParallel)")
    p.add_argument("--dist_backend", default="nccl", help="process-group backend of --world_size > 1 (nccl = RCCL)")
    p.add_argument("--one_device", action="store_true",
                   help="rehearsal only: every rank uses cuda:0 (needs --dist_backend gloo; RCCL wants one GPU per rank)")
    p.add_argument("--rank_timeout", type=float, default=None, help="--world_size > 1: give up after this many seconds")
    p.add_argument("--seed", type=int, default=None, help="seed of the host-side random draws (dataset shuffles, model init); "
                                                       "with --world_size > 1 every rank uses the same one (default 0 there)")
    p.add_argument("--max_steps", type=int, default=None, help="bound the training loops (tests)")
    p.add_argument("--train_precision", choices=["fp16", "fp32"], default="fp16",
                   help="arithmetic of the classifier training step: fp16 = the reference's autocast + GradScaler "
                        "(src/main.py:499-508), fp32 = exact f32 MFMA")
    p.add_argument("--device_aug", action="store_true",
                   help="training loops: keep the decoded patches in HBM and make every batch's augmented views on the device "
                        "(hipac_augment_views; classifier loops: 224-pixel patches) instead of in DataLoader workers")
    p.add_argument("--from_slides", action="store_true",
                   help="--train / --train_strategy: no PNG tree -- the kept windows go from the slides in HBM straight into the "
                        "training pool (implies the device input pipeline)")
    p.add_argument("--simclr_precision", choices=["fp16", "fp32"], default="fp32",
                   help="arithmetic of the SimCLR pre-training step (the reference's loop is fp32, src/models/simclr.py:85-96)")
    p.add_argument("--_child", action="store_true", help=argparse.SUPPRESS)
    return p


def data_root(args) -> str:
    return args.data_root or os.path.join(os.getcwd(), "data", "camelyon16")


def levels_of(args) -> List[int]:
    return [0, 1, 2, 3] if args.patch_level == "all" else [int(args.patch_level)]


def list_slides(args):
    """[(name, opener)] of every slide in processing order -- synthetic specs first, then the files in train/img --
    without loading any: ``opener()`` puts the slide into HBM (a rank opens only the slides it owns)."""
    from .extract import DeviceSlide, parse_annotation_xml

    out = []
    for spec in args.synthetic:
        parts = spec.split(",")
        w, h, seed = int(parts[0]), int(parts[1]), int(parts[2])
        name = parts[3] if len(parts) > 3 else f"synthetic_{seed}"
        out.append((name, lambda w=w, h=h, seed=seed, name=name: DeviceSlide.synthetic(w, h, seed=seed, name=name)))
    img_dir = os.path.join(data_root(args), "train", "img")
    if not os.path.isdir(img_dir):
        return out
    ann_dir = os.path.join(data_root(args), "train", "mask", "annotations")

    def open_file(path, stem, ext):
        if ext == ".npz":
            z = np.load(path)
            levels = [torch.from_numpy(z[f"level{i}"]) for i in range(len(z.files)) if f"level{i}" in z.files]
            slide = DeviceSlide(levels, name=stem)
        else:
            # tiled pyramidal TIFF / BigTIFF (the CAMELYON16 container): own reader, openslide is not needed
            slide = DeviceSlide.from_tiff(path, name=stem)
        xml = os.path.join(ann_dir, stem + ".xml")
        if os.path.exists(xml):
            try:  # a bad annotation file costs the slide its tumour labels, not the run (src/main.py:670-675)
                slide.polygons = parse_annotation_xml(xml)
            except Exception as e:
                print(f"[WARNING] Failed to parse XML for {os.path.basename(path)}: {e}")
        return slide

    for file in sorted(os.listdir(img_dir)):
        stem, ext = os.path.splitext(file)
        if ext in (".npz", ".tif", ".tiff"):
            out.append((stem, lambda p=os.path.join(img_dir, file), stem=stem, ext=ext: open_file(p, stem, ext)))
    return out


def try_open(name: str, make):
    """The slide, or None after ``[ERROR] Could not open ...`` -- the reference opens every slide under try / except ...
    continue (src/main.py:649-653): one unreadable file costs that slide, not the run (under --world_size N: not the
    other N - 1 ranks either; the owning rank contributes zero rows for it)."""
    try:
        return make()
    except Exception as e:  # noqa: BLE001 -- whatever the reader raises for a truncated / foreign file
        print(f"[ERROR] Could not open {name}: {type(e).__name__}: {e}", flush=True)
        if int(os.environ.get("RANK", "0")) != 0:  # the launcher relays rank 0's stdout only: the other ranks report on stderr too
            print(f"[ERROR] rank {os.environ['RANK']}: could not open {name}: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        return None


def open_slides(args, rank: int = 0, world: int = 1):
    """Yield (index, DeviceSlide) of the slides this rank owns (slide i -> rank i mod world); slides that
    cannot be opened are reported and skipped."""
    from .dist import shard_units

    slides = list_slides(args)
    for i in shard_units(len(slides), rank, world):
        slide = try_open(slides[i][0], slides[i][1])
        if slide is not None:
            yield i, slide


def cmd_patch(args):
    from .dist import rank_world
    from .extract import save_patch_pngs, scan_level

    rank, world = rank_world()  # N > 1: every rank extracts the slides it owns; the outputs are per-slide directories
    for _, slide in open_slides(args, rank, world):
        for level in levels_of(args):
            level_dir = os.path.join(data_root(args), "patches", f"level_{level}")
            save_dir = os.path.join(level_dir, slide.name)
            if os.path.isdir(save_dir) and os.listdir(save_dir):  # src/main.py:634-640
                print(f"[INFO] Patches for {slide.name} already extracted, skipping.")
                continue
            os.makedirs(save_dir, exist_ok=True)
            scan = scan_level(slide, level, stride=args.stride)
            np.savez(os.path.join(save_dir, "manifest.npz"), xy=scan.xy.cpu().numpy(),
                     keep=scan.keep.cpu().numpy(), labels=scan.labels.cpu().numpy(),
                     sums=scan.sums.cpu().numpy().astype(np.uint32), patch_size=scan.patch_size, level=level)
            n = int(scan.keep.sum().item())
            if args.write_png:
                save_patch_pngs(slide, level, level_dir, stride=args.stride)
            print(f"[INFO] Patch extraction complete for {slide.name} at level {level}. Total patches: {n}")


def load_net(args, num_classes: Optional[int] = None):
    from . import capi, synth
    from .weights import canonical_state_dict

    if args.weights:
        sd = canonical_state_dict(torch.load(args.weights, map_location="cpu", weights_only=True))
    else:
        print("[WARNING] no --weights given: using seeded random-init ResNet18 "
              "(the reference's own transplant loads nothing either, SURVEY.md F4)")
        sd = synth.seeded_resnet18_state_dict(0, num_classes=num_classes)
    if num_classes is None:
        sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}
    return capi.PackedResNet18(sd, precision=args.precision)


def cmd_extract_features(args):
    from .features import extract_features_from_pngs, save_feature_files

    level = int(args.patch_level) if args.patch_level != "all" else 3  # src/main.py:1134
    if args.simclr_encoder:  # extract_features_with_simclr: UnifiedResNet(encoder checkpoint, classifier=False)
        args.weights = args.simclr_encoder
    net = load_net(args, num_classes=None)
    patch_dir = os.path.join(data_root(args), "patches", f"level_{level}")
    has_png = os.path.isdir(patch_dir) and any(
        f.endswith(".png") for _, _, fs in os.walk(patch_dir) for f in fs)
    from .dist import rank_world, score_sharded
    from .extract import score_slide

    rank, world = rank_world()
    if has_png:
        if world > 1 and rank == 0:
            print("[INFO] PNG patch tree: scored by rank 0 (the fused slide path is the one that shards)")
        if rank != 0:
            return 0
        feats, labels, paths = extract_features_from_pngs(patch_dir, net, batch_size=args.batch_size)
    else:
        # fused path: slides sharded over the ranks, one ragged all-gather, rows back in slide order -- the files rank 0
        # writes are the single-process files (DataParallel's gather order, src/main.py:841-842, :870-893)
        slides = list_slides(args)
        if not slides:
            print(f"[ERROR] Patches must be extracted at level {level} before extracting features.")
            return 1

        def score(i):
            slide = try_open(slides[i][0], slides[i][1])
            if slide is None:  # zero rows for this unit; the exchange and the other slides go on
                dev = torch.device("cuda", torch.cuda.current_device())
                return torch.zeros((0, 512), dtype=torch.float32, device=dev), None, torch.zeros((0, 4), dtype=torch.int32, device=dev)
            f, _, _, meta = score_slide(slide, net, levels=(level,), batch_windows=512, stride=args.stride,
                                        want_logits=False)
            return f, None, meta

        f_all, _, meta = score_sharded(len(slides), score, rank, world)
        if rank != 0:
            return 0
        m = meta.cpu().numpy()
        feats, labels = f_all.cpu(), m[:, 3].astype(np.int64)
        names = [slides[u][0] for u in m[:, 4]]
        paths = [f"{n}/{n}_x{x}_y{y}_{'tumor' if lab else 'normal'}.png" for n, (x, y, lab) in zip(names, m[:, 1:4])]
    save_feature_files(level, feats.numpy(), labels, paths)
    print(f"[INFO] Features saved to patch_features_{level}.npy ({feats.shape[0]} patches)")
    return 0


def cmd_train(args, strategy: Optional[str]):
    from .train import train_resnet_classifier

    level = int(args.patch_level) if args.patch_level != "all" else 3
    patch_dir = os.path.join(data_root(args), "patches", f"level_{level}")
    if args.from_slides:
        # every rank holds every slide: batches are shared out, not slides (an unreadable file is skipped on every rank alike)
        slides = [s for s in (try_open(name, make) for name, make in list_slides(args)) if s is not None]
        train_resnet_classifier(None, strategy=strategy, epochs=args.epochs, batch_size=args.batch_size, precision=args.precision,
                                simclr_epochs=args.simclr_epochs, max_steps=args.max_steps, train_precision=args.train_precision,
                                simclr_precision=args.simclr_precision, slides=slides, level=level)
        return 0
    if not (os.path.isdir(patch_dir) and os.listdir(patch_dir)):
        print("[ERROR] Patches must be extracted before training.")
        return 1
    train_resnet_classifier(patch_dir, strategy=strategy, epochs=args.epochs, batch_size=args.batch_size,
                            precision=args.precision, simclr_epochs=args.simclr_epochs, max_steps=args.max_steps,
                            train_precision=args.train_precision, simclr_precision=args.simclr_precision,
                            device_aug=args.device_aug)
    return 0


def _seed_everything(seed: int):
    import random

    random.seed(seed)
    np.random.seed(seed & 0x7FFFFFFF)
    torch.manual_seed(seed)


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    for name in OUT_OF_SCOPE:
        if getattr(args, name):
            print(f"[ERROR] --{name} is outside the accelerated hot path (see DESIGN.md 'Out of scope').")
            return 2
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.world_size > 1 and not under_launcher:
        # parent: start one fresh process per GPU and supervise them; nothing here may initialise the GPU
        from . import launch

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # importable from any working directory
        entry = ["-c", f"import sys; sys.path.insert(0, {root!r}); from {__package__}.main import main; sys.exit(main())"]
        cmds = launch.child_commands(entry, argv, args.world_size, launch.free_port())
        return launch.launch_ranks(cmds, rank_timeout=args.rank_timeout, name="main.py")
    if under_launcher and int(os.environ["WORLD_SIZE"]) > 1:
        from . import dist as hdist

        rank, world, local = hdist.init_from_env(args.dist_backend)
        if args.world_size not in (1, world):
            print(f"[ERROR] --world_size {args.world_size} but WORLD_SIZE={world}")
            return 2
        torch.cuda.set_device(0 if args.one_device else local)
        _seed_everything(0 if args.seed is None else args.seed)  # the same host-side draws on every rank
    elif args.seed is not None:
        _seed_everything(args.seed)
    try:
        return _dispatch(args)
    finally:
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


def _rendezvous():
    """--world_size N: the commands of one invocation run back to back on every rank, and the next one lists what the
    previous one wrote (the PNG tree, the feature files): nobody moves on before everybody is done.  Also keeps the ranks
    that sit out a rank-0-only phase (the PNG branch of --extract_features) out of the next command's collectives."""
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        torch.distributed.barrier()


def _dispatch(args) -> int:
    rc = 0
    if args.patch:
        cmd_patch(args)
        _rendezvous()
    if args.extract_features:
        rc = cmd_extract_features(args) or rc
        _rendezvous()
    if args.train:
        rc = cmd_train(args, None) or rc
    if args.train_strategy:
        rc = cmd_train(args, args.strategy) or rc
    return rc


if __name__ == "__main__":
    sys.exit(main())
