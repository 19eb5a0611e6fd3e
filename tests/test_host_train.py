"""CPU tests of the host side of the training path and of bench.py's launcher: loss weights of every strategy,
data loaders on a PNG tree, the deterministic cores of the host augmentations, the parameter table of the native
encoder, the argv / environment of the per-rank processes."""
import math
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ss25_hierarchical_multiscale_image_classification_amd import train, train_native, transforms as T  # noqa: E402
from ss25_hierarchical_multiscale_image_classification_amd.patch_dataset import PatchDataset  # noqa: E402


def png_tree(root, slides=3, per_slide=8, size=224, seed=0):
    rng = np.random.RandomState(seed)
    for s in range(slides):
        d = os.path.join(root, f"tumor_{s:03d}")
        os.makedirs(d, exist_ok=True)
        for k in range(per_slide):
            lab = "tumor" if (k + s) % 3 == 0 else "normal"
            Image.fromarray(rng.randint(0, 256, (size, size, 3), dtype=np.uint8), "RGB").save(
                os.path.join(d, f"tumor_{s:03d}_x{224 * k}_y0_{lab}.png"))
    return root


def test_class_weights_follow_the_reference_per_strategy(tmp_path):
    ds = PatchDataset(png_tree(str(tmp_path)), verbose=False)
    counts = ds.get_class_counts()
    c0, c1 = counts[0], counts[1]
    w = train.class_weights(ds, None)  # src/main.py:485-490: 1/count normalised by the smaller weight
    assert torch.allclose(w, torch.tensor([1.0 / c0, 1.0 / c1]) / min(1.0 / c0, 1.0 / c1))
    for strat in ("weighted_loss", "self_supervised"):  # :549-552, applied at :558 / :572
        assert torch.allclose(train.class_weights(ds, strat), torch.tensor([(c0 + c1) / c0, (c0 + c1) / c1]))
    assert train.class_weights(ds, "balanced") is None  # :566: unweighted


def test_dataloaders_split_by_slide_and_augment_only_tumour(tmp_path):
    root = png_tree(str(tmp_path), slides=5)
    tl, vl, tds, vds = train.get_dataloaders(root, 0.2, batch_size=4)
    train_slides = {os.path.relpath(p, root).split(os.sep)[0] for p in tds.image_paths}
    base = vds.dataset if hasattr(vds, "dataset") else vds
    val_slides = {os.path.relpath(p, root).split(os.sep)[0] for p in base.image_paths}
    assert train_slides and val_slides and not (train_slides & val_slides)  # slide-level split (:413-414)
    x, y, paths = next(iter(tl))
    assert x.shape == (4, 3, 224, 224) and x.dtype == torch.float32 and y.dtype == torch.int64 and len(paths) == 4
    labels = np.array(base.labels)[vds.indices] if hasattr(vds, "indices") else np.array(base.labels)
    assert (labels == 0).sum() == (labels == 1).sum()  # validation set balanced (:446-458)


def test_dataloaders_share_every_global_batch_over_the_ranks(tmp_path):
    """world > 1 (`main.py --world_size N`): every rank builds the same datasets (same host-side seed) and draws ITS
    chunk of every global batch -- what DataParallel's scatter hands replica r (src/main.py:481-482); the validation
    samples are shared out as contiguous ranges, each scored exactly once."""
    import random

    root = png_tree(str(tmp_path), slides=5, per_slide=9)
    world, B = 3, 6
    per_rank = []
    for r in range(world):
        random.seed(0)  # main.py seeds every rank identically before the datasets are built
        tl, vl, tds, vds = train.get_dataloaders(root, 0.2, batch_size=B, rank=r, world=world)
        per_rank.append(([p for _, _, paths in tl for p in paths], [len(paths) for _, _, paths in tl],
                         [p for _, _, paths in vl for p in paths], list(tds.image_paths), len(vds)))
    assert all(pr[3] == per_rank[0][3] for pr in per_rank)  # the same dataset order on every rank
    n_train = len(per_rank[0][3])
    assert all(pr[1] == per_rank[0][1] for pr in per_rank) and set(per_rank[0][1]) <= {B // world, (n_train % B) // world}
    seen = [p for pr in per_rank for p in pr[0]]
    assert len(seen) == len(set(seen)) and len(seen) >= n_train - (world - 1)  # each sample once; only a remainder is left out
    val = [p for pr in per_rank for p in pr[2]]
    assert len(val) == len(set(val)) == per_rank[0][4]  # the validation set, partitioned


def test_augmentation_cores():
    img = Image.fromarray(np.random.RandomState(1).randint(0, 256, (64, 48, 3), dtype=np.uint8), "RGB")
    # hue shift by +-0.5 twice is the identity on the H channel (uint8 wrap), and 0 leaves HSV round-trip only
    h0 = np.array(img.convert("HSV"))[..., 0]
    h1 = np.array(T.adjust_hue(img, 0.25).convert("HSV"))[..., 0]
    assert np.abs(((h1.astype(int) - h0.astype(int) + 128) % 256) - 128 - 63).max() <= 6  # 0.25 * 255 = 63, HSV rounding
    with pytest.raises(ValueError):
        T.adjust_hue(img, 0.7)
    g = np.array(T.RandomGrayscale(p=1.0)(img))
    assert (g[..., 0] == g[..., 1]).all() and (g[..., 1] == g[..., 2]).all()
    assert np.array_equal(g[..., 0], np.array(img.convert("L")))
    torch.manual_seed(0)
    for _ in range(50):
        i, j, h, w = T.RandomResizedCrop.get_params(48, 64, (0.08, 1.0), (3 / 4, 4 / 3))
        assert 0 <= i and i + h <= 64 and 0 <= j and j + w <= 48 and 0.08 * 48 * 64 * 0.9 <= h * w <= 48 * 64
        assert 3 / 4 - 0.1 <= w / h <= 4 / 3 + 0.1
    out = T.simclr_transform()(img)
    assert out.shape == (3, 224, 224)
    torch.manual_seed(3)
    r = T.RandomRotation(90)(img)
    assert r.size == img.size
    jit = T.ColorJitter(0.2, 0.2, 0.2, 0.1)
    assert jit.brightness == (0.8, 1.2) and jit.hue == (-0.1, 0.1) and jit(img).size == img.size


def test_native_encoder_parameter_table():
    table = train_native.conv_table()
    assert len(table) == 20 and table[0]["conv"] == "conv1" and table[7]["conv"] == "layer2.0.downsample.0"
    n = sum(e["cout"] * e["cin"] * e["ks"] ** 2 + 2 * e["cout"] for e in table)
    assert n == 11_176_512  # torchvision resnet18 without fc
    from ss25_hierarchical_multiscale_image_classification_amd import capi
    lib = capi.load_library()
    assert lib.hipac_train_param_floats() == n and lib.hipac_train_stat_floats() == 2 * 4800
    offs = [e["param_off"] for e in table]
    assert offs == sorted(offs) and offs[0] == 0
    per_image = lib.hipac_train_workspace_bytes(2) - lib.hipac_train_workspace_bytes(1)
    assert 25e6 < per_image < 40e6  # ~31 MB of saved activations + gradient scratch per image


def test_bench_launcher_builds_one_process_per_rank():
    import bench

    cmds = bench.child_commands(["--gpus", "4", "--steps", "3", "--workload", "wsi"], 4, 29511, python="py", base_env={"A": "1"})
    assert len(cmds) == 4
    for r, (argv, env) in enumerate(cmds):
        assert argv[0] == "py" and argv[1].endswith("bench.py") and argv[-1] == "--_child" and argv.count("--_child") == 1
        assert argv[2:-1] == ["--gpus", "4", "--steps", "3", "--workload", "wsi"]
        assert env["RANK"] == env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29511" and env["A"] == "1"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    args = bench.build_parser().parse_args(["--gpus", "2", "--_child"])
    assert args.gpus == 2 and args._child and args.workload == "resnet" and args.steps == 8 and args.warmup == 2
    assert bench.host_threads() >= 1
    assert bench.kernel_of("l1b0") == "block16_c64_kernel" and bench.kernel_of("l3b1c1") == "conv3x3_halo16_kernel"
    assert [n for n, _ in bench.OPS][2:6] == ["l1b0", "(fused)", "l1b1", "(fused)"]  # a layer1 BasicBlock is one launch
