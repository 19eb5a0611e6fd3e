"""End-to-end: slide in HBM -> windows -> preprocess -> ResNet18 -> per-patch results,
against the oracle's extractor + transform + network on the same seeded slide."""
import numpy as np
import pytest
import torch

from oracle import extractor_ref as E, resnet18_ref as R, transform_ref as T
from ss25_hierarchical_multiscale_image_classification_amd import capi, extract, features, synth

pytestmark = pytest.mark.gpu


def test_score_slide_matches_oracle_pipeline():
    W, H, seed = 1900, 1850, 21
    l0 = synth.synth_level0(W, H, seed=seed, n_blobs=4)
    levels = synth.build_pyramid(l0, 4)
    polys = synth.synth_polygons(W, H, seed=seed)
    slide = extract.DeviceSlide(levels, device="cuda", name="tumor_901")
    slide.polygons = polys
    sd = synth.seeded_resnet18_state_dict(4, num_classes=2)
    net = capi.PackedResNet18(sd, precision="fp16")
    feats, logits, preds, meta = extract.score_slide(slide, net, levels=(1, 2, 3), batch_windows=7)
    ref_slide = E.ArraySlide([l.numpy() for l in levels])
    rows, ref_x = [], []
    for level in (1, 2, 3):
        wins, pix = E.extract_patches_ref(ref_slide, level, polygons_l0=polys)
        for w, p in zip([w for w in wins if w.keep], pix):
            rows.append((level, w.x, w.y, w.label))
            ref_x.append(torch.from_numpy(T.eval_transform(p)))
    assert np.array_equal(meta.cpu().numpy(), np.array(rows, np.int32))  # same windows, same order
    ref_f, ref_l = R.resnet18_forward(torch.stack(ref_x), sd)
    rel = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
    assert rel(feats, ref_f) <= 1e-3 and rel(logits, ref_l) <= 2e-3  # fp16: features at north_star's bound
    margin = (ref_l[:, 0] - ref_l[:, 1]).abs()
    decided = margin > 2 * 2e-3 * ref_l.abs().max()
    assert torch.equal(preds.cpu()[decided], ref_l.argmax(1)[decided])
    # the fused feature-extraction entry point names patches like the reference (main.py:722)
    f3, lab3, paths = features.extract_features_from_slide(slide, net, 3)
    n3 = sum(1 for r in rows if r[0] == 3)
    assert f3.shape == (n3, 512) and len(paths) == n3 and lab3.dtype == np.int64
    first = next(r for r in rows if r[0] == 3)
    assert paths[0] == f"tumor_901/tumor_901_x{first[1]}_y{first[2]}_{'tumor' if first[3] else 'normal'}.png"


def _oracle_rows(levels_np, polys, sd, levels):
    ref_slide = E.ArraySlide(levels_np)
    rows, ref_x = [], []
    for level in levels:
        wins, pix = E.extract_patches_ref(ref_slide, level, polygons_l0=polys)
        for w, p in zip([w for w in wins if w.keep], pix):
            rows.append((level, w.x, w.y, w.label))
            ref_x.append(torch.from_numpy(T.eval_transform(p)))
    ref_f, ref_l = R.resnet18_forward(torch.stack(ref_x), sd)
    return np.array(rows, np.int32), ref_f, ref_l


@pytest.mark.parametrize("prec,tol", [("fp16x3", 2e-5), ("fp16q8", 1e-4), ("fp32", 2e-5), ("fp16", 2e-3), ("bf16", 2.5e-2)])
def test_score_slide_all_levels_matches_oracle(prec, tol):
    """(fp16x3 = the parity mode: north_star's 1e-3 on features AND logits, asserted here at 2e-5 (measured 1.5e-6 /
    3.7e-6), labels identical.)
    BASELINE configs[2] end to end, level 0 included (`--patch_level all`, src/main.py:1120-1122): a slide with
    ragged right / bottom edges (level-0 windows of 1792 px clipped at both borders, 132 of them on the stride-224
    grid), windows / order / labels identical to the oracle's extractor, features and logits within the
    precision's tolerance.  Level 0 is the BIG level of score_slide's schedule: its windows are decided and
    gathered on the side stream under the forwards of levels 1-3 and the results are permuted back into
    level-major order -- this is the check of that reorder logic (small forward batches force several
    forwards per level)."""
    W, H, seed = 2600, 2300, 5
    levels = synth.build_pyramid(synth.synth_level0(W, H, seed=seed, n_blobs=4), 4)
    polys = synth.synth_polygons(W, H, seed=seed)
    slide = extract.DeviceSlide(levels, device="cuda", name="tumor_902")
    slide.polygons = polys
    sd = synth.seeded_resnet18_state_dict(5, num_classes=2)
    rows, ref_f, ref_l = _oracle_rows([l.numpy() for l in levels], polys, sd, (0, 1, 2, 3))
    assert (rows[:, 0] == 0).sum() >= 20 and len(set(rows[:, 0].tolist())) == 4  # level 0 really contributes, all levels present
    net = capi.PackedResNet18(sd, precision=prec)
    rel = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
    for kw in (dict(), dict(batch_windows=13, fwd_batch=32)):
        feats, logits, preds, meta = extract.score_slide(slide, net, levels=(0, 1, 2, 3), **kw)
        torch.cuda.synchronize()
        assert np.array_equal(meta.cpu().numpy(), rows)  # same windows, same labels, reference order
        assert rel(feats, ref_f) <= tol and rel(logits, ref_l) <= tol, (prec, kw)
        if prec in ("fp16x3", "fp16q8"):
            assert rel(feats, ref_f) <= 1e-3 and rel(logits, ref_l) <= 1e-3
        margin = (ref_l[:, 0] - ref_l[:, 1]).abs()
        decided = margin > 2 * tol * ref_l.abs().max()
        assert torch.equal(preds.cpu()[decided], ref_l.argmax(1)[decided])
    # listing order of the levels is the output order, whatever the processing order
    f2, l2, _, m2 = extract.score_slide(slide, net, levels=(2, 0), fwd_batch=32)
    sel = np.concatenate([np.nonzero(rows[:, 0] == 2)[0], np.nonzero(rows[:, 0] == 0)[0]])
    assert np.array_equal(m2.cpu().numpy(), rows[sel])
    assert rel(f2, ref_f[sel]) <= tol


def test_score_slide_non_lattice_stride_uses_the_per_window_kernel():
    """A stride that is not a multiple of 224 leaves the whole-level resampler: windows go through the per-window
    kernel in chunks and only kept windows keep their pixels; decisions and features equal the oracle's."""
    W, H, seed = 1500, 1250, 9
    levels = synth.build_pyramid(synth.synth_level0(W, H, seed=seed, n_blobs=3), 4)
    slide = extract.DeviceSlide(levels, device="cuda", name="normal_11")
    sd = synth.seeded_resnet18_state_dict(6, num_classes=2)
    net = capi.PackedResNet18(sd, precision="fp32")
    old = extract.LevelWindows.CHUNK
    extract.LevelWindows.CHUNK = 7  # several chunks
    try:
        feats, logits, _, meta = extract.score_slide(slide, net, levels=(1, 2), stride=160)
    finally:
        extract.LevelWindows.CHUNK = old
    ref_slide = E.ArraySlide([l.numpy() for l in levels])
    rows, ref_x = [], []
    for level in (1, 2):
        wins, pix = E.extract_patches_ref(ref_slide, level, stride=160)
        for w, p in zip([w for w in wins if w.keep], pix):
            rows.append((level, w.x, w.y, w.label))
            ref_x.append(torch.from_numpy(T.eval_transform(p)))
    assert np.array_equal(meta.cpu().numpy(), np.array(rows, np.int32))
    ref_f, _ = R.resnet18_forward(torch.stack(ref_x), sd)
    assert float((feats.cpu() - ref_f).abs().max() / ref_f.abs().max()) <= 2e-5


def test_png_tree_flow_equals_fused_flow(tmp_path):
    """extract_patches -> PNG tree -> PatchDataset -> features == fused features
    (PNG is lossless), compared keyed by path because the loader shuffles."""
    W, H, seed = 1500, 1300, 33
    slide = extract.DeviceSlide(synth.build_pyramid(synth.synth_level0(W, H, seed=seed, n_blobs=3), 4),
                                device="cuda", name="normal_777")
    net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=None), precision="bf16")
    level = 2
    n = extract.save_patch_pngs(slide, level, str(tmp_path))
    assert n > 0
    f_png, lab_png, paths_png = features.extract_features_from_pngs(str(tmp_path), net, batch_size=5, num_workers=0)
    f_fused, lab_fused, paths_fused = features.extract_features_from_slide(slide, net, level)
    assert len(paths_png) == len(paths_fused) == n
    by_name = {p.split("/")[-1]: i for i, p in enumerate(paths_fused)}
    order = [by_name[p.replace("\\", "/").split("/")[-1]] for p in paths_png]
    assert torch.equal(f_png, f_fused[order])
    assert np.array_equal(lab_png, lab_fused[order])


def test_cli_patch_then_extract_features(tmp_path, monkeypatch):
    """The reference's flag surface end to end on a synthetic slide: --patch writes the
    level tree (manifest + PNGs named like src/main.py:722), --extract_features writes the
    three files of src/main.py:885-893."""
    from ss25_hierarchical_multiscale_image_classification_amd.main import main

    monkeypatch.chdir(tmp_path)
    root = str(tmp_path / "data" / "camelyon16")
    assert main(["--patch", "--patch_level", "1", "--synthetic", "1500,1300,33,normal_042", "--write_png",
                 "--data_root", root]) == 0
    level_dir = tmp_path / "data" / "camelyon16" / "patches" / "level_1" / "normal_042"
    man = np.load(level_dir / "manifest.npz")
    pngs = sorted(p.name for p in level_dir.glob("*.png"))
    assert len(pngs) == int(man["keep"].sum()) > 1
    assert all(n.startswith("normal_042_x") and (n.endswith("_normal.png") or n.endswith("_tumor.png")) for n in pngs)
    # resume rule (src/main.py:634-640): a non-empty slide directory is skipped
    assert main(["--patch", "--patch_level", "1", "--synthetic", "1500,1300,33,normal_042", "--data_root", root]) == 0
    assert main(["--extract_features", "--patch_level", "1", "--data_root", root, "--precision", "fp16"]) == 0
    feats = np.load(tmp_path / "patch_features_1.npy")
    labels = np.load(tmp_path / "patch_labels_1.npy")
    paths = (tmp_path / "patch_paths_1.txt").read_text().split()
    assert sorted(int("_tumor" in p) for p in paths) == sorted(labels.tolist())
    assert feats.shape == (len(pngs), 512) and feats.dtype == np.float32
    assert labels.shape == (len(pngs),) and labels.dtype == np.int64 and len(paths) == len(pngs)
    assert np.isfinite(feats).all() and float(np.abs(feats).max()) > 0


def test_cli_extract_features_with_simclr_encoder(tmp_path, monkeypatch):
    """`--extract_features --simclr_encoder CKPT` (extract_features_with_simclr, src/main.py:897-932): the encoder of a
    SimCLRModel checkpoint (`encoder.*`, projector dropped, fc = Identity) scores the patches -- the features equal a
    direct forward of that encoder, and differ from those of the default seeded weights."""
    from ss25_hierarchical_multiscale_image_classification_amd.main import main
    from ss25_hierarchical_multiscale_image_classification_amd.simclr import SimCLRModel
    from ss25_hierarchical_multiscale_image_classification_amd.weights import canonical_state_dict

    monkeypatch.chdir(tmp_path)
    torch.manual_seed(77)
    sim = SimCLRModel()
    ckpt = tmp_path / "simclr_encoder.pth"
    torch.save(sim.state_dict(), ckpt)
    spec = ["--synthetic", "1500,1300,33,normal_042", "--data_root", str(tmp_path / "none")]
    assert main(["--extract_features", "--patch_level", "1", "--precision", "fp16", "--simclr_encoder", str(ckpt)] + spec) == 0
    feats = np.load(tmp_path / "patch_features_1.npy")
    bare = {k: v for k, v in canonical_state_dict(sim.state_dict()).items() if not k.startswith(("projector.", "fc."))}
    net = capi.PackedResNet18(bare, precision="fp16")
    slide = extract.DeviceSlide.synthetic(1500, 1300, seed=33, name="normal_042")
    ref, _, _ = features.extract_features_from_slide(slide, net, 1)
    assert feats.shape == tuple(ref.shape) and feats.shape[0] > 1 and np.array_equal(feats, ref.numpy())
    assert main(["--extract_features", "--patch_level", "1", "--precision", "fp16"] + spec) == 0  # seeded default weights
    assert not np.allclose(np.load(tmp_path / "patch_features_1.npy"), feats, atol=1e-3)


def test_train_mode_forward_on_gpu_points_to_the_native_step():
    """No silent torch-autograd path on the GPU: a train()-mode forward on a ROCm tensor raises and names the native
    trainers; on a CPU tensor it is the torch graph the training oracle is pinned on; eval() runs the HIP path."""
    from ss25_hierarchical_multiscale_image_classification_amd.resnet import ResNet18Classifier, ResNet18FeatureExtractor
    from ss25_hierarchical_multiscale_image_classification_amd.simclr import SimCLRModel

    x = torch.randn(2, 3, 224, 224)
    for m in (ResNet18Classifier(), ResNet18FeatureExtractor(weight_path=None), SimCLRModel()):
        m = m.cuda().train()
        with pytest.raises(capi.HipacError, match="NativeClassifierTrainer"):
            m(x.cuda())
        assert m.cpu()(x).shape[0] == 2  # CPU tensor: the module's own graph
        m = m.cuda().eval()
        with torch.no_grad():
            assert m(x.cuda()).shape[0] == 2


def test_tiff_slide_scans_like_the_same_pixels_from_memory(tmp_path):
    # a tiled (deflate, lossless) TIFF pyramid read through tiff_pyramid -> DeviceSlide.from_tiff gives the
    # same extractor decisions and resized patches as the same levels handed over as tensors
    from ss25_hierarchical_multiscale_image_classification_amd import tiff_pyramid as tp

    ref = extract.DeviceSlide.synthetic(2300, 1900, seed=4, with_polygons=False)
    levels = [l[:, :w].cpu().numpy() for l, (w, _) in zip(ref.levels, ref.level_dimensions)]
    path = str(tmp_path / "slide_007.tif")
    tp.write_tiled_tiff(path, levels, tile=512, compression="deflate", bigtiff=True)
    got = extract.DeviceSlide.from_tiff(path)
    assert got.name == "slide_007" and got.level_dimensions == ref.level_dimensions
    for level in (1, 2, 3):
        a, b = extract.LevelWindows(ref, level), extract.LevelWindows(got, level)
        assert torch.equal(a.xy, b.xy) and torch.equal(a.keep, b.keep) and torch.equal(a.sums, b.sums)
        k = a.kept_index()
        assert torch.equal(a.patches(k), b.patches(k))
