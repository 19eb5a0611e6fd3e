"""BASELINE.json's FULL sizes through size-independent properties (the oracle cannot run them in seconds):
configs[1] -- 65 536 synthetic 224x224 patches through the batched ResNet18 -- and configs[2] -- the full hierarchical scan of
one synthetic 50 000 x 50 000 level-0 slide across levels 0-3."""
import numpy as np
import pytest
import torch

from oracle import resnet18_ref as R, transform_ref as T
from ss25_hierarchical_multiscale_image_classification_amd import capi, extract, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["bf16", "fp16x3", "fp16q8"])
def test_configs1_64k_patches_batch_invariance_and_oracle_spot_check(prec):
    """65 536 patches in 8 steps of 8 192 (the benchmark's shape).  Properties: a patch's result does not depend on the
    batch it travels in (two launch lanes, 16 sub-batches, 2 groups per step vs small ragged slices: bit-identical);
    permuting the batch permutes the outputs; argmax == the library's labels; and 64 patches spread over the whole set
    agree with the CPU oracle within the mode's bound."""
    sd = synth.seeded_resnet18_state_dict(0, num_classes=2)
    net = capi.PackedResNet18(sd, precision=prec)
    n, step = 65536, 8192
    feats, logits, labels = [], [], []
    u8_keep = []
    for i in range(n // step):
        u8 = synth.synth_patches_u8(step, seed=100 + i, device="cuda")
        f, l, lab = net.forward(u8, want_logits=True, want_labels=True)
        feats.append(f.clone()), logits.append(l.clone()), labels.append(lab.clone())
        if i in (0, 7):
            u8_keep.append((i, u8))
    F_, L_, Y_ = torch.cat(feats), torch.cat(logits), torch.cat(labels)
    assert F_.shape == (n, 512) and L_.shape == (n, 2) and bool(torch.isfinite(F_).all())
    assert torch.equal(Y_, L_.argmax(1))
    for i, u8 in u8_keep:
        base = i * step
        # the same patches in ragged slices on one lane
        for lo, hi in ((0, 37), (37, 700), (4000, 4513), (8000, 8192)):
            f, l, _ = net.forward(u8[lo:hi].contiguous(), want_logits=True)
            assert torch.equal(f, F_[base + lo:base + hi]) and torch.equal(l, L_[base + lo:base + hi])
        # a permutation of a slice permutes the results
        perm = torch.randperm(600, generator=torch.Generator().manual_seed(i)).cuda()
        f, _, _ = net.forward(u8[:600][perm].contiguous())
        assert torch.equal(f, F_[base:base + 600][perm])
    # oracle spot check: 64 patches from the first and the last step
    idx = torch.arange(0, 32) * 251
    sel = torch.cat([u8_keep[0][1][idx], u8_keep[1][1][idx]]).cpu()
    lut = torch.from_numpy(T.normalize_lut())
    x = torch.stack([lut[c][sel[..., c].long()] for c in range(3)], dim=1)
    ref_f, ref_l = R.resnet18_forward(x, sd)
    got_f = torch.cat([F_[idx.cuda()], F_[7 * step + idx.cuda()]]).cpu()
    got_l = torch.cat([L_[idx.cuda()], L_[7 * step + idx.cuda()]]).cpu()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    tol_f, tol_l = {"fp16x3": (2e-5, 2e-5), "fp16q8": (1e-4, 1e-4)}.get(prec, (2.5e-2, 2.5e-2))
    assert rel(got_f, ref_f) <= tol_f and rel(got_l, ref_l) <= tol_l


def test_configs2_full_50k_slide_properties():
    """One synthetic 50 000^2 slide, levels 0-3, reference grid (stride 224): the window counts SURVEY 8a-1 derives
    (50 176 + 12 544 + 3 136 + 784 = 66 640), level-major reference order, idempotence, independence from the gather /
    forward batch sizes, every level alone == its rows of the full scan, whiteness and labels recomputed per window."""
    side = 50000
    slide = extract.DeviceSlide.synthetic(side, side, seed=10, with_polygons=True)
    assert [d for d in slide.level_dimensions] == [(50000, 50000), (25000, 25000), (12500, 12500), (6250, 6250)]
    counts = [len(extract.window_grid(w, h, lv)[2]) for lv, (w, h) in enumerate(slide.level_dimensions)]
    assert counts == [50176, 12544, 3136, 784]
    net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision="bf16")
    f, l, p, meta = extract.score_slide(slide, net)
    n = f.shape[0]
    assert 0.3 * sum(counts) < n < 0.9 * sum(counts) and meta.shape == (n, 4)
    m = meta.cpu().numpy()
    assert np.all(np.diff(m[:, 0]) >= 0)  # level-major
    for lv in range(4):  # x-outer / y-inner visiting order inside a level (src/main.py:682-683)
        rows = m[m[:, 0] == lv]
        key = rows[:, 1].astype(np.int64) * 10**6 + rows[:, 2]
        assert np.all(np.diff(key) > 0)
    f2, l2, p2, meta2 = extract.score_slide(slide, net, batch_windows=3001, fwd_batch=5000)
    assert torch.equal(meta, meta2) and torch.equal(f, f2) and torch.equal(l, l2) and torch.equal(p, p2)
    # one level alone == its rows of the full scan
    f3, _, _, meta3 = extract.score_slide(slide, net, levels=(2,))
    sel = torch.nonzero(meta[:, 0] == 2).flatten()
    assert torch.equal(meta3, meta[sel]) and torch.equal(f3, f[sel])
    # decisions recomputed window by window (per-window kernel) on a sample of level-1 windows
    lw = extract.LevelWindows(slide, 1)
    pick = torch.arange(0, lw.xy.shape[0], 97, device="cuda")
    xy = lw.xy[pick].contiguous()
    _, sums, keep = capi.tile_preprocess(slide.levels[1], xy, 896, "u8", width=slide.level_dimensions[1][0])
    assert torch.equal(sums, lw.sums[pick]) and torch.equal(keep, lw.keep[pick])
    assert torch.equal(capi.window_labels(slide.mask(1), xy, 896), lw.labels[pick])
    kept1 = set(map(tuple, m[m[:, 0] == 1][:, 1:3].tolist()))
    assert kept1 == set(map(tuple, lw.xy[lw.keep.bool()].cpu().tolist()))
