"""SimCLR's view augmentation on the device (csrc/augment.hip, augment.py) against Pillow itself -- the library the
reference's torchvision transforms call (src/models/simclr.py:57-66): every colour operation, the crop + resize, and the
whole pipeline under the same random draws as the host transforms.  Bit-exact: these are uint8 / integer results."""
import random

import numpy as np
import pytest
import torch
from PIL import Image, ImageEnhance

from ss25_hierarchical_multiscale_image_classification_amd import augment, capi, synth, transforms

pytestmark = pytest.mark.gpu


def _images(n, P, seed):
    """Half uniform noise, half smooth tissue-like windows of the synthetic slide generator (saturated and gray pixels too)."""
    g = torch.Generator().manual_seed(seed)
    noise = torch.randint(0, 256, (n, P, P, 3), generator=g, dtype=torch.uint8)
    l0 = synth.synth_level0(4 * P, 4 * P, seed=seed, device="cpu")
    for i in range(0, n, 2):
        y, x = (i * 37) % (3 * P), (i * 91) % (3 * P)
        noise[i] = l0[y:y + P, x:x + P]
    noise[0, :8] = 0
    noise[0, 8:16] = 255
    noise[1, :, :8] = noise[1, :, :8, :1]  # gray pixels
    return noise


def _view(idx, P, ops=(-1, -1, -1, -1), b=1.0, c=1.0, s=1.0, hue=0, gray=0, crop=None, flip=0):
    top, left, h, w = crop if crop is not None else (0, 0, P, P)
    return [idx, top, left, h, w, flip, *ops, gray, augment._f32_bits(b), augment._f32_bits(c), augment._f32_bits(s), hue & 255, 0,
            0, *augment.IDENTITY_FIX, 0]


def test_colour_operations_equal_pillow():
    imgs = _images(6, 224, 3)
    pool = augment.DevicePatchPool(imgs.cuda())
    rows, refs = [], []
    factors = [0.0, 1.0, 0.6, 1.4, 0.5, 1.7, 0.9999, 0.013, 1.25, 2.5]
    for i in range(imgs.shape[0]):
        pil = Image.fromarray(imgs[i].numpy(), "RGB")
        for f in factors:
            rows.append(_view(i, 224, ops=(0, -1, -1, -1), b=f)), refs.append(np.array(ImageEnhance.Brightness(pil).enhance(f)))
            rows.append(_view(i, 224, ops=(1, -1, -1, -1), c=f)), refs.append(np.array(ImageEnhance.Contrast(pil).enhance(f)))
            rows.append(_view(i, 224, ops=(2, -1, -1, -1), s=f)), refs.append(np.array(ImageEnhance.Color(pil).enhance(f)))
        for hf in (-0.1, -0.05, 0.0, 0.031, 0.1, 0.5, -0.5):
            rows.append(_view(i, 224, ops=(3, -1, -1, -1), hue=int(hf * 255) & 0xFF)), refs.append(np.array(transforms.adjust_hue(pil, hf)))
        g = np.array(pil.convert("L"))
        rows.append(_view(i, 224, gray=1)), refs.append(np.dstack([g, g, g]))
        # a chain in a fixed order, as ColorJitter applies it
        chain = ImageEnhance.Color(pil).enhance(1.3)
        chain = transforms.adjust_hue(chain, -0.07)
        chain = ImageEnhance.Brightness(chain).enhance(0.7)
        chain = ImageEnhance.Contrast(chain).enhance(1.35)
        rows.append(_view(i, 224, ops=(2, 3, 0, 1), b=0.7, c=1.35, s=1.3, hue=int(-0.07 * 255) & 0xFF)), refs.append(np.array(chain))
    _, got = pool.augment(np.array(rows, np.int32), want_u8=True)
    got = got.cpu().numpy()
    for k, ref in enumerate(refs):
        assert np.array_equal(got[k], ref), (k, rows[k][6:15], int(np.abs(got[k].astype(int) - ref.astype(int)).max()))


def test_hue_over_the_whole_colour_cube():
    """All 2^24 RGB triples through Pillow's RGB -> HSV -> (H + shift) -> RGB and through the kernel."""
    v = np.arange(256, dtype=np.uint8)
    r, g, b = np.meshgrid(v, v, v, indexing="ij")
    cube = np.stack([r.ravel(), g.ravel(), b.ravel()], 1)
    n = -(-cube.shape[0] // (224 * 224))
    flat = np.zeros((n * 224 * 224, 3), np.uint8)
    flat[:cube.shape[0]] = cube
    imgs = torch.from_numpy(flat.reshape(n, 224, 224, 3))
    pool = augment.DevicePatchPool(imgs.cuda())
    for hf in (0.0, 0.1, -0.1, 0.037):
        shift = int(hf * 255) & 0xFF
        rows = np.array([_view(i, 224, ops=(3, -1, -1, -1), hue=shift) for i in range(n)], np.int32)
        _, got = pool.augment(rows, want_u8=True)
        ref = np.array(transforms.adjust_hue(Image.fromarray(flat.reshape(n * 224, 224, 3), "RGB"), hf)).reshape(n, 224, 224, 3)
        assert np.array_equal(got.cpu().numpy(), ref), hf


@pytest.mark.parametrize("P", [224, 448, 896, 1792])
def test_resized_crop_and_flip_equal_pillow(P):
    imgs = _images(4 if P < 1792 else 2, P, 5 + P)
    pool = augment.DevicePatchPool(imgs.cuda())
    rng = np.random.default_rng(P)
    crops = [(0, 0, P, P), (0, 0, 1, 1), (P - 1, P - 1, 1, 1), (3, 5, P - 3, P - 5), (0, P - 57, P, 57), (P - 60, 0, 60, P), (10, 20, min(223, P - 10), min(225, P - 20))]
    for _ in range(20):
        h, w = int(rng.integers(1, P + 1)), int(rng.integers(1, P + 1))
        crops.append((int(rng.integers(0, P - h + 1)), int(rng.integers(0, P - w + 1)), h, w))
    rows, refs = [], []
    for k, (top, left, h, w) in enumerate(crops):
        i, flip = k % imgs.shape[0], k % 2
        pil = Image.fromarray(imgs[i].numpy(), "RGB").crop((left, top, left + w, top + h)).resize((224, 224), Image.BILINEAR)
        refs.append(np.array(pil.transpose(Image.FLIP_LEFT_RIGHT) if flip else pil))
        rows.append(_view(i, P, crop=(top, left, h, w), flip=flip))
    _, got = pool.augment(np.array(rows, np.int32), want_u8=True)
    got = got.cpu().numpy()
    for k, ref in enumerate(refs):
        assert np.array_equal(got[k], ref), (P, crops[k], int(np.abs(got[k].astype(int) - ref.astype(int)).max()))


@pytest.mark.parametrize("P", [224, 448])
def test_pipeline_equals_host_transforms_under_the_same_draws(P):
    """`transforms.simclr_transform()` (the reference's Compose on Pillow, host) and the device pipeline with the same seeds:
    the same parameters are drawn in the same order, and the normalised float views are equal bit for bit."""
    imgs = _images(12, P, 11)
    pool = augment.DevicePatchPool(imgs.cuda())
    T = transforms.simclr_transform()
    torch.manual_seed(123), random.seed(123)
    host = []
    for i in range(imgs.shape[0]):
        pil = Image.fromarray(imgs[i].numpy(), "RGB")
        host.append(T(pil)), host.append(T(pil))  # x_i, x_j (src/datasets/simclr_dataset.py:10-11)
    torch.manual_seed(123), random.seed(123)
    rows = []
    for i in range(imgs.shape[0]):
        rows.append(augment.draw_simclr_view(i, P, P)), rows.append(augment.draw_simclr_view(i, P, P))
    assert any(r[6] >= 0 for r in rows) and any(r[6] < 0 for r in rows) and any(r[5] for r in rows)
    got = pool.augment(np.array(rows, np.int32)).cpu()
    for k, ref in enumerate(host):
        assert torch.equal(got[k], ref), (k, rows[k], float((got[k] - ref).abs().max()))


def test_bad_parameters_are_refused_on_the_host():
    pool = augment.DevicePatchPool(_images(2, 224, 1).cuda())
    for bad in (_view(2, 224), _view(0, 224, crop=(0, 0, 225, 10)), _view(0, 224, crop=(200, 0, 30, 10)), _view(0, 224, crop=(0, 0, 0, 5)),
                _view(0, 224, ops=(4, -1, -1, -1))):
        with pytest.raises(capi.HipacError):
            pool.augment(np.array([bad], np.int32))


def test_device_loader_feeds_the_native_simclr_step(tmp_path):
    """PNG tree -> PatchDataset -> DevicePatchPool -> DeviceSimCLRLoader -> NativeSimCLRTrainer, and through `pretrain_simclr`."""
    from ss25_hierarchical_multiscale_image_classification_amd import simclr as S, train_native as TN
    from ss25_hierarchical_multiscale_image_classification_amd.patch_dataset import PatchDataset

    imgs = _images(10, 224, 21)
    d = tmp_path / "level_3" / "tumor_001"
    d.mkdir(parents=True)
    for i in range(imgs.shape[0]):
        Image.fromarray(imgs[i].numpy(), "RGB").save(d / f"tumor_001_x{i * 224}_y0_{'tumor' if i % 2 else 'normal'}.png")
    base = PatchDataset(str(tmp_path / "level_3"), transform=None)
    pool = augment.DevicePatchPool.from_patch_dataset(base)
    assert len(pool) == 10 and pool.P == 224
    order = {p: i for i, p in enumerate(base.image_paths)}
    for p, i in order.items():
        assert np.array_equal(pool.patches[i].cpu().numpy(), np.array(Image.open(p).convert("RGB")))
    loader = augment.DeviceSimCLRLoader(pool, batch_size=4, seed=1)
    batches = list(loader)
    assert [b[0].shape[0] for b in batches] == [4, 4, 2] and all(b[0].is_cuda and b[0].dtype == torch.float32 for b in batches)
    torch.manual_seed(3)
    tr = TN.NativeSimCLRTrainer({k: v.clone() for k, v in S.SimCLRModel().state_dict().items()}, device="cuda", precision="fp16")
    for x_i, x_j in batches[:2]:
        assert float(tr.step(x_i, x_j)) > 0
    model, hist = S.pretrain_simclr(str(tmp_path / "level_3"), epochs=1, batch_size=4, out_dir=str(tmp_path), max_steps=2, verbose=False,
                                    device_aug=True, precision="fp16")
    assert len(hist) == 1 and np.isfinite(hist[0]) and (tmp_path / "simclr_encoder.pth").exists()


def test_flips_and_rotation_equal_pillow():
    """Geometry 1: RandomHorizontalFlip / RandomVerticalFlip / Image.rotate(angle, NEAREST, fillcolor=0) -- Pillow's affine_fixed
    path with the matrix Image.rotate builds, the transpose fast paths of 0 / 90 / 180 / 270 degrees included."""
    imgs = _images(3, 224, 9)
    pool = augment.DevicePatchPool(imgs.cuda())
    rng = np.random.default_rng(2)
    angles = list(rng.uniform(-90, 90, 40)) + [0.0, 90.0, -90.0, 180.0, 270.0, 360.0, 45.0, -45.0, 1e-9, -1e-9, 89.999999, 0.5, -0.5]
    rows, refs = [], []
    for k, a in enumerate(angles):
        i, hf, vf = k % 3, (k >> 1) & 1, (k >> 2) & 1
        pil = Image.fromarray(imgs[i].numpy(), "RGB")
        pil = pil.transpose(Image.FLIP_LEFT_RIGHT) if hf else pil
        pil = pil.transpose(Image.FLIP_TOP_BOTTOM) if vf else pil
        refs.append(np.array(pil.rotate(float(a), resample=Image.NEAREST, expand=False, fillcolor=0)))
        row = augment.identity_view(i)
        row[5], row[16] = hf, vf
        row[17:23] = augment.pil_rotate_fixed(float(a), 224, 224)
        rows.append(row)
    _, got = pool.augment(np.array(rows, np.int32), want_u8=True, geometry=1)
    got = got.cpu().numpy()
    for k, ref in enumerate(refs):
        assert np.array_equal(got[k], ref), (angles[k], int((got[k] != ref).any(-1).sum()))


def test_train_transform_equals_host_transforms_under_the_same_draws():
    """`transforms.train_transform()` (src/main.py:417-425 on Pillow, host) against geometry 1 with `draw_train_view`, same seeds;
    and the eval transform of a 224-pixel patch = the identity row."""
    imgs = _images(10, 224, 31)
    pool = augment.DevicePatchPool(imgs.cuda())
    T, E = transforms.train_transform(), transforms.eval_transform()
    torch.manual_seed(77), random.seed(77)
    host = [T(Image.fromarray(imgs[i].numpy(), "RGB")) for i in range(imgs.shape[0])]
    torch.manual_seed(77), random.seed(77)
    rows = [augment.draw_train_view(i) for i in range(imgs.shape[0])]
    got = pool.augment(np.array(rows, np.int32), geometry=1).cpu()
    for k, ref in enumerate(host):
        assert torch.equal(got[k], ref), (k, rows[k], float((got[k] - ref).abs().max()))
    got = pool.augment(np.array([augment.identity_view(i) for i in range(3)], np.int32), geometry=1).cpu()
    for i in range(3):
        assert torch.equal(got[i], E(Image.fromarray(imgs[i].numpy(), "RGB")))
    with pytest.raises(capi.HipacError):  # flips + rotation act at 224 pixels only
        augment.DevicePatchPool(_images(2, 448, 1).cuda()).augment(np.array([augment.identity_view(0)], np.int32), geometry=1)


def test_classifier_loop_on_device_loaders(tmp_path):
    """`train_resnet_classifier(device_aug=True)`: both sets' patches in HBM, batches made on the device, the native step."""
    from ss25_hierarchical_multiscale_image_classification_amd import train
    from ss25_hierarchical_multiscale_image_classification_amd.patch_dataset import PatchDataset

    imgs = _images(24, 224, 41)
    root = tmp_path / "level_3"
    for s in range(3):
        d = root / f"tumor_00{s}"
        d.mkdir(parents=True)
        for i in range(8):
            k = 8 * s + i
            Image.fromarray(imgs[k].numpy(), "RGB").save(d / f"tumor_00{s}_x{i * 224}_y0_{'tumor' if i % 2 else 'normal'}.png")
    ds = PatchDataset(str(root), transform=None)
    pool = augment.DevicePatchPool.from_patch_dataset(ds)
    ld = augment.DeviceClassifierLoader(pool, 6, shuffle=True, augment=True, seed=3)
    seen = []
    for x, labels, idx in ld:
        assert x.is_cuda and x.shape[1:] == (3, 224, 224) and labels.tolist() == [pool.labels[j] for j in idx]
        seen += idx
        for j, lab, xi in zip(idx, labels.tolist(), x):
            if lab == 0:  # normal patches: the eval transform, i.e. the normalised pixels themselves
                assert torch.equal(xi.cpu(), transforms.eval_transform()(Image.open(pool.paths[j]).convert("RGB")))
    assert sorted(seen) == list(range(24))
    model, hist = train.train_resnet_classifier(str(root), strategy="weighted_loss", epochs=1, batch_size=8, max_steps=2,
                                                save_path=str(tmp_path / "clf.pth"), device_aug=True)
    assert len(hist) == 1 and np.isfinite(hist[0][0]) and 0.0 <= hist[0][2] <= 1.0 and (tmp_path / "clf.pth").exists()


def test_pool_from_slides_equals_the_png_tree(tmp_path):
    """Level 3: the pool made straight from the slides in HBM holds the pixels `--patch` writes as PNGs and `PatchDataset`
    reads back, with the same labels (src/main.py:705-726)."""
    from ss25_hierarchical_multiscale_image_classification_amd import extract
    from ss25_hierarchical_multiscale_image_classification_amd.patch_dataset import PatchDataset

    slides = [extract.DeviceSlide.synthetic(9000, 7000, seed=61 + i, name=f"tumor_06{i}", with_polygons=True) for i in range(2)]
    pool = augment.DevicePatchPool.from_slides(slides, level=3)
    root = tmp_path / "level_3"
    for sl in slides:
        extract.save_patch_pngs(sl, 3, str(root))
    ds = PatchDataset(str(root), transform=None)
    assert len(ds) == len(pool) > 20 and pool.P == 224
    by_key = {(n, int(m[1]), int(m[2])): i for i, (n, m) in enumerate(zip(pool.slide_names, pool.meta))}
    import re
    for path, lab in zip(ds.image_paths, ds.labels):
        mm = re.search(r"(tumor_\d+)_x(\d+)_y(\d+)_", path)
        i = by_key[(mm.group(1), int(mm.group(2)), int(mm.group(3)))]
        assert pool.labels[i] == lab
        assert np.array_equal(pool.patches[i].cpu().numpy(), np.array(Image.open(path).convert("RGB")))
    ld = augment.DeviceClassifierLoader(pool, 16, seed=2)
    x, labels, idx = next(iter(ld))
    assert x.shape == (16, 3, 224, 224) and labels.tolist() == [pool.labels[j] for j in idx]


def test_training_loops_straight_from_slides(tmp_path):
    """No PNG tree: `train_resnet_classifier(slides=...)` builds its pools from the pyramids in HBM (slide-level split as the
    reference's), runs the native steps on device-made batches, validates, saves; `self_supervised` pre-trains SimCLR on the
    same pool first."""
    from ss25_hierarchical_multiscale_image_classification_amd import extract, train

    slides = [extract.DeviceSlide.synthetic(5000, 4500, seed=71 + i, name=f"tumor_07{i}", with_polygons=True) for i in range(3)]
    tl, vl, tds, vds, pool = train.get_device_loaders(slides, level=3, batch_size=8)
    names = sorted(set(pool.slide_names))
    assert len(names) == 3 and len(tds) > 0 and len(vds) > 0
    tr_names = {pool.slide_names[i] for i in tds.indices}
    va_names = {pool.slide_names[i] for i in vds.indices}
    assert tr_names.isdisjoint(va_names) and len(va_names) == 1  # the reference's slide-level split (src/main.py:412-416)
    c = vds.get_class_counts()
    assert len(c) < 2 or c[0] == c[1]
    model, hist = train.train_resnet_classifier(None, strategy="weighted_loss", epochs=1, batch_size=8, max_steps=2,
                                                save_path=str(tmp_path / "clf.pth"), slides=slides, level=3)
    assert len(hist) == 1 and np.isfinite(hist[0][0]) and (tmp_path / "clf.pth").exists()
    model, hist = train.train_resnet_classifier(None, strategy="self_supervised", epochs=1, batch_size=8, max_steps=2, simclr_epochs=1,
                                                simclr_path=str(tmp_path / "enc.pth"), save_path=str(tmp_path / "clf2.pth"),
                                                slides=slides, level=3, simclr_precision="fp16")
    assert (tmp_path / "enc.pth").exists() and (tmp_path / "clf2.pth").exists() and np.isfinite(hist[0][0])
