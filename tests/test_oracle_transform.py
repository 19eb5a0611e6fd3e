"""Oracle pin: the numpy restatement of Pillow's resampler == Pillow itself, bit for
bit, for the four window sizes of the reference (src/main.py:614) and a few others."""
import numpy as np
import pytest

from oracle import transform_ref as T


@pytest.mark.parametrize("P", [224, 448, 896, 1792, 672])
def test_restated_resample_equals_pillow(P):
    rng = np.random.RandomState(P)
    for kind in ("noise", "smooth", "extremes"):
        if kind == "noise":
            a = rng.randint(0, 256, (P, P, 3), dtype=np.uint8)
        elif kind == "smooth":
            yy, xx = np.mgrid[0:P, 0:P]
            a = np.stack([(xx * 255 // P), (yy * 255 // P), ((xx + yy) % 256)], -1).astype(np.uint8)
        else:
            a = (rng.randint(0, 2, (P, P, 3)) * 255).astype(np.uint8)
        assert np.array_equal(T.resample_restated(a), T.pillow_resize(a)), (P, kind)


def test_identity_at_224():
    a = np.random.RandomState(0).randint(0, 256, (224, 224, 3), dtype=np.uint8)
    assert np.array_equal(T.pillow_resize(a), a)


def test_interior_coefficients_are_dyadic_triangle():
    # known structure for integer scale s: weights (2t+1)/(2 s^2) -> exact in 22-bit fixed point
    for s in (2, 4, 8):
        bounds, kk, ksize = T.precompute_coeffs(224 * s, 224)
        assert ksize == 2 * s + 1
        j = 100
        assert bounds[j, 0] == s * j - s // 2 and bounds[j, 1] == 2 * s
        w = [(2 * t + 1) if t < s else (2 * (2 * s - 1 - t) + 1) for t in range(2 * s)]
        expect = np.array(w, np.int64) * (1 << 22) // (2 * s * s)
        assert np.array_equal(kk[j, : 2 * s], expect)
        assert kk[j].sum() == 1 << 22


def test_lut_equals_torch_ops():
    a = np.arange(256, dtype=np.uint8).reshape(1, 256, 1).repeat(3, axis=2)
    img = np.zeros((224, 224, 3), np.uint8)
    img[0, :] = 0
    img[:1, :224] = a[:, :224]
    img[1:2, :32] = a[:, 224:]
    t = T.to_tensor_normalize(img)
    lut = T.normalize_lut()
    for c in range(3):
        assert np.array_equal(t[c, 0, :224], lut[c, :224])
        assert np.array_equal(t[c, 1, :32], lut[c, 224:])
