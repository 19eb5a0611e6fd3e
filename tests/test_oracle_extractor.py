"""Oracle pin: grid arithmetic against the known answers recorded in the reference's
notebooks, and the documented quirks of src/main.py:609-732."""
import numpy as np

from oracle import extractor_ref as E


def test_known_answer_level0_grid_02_notebook():
    # src/02_patch_extraction.ipynb:88-90: 97792 x 221184 level 0, P=1792 -> 6642 full
    # patches, 1.39 % of the area lost.
    W, H, P = 97792, 221184, 1792
    _, _, _, grid = E.window_grid(W, H, 0, stride=P, pad=False)
    full = [g for g in grid if g[2] == P and g[3] == P]
    assert len(full) == 6642
    lost = 1.0 - len(full) * P * P / (W * H)
    assert abs(lost * 100 - 1.39) < 0.005


def test_known_answer_tumor_076_level3_grid():
    # src/01_02_data_explor_patch_extraction.ipynb:479 (level-6 dims 1792x1568, downsample
    # 64) => level-3 dims 14336 x 12544; :416 records 2666 + 912 kept = 3578 of a 64x56 grid.
    W3, H3 = 1792 * 8, 1568 * 8
    P, stride, pad, grid = E.window_grid(W3, H3, 3)
    assert (P, stride, pad) == (224, 224, (0, 0))
    assert len(grid) == 64 * 56 == 3584
    assert 2666 + 912 <= len(grid)


def test_stride_quirk_is_224_at_every_level():
    for level, P in E.PATCH_SIZES.items():
        p, stride, _, _ = E.window_grid(5000, 4000, level)
        assert p == P and stride == 224


def test_loop_order_x_outer_y_inner_and_skip_rule():
    _, _, (pad_w, pad_h), grid = E.window_grid(500, 700, 2)  # P = 448
    assert (pad_w, pad_h) == (396, 196)
    xs = [g[0] for g in grid]
    assert xs == sorted(xs)
    first_col = [g[1] for g in grid if g[0] == 0]
    assert first_col == list(range(0, 700, 224))
    assert all(g[0] < 500 and g[1] < 700 for g in grid)
    assert (448, 672, 52, 28) in grid  # clipped corner window


def _slide_l3(l3):
    """A 4-level pyramid whose level 3 is exactly ``l3`` (finer levels by pixel repeat)."""
    return E.ArraySlide([np.repeat(np.repeat(l3, 2 ** (3 - k), axis=0), 2 ** (3 - k), axis=1) for k in range(4)])


def _slide(w, h, seed=0):
    rng = np.random.RandomState(seed)
    return _slide_l3(rng.randint(0, 256, (h, w, 3), dtype=np.uint8))


def test_border_padding_is_white_and_counts_in_mean():
    s = _slide_l3(np.random.RandomState(0).randint(0, 100, (380, 400, 3), dtype=np.uint8))
    wins, pix = E.extract_patches_ref(s, 3)
    by = {(w.x, w.y): (w, p) for w, p in zip([w for w in wins if w.keep], pix)}
    w, p = by[(224, 224)]
    assert (w.pw, w.ph) == (176, 156)
    assert (p[156:, :, :] == 255).all() and (p[:, 176:, :] == 255).all()
    assert np.array_equal(p[:156, :176], s.levels[3][224:380, 224:400])
    assert w.pixel_sum == int(p.sum(dtype=np.int64))


def test_whiteness_integer_form_equals_float_mean():
    rng = np.random.RandomState(1)
    for P in (224, 448):
        thr = 240 * 3 * P * P
        for delta in (-2, -1, 0, 1, 2):
            a = np.full((P, P, 3), 240, np.uint8)
            flat = a.reshape(-1)
            if delta > 0:
                flat[:delta] = 241
            elif delta < 0:
                flat[: -delta] = 239
            s = int(a.sum(dtype=np.int64))
            assert s == thr + delta
            assert (not (np.mean(a) > 240)) == E.whiteness_keep_integer(s, P)
    a = rng.randint(0, 256, (224, 224, 3), dtype=np.uint8)
    assert (not (np.mean(a) > 240)) == E.whiteness_keep_integer(int(a.sum(dtype=np.int64)), 224)


def test_all_white_window_is_dropped_and_labels_from_mask():
    l0 = np.full((448, 448, 3), 255, np.uint8)
    l0[:224, :224] = 100
    s = _slide_l3(l0)
    poly = [[(80.0, 80.0), (800.0, 80.0), (800.0, 800.0), (80.0, 800.0)]]  # level-0 coordinates
    wins, pix = E.extract_patches_ref(s, 3, polygons_l0=poly)
    d = {(w.x, w.y): w for w in wins}
    assert d[(0, 0)].keep and d[(0, 0)].label == E.LABEL_TUMOR
    assert not d[(224, 224)].keep and d[(224, 224)].label == E.LABEL_NORMAL
    assert len(pix) == 1
    assert E.patch_file_name("tumor_001", 0, 224, 1) == "tumor_001_x0_y224_tumor.png"


def test_mask_truncation_and_outline():
    m = np.array(E.parse_mask([[(15.9, 15.9), (40.2, 15.9), (40.2, 40.9)]], (50, 50), (100, 100)))
    assert m[7, 7] == 255 and m[7, 20] == 255 and m[20, 20] == 255  # int() truncation: (7,7),(20,7),(20,20)
    assert m[20, 7] == 0
