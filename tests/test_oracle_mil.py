"""CPU: the MIL oracle against the fixtures produced by the reference's own modules
(tests/golden/make_golden_mil.py), and the host-side bag builder against the same."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import mil_ref
from ss25_hierarchical_multiscale_image_classification_amd import mil

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "mil_golden.npz"))


def sd_of(gold, pooling):
    pre = f"{pooling}.sd."
    return {k[len(pre):]: gold[k] for k in gold.files if k.startswith(pre)}


@pytest.mark.parametrize("pooling", ["attention", "mean", "max"])
def test_oracle_matches_reference_outputs(gold, pooling):
    sd = sd_of(gold, pooling)
    for i in range(3):
        logits, attn, _ = mil_ref.mil_forward_ref(sd, gold[f"bag{i}"], pooling)
        np.testing.assert_allclose(logits, gold[f"{pooling}.logits{i}"], rtol=2e-5, atol=2e-6)
        if pooling == "attention":
            np.testing.assert_allclose(attn, gold[f"{pooling}.attn{i}"], rtol=2e-5, atol=1e-8)
            assert abs(float(attn.sum()) - 1.0) < 1e-5
        else:
            assert attn is None


def test_grouping_matches_reference_dataset():
    ref = json.load(open(os.path.join(GOLD, "mil_dataset_ref.json")))
    got = mil_ref.group_patches_ref(ref["paths"], ref["labels"])
    assert [(rows, lab) for _, rows, lab in got] == [(b["rows"], b["wsi_label"]) for b in ref["bags"]]
    # the host-side builder (product code) gives the same bags, as CSR offsets over a row order
    order, offsets, names, wsi_labels = mil.group_patches_by_wsi(ref["paths"], ref["labels"])
    assert [order[offsets[i]:offsets[i + 1]].tolist() for i in range(len(names))] == [b["rows"] for b in ref["bags"]]
    assert wsi_labels.tolist() == [b["wsi_label"] for b in ref["bags"]]
    # documented intent (slide name only): 3 slides
    _, off2, names2, lab2 = mil.group_patches_by_wsi(ref["paths"], ref["labels"], by_slide=True)
    assert sorted(names2) == ["normal_002", "test_010", "tumor_001"] and len(off2) == 4
    assert dict(zip(names2, lab2.tolist())) == {"tumor_001": 1, "normal_002": 0, "test_010": 0}


def test_module_mirror_state_dict_and_train_path(gold):
    # same constructor, same state_dict keys; the autograd (train-mode / CPU) path equals the reference outputs
    for pooling in ("attention", "mean", "max"):
        m = mil.MILClassifier(512, num_classes=2, pooling=pooling)
        sd = {k: torch.from_numpy(v) for k, v in sd_of(gold, pooling).items()}
        assert set(m.state_dict().keys()) == set(sd.keys())
        m.load_state_dict(sd)
        m.train()
        logits, attn = m(torch.from_numpy(gold["bag1"]))
        np.testing.assert_allclose(logits.detach().numpy(), gold[f"{pooling}.logits1"], rtol=2e-5, atol=2e-6)
        assert (attn is None) == (pooling != "attention")
    with pytest.raises(ValueError):
        mil.MILClassifier(512, pooling="median")
    m = mil.MILClassifier(512).eval()
    with pytest.raises(Exception):  # eval on a CPU tensor: the HIP path has no CPU fallback
        m(torch.zeros(3, 512))


def test_dataset_mirror(tmp_path):
    ref = json.load(open(os.path.join(GOLD, "mil_dataset_ref.json")))
    feats = np.arange(len(ref["paths"]) * 512, dtype=np.float32).reshape(-1, 512)
    np.save(tmp_path / "f.npy", feats), np.save(tmp_path / "l.npy", np.array(ref["labels"]))
    (tmp_path / "p.txt").write_text("\n".join(ref["paths"]) + "\n")
    ds = mil.WSIMILDDataset(str(tmp_path / "f.npy"), str(tmp_path / "l.npy"), str(tmp_path / "p.txt"))
    assert len(ds) == len(ref["bags"])
    for i, b in enumerate(ref["bags"]):
        f, y = ds[i]
        assert f.dtype == torch.float32 and torch.equal(f, torch.from_numpy(feats[b["rows"]])) and int(y) == b["wsi_label"]
        assert y.dtype == torch.long
