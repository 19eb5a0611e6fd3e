"""Generate tests/golden/mil_golden.npz + mil_dataset_ref.json.  Run in the BUILD container:

    python tests/golden/make_golden_mil.py

Source of truth: the REFERENCE's own src/models/mil_classifier.py and
src/datasets/mildataset.py, imported by file location (pure torch / numpy: they import
cleanly here).  Only data is stored: the seeded parameters that were loaded
into the reference modules, the input bags, and the outputs the reference produced.
"""
import importlib.util
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    mc = load(os.path.join(REF, "models", "mil_classifier.py"), "ref_mil_classifier")
    md = load(os.path.join(REF, "datasets", "mildataset.py"), "ref_mildataset")
    out = {}
    g = torch.Generator().manual_seed(11)
    bags = [torch.randn(n, 512, generator=g) * (0.5 + 0.5 * i) for i, n in enumerate((1, 37, 300))]
    for i, b in enumerate(bags):
        out[f"bag{i}"] = b.numpy()
    for pooling in ("attention", "mean", "max"):
        torch.manual_seed(5)
        m = mc.MILClassifier(512, num_classes=2, pooling=pooling).eval()
        sd = m.state_dict()
        for k, v in sd.items():
            out[f"{pooling}.sd.{k}"] = v.numpy()
        with torch.no_grad():
            for i, b in enumerate(bags):
                logits, attn = m(b)
                out[f"{pooling}.logits{i}"] = logits.numpy()
                if attn is not None:
                    out[f"{pooling}.attn{i}"] = attn.numpy()
    np.savez_compressed(os.path.join(HERE, "mil_golden.npz"), **out)

    # dataset grouping: the reference's WSIMILDDataset on a small (features, labels, paths) triple
    names = []
    for slide, cols in (("tumor_001", 3), ("normal_002", 2), ("test_010", 2)):
        for x in range(cols):
            for y in range(2):
                lab = "tumor" if (slide.startswith("tumor") and x == 1 and y == 0) else "normal"
                names.append(f"level_3/{slide}/{slide}_x{224 * x}_y{224 * y}_{lab}.png")
    rng = np.random.RandomState(3)
    order = rng.permutation(len(names))
    names = [names[i] for i in order]
    feats = rng.randn(len(names), 512).astype(np.float32)
    labels = np.array([1 if "_tumor" in n else 0 for n in names], np.int64)
    with tempfile.TemporaryDirectory() as td:
        fp, lp, pp = (os.path.join(td, f) for f in ("f.npy", "l.npy", "p.txt"))
        np.save(fp, feats), np.save(lp, labels)
        open(pp, "w").write("\n".join(names) + "\n")
        ds = md.WSIMILDDataset(fp, lp, pp)
        bags_out = []
        for i in range(len(ds)):
            f, y = ds[i]
            # identify the member rows by matching feature vectors (rows are unique)
            idx = [int(np.where((feats == r.numpy()).all(axis=1))[0][0]) for r in f]
            bags_out.append({"rows": idx, "wsi_label": int(y)})
    json.dump({"paths": names, "labels": labels.tolist(), "bags": bags_out},
              open(os.path.join(HERE, "mil_dataset_ref.json"), "w"), indent=0)
    print("wrote mil_golden.npz, mil_dataset_ref.json:", len(out), "arrays,", len(bags_out), "bags")


if __name__ == "__main__":
    sys.exit(main())
