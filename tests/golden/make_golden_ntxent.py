"""Generate tests/golden/ntxent_golden.npz.  Run in the BUILD container:

    python tests/golden/make_golden_ntxent.py

Source of truth: the REFERENCE's own ``nt_xent_loss`` (src/models/simclr.py:31-54).  The module cannot be imported here
(its top-level ``import torchvision.models`` raises an ordinary ModuleNotFoundError: torchvision is not installed), but the
function itself is pure torch: this script parses the file, compiles ONLY that function definition and calls it with the
real ``torch`` / ``torch.nn.functional`` -- nothing is stubbed, nothing else of the file runs.  Stored: seeded inputs, the
loss value and its autograd gradient.  The reference's source never enters this repository; the fixture is data only."""
import ast
import os

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/models/simclr.py"
CASES = [(8, 128, 0.5, 0), (37, 64, 0.2, 1), (5, 256, 1.0, 2), (64, 128, 0.5, 3)]  # (N, D, temperature, seed)


def reference_function():
    tree = ast.parse(open(REF).read(), REF)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "nt_xent_loss")
    ns = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
    return ns["nt_xent_loss"]


def make_inputs(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    zi = torch.randn(n, d, generator=g) * 1.7
    zj = zi * 0.6 + torch.randn(n, d, generator=g)
    return zi, zj


def main():
    fn = reference_function()
    out = {}
    for k, (n, d, t, seed) in enumerate(CASES):
        zi, zj = make_inputs(n, d, seed)
        a, b = zi.clone().requires_grad_(True), zj.clone().requires_grad_(True)
        loss = fn(a, b, temperature=t)
        loss.backward()
        out[f"c{k}_meta"] = np.array([n, d, seed], np.int64)
        out[f"c{k}_t"] = np.float64(t)
        out[f"c{k}_zi"], out[f"c{k}_zj"] = zi.numpy(), zj.numpy()
        out[f"c{k}_loss"] = np.float64(float(loss))
        out[f"c{k}_gi"], out[f"c{k}_gj"] = a.grad.numpy(), b.grad.numpy()
        print(f"case {k}: N={n} D={d} T={t}: loss {float(loss):.6f}")
    np.savez_compressed(os.path.join(HERE, "ntxent_golden.npz"), **out)


if __name__ == "__main__":
    main()
