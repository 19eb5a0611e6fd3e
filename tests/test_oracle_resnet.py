"""Oracle self-checks for the ResNet18 restatement (the graph lives in torchvision,
absent here: PARITY UNPINNED by reference fixtures -- these are the known answers of
the published architecture and internal consistency checks)."""
import torch

from oracle import ntxent_ref, resnet18_ref as R
from ss25_hierarchical_multiscale_image_classification_amd import synth


def test_param_count_and_shapes_of_published_resnet18():
    sd = synth.seeded_resnet18_state_dict(0, num_classes=2)
    shapes = dict(R.expected_param_shapes(2))
    learnable = 0
    for k, shape in shapes.items():
        assert tuple(sd[k].shape) == shape, k
        if "running" not in k:
            learnable += sd[k].numel()
    assert learnable == R.N_CONV_BN_PARAMS + R.N_FC_PARAMS_2CLASS == 11_177_538
    # ImageNet head: 11,689,512 is the well-known total
    assert R.N_CONV_BN_PARAMS + 512 * 1000 + 1000 == 11_689_512


def test_functional_equals_module_form_and_shapes():
    sd = synth.seeded_resnet18_state_dict(3, num_classes=2)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    taps = {}
    feats, logits = R.resnet18_forward(x, sd, taps)
    assert feats.shape == (2, 512) and logits.shape == (2, 2)
    assert taps["stem"].shape == (2, 64, 112, 112) and taps["maxpool"].shape == (2, 64, 56, 56)
    assert taps["layer2.0"].shape == (2, 128, 28, 28) and taps["layer4.1"].shape == (2, 512, 7, 7)
    m = R.ModuleResNet18(2)
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        assert torch.allclose(m(x), logits, atol=1e-5, rtol=1e-5)
    assert torch.equal(R.predict_labels(logits), logits.argmax(1))


def test_all_reference_key_layouts_give_same_result():
    bare = synth.seeded_resnet18_state_dict(1, num_classes=2)
    x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    f0, l0 = R.resnet18_forward(x, bare)
    for prefix in ("model.", "module.model.", "encoder."):
        f, l = R.resnet18_forward(x, {prefix + k: v for k, v in bare.items()})
        assert torch.equal(f, f0) and torch.equal(l, l0)
    idx = {"conv1": "0", "bn1": "1", "layer1": "4", "layer2": "5", "layer3": "6", "layer4": "7"}
    ext = {}
    for k, v in bare.items():
        head, _, tail = k.partition(".")
        if head in idx:
            ext[f"features.{idx[head]}.{tail}"] = v
    f, l = R.resnet18_forward(x, ext)
    assert torch.equal(f, f0) and l is None


def test_nt_xent_closed_form():
    # orthonormal z: S = I/T off-diagonal 0 => loss = log(1 + (2N-2) e^{0}... ) closed form
    n, d, T = 4, 16, 0.5
    z = torch.eye(2 * n, d)
    zi, zj = z[:n], z[n:]
    loss = ntxent_ref.nt_xent_loss_ref(zi, zj, T)
    # every row: positives sim = 0, all other 2N-2 off-diagonal sims = 0 -> -0 + log(2N-1)
    assert torch.allclose(loss, torch.log(torch.tensor(2.0 * n - 1)))
    # identical pairs: positive sim = 1/T
    zi = torch.randn(n, d, generator=torch.Generator().manual_seed(0))
    l2 = ntxent_ref.nt_xent_loss_ref(zi, zi.clone(), T)
    zz = torch.nn.functional.normalize(torch.cat([zi, zi]), dim=1)
    s = zz @ zz.T / T
    s.fill_diagonal_(float("-inf"))
    manual = (-1.0 / T + torch.logsumexp(s, 1)).mean()
    assert torch.allclose(l2, manual, atol=1e-6)


def test_ntxent_restatement_equals_the_reference_function(golden_dir):
    """tests/golden/ntxent_golden.npz was produced by the REFERENCE's own nt_xent_loss (src/models/simclr.py:31-54,
    compiled from its source by make_golden_ntxent.py): value and autograd gradient on four seeded cases."""
    import numpy as np

    g = np.load(f"{golden_dir}/ntxent_golden.npz")
    for k in range(4):
        zi, zj = torch.from_numpy(g[f"c{k}_zi"]).requires_grad_(True), torch.from_numpy(g[f"c{k}_zj"]).requires_grad_(True)
        loss = ntxent_ref.nt_xent_loss_ref(zi, zj, float(g[f"c{k}_t"]))
        loss.backward()
        assert abs(float(loss) - float(g[f"c{k}_loss"])) <= 1e-6 * max(1.0, abs(float(g[f"c{k}_loss"])))
        for got, want in ((zi.grad, g[f"c{k}_gi"]), (zj.grad, g[f"c{k}_gj"])):
            assert float((got - torch.from_numpy(want)).abs().max()) <= 1e-6 * float(np.abs(want).max()) + 1e-12
