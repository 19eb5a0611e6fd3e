"""Native training step (csrc/train.hip through the C ABI) against the autograd oracle (torch-CPU fp32,
oracle/train_ref.py): every parameter gradient of a train-mode step, the loss, the updated running statistics.

The oracle is given the ACTIVATION PATTERN of the run under test (which ReLU units are on, which element won
each max-pool window; oracle/train_ref.py::_ReluGiven): a ReLU's derivative is discontinuous at 0, two fp32
implementations disagree on a handful of the ~10 M units of a step, and each disagreement moves the heavily
cancelling gradient sums by percents (torch's own fp32 and fp64 runs differ by 1e-2 on layer1-3 gradients for
that reason).  The oracle's VALUES stay its own; the test also bounds how many units disagree.

Tolerances (norm-relative per tensor, max|a-b| / max|b|): fp32 on both sides (exact f32 MFMA here, fp32 FMA on
the CPU), so what remains is summation order through 20 train-mode batch-norms and their backward passes:
    loss 1e-5, every gradient tensor 5e-4 [measured: <= 5e-5], running statistics 1e-4.
"""
import os

import numpy as np
import pytest
import torch

from oracle import train_ref as TR
from ss25_hierarchical_multiscale_image_classification_amd import capi, train_native as TN
from ss25_hierarchical_multiscale_image_classification_amd.resnet import ResNet18Classifier
from ss25_hierarchical_multiscale_image_classification_amd.simclr import SimCLRModel

pytestmark = pytest.mark.gpu
GRAD_TOL = 5e-4


def rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _randomise_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=g)
            m.bias.data = 0.1 * torch.randn(m.bias.shape, generator=g)
            m.running_mean.data = 0.1 * torch.randn(m.bias.shape, generator=g)
            m.running_var.data = 0.5 + torch.rand(m.bias.shape, generator=g)


def patterns(enc, slot, hidden=None):
    """Activation pattern of the forward in workspace ``slot`` (and the projector's hidden layer), keyed like the
    oracle's taps."""
    m = {"pool_idx": enc.tap(slot, "pool_idx", 0).cpu()}
    for i, e in enumerate(enc.table):
        if "downsample" not in e["conv"]:
            m[e["conv"] + ".post"] = enc.tap(slot, "post", i).cpu() > 0
    if hidden is not None:
        m["projector.hidden"] = hidden.cpu() > 0
    return m


def count_disagreements(sd_bare, x, masks):
    """How many ReLU units the oracle's own forward would switch differently (bounded, reported)."""
    p, st = TR._split(sd_bare)
    taps = {}
    with torch.no_grad():
        TR.encoder_train_forward(x, p, st, taps)
    return sum(int(((taps[k] > 0) != v).sum()) for k, v in masks.items() if k in taps and k != "pool_idx")


def test_linear_cross_entropy_adam_against_torch():
    torch.manual_seed(0)
    dev = torch.device("cuda")
    for (M, N, K, relu) in ((16, 512, 512, True), (16, 128, 512, False), (37, 2, 512, False), (130, 70, 33, True)):
        flat = TN.FlatAdam(TN.NativeLinear.floats(N, K), dev, 1e-3)
        lin = TN.NativeLinear(flat, 0, N, K, relu=relu)
        w, b, x = torch.randn(N, K) * 0.05, torch.randn(N) * 0.1, torch.randn(M, K)
        lin.w.copy_(w), lin.b.copy_(b)
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = torch.nn.functional.linear(xr, wr, br)
        yr = torch.relu(yr) if relu else yr
        dy = torch.randn(M, N)
        yr.backward(dy)
        xd = x.to(dev)
        y = lin.forward(xd)
        dx = lin.backward(xd, y, dy.to(dev), accumulate=False)
        assert rel(y, yr.detach()) < 1e-5 and rel(dx, xr.grad) < 1e-5
        assert rel(lin.dw, wr.grad) < 1e-5 and rel(lin.db, br.grad) < 1e-5
        lin.backward(xd, y, dy.to(dev), accumulate=True, need_dx=False)
        assert rel(lin.dw, 2 * wr.grad) < 1e-5 and rel(lin.db, 2 * br.grad) < 1e-5
    # weighted cross-entropy
    lib = capi.load_library()
    for cw in (None, torch.tensor([1.0, 4.2, 0.5])):
        logits = torch.randn(300, 3)
        labels = torch.randint(0, 3, (300,))
        lr_ = logits.clone().requires_grad_(True)
        loss_r = torch.nn.functional.cross_entropy(lr_, labels, weight=cw)
        loss_r.backward()
        ld, lab = logits.to(dev), labels.to(dev)
        loss, dl, scratch = torch.empty((), device=dev), torch.empty_like(ld), torch.zeros(2 + 8 * 2, device=dev)
        capi._check(lib.hipac_cross_entropy_fwd_bwd(ld.data_ptr(), lab.data_ptr(), None if cw is None else cw.to(dev).data_ptr(),
                                                    300, 3, loss.data_ptr(), dl.data_ptr(), scratch.data_ptr(), capi._stream()), "ce")
        assert abs(float(loss) - float(loss_r)) < 1e-5 and rel(dl, lr_.grad) < 1e-5
    # Adam
    flat = TN.FlatAdam(5000, dev, 1e-3)
    p0 = torch.randn(5000)
    flat.params.copy_(p0)
    q, m, v = p0.clone(), torch.zeros(5000), torch.zeros(5000)
    for t in range(1, 4):
        g = torch.randn(5000)
        flat.grads.copy_(g)
        flat.step()
        q, m, v = TR.adam_ref(q, g, m, v, t, 1e-3)
        assert torch.allclose(flat.params.cpu(), q, rtol=1e-5, atol=1e-7)


def test_simclr_step_every_gradient_against_oracle_autograd():
    """B = 8 views per side (SURVEY a-12): z_i = model(x_i); z_j = model(x_j); NT-Xent; backward -- per-replica BN."""
    torch.manual_seed(11)
    model = SimCLRModel()
    _randomise_bn(model, 4)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x_i, x_j = torch.randn(8, 3, 224, 224), torch.randn(8, 3, 224, 224)
    tr = TN.NativeSimCLRTrainer(sd, device="cuda", lr=1e-3)
    loss = tr.forward_backward(x_i.cuda(), x_j.cuda())
    torch.cuda.synchronize()
    m_i, m_j = patterns(tr.encoder, 0, tr.last_hidden[0]), patterns(tr.encoder, 1, tr.last_hidden[1])
    from oracle.resnet18_ref import canonical_state_dict
    bare = canonical_state_dict({k: v for k, v in sd.items() if not k.startswith("projector.")})
    n_dis, n_dis_j = count_disagreements(bare, x_i, m_i), count_disagreements(bare, x_j, m_j)
    print(f"ReLU units switched differently by the oracle's own forward: view i {n_dis}, view j {n_dis_j} of ~5.3 M each")
    assert n_dis <= 40 and n_dis_j <= 40
    loss_ref, grads_ref, stats_ref = TR.simclr_step_ref(sd, x_i, x_j, masks_i=m_i, masks_j=m_j)
    assert abs(float(loss) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref)) + 1e-6
    # and WITHOUT handing over any activation pattern (the oracle decides every ReLU / max-pool itself): the loss and
    # the gradients nearest to it are insensitive to the few units that flip (deep layers are not: module docstring)
    loss_free, grads_free, _ = TR.simclr_step_ref(sd, x_i, x_j)
    assert abs(float(loss) - float(loss_free)) <= 1e-5 * abs(float(loss_free)) + 1e-6
    for k in ("projector.2.weight", "projector.2.bias", "projector.0.weight", "projector.0.bias"):
        assert rel(tr.grad_dict()[k], grads_free[k]) <= 5e-3, (k, rel(tr.grad_dict()[k], grads_free[k]))
    got = tr.grad_dict()
    errs = {name: rel(got[name], g) for name, g in grads_ref.items()}
    print({k: f"{v:.1e}" for k, v in errs.items() if k.endswith("weight") and ("conv" in k or "downsample.0" in k or "projector" in k)})
    worst = max(errs.items(), key=lambda t: t[1])
    for name, g in grads_ref.items():
        assert got[name].shape == g.shape and errs[name] <= GRAD_TOL, (name, errs[name])
    print(f"simclr step: loss {float(loss):.6f} (oracle {float(loss_ref):.6f}); worst gradient {worst[0]} {worst[1]:.2e}")
    new = tr.state_dict()
    for k, v in stats_ref.items():
        assert rel(new[k], v) <= 1e-4, k
    assert int(new["encoder.bn1.num_batches_tracked"]) == 2
    # one optimizer step, then the parameters against Adam applied to the oracle's gradients
    tr.encoder.opt.step(), tr.head.step()
    after = tr.state_dict()
    for name in ("encoder.conv1.weight", "encoder.layer3.0.downsample.0.weight", "encoder.layer4.1.bn2.bias", "projector.2.weight"):
        p, _, _ = TR.adam_ref(sd[name], grads_ref[name], torch.zeros_like(sd[name]), torch.zeros_like(sd[name]), 1, 1e-3)
        # Adam's first step moves every entry by ~lr * sign(g): compare the moves where the gradient is not negligible
        big = grads_ref[name].abs() > 1e-3 * grads_ref[name].abs().max()
        assert torch.allclose(after[name][big], p[big], atol=2e-5), name


def test_classifier_step_against_oracle_autograd():
    """The fine-tune step (src/main.py:499-506): CE with class weights, B = 8, every gradient."""
    torch.manual_seed(13)
    model = ResNet18Classifier()
    _randomise_bn(model, 6)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x, y = torch.randn(8, 3, 224, 224), torch.tensor([0, 1, 1, 0, 0, 0, 1, 0])
    w = torch.tensor([1.0, 2.5])
    tr = TN.NativeClassifierTrainer(sd, device="cuda", lr=1e-4, class_weights=w, precision="fp32")  # (fp16: test_gpu_train_amp.py)
    loss, logits = tr.forward_backward(x.cuda(), y.cuda())
    masks = patterns(tr.encoder, 0)
    from oracle.resnet18_ref import canonical_state_dict
    n_dis = count_disagreements({k: v for k, v in canonical_state_dict(sd).items() if not k.startswith("fc.")}, x, masks)
    print(f"classifier step: ReLU units switched differently by the oracle's own forward: {n_dis}")
    assert n_dis <= 40
    loss_ref, logits_ref, grads_ref, _ = TR.classifier_step_ref(sd, x, y, w, masks=masks)
    assert abs(float(loss) - float(loss_ref)) <= 1e-5 and rel(logits, logits_ref) <= 1e-4
    # unmasked: loss, logits and the fc gradients do not depend on which of the near-zero units flipped
    loss_free, logits_free, grads_free, _ = TR.classifier_step_ref(sd, x, y, w)
    assert abs(float(loss) - float(loss_free)) <= 1e-5 and rel(logits, logits_free) <= 1e-4
    for k in ("fc.weight", "fc.bias"):
        assert rel(tr.grad_dict()[k], grads_free[k]) <= 5e-3, k
    got = tr.grad_dict()
    for name, g in grads_ref.items():
        assert rel(got[name], g) <= GRAD_TOL, (name, rel(got[name], g))
    out = tr.state_dict()
    assert set(out) == set(sd)  # the reference's key layout (model.*), ready for torch.save / load_state_dict
    model.load_state_dict(out)


def test_odd_batch_and_workspace_reuse():
    """Batch sizes that are not multiples of the kernels' tiles; a second step in the same workspaces."""
    torch.manual_seed(17)
    model = SimCLRModel()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    tr = TN.NativeSimCLRTrainer(sd, device="cuda")
    for n in (3, 5):
        x_i, x_j = torch.randn(n, 3, 224, 224), torch.randn(n, 3, 224, 224)
        cur = tr.state_dict()
        loss = tr.forward_backward(x_i.cuda(), x_j.cuda())
        loss_ref, grads_ref, _ = TR.simclr_step_ref(cur, x_i, x_j, masks_i=patterns(tr.encoder, 0, tr.last_hidden[0]),
                                                    masks_j=patterns(tr.encoder, 1, tr.last_hidden[1]))
        assert abs(float(loss) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref)) + 1e-6
        got = tr.grad_dict()
        for name in ("encoder.conv1.weight", "encoder.layer1.0.conv1.weight", "encoder.layer2.0.downsample.0.weight",
                     "encoder.layer4.1.conv2.weight", "encoder.bn1.weight", "projector.0.bias"):
            assert rel(got[name], grads_ref[name]) <= GRAD_TOL, (n, name)


def test_training_loops_end_to_end(tmp_path, monkeypatch):
    """pretrain_simclr (src/models/simclr.py:68-124) and the self_supervised fine-tune loop (src/main.py:536-606) on a
    tiny PNG tree, a few native steps each: checkpoints in the reference's key layouts, the classifier really starts
    from the pre-trained encoder, validation runs on the HIP inference path."""
    import os

    import numpy as np
    from PIL import Image

    from ss25_hierarchical_multiscale_image_classification_amd import train
    from ss25_hierarchical_multiscale_image_classification_amd.simclr import pretrain_simclr

    rng = np.random.RandomState(0)
    root = tmp_path / "patches"
    for s in range(4):
        d = root / f"tumor_{s:03d}"
        d.mkdir(parents=True)
        for k in range(6):
            lab = "tumor" if (k + s) % 2 == 0 else "normal"
            Image.fromarray(rng.randint(0, 256, (224, 224, 3), dtype=np.uint8), "RGB").save(d / f"tumor_{s:03d}_x{224 * k}_y0_{lab}.png")
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    model, hist = pretrain_simclr(str(root), epochs=2, batch_size=4, lr=1e-3, num_workers=0, out_dir=str(tmp_path), max_steps=3,
                                  verbose=False)
    assert len(hist) >= 1 and all(np.isfinite(hist))
    sd = torch.load(tmp_path / "simclr_encoder.pth", map_location="cpu", weights_only=True)
    assert set(sd) == set(model.state_dict()) and "encoder.conv1.weight" in sd and "projector.2.bias" in sd
    assert os.path.exists(tmp_path / "simclr_encoder_best.pth")
    init = SimCLRModel().state_dict()
    assert not torch.equal(sd["encoder.layer1.0.conv1.weight"], init["encoder.layer1.0.conv1.weight"])  # it trained
    clf, history = train.train_resnet_classifier(str(root), strategy="self_supervised", epochs=1, batch_size=4, precision="fp16",
                                                 save_path=str(tmp_path / "clf.pth"), simclr_path=str(tmp_path / "simclr_encoder.pth"),
                                                 max_steps=2)
    assert len(history) == 1 and np.isfinite(history[0][0]) and 0.0 <= history[0][2] <= 1.0
    out = torch.load(tmp_path / "clf.pth", map_location="cpu", weights_only=True)
    assert set(out) == set(ResNet18Classifier().state_dict())  # model.* layout incl. model.fc
    # two Adam steps at lr 1e-4 away from the SimCLR encoder: close to it, far from a fresh init
    d_pre = float((out["model.layer2.0.conv1.weight"] - sd["encoder.layer2.0.conv1.weight"]).abs().max())
    assert 0 < d_pre < 1e-3
    # the plain --train loop as well (class weights 1/count)
    clf2, h2 = train.train_resnet_classifier(str(root), strategy=None, epochs=1, batch_size=4, precision="bf16",
                                             save_path=str(tmp_path / "clf2.pth"), max_steps=1)
    assert np.isfinite(h2[0][0])


def test_two_rank_simclr_step(tmp_path):
    """Two ranks (gloo, both on cuda:0), 4 view pairs each: NT-Xent over the ALL-GATHERED z, batch-norm per replica,
    gradients all-reduced (SUM) -- i.e. what nn.DataParallel computes in the reference.  Oracle: one
    autograd graph in which each replica's images pass the encoder separately (own batch statistics) and the loss
    sees all z."""
    import subprocess
    import sys

    import bench
    from oracle.ntxent_ref import nt_xent_loss_ref
    from oracle.resnet18_ref import canonical_state_dict

    torch.manual_seed(23)
    model = SimCLRModel()
    _randomise_bn(model, 8)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.randn(2, 2, 4, 3, 224, 224)
    torch.save(sd, tmp_path / "sd.pt")
    torch.save(x, tmp_path / "x.pt")
    helper = os.path.join(os.path.dirname(__file__), "helpers", "simclr_rank.py")
    procs = []
    for r, (_, env) in enumerate(bench.child_commands([], 2, bench._free_port())):
        procs.append(subprocess.Popen([sys.executable, helper, str(tmp_path)], env=env))
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    res = [torch.load(tmp_path / f"rank{r}.pt", weights_only=True) for r in range(2)]
    # both ranks hold the same all-reduced gradients and the same (global) loss
    for k, g in res[0]["grads"].items():
        assert torch.equal(g, res[1]["grads"][k]), k
    assert res[0]["loss"] == res[1]["loss"]
    # oracle: replicas separately through the encoder, one loss over everything
    enc = canonical_state_dict({k: v for k, v in sd.items() if not k.startswith("projector.")})
    p, _ = TR._split(enc)
    proj = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if k.startswith("projector.")}
    import torch.nn.functional as F

    def model_fn(xx, stats):
        f = TR.encoder_train_forward(xx, p, stats)
        return F.linear(F.relu(F.linear(f, proj["projector.0.weight"], proj["projector.0.bias"])), proj["projector.2.weight"],
                        proj["projector.2.bias"])

    zi, zj = [], []
    for r in range(2):
        _, st = TR._split(enc)
        zi.append(model_fn(x[r, 0], st))
        zj.append(model_fn(x[r, 1], st))
    loss = nt_xent_loss_ref(torch.cat(zi), torch.cat(zj), 0.5)
    loss.backward()
    assert abs(res[0]["loss"] - float(loss)) <= 1e-5 * abs(float(loss)) + 1e-6
    ref = {"encoder." + k: v.grad for k, v in p.items()}
    ref.update({k: v.grad for k, v in proj.items()})
    # without handing over activation patterns the deep layers can differ by percents (see the module docstring):
    # the check here is the collective semantics -- loss, and the gradients nearest to the loss
    for k in ("projector.2.weight", "projector.2.bias", "projector.0.weight", "projector.0.bias"):
        assert rel(res[0]["grads"][k], ref[k]) <= 5e-3, (k, rel(res[0]["grads"][k], ref[k]))


def test_fp32_steps_run_twice_give_the_same_bits():
    """The fp32 steps reduce without atomics too (BN statistics and their backward sums through per-workgroup partials added
    in a fixed order, the weight gradients' split-K slices likewise, the cross-entropy sums per wave): the same SimCLR step and
    the same classifier step from the same state give the same parameters bit for bit, the loss included."""
    from ss25_hierarchical_multiscale_image_classification_amd.resnet import ResNet18Classifier

    torch.manual_seed(23)
    sd = {k: v.clone() for k, v in SimCLRModel().state_dict().items()}
    x_i, x_j = torch.randn(12, 3, 224, 224).cuda(), torch.randn(12, 3, 224, 224).cuda()
    outs = []
    for _ in range(2):
        tr = TN.NativeSimCLRTrainer(sd, device="cuda", lr=1e-3, precision="fp32")
        l1, l2 = float(tr.step(x_i, x_j)), float(tr.step(x_j, x_i))
        outs.append((l1, l2, tr.state_dict()))
    assert outs[0][:2] == outs[1][:2]
    for k, v in outs[0][2].items():
        assert torch.equal(v, outs[1][2][k]), k
    csd = {k: v.clone() for k, v in ResNet18Classifier().state_dict().items()}
    x, y = torch.randn(300, 3, 224, 224).cuda(), torch.randint(0, 2, (300,))
    outs = []
    for _ in range(2):
        tr = TN.NativeClassifierTrainer(csd, device="cuda", lr=1e-4, class_weights=torch.tensor([1.0, 2.5]), precision="fp32")
        loss, logits = tr.step(x, y)
        outs.append((float(loss), logits.cpu(), tr.state_dict()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
    for k, v in outs[0][2].items():
        assert torch.equal(v, outs[1][2][k]), k
