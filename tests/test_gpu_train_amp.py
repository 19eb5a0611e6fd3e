"""Mixed-precision native training step (csrc/train_amp.hip: fp16 operands and maps on the fp16 MFMA, fp32
accumulation, fp32 master weights, GradScaler) -- the arithmetic of the reference's fine-tune loops under
torch.cuda.amp.autocast() (src/main.py:499-508) -- against the fp32 autograd oracle.

Bounds (norm-relative max|a-b| / max|b| per tensor, measured values printed): the forward's fp16 maps differ from the
fp32 oracle by the 11-bit roundings of ~20 stored maps, which the backward then carries through 20 layers: logits
5e-3 (measured 1.7e-3), loss 5e-3 (5e-5), gradients 3e-2 (measured 1.6e-3 at the stem rising to 7.6e-3 at layer4's convolutions;
worst tensor 8.9e-3 / 1.3e-2, a layer4 batch-norm weight).  The oracle gets the
step's own ReLU / max-pool patterns, as in test_gpu_train.py.  Determinism is exact: a step run twice gives the same bits."""
import numpy as np
import pytest
import torch

from oracle import train_ref as TR
from ss25_hierarchical_multiscale_image_classification_amd import capi, train_native as TN
from ss25_hierarchical_multiscale_image_classification_amd.resnet import ResNet18Classifier
from ss25_hierarchical_multiscale_image_classification_amd.simclr import SimCLRModel
from test_gpu_train import _randomise_bn, patterns, rel

pytestmark = pytest.mark.gpu
AMP_GRAD_TOL, AMP_OUT_TOL = 3e-2, 5e-3


def test_amp_classifier_step_against_oracle_autograd():
    torch.manual_seed(13)
    model = ResNet18Classifier()
    _randomise_bn(model, 6)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x, y = torch.randn(8, 3, 224, 224), torch.tensor([0, 1, 1, 0, 0, 0, 1, 0])
    w = torch.tensor([1.0, 2.5])
    tr = TN.NativeClassifierTrainer(sd, device="cuda", lr=1e-4, class_weights=w, precision="fp16",
                                    scaler=TN.GradScaler(init_scale=1024.0))
    loss, logits = tr.forward_backward(x.cuda(), y.cuda())
    assert tr.scaler.unscale_and_check(tr.encoder.opt, tr.head)  # finite; the loss scale is divided out
    loss_ref, logits_ref, grads_ref, stats_ref = TR.classifier_step_ref(sd, x, y, w, masks=patterns(tr.encoder, 0))
    print(f"amp classifier step: loss {float(loss):.6f} (oracle {float(loss_ref):.6f}), logits {rel(logits, logits_ref):.2e}")
    assert abs(float(loss) - float(loss_ref)) <= AMP_OUT_TOL * max(1.0, abs(float(loss_ref)))
    assert rel(logits, logits_ref) <= AMP_OUT_TOL
    got = tr.grad_dict()
    errs = {name: rel(got[name], g) for name, g in grads_ref.items()}
    worst = max(errs.items(), key=lambda t: t[1])
    print({k: f"{v:.1e}" for k, v in errs.items() if k.endswith("weight") and ("conv" in k or "downsample.0" in k or k.startswith("fc"))})
    print(f"worst gradient {worst[0]} {worst[1]:.2e}")
    for name, g in grads_ref.items():
        assert got[name].shape == g.shape and errs[name] <= AMP_GRAD_TOL, (name, errs[name])
    new = tr.state_dict()
    for k, v in stats_ref.items():  # running statistics from the fp16 maps
        assert rel(new["model." + k], v) <= 2e-3, k


def test_amp_simclr_step_against_oracle_and_twice_bit_identical():
    torch.manual_seed(11)
    model = SimCLRModel()
    _randomise_bn(model, 4)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x_i, x_j = torch.randn(6, 3, 224, 224), torch.randn(6, 3, 224, 224)
    runs = []
    for rep in range(2):
        tr = TN.NativeSimCLRTrainer(sd, device="cuda", lr=1e-3, precision="fp16", scaler=TN.GradScaler(init_scale=256.0))
        loss = tr.forward_backward(x_i.cuda(), x_j.cuda())
        assert tr.scaler.unscale_and_check(tr.encoder.opt, tr.head)
        torch.cuda.synchronize()
        runs.append((float(loss), tr.grad_dict(), tr.state_dict(), tr))
    # determinism: no atomics anywhere in the mixed-precision step -- the same bits, gradient by gradient
    assert runs[0][0] == runs[1][0]
    for k, g in runs[0][1].items():
        assert torch.equal(g, runs[1][1][k]), k
    for k, v in runs[0][2].items():
        assert torch.equal(v, runs[1][2][k]), k
    tr = runs[1][3]
    m_i, m_j = patterns(tr.encoder, 0, tr.last_hidden[0]), patterns(tr.encoder, 1, tr.last_hidden[1])
    loss_ref, grads_ref, _ = TR.simclr_step_ref(sd, x_i, x_j, masks_i=m_i, masks_j=m_j)
    assert abs(runs[1][0] - float(loss_ref)) <= AMP_OUT_TOL * abs(float(loss_ref))
    errs = {name: rel(runs[1][1][name], g) for name, g in grads_ref.items()}
    worst = max(errs.items(), key=lambda t: t[1])
    print(f"amp simclr step: loss {runs[1][0]:.6f} (oracle {float(loss_ref):.6f}); worst gradient {worst[0]} {worst[1]:.2e}")
    for name in grads_ref:
        assert errs[name] <= AMP_GRAD_TOL, (name, errs[name])


def test_amp_odd_batches_and_stem_weight_gradient():
    """Batches that are not multiples of the kernels' tiles (the weight-gradient kernel's 32-pixel sub-chunks and split-K
    slices end ragged), checked on the tensors whose kernels differ most: stem, a 64-channel, a stride-2, a 1x1, a 512-channel."""
    torch.manual_seed(17)
    model = ResNet18Classifier()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    for n in (1, 3, 5):
        x, y = torch.randn(n, 3, 224, 224), torch.randint(0, 2, (n,))
        tr = TN.NativeClassifierTrainer(sd, device="cuda", precision="fp16", scaler=TN.GradScaler(init_scale=128.0))
        loss, _ = tr.forward_backward(x.cuda(), y.cuda())
        assert tr.scaler.unscale_and_check(tr.encoder.opt, tr.head)
        if n == 1:
            continue  # batch statistics of a single image at 7x7 are too thin for a tight comparison; it must only run
        loss_ref, _, grads_ref, _ = TR.classifier_step_ref(sd, x, y, None, masks=patterns(tr.encoder, 0))
        got = tr.grad_dict()
        for name in ("conv1.weight", "layer1.0.conv1.weight", "layer2.0.conv1.weight", "layer3.0.downsample.0.weight",
                     "layer4.1.conv2.weight", "bn1.weight", "fc.weight"):
            assert rel(got[name], grads_ref[name]) <= AMP_GRAD_TOL, (n, name, rel(got[name], grads_ref[name]))


def test_grad_scaler_skips_an_overflowing_step_and_backs_off():
    """GradScaler semantics (src/main.py:506-508): a loss scale that overflows the fp16 gradient maps -> inf in the
    gradients -> the optimizer step is skipped (parameters unchanged) and the scale halves; with a sane scale the step
    is taken and, after `growth_interval` good steps, the scale doubles."""
    torch.manual_seed(3)
    model = ResNet18Classifier()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x, y = torch.randn(4, 3, 224, 224).cuda(), torch.tensor([0, 1, 1, 0])
    tr = TN.NativeClassifierTrainer(sd, device="cuda", precision="fp16", scaler=TN.GradScaler(init_scale=2.0 ** 40, growth_interval=2))
    before = tr.encoder.opt.params.clone()
    tr.step(x, y)
    assert tr.scaler.skipped == 1 and tr.scaler.get_scale() == 2.0 ** 39 and tr.encoder.opt.t == 0
    assert torch.equal(tr.encoder.opt.params, before)
    tr.scaler.scale = 1024.0
    tr.step(x, y), tr.step(x, y)
    assert tr.encoder.opt.t == 2 and tr.scaler.get_scale() == 2048.0 and not torch.equal(tr.encoder.opt.params, before)
    assert bool(torch.isfinite(tr.encoder.opt.params).all())


def test_amp_loss_and_head_gradient_agree_with_fp32_native_step():
    """The two native arithmetics on the same step with NO shared activation patterns: loss within 5e-3 and the fc
    gradient within 5e-2 (deeper gradients depend on which near-zero ReLU units each forward switched: test_gpu_train.py)."""
    torch.manual_seed(21)
    model = ResNet18Classifier()
    _randomise_bn(model, 9)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x, y = torch.randn(8, 3, 224, 224).cuda(), torch.tensor([1, 0, 1, 1, 0, 0, 1, 0])
    a = TN.NativeClassifierTrainer(sd, device="cuda", precision="fp32")
    b = TN.NativeClassifierTrainer(sd, device="cuda", precision="fp16", scaler=TN.GradScaler(init_scale=512.0))
    la, _ = a.forward_backward(x, y)
    lb, _ = b.forward_backward(x, y)
    assert b.scaler.unscale_and_check(b.encoder.opt, b.head)
    assert abs(float(la) - float(lb)) <= 5e-3 * max(1.0, abs(float(la)))
    ga, gb = a.grad_dict(), b.grad_dict()
    for k in ("fc.weight", "fc.bias"):
        assert rel(gb[k], ga[k]) <= 5e-2, (k, rel(gb[k], ga[k]))
