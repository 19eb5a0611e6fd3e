"""CPU emulation of a cheaper parity mode: every value is the fp16 pair (hi, lo) of the fp16x3 mode,
the product hi*hi stays on the fp16 MFMA, and the two cross terms hi*lo + lo*hi run on the block-scaled MX MFMA
(v_mfma_scale_f32_32x32x64_f8f6f4: one E8M0 scale per 32 K-elements) with e4m3 (2x the fp16 rate) or e2m3 (4x) operands.
Prints the error of features / logits against the fp32 oracle for: fp16x3 (exact cross terms), MXFP8 and MXFP6 cross
terms, and the single fp16 product.  usage: python tests/tools/prec_mx.py [n_patches]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import resnet18_ref as R, transform_ref as T  # noqa: E402
from ss25_hierarchical_multiscale_image_classification_amd import synth  # noqa: E402

torch.set_num_threads(8)
H = torch.float16


def fold(sd, conv, bn):
    w = sd[conv + '.weight'].double(); g = sd[bn + '.weight'].double(); b = sd[bn + '.bias'].double()
    m = sd[bn + '.running_mean'].double(); v = sd[bn + '.running_var'].double()
    s = g / torch.sqrt(v + 1e-5)
    return (w * s[:, None, None, None]).float(), (b - m * s).float()


def pair(v):
    hi = v.to(H).float()
    return hi, (v - hi).to(H).float()


def mx_quant(v, fmt):
    """Block-scaled quantisation along dim 1 (channels) in blocks of 32: shared power-of-two scale, e4m3 / e2m3 elements."""
    if fmt is None:
        return v
    n, c = v.shape[0], v.shape[1]
    pad = (-c) % 32
    x = F.pad(v, (0, 0) * (v.dim() - 2) + (0, pad)) if pad else v
    shp = x.shape
    x = x.reshape(n, (c + pad) // 32, 32, *shp[2:])
    amax = x.abs().amax(dim=2, keepdim=True).clamp_min(1e-38)
    emax = 8 if fmt == 'e4m3' else 2
    scale = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    y = x / scale
    if fmt == 'e4m3':
        q = y.clamp(-448, 448).to(torch.float8_e4m3fn).float()
    else:  # e2m3: 3 mantissa bits, exponents 0..2 (+ subnormals of exponent 0), max 7.5
        a = y.abs().clamp(max=7.5)
        e = torch.floor(torch.log2(a.clamp_min(1e-30))).clamp(0, 2)
        step = torch.exp2(e - 3)
        q = torch.sign(y) * torch.round(a / step) * step
    out = (q * scale).reshape(shp)
    return out[:, :c] if pad else out


def const_quant(v, shift):
    """e4m3 with ONE constant power-of-two scale for the whole tensor (no block scales): fp8(v * 2^shift) / 2^shift."""
    return (v * 2.0 ** shift).clamp(-448, 448).to(torch.float8_e4m3fn).float() / 2.0 ** shift


def conv3(x, w, b, fmt, mode, **kw):
    """One convolution in the emulated arithmetic.  mode: 'x3' three products, 'x1' single fp16 product."""
    xh, xl = pair(x)
    wh, wl = pair(w)
    y = F.conv2d(xh.double(), wh.double(), None, **kw)
    if mode == 'x3' and fmt == 'e4m3const':
        # hi parts as they are, lo parts times 2^11 (|lo| <= 2^-11 |hi|), every tensor with the same constants
        y = y + F.conv2d(const_quant(xl, 11).double(), const_quant(wh, 4).double(), None, **kw) \
              + F.conv2d(const_quant(xh, 0).double(), const_quant(wl, 15).double(), None, **kw)
    elif mode == 'x3':
        wq = lambda t: mx_quant(t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]), fmt).reshape(t.shape[0], t.shape[2], t.shape[3], t.shape[1]).permute(0, 3, 1, 2)
        aq = lambda t: mx_quant(t, fmt)
        y = y + F.conv2d(aq(xl).double(), wq(wh).double(), None, **kw) + F.conv2d(aq(xh).double(), wq(wl).double(), None, **kw)
    return (y + b.double()[None, :, None, None]).float()


def sim(x, sd, fmt, mode, stem_single=False):
    sd = R.canonical_state_dict(sd)
    w, b = fold(sd, 'conv1', 'bn1')
    if stem_single:
        # the stem as ONE product: input exact (bytes), weights rounded to fp16 (no lo halves)
        y = F.relu((F.conv2d(x.double(), pair(w)[0].double(), None, stride=2, padding=3) + b.double()[None, :, None, None]).float())
    else:
        # the stem keeps the fp16x3 kernel's arithmetic (bytes exact, weights as pairs: two exact products)
        y = F.relu(conv3(x, w, b, None, 'x3' if mode == 'x3' else 'x1', stride=2, padding=3))
    y = F.max_pool2d(y, 3, 2, 1)
    for name, _, stride in R.STAGES:
        for blk in (0, 1):
            p = f'{name}.{blk}'
            st = stride if blk == 0 else 1
            w1, b1 = fold(sd, p + '.conv1', p + '.bn1'); w2, b2 = fold(sd, p + '.conv2', p + '.bn2')
            t = F.relu(conv3(y, w1, b1, fmt, mode, stride=st, padding=1))
            if (p + '.downsample.0.weight') in sd:
                wd, bd = fold(sd, p + '.downsample.0', p + '.downsample.1')
                ident = conv3(y, wd, bd, fmt, mode, stride=st)
            else:
                ident = y
            y = F.relu(conv3(t, w2, b2, fmt, mode, padding=1) + ident)
    f = torch.flatten(F.adaptive_avg_pool2d(y, 1), 1)
    return f, F.linear(f, sd['fc.weight'], sd['fc.bias'])


rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sd = synth.seeded_resnet18_state_dict(0, num_classes=2)
u8 = synth.synth_patches_u8(N, seed=1)
lut = torch.from_numpy(T.normalize_lut())
x = torch.stack([lut[c][u8[..., c].long()] for c in range(3)], dim=1)
with torch.no_grad():
    rf, rl = R.resnet18_forward(x, sd)
    for nm, fmt, mode in [('cross terms e4m3, constant scales', 'e4m3const', 'x3'),
                          ('fp16x3 (exact cross terms)', None, 'x3'), ('cross terms MXFP8 e4m3', 'e4m3', 'x3'),
                          ('cross terms MXFP6 e2m3', 'e2m3', 'x3'), ('single fp16 product', None, 'x1')]:
        f, l = sim(x, sd, fmt, mode)
        if fmt == 'e4m3const':
            f1, l1 = sim(x, sd, fmt, mode, stem_single=True)
            print(f"{'... with a one-product stem':30s} feats {rel(f1, rf):.2e} logits {rel(l1, rl):.2e} labels equal {bool((l1.argmax(1) == rl.argmax(1)).all())}", flush=True)
        print(f'{nm:30s} feats {rel(f, rf):.2e} logits {rel(l, rl):.2e} labels equal {bool((l.argmax(1) == rl.argmax(1)).all())}', flush=True)
