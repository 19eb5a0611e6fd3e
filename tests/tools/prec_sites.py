import sys, torch, numpy as np, torch.nn.functional as F
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import resnet18_ref as R, transform_ref as T
from ss25_hierarchical_multiscale_image_classification_amd import synth
torch.set_num_threads(8)
H = torch.float16
def fold(sd, conv, bn):
    w = sd[conv + '.weight'].double(); g = sd[bn + '.weight'].double(); b = sd[bn + '.bias'].double()
    m = sd[bn + '.running_mean'].double(); v = sd[bn + '.running_var'].double()
    s = g / torch.sqrt(v + 1e-5)
    return (w * s[:, None, None, None]).float(), (b - m * s).float()
def rn(w): return w.to(H).float()
def diffuse(w):
    # error-diffusion rounding along K ordered (cin outer, kh, kw inner): w [co, ci, kh, kw]
    co = w.shape[0]
    flat = w.reshape(co, -1).double()
    out = torch.empty_like(flat)
    err = torch.zeros(co, dtype=torch.float64)
    for k in range(flat.shape[1]):
        t = flat[:, k] + err
        r = t.to(torch.float32).to(H).double()
        # keep within 1 ulp of original: r is RN of (w+err); fine
        out[:, k] = r
        err = t - r
    return out.float().reshape(w.shape)
def sim(x, sd, wq, aq):
    """wq(name, w) -> w'; aq(name, x) -> x'"""
    sd = R.canonical_state_dict(sd)
    x = aq('input', x)
    w, b = fold(sd, 'conv1', 'bn1')
    y = aq('stem', F.relu(F.conv2d(x, wq('conv1', w), b, stride=2, padding=3)))
    y = F.max_pool2d(y, 3, 2, 1)
    for name, _, stride in R.STAGES:
        for blk in (0, 1):
            p = f'{name}.{blk}'
            st = stride if blk == 0 else 1
            w1, b1 = fold(sd, p + '.conv1', p + '.bn1'); w2, b2 = fold(sd, p + '.conv2', p + '.bn2')
            t = aq(p + '.c1', F.relu(F.conv2d(y, wq(p + '.conv1', w1), b1, stride=st, padding=1)))
            if (p + '.downsample.0.weight') in sd:
                wd, bd = fold(sd, p + '.downsample.0', p + '.downsample.1')
                ident = aq(p + '.ds', F.conv2d(y, wq(p + '.ds', wd), bd, stride=st))
            else:
                ident = y
            o = F.relu(F.conv2d(t, wq(p + '.conv2', w2), b2, padding=1) + ident)
            last = (name == 'layer4' and blk == 1)
            y = o if last else aq(p + '.out', o)
    f = torch.flatten(F.adaptive_avg_pool2d(y, 1), 1)
    return f, F.linear(f, sd['fc.weight'], sd['fc.bias'])
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = sys.argv[2] if len(sys.argv) > 2 else 'diffuse'
seed = 0
sd = synth.seeded_resnet18_state_dict(seed, num_classes=2)
u8 = synth.synth_patches_u8(N, seed=1)
lut = torch.from_numpy(T.normalize_lut())
x = torch.stack([lut[c][u8[..., c].long()] for c in range(3)], dim=1)
ident_w = lambda n, w: w
ident_a = lambda n, a: a
with torch.no_grad():
    rf, rl = R.resnet18_forward(x, sd)
    if mode == 'diffuse':
        cache = {}
        def wd(n, w):
            if n not in cache: cache[n] = diffuse(w)
            return cache[n]
        for nm, wq, aq in [('w RN only', lambda n, w: rn(w), ident_a), ('w diffuse only', wd, ident_a),
                           ('w diffuse + a RN', wd, lambda n, a: rn(a)), ('w RN + a RN', lambda n, w: rn(w), lambda n, a: rn(a)),
                           # the two-product variant (weights as (hi, lo) pairs = exact here, activations single fp16)
                           ('w exact + a RN', ident_w, lambda n, a: rn(a))]:
            f, l = sim(x, sd, wq, aq)
            print(f'{nm:20s} feats {rel(f, rf):.2e} logits {rel(l, rl):.2e}')
    else:
        names_w = ['conv1'] + [f'{n}.{b}.{c}' for n, _, _ in R.STAGES for b in (0, 1) for c in ('conv1', 'conv2')] + [f'layer{i}.0.ds' for i in (2, 3, 4)]
        for nw in names_w:
            f, l = sim(x, sd, lambda n, w: rn(w) if n == nw else w, ident_a)
            print(f'W {nw:20s} feats {rel(f, rf):.2e} logits {rel(l, rl):.2e}')
        names_a = ['input', 'stem'] + [f'{n}.{b}.{c}' for n, _, _ in R.STAGES for b in (0, 1) for c in ('c1', 'out')] + [f'layer{i}.0.ds' for i in (2, 3, 4)]
        for na in names_a:
            f, l = sim(x, sd, ident_w, lambda n, a: rn(a) if n == na else a)
            print(f'A {na:20s} feats {rel(f, rf):.2e} logits {rel(l, rl):.2e}')
