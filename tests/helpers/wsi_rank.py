"""Helper of tests/test_gpu_dist.py::test_two_rank_sharded_slides_equal_single_process: one rank of a 2-rank run of
BASELINE configs[3] on the box's single GPU (gloo process group; the exchange is staged through the host): the slides this
rank owns (slide i -> rank i mod 2) are scored with extract.score_slide and the rows of both ranks are gathered."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ss25_hierarchical_multiscale_image_classification_amd import capi, dist as hdist, extract, synth  # noqa: E402

out_dir, precision = sys.argv[1], sys.argv[2]
specs = torch.load(os.path.join(out_dir, "specs.pt"), weights_only=True)  # [[W, H, seed], ...]
rank, world, _ = hdist.init_from_env("gloo")
torch.cuda.set_device(0)
net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision=precision)


def score(i):
    w, h, seed = specs[i]
    slide = extract.DeviceSlide.synthetic(int(w), int(h), seed=int(seed), name=f"slide_{i}", with_polygons=True)
    f, l, _, meta = extract.score_slide(slide, net, levels=(1, 2, 3), fwd_batch=64)
    return f, l, meta


# the raw exchange (rank-major order) ...
mine = [score(i) for i in hdist.shard_units(len(specs), rank, world)]
f = torch.cat([m[0] for m in mine]) if mine else torch.zeros((0, 512), device="cuda")
l = torch.cat([m[1] for m in mine]) if mine else torch.zeros((0, 2), device="cuda")
meta = torch.cat([m[2] for m in mine]) if mine else torch.zeros((0, 4), dtype=torch.int32, device="cuda")
gf, gl, gm = hdist.gather_results(f, l, meta)
# ... and the product's entry point (rows back in slide order)
sf, sl, sm = hdist.score_sharded(len(specs), score, rank, world)
torch.cuda.synchronize()
torch.save({"rank_major": (gf.cpu(), gl.cpu(), gm.cpu()), "slide_order": (sf.cpu(), sl.cpu(), sm.cpu())},
           os.path.join(out_dir, f"rank{rank}.pt"))
torch.distributed.destroy_process_group()
