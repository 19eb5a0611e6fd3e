"""Helper of tests/test_gpu_dist.py::test_single_rank_rccl_drives_every_exchange: ONE rank, backend "nccl" (= RCCL) on
cuda:0, HIPAC_DIST_FORCE=1 so that no exchange of the product path returns early -- the slide-sharded scan's count /
padded all-gathers, the SimCLR step's all-gather of z and all-reduce of the gradients, the rank-0 broadcast, the loss-scale
flag's MAX all-reduce and the loops' scalar all-reduce all go through RCCL on device tensors (the one-GPU box has no second
device for a second rank).  Writes what a process without a process group must reproduce bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["HIPAC_DIST_FORCE"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from ss25_hierarchical_multiscale_image_classification_amd import capi, dist as hdist, extract, synth, train_native as TN  # noqa: E402

out_dir, port = sys.argv[1], sys.argv[2]
os.environ["MASTER_PORT"] = port
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert hdist.active() and dist.get_backend() == "nccl"

specs = torch.load(os.path.join(out_dir, "specs.pt"), weights_only=True).tolist()
net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision="fp16x3")


def score(i):
    w, h, seed = specs[i]
    slide = extract.DeviceSlide.synthetic(w, h, seed=seed, name=f"slide_{i}", with_polygons=True)
    f, l, _, meta = extract.score_slide(slide, net, levels=(1, 2, 3), fwd_batch=64)
    return f, l, meta


sf, sl, sm = hdist.score_sharded(len(specs), score)
eq = hdist.all_gather_equal(sf[:7].contiguous())
scal = hdist.all_reduce_sum_scalars([1.5, 2.0, -3.25], device=torch.device("cuda", 0))

sd = torch.load(os.path.join(out_dir, "sd.pt"), weights_only=True)
x = torch.load(os.path.join(out_dir, "x.pt"), weights_only=True)  # [2, n, 3, 224, 224]
res = {}
for prec in ("fp32", "fp16"):
    tr = TN.NativeSimCLRTrainer(sd, device="cuda:0", precision=prec)
    tr.sync_from_rank0()
    if prec == "fp16":  # deterministic: the whole step (incl. the loss-scale flag's MAX all-reduce) must be reproducible
        loss = tr.step(x[0].cuda().contiguous(), x[1].cuda().contiguous())
        torch.cuda.synchronize()
        res[prec] = {"loss": float(loss), "params": {k: v.cpu() for k, v in tr.state_dict().items()}}
    else:  # fp32: atomics in its reductions; compare the all-reduced gradients (Adam's first step is sign(g): noise-sensitive)
        loss = tr.forward_backward(x[0].cuda().contiguous(), x[1].cuda().contiguous())
        torch.cuda.synchronize()
        res[prec] = {"loss": float(loss), "grads": {k: v.cpu() for k, v in tr.grad_dict().items()}}
torch.save({"slides": (sf.cpu(), sl.cpu(), sm.cpu()), "eq": eq.cpu(), "scal": scal, "simclr": res},
           os.path.join(out_dir, "rccl.pt"))
dist.destroy_process_group()
