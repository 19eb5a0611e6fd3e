"""Helper of tests/test_gpu_train.py::test_two_rank_simclr_step: one rank of a 2-rank native SimCLR step on the box's
single GPU (gloo process group; collectives staged through the host).  Writes the rank's gradients after the all-reduce."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ss25_hierarchical_multiscale_image_classification_amd import dist as hdist, train_native as TN  # noqa: E402

out_dir = sys.argv[1]
rank, world, _ = hdist.init_from_env("gloo")
sd = torch.load(os.path.join(out_dir, "sd.pt"), weights_only=True)
x = torch.load(os.path.join(out_dir, "x.pt"), weights_only=True)  # [world, 2, n, 3, 224, 224]
tr = TN.NativeSimCLRTrainer(sd, device="cuda:0")
loss = tr.forward_backward(x[rank, 0].cuda().contiguous(), x[rank, 1].cuda().contiguous())
torch.cuda.synchronize()
torch.save({"loss": float(loss), "grads": tr.grad_dict(), "stats": {k: v for k, v in tr.state_dict().items() if "running" in k}},
           os.path.join(out_dir, f"rank{rank}.pt"))
torch.distributed.destroy_process_group()
