"""Helper of tests/test_gpu_dist.py::test_two_rank_classifier_step: one rank of a 2-rank native classifier step on the
box's single GPU (gloo).  Writes the rank's loss and gradients after the all-reduce."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ss25_hierarchical_multiscale_image_classification_amd import dist as hdist, train_native as TN  # noqa: E402

out_dir = sys.argv[1]
rank, world, _ = hdist.init_from_env("gloo")
sd = torch.load(os.path.join(out_dir, "sd.pt"), weights_only=True)
d = torch.load(os.path.join(out_dir, "xy.pt"), weights_only=True)  # x [world, n, 3, 224, 224], y [world, n], w [2]
tr = TN.NativeClassifierTrainer(sd, device="cuda:0", lr=1e-4, class_weights=d["w"], precision="fp32")
loss, logits = tr.forward_backward(d["x"][rank].cuda().contiguous(), d["y"][rank])
torch.cuda.synchronize()
torch.save({"loss": float(loss), "logits": logits.cpu(), "grads": tr.grad_dict()}, os.path.join(out_dir, f"rank{rank}.pt"))
torch.distributed.destroy_process_group()
