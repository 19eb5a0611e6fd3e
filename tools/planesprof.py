"""Developer tool (GPU box): a few launches of hipac_level_build_planes for rocprofv3 (--kernel-trace / --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ss25_hierarchical_multiscale_image_classification_amd import capi
side = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1792
img = torch.randint(0, 256, (side, (side + 15) // 16 * 16, 3), dtype=torch.uint8, device="cuda")
for _ in range(4):
    capi.LevelPlanes(img, P, width=side)
torch.cuda.synchronize()
