"""Developer tool (GPU box): phase breakdown of block_c64_kernel (layer1 BasicBlock in one kernel) from a
-DHIPAC_HALO_STAMPS build.  usage: HIPAC_LIB_NAME=lib_stamps.so python tools/blockstamps.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ss25_hierarchical_multiscale_image_classification_amd import capi, synth  # noqa: E402

dev = torch.device("cuda:0")
net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision="bf16")
lib = capi.load_library()
fh, fb = lib.hipac_debug_halo_stamps, lib.hipac_debug_blk_stamps
fh.argtypes = fb.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
h8, b4 = (ctypes.c_ulonglong * 8)(), (ctypes.c_ulonglong * 4)()
u8 = synth.synth_patches_u8(512, seed=1, device=dev)
net.forward(u8)
torch.cuda.synchronize()
i = [n for n, _ in bench.OPS].index("l1b0")
fh(h8, 1), fb(b4, 1)
net.run_ops(u8, i, i)
torch.cuda.synchronize()
fh(h8, 1), fb(b4, 1)
na, nb = max(1, b4[2]), max(1, b4[3])  # wave-steps incl. the idle one per workgroup
print(f"A (conv1) per wave-step: x DMA issue {h8[0]/na:.0f}  MFMA loop (72) {h8[1]/na:.0f}  epilogue {h8[2]/na:.0f}  "
      f"DMA wait {h8[3]/na:.0f}  barrier {b4[0]/na:.0f}")
print(f"B (conv2) per wave-step: deferred epilogue {h8[4]/nb:.0f}  setup + shortcut DMA issue {h8[7]/nb:.0f}  MFMA pair loop {h8[5]/nb:.0f}  "
      f"shortcut wait + MFMAs {h8[6]/nb:.0f}  barrier {b4[1]/nb:.0f}")
