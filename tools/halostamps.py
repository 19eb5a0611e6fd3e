"""Developer tool (GPU box): phase breakdown of conv3x3_halo_kernel from a -DHIPAC_HALO_STAMPS build.
usage: HIPAC_LIB_NAME=libhipac_stamps.so python tools/halostamps.py"""
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
fn = lib.hipac_debug_halo_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
u8 = synth.synth_patches_u8(4096, seed=1, device=dev)
net.forward(u8)
torch.cuda.synchronize()
names = [n for n, _ in bench.OPS]
for i, name in enumerate(names):
    if not (name[0] == "l" and name[1] in "234"):
        continue
    fn(buf, 1)
    net.run_ops(u8, i, i)
    torch.cuda.synchronize()
    fn(buf, 1)
    n = max(1, buf[3])
    tot = max(1, buf[0] + buf[1] + buf[2])
    print(f"{name}: workgroups {buf[3]}  per workgroup (s_memtime = shader cycles): prologue {buf[0]/n:.0f}  "
          f"K loop {buf[1]/n:.0f}  epilogue {buf[2]/n:.0f}   shares {buf[0]/tot:.2f} / {buf[1]/tot:.2f} / {buf[2]/tot:.2f}"
          + (f"   epilogue = barrier {buf[4]/n:.0f} + next band issue {buf[5]/n:.0f} + staged groups {buf[6]/n:.0f}" if buf[6] else "")
          + (f"   prologue = setup {buf[7]/n:.0f} + wait for DMA / other waves {(buf[0]-buf[7])/n:.0f}" if buf[7] else ""))

for i, name in enumerate(names):
    if not name.startswith("l1"):
        continue
    fn(buf, 1)
    net.run_ops(u8, i, i)
    torch.cuda.synchronize()
    fn(buf, 1)
    n = max(1, buf[3])
    tot = max(1, buf[4] + buf[5] + buf[6] + buf[7])
    print(f"{name}: wave-units {buf[3]}  cycles per unit and wave: vmcnt drain {buf[4]/n:.0f}  barrier {buf[5]/n:.0f}  "
          f"DMA issue + MFMA loop {buf[6]/n:.0f}  epilogue {buf[7]/n:.0f}   total {tot/n:.0f}")
