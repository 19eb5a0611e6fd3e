"""Developer tool (GPU box): phase breakdown of stem_pool_strip2_kernel from a -DHIPAC_HALO_STAMPS build.
usage: HIPAC_LIB_NAME=lib_stamps.so python tools/stripstamps.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ss25_hierarchical_multiscale_image_classification_amd import capi, synth  # noqa: E402

dev = torch.device("cuda:0")
net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision="bf16")
lib = capi.load_library()
fn = lib.hipac_debug_halo_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
u8 = synth.synth_patches_u8(512, seed=1, device=dev)
net.forward(u8)
for _ in range(5):
    net.run_ops(u8, 0, 0)
torch.cuda.synchronize()
fn(buf, 1)
net.run_ops(u8, 0, 0)
torch.cuda.synchronize()
fn(buf, 1)
n = max(1, buf[3])
print(f"wave-steps {buf[3]}; per wave and step, s_memtime ticks: H1 (DMA issue + MFMA loop) {buf[0]/n:.0f}  "
      f"epilogue {buf[1]/n:.0f}  raw-row wait {buf[2]/n:.0f}  conversion {buf[4]/n:.0f}  barrier waits (2 per step) {buf[5]/n:.0f}")
