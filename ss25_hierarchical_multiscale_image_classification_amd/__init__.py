"""MI355X-native implementation of HiPAC's hierarchical multiscale patch-inference hot
path: sliding-window extraction over WSI pyramid levels, Pillow-exact resize/normalise
and batched ResNet18 scoring, as hand-written HIP kernels behind a C ABI
(include/hipac.h), with a Python host layer that mirrors the reference's interface.

Import is cheap and GPU-free; the shared library is loaded on first use
(``capi.load_library()``) and its absence is an error, never a silent CPU fallback.
"""
__all__ = ["capi", "synth", "weights", "resnet", "simclr", "patch_dataset", "simclr_dataset", "extract",
           "features", "dist", "main", "train", "transforms", "build"]
__version__ = "0.1.0"
