"""ctypes binding of libhipac_hip.so (the C ABI in include/hipac.h).

PyTorch appears here only as the owner of device memory and streams: tensors are
handed to the library as raw pointers, and every launch goes on torch's current
HIP stream.  There is no CPU fallback: if the library is missing or a call
fails, a ``HipacError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np
import torch

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / os.environ.get("HIPAC_LIB_NAME", "libhipac_hip.so")  # override: A/B builds of the same ABI

PREC_BF16, PREC_FP16, PREC_FP32, PREC_FP16X3, PREC_FP16Q8 = 0, 1, 2, 3, 4
IN_NCHW_F32, IN_NHWC4_PAD, IN_U8_HWC = 0, 1, 2
OUT_NCHW_F32, OUT_NHWC4_PAD_BF16, OUT_NHWC4_PAD_FP16, OUT_U8_HWC = 0, 1, 2, 3
PATCH, PAD_H, PAD_W = 224, 230, 232
ABI_VERSION = 8  # include/hipac.h HIPAC_ABI_VERSION this binding was written against

# "fp16x3" is the parity mode (fp16 (hi, lo) pairs, three MFMA products per term: the reference's fp32 results to 1e-3);
# "fp32" the debugging reference: fp32 storage and the exact f32 MFMA (about 1/16 of the bf16 rate)
PRECISIONS = {"bf16": PREC_BF16, "fp16": PREC_FP16, "fp32": PREC_FP32, "fp16x3": PREC_FP16X3, "fp16q8": PREC_FP16Q8}
TORCH_DTYPE = {PREC_BF16: torch.bfloat16, PREC_FP16: torch.float16, PREC_FP32: torch.float32, PREC_FP16X3: torch.float32, PREC_FP16Q8: torch.float32}


class HipacError(RuntimeError):
    pass


class ConvBN(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("conv_w", "bn_gamma", "bn_beta", "bn_mean", "bn_var")]


class JpegLevel(C.Structure):
    """hipac_jpeg_level (include/hipac.h)."""
    _fields_ = [("pixels", C.c_void_p), ("pitch_bytes", C.c_int64), ("W", C.c_int32), ("H", C.c_int32), ("tile_w", C.c_int32),
                ("tile_h", C.c_int32), ("photometric", C.c_int32), ("reserved", C.c_int32), ("jpeg_tables", C.c_void_p),
                ("jpeg_tables_len", C.c_int64)]


class MilParams(C.Structure):  # hipac_mil_params_t
    _fields_ = [(n, C.c_void_p) for n in ("attn_V_w", "attn_V_b", "attn_U_w", "attn_U_b", "fc1_w", "fc1_b", "fc2_w",
                                           "fc2_b")] + [(n, C.c_int32) for n in ("feature_dim", "attn_dim",
                                                                                 "hidden_dim", "num_classes")]


class ResNet18Params(C.Structure):
    _fields_ = [
        ("stem", ConvBN),
        ("block", (ConvBN * 2) * 8),
        ("down", ConvBN * 3),
        ("fc_w", C.c_void_p),
        ("fc_b", C.c_void_p),
        ("num_classes", C.c_int32),
        ("bn_eps", C.c_float),
    ]


# name -> (restype, argtypes); must list every symbol include/hipac.h declares.
SYMBOLS = {
    "hipac_abi_version": (C.c_int, []),
    "hipac_last_error": (C.c_char_p, []),
    "hipac_resnet18_pack": (C.c_int, [C.POINTER(ResNet18Params), C.c_int, C.POINTER(C.c_void_p)]),
    "hipac_weights_free": (None, [C.c_void_p]),
    "hipac_weights_precision": (C.c_int, [C.c_void_p]),
    "hipac_weights_num_classes": (C.c_int, [C.c_void_p]),
    "hipac_resnet18_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "hipac_resnet18_forward": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "hipac_resnet18_run_ops": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hipac_resnet18_tap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hipac_resample_coeffs": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "hipac_tile_preprocess": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
         C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "hipac_level_planes_sizes": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                           C.POINTER(C.c_size_t)]),
    "hipac_level_build_planes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hipac_level_window_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "hipac_level_gather": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_void_p]),
    "hipac_mask_cells": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "hipac_window_labels_cells": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_void_p]),
    "hipac_window_labels": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    ),
    "hipac_patches_normalize": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "hipac_ntxent_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "hipac_ntxent_fwd_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    "hipac_mil_forward": (C.c_int, [C.POINTER(MilParams), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hipac_train_num_convs": (C.c_int, []),
    "hipac_train_conv_desc": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hipac_train_param_floats": (C.c_size_t, []),
    "hipac_train_stat_floats": (C.c_size_t, []),
    "hipac_train_workspace_bytes": (C.c_size_t, [C.c_int]),
    "hipac_train_debug_offset": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "hipac_train_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p,
                                              C.c_void_p, C.c_size_t, C.c_void_p]),
    "hipac_train_encoder_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                               C.c_void_p]),
    "hipac_train_amp_workspace_bytes": (C.c_size_t, [C.c_int]),
    "hipac_train_amp_debug_offset": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "hipac_train_amp_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p,
                                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "hipac_train_amp_encoder_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                                   C.c_void_p]),
    "hipac_grads_unscale_check": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "hipac_jpeg_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "hipac_jpeg_decode_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "hipac_augment_views": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hipac_linear_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p]),
    "hipac_linear_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hipac_cross_entropy_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "hipac_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_int, C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


def load_library(path: Optional[os.PathLike] = None):
    """dlopen the library and bind every symbol.  Raises HipacError if absent."""
    global _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = Path(path) if path else LIB_PATH
        if not p.exists():
            raise HipacError(
                f"{p} not found: build it with `python -m {__package__}.build` "
                "(there is no CPU fallback for the HIP path)"
            )
        lib = C.CDLL(str(p))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.hipac_abi_version() != ABI_VERSION:
            raise HipacError(f"ABI version mismatch: library {lib.hipac_abi_version()}, binding {ABI_VERSION}")
        if path is None:
            _lib = lib
        return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = load_library().hipac_last_error().decode(errors="replace")
        raise HipacError(f"{what} failed (code {rc}): {msg}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _require_gpu(*tensors: torch.Tensor):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipacError("HIP path called with a CPU tensor: there is no CPU fallback (move inputs to cuda)")
        if t is not None and not t.is_contiguous():
            raise HipacError("HIP path needs contiguous tensors")


# ----------------------------------------------------------------------------
# resampling tables / LUT
# ----------------------------------------------------------------------------


def resample_coeffs(in_size: int, out_size: int = PATCH) -> Tuple[np.ndarray, np.ndarray, int]:
    """Host tables (bounds int32[out,2], kk int32[out,ksize], ksize) from the
    library's C++ restatement of Pillow's precompute_coeffs."""
    lib = load_library()
    ksize = lib.hipac_resample_coeffs(in_size, out_size, None, None, 0)
    if ksize <= 0:
        _check(ksize, "hipac_resample_coeffs")
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    rc = lib.hipac_resample_coeffs(in_size, out_size, bounds.ctypes.data, kk.ctypes.data, ksize)
    if rc != ksize:
        _check(rc if rc < 0 else -1, "hipac_resample_coeffs")
    return bounds, kk, ksize


def normalize_lut() -> np.ndarray:
    """float32[3,256] = (v/255 - mean_c)/std_c in fp32, torchvision's op order
    (ToTensor .div(255), Normalize sub/div), reference src/main.py:815-816."""
    v = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float32)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float32)
    return ((v[None, :] - mean[:, None]) / std[:, None]).contiguous().numpy()


_dev_tables: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor, int]] = {}
_dev_lut: Dict[int, torch.Tensor] = {}


def device_tables(P: int, device: torch.device):
    key = (P, device.index or 0)
    if key not in _dev_tables:
        b, k, ksize = resample_coeffs(P, PATCH)
        _dev_tables[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ksize)
    return _dev_tables[key]


def device_lut(device: torch.device) -> torch.Tensor:
    key = device.index or 0
    if key not in _dev_lut:
        _dev_lut[key] = torch.from_numpy(normalize_lut()).to(device)
    return _dev_lut[key]


# ----------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------

_STAGE_NAMES = ("layer1", "layer2", "layer3", "layer4")


class PackedResNet18:
    """Owner of a hipac_weights_t handle (BN-folded, repacked, on device)."""

    def __init__(self, sd: Dict[str, torch.Tensor], precision: str = "bf16", bn_eps: float = 1e-5,
                 device: Optional[torch.device] = None):
        """``sd``: bare torchvision-named tensors (conv1.weight, bn1.*, layerK.B.*,
        optional fc.*).  Use ``weights.canonical_state_dict`` to get there from
        the reference's key layouts.  ``device``: the ROCm device the handle lives on
        (default: torch's current device); forwards must be given tensors of that device."""
        if precision not in PRECISIONS:
            raise HipacError(f"precision must be one of {sorted(PRECISIONS)}")
        lib = load_library()
        keep = []

        def host(name):
            if name not in sd:
                raise HipacError(f"state dict lacks '{name}'")
            a = np.ascontiguousarray(sd[name].detach().to("cpu", torch.float32).numpy())
            keep.append(a)
            return a.ctypes.data

        def convbn(conv, bn):
            return ConvBN(host(conv + ".weight"), host(bn + ".weight"), host(bn + ".bias"),
                          host(bn + ".running_mean"), host(bn + ".running_var"))

        p = ResNet18Params()
        p.stem = convbn("conv1", "bn1")
        for s, name in enumerate(_STAGE_NAMES):
            for b in (0, 1):
                for c in (0, 1):
                    p.block[2 * s + b][c] = convbn(f"{name}.{b}.conv{c + 1}", f"{name}.{b}.bn{c + 1}")
            if s > 0:
                p.down[s - 1] = convbn(f"{name}.0.downsample.0", f"{name}.0.downsample.1")
        if "fc.weight" in sd:
            p.num_classes = int(sd["fc.weight"].shape[0])
            p.fc_w = host("fc.weight")
            p.fc_b = host("fc.bias")
        else:
            p.num_classes = 0
            p.fc_w = None
            p.fc_b = None
        p.bn_eps = bn_eps
        handle = C.c_void_p()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(self.device):
            _check(lib.hipac_resnet18_pack(C.byref(p), PRECISIONS[precision], C.byref(handle)), "hipac_resnet18_pack")
        self._lib = lib
        self.handle = handle
        self.precision = precision
        self.num_classes = p.num_classes
        self._ws: Optional[torch.Tensor] = None
        self._ws_batch = 0
        self._stage: Optional[torch.Tensor] = None  # fp32 parity mode: normalised copy of a uint8 batch

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and self._lib is not None:
            self._lib.hipac_weights_free(h)
            self.handle = None

    def batch_buffer(self, n: int, device: torch.device) -> torch.Tensor:
        """uint8[n,224,224,3] view of a grow-only buffer kept on the handle (a whole slide's kept windows: multi-GB;
        allocating it per slide would go through hipMalloc whenever the caching allocator had let go of the block).
        Valid until the next call; a consumer on another stream must order itself after the forwards that read it."""
        need = n * PATCH * PATCH * 3
        device = torch.device(device)
        if device.index is None:  # "cuda" and "cuda:0" must compare equal, or the buffer is re-made on every call
            device = torch.device(device.type, torch.cuda.current_device())
        buf = getattr(self, "_batch", None)
        if buf is None or buf.numel() < need or buf.device != device:
            buf = self._batch = None  # release BOTH references before growing: the old block must be free for the new one
            buf = self._batch = torch.empty(need, dtype=torch.uint8, device=device)
        return buf[:need].view(n, PATCH, PATCH, 3)

    def workspace(self, batch: int, device: torch.device) -> torch.Tensor:
        need = self._lib.hipac_resnet18_workspace_bytes(batch, PRECISIONS[self.precision])
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        self._ws_batch = batch
        return self._ws

    def forward(
        self,
        x: torch.Tensor,
        want_feats: bool = True,
        want_logits: bool = False,
        want_labels: bool = False,
        native_layout: bool = False,
    ):
        """x: float32[B,3,224,224] on cuda, or (native_layout) T[B,230,232,4], or raw
        uint8[B,224,224,3] patches (ToTensor/Normalize fused into the stem kernel).
        Returns (feats|None, logits|None, labels|None), all on x's device."""
        _require_gpu(x)
        layout = IN_NHWC4_PAD if native_layout else IN_NCHW_F32
        if x.dtype == torch.uint8:
            if tuple(x.shape[1:]) != (PATCH, PATCH, 3):
                raise HipacError(f"uint8 input must be [B,224,224,3], got {tuple(x.shape)}")
            layout = IN_U8_HWC
            if self.precision == "fp32":  # the debugging mode has no fused uint8 stem: normalise first (bit-exact LUT)
                # the staging buffer is kept on the object (grow-only), like the workspace: a multi-GB block freed
                # here while the kernels that read it are still queued could be handed to a caller that fills
                # its next buffer from ANOTHER stream
                n_el = x.shape[0] * 3 * PATCH * PATCH
                if self._stage is None or self._stage.numel() < n_el or self._stage.device != x.device:
                    self._stage = torch.empty(n_el, dtype=torch.float32, device=x.device)
                stage = self._stage[:n_el].view(x.shape[0], 3, PATCH, PATCH)
                x, layout, native_layout = patches_normalize(x, "nchw_f32", out=stage), IN_NCHW_F32, False
        elif native_layout:
            if tuple(x.shape[1:]) != (PAD_H, PAD_W, 4) or x.dtype != TORCH_DTYPE[PRECISIONS[self.precision]]:
                raise HipacError(f"native input must be {self.precision}[B,{PAD_H},{PAD_W},4], got {x.dtype}{tuple(x.shape)}")
        else:
            if tuple(x.shape[1:]) != (3, PATCH, PATCH) or x.dtype != torch.float32:
                raise HipacError(f"input must be float32[B,3,224,224], got {x.dtype}{tuple(x.shape)}")
        B = x.shape[0]
        dev = x.device
        if (want_logits or want_labels) and self.num_classes == 0:
            raise HipacError("logits requested but the model has fc = Identity")
        feats = torch.empty((B, 512), dtype=torch.float32, device=dev) if want_feats else None
        logits = torch.empty((B, self.num_classes), dtype=torch.float32, device=dev) if want_logits else None
        labels = torch.empty((B,), dtype=torch.int64, device=dev) if want_labels else None
        if B == 0:
            return feats, logits, labels
        ws = self.workspace(B, dev)
        with torch.cuda.device(dev):
            rc = self._lib.hipac_resnet18_forward(
                self.handle, x.data_ptr(), B, layout,
                _ptr(feats), _ptr(logits), _ptr(labels), ws.data_ptr(), ws.numel(), _stream())
        _check(rc, "hipac_resnet18_forward")
        return feats, logits, labels

    def run_ops(self, x: torch.Tensor, first: int, last: int):
        """Re-run trunk ops first..last on the activations the last forward(x) left behind
        (profiling aid; x must be the same batch, it is re-read by op 0)."""
        layout = IN_U8_HWC if x.dtype == torch.uint8 else (IN_NCHW_F32 if x.dtype == torch.float32 else IN_NHWC4_PAD)
        with torch.cuda.device(self._ws.device):
            _check(self._lib.hipac_resnet18_run_ops(self.handle, x.data_ptr(), layout, self._ws.data_ptr(),
                                                    self._ws.numel(), x.shape[0], first, last, _stream()),
                   "hipac_resnet18_run_ops")

    def tap(self, batch: int, tap: int) -> torch.Tensor:
        """Intermediate activation of the last forward (float32 NCHW); tests only."""
        shapes = [(64, 112), (64, 56), (64, 56), (64, 56), (128, 28), (128, 28), (256, 14), (256, 14), (512, 7), (512, 7)]
        if not 0 <= tap < len(shapes):
            raise HipacError(f"tap index {tap} out of range")
        # the C entry point trusts `batch` (it addresses the workspace with that batch's plan): never let it
        # exceed what the workspace of the last forward was sized for
        if self._ws is None or batch < 1 or \
                self._lib.hipac_resnet18_workspace_bytes(batch, self._lib.hipac_weights_precision(self.handle)) > self._ws.numel():
            raise HipacError(f"tap: batch {batch} does not fit the workspace of the last forward")
        c, h = shapes[tap]
        dst = torch.empty((batch, c, h, h), dtype=torch.float32, device=self._ws.device)
        with torch.cuda.device(dst.device):
            _check(self._lib.hipac_resnet18_tap(self.handle, self._ws.data_ptr(), batch, tap, dst.data_ptr(), _stream()),
                   "hipac_resnet18_tap")
        return dst


# ----------------------------------------------------------------------------
# tile preprocess
# ----------------------------------------------------------------------------

_OUT_FMT = {"nchw_f32": OUT_NCHW_F32, "bf16": OUT_NHWC4_PAD_BF16, "fp16": OUT_NHWC4_PAD_FP16, "u8": OUT_U8_HWC}


def _alloc_shape(n: int, fmt: str):
    return {"nchw_f32": (n, 3, PATCH, PATCH), "u8": (n, PATCH, PATCH, 3)}.get(fmt, (n, PAD_H, PAD_W, 4))


def _alloc_dtype(fmt: str):
    return {"nchw_f32": torch.float32, "u8": torch.uint8, "bf16": torch.bfloat16}.get(fmt, torch.float16)


def _alloc_out(n: int, fmt: str, device) -> torch.Tensor:
    return torch.empty(_alloc_shape(n, fmt), dtype=_alloc_dtype(fmt), device=device)


def tile_preprocess(
    level: torch.Tensor,
    xy: torch.Tensor,
    P: int,
    out_format: str = "nchw_f32",
    width: Optional[int] = None,
    want_sums: bool = True,
    out: Optional[torch.Tensor] = None,
):
    """level: uint8[H, Wp, C] on cuda (C = 3|4; Wp >= width is the padded row
    length, Wp*C a multiple of 16 when P > 224); xy: int32[n,2] on cuda.
    Returns (out, sums uint32-as-int64|None, keep uint8|None)."""
    _require_gpu(level, xy)
    if level.dtype != torch.uint8 or level.dim() != 3 or level.shape[2] not in (3, 4):
        raise HipacError("level must be uint8[H,W,3|4]")
    if xy.dtype != torch.int32 or xy.dim() != 2 or xy.shape[1] != 2:
        raise HipacError("xy must be int32[n,2]")
    if out_format not in _OUT_FMT:
        raise HipacError(f"out_format must be one of {sorted(_OUT_FMT)}")
    H, Wp, ch = level.shape
    W = Wp if width is None else int(width)
    n = xy.shape[0]
    dev = level.device
    if out is None:
        out = _alloc_out(n, out_format, dev)
    sums = torch.empty((n,), dtype=torch.int32, device=dev) if want_sums else None
    keep = torch.empty((n,), dtype=torch.uint8, device=dev) if want_sums else None
    if n == 0:
        return out, sums, keep
    lib = load_library()
    if P > PATCH:
        b, k, ksize = device_tables(P, dev)
        bp, kp = b.data_ptr(), k.data_ptr()
    else:
        bp, kp, ksize = None, None, 0
    lut = device_lut(dev)
    with torch.cuda.device(dev):
        rc = lib.hipac_tile_preprocess(level.data_ptr(), W, H, Wp * ch, ch, xy.data_ptr(), n, P, bp, kp, ksize,
                                       lut.data_ptr(), out.data_ptr(), _OUT_FMT[out_format], _ptr(sums), _ptr(keep),
                                       _stream())
    _check(rc, "hipac_tile_preprocess")
    return out, sums, keep


class LevelPlanes:
    """Whole-level resampling for windows on the 224-pixel lattice (P = 448 / 896 / 1792):
    build once per level, then ``stats`` (sums, keep) and ``gather`` (uint8 patches) for any
    list of lattice windows.  Bit-identical to ``tile_preprocess`` on the same windows."""

    def __init__(self, level: torch.Tensor, P: int, width: Optional[int] = None):
        _require_gpu(level)
        if level.dtype != torch.uint8 or level.dim() != 3 or level.shape[2] != 3:
            raise HipacError("level must be uint8[H,W,3]")
        lib = load_library()
        H, Wp, _ = level.shape
        self.W, self.H, self.P = (Wp if width is None else int(width)), H, P
        hb, db, cb = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _check(lib.hipac_level_planes_sizes(self.W, self.H, P, C.byref(hb), C.byref(db), C.byref(cb)),
               "hipac_level_planes_sizes")
        dev = level.device
        himg = torch.empty(hb.value, dtype=torch.uint8, device=dev)  # only needed while building
        self.dimg = torch.empty(db.value, dtype=torch.uint8, device=dev)
        self.cells = torch.empty(cb.value // 4, dtype=torch.int32, device=dev)
        b, k, ksize = device_tables(P, dev)
        with torch.cuda.device(dev):
            rc = lib.hipac_level_build_planes(level.data_ptr(), self.W, self.H, Wp * 3, P, b.data_ptr(), k.data_ptr(),
                                              ksize, himg.data_ptr(), self.dimg.data_ptr(), self.cells.data_ptr(),
                                              _stream())
        _check(rc, "hipac_level_build_planes")
        self._lib, self.device = lib, dev
        himg.record_stream(torch.cuda.current_stream(dev))
        if os.environ.get("HIPAC_KEEP_HIMG"):
            self.himg = himg  # debugging aid

    def stats(self, xy: torch.Tensor):
        n = xy.shape[0]
        sums = torch.empty((n,), dtype=torch.int32, device=self.device)
        keep = torch.empty((n,), dtype=torch.uint8, device=self.device)
        if n:
            with torch.cuda.device(self.device):
                _check(self._lib.hipac_level_window_stats(self.cells.data_ptr(), self.W, self.H, self.P, xy.data_ptr(),
                                                          n, sums.data_ptr(), keep.data_ptr(), _stream()),
                       "hipac_level_window_stats")
        return sums, keep

    def gather(self, xy: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """uint8[n,224,224,3] resized pixels of the windows at ``xy``; ``out``: write into this
        (contiguous, e.g. a slice of a larger batch buffer) instead of allocating."""
        n = xy.shape[0]
        if out is None:
            out = torch.empty((n, PATCH, PATCH, 3), dtype=torch.uint8, device=self.device)
        elif out.dtype != torch.uint8 or tuple(out.shape) != (n, PATCH, PATCH, 3) or not out.is_contiguous() or \
                out.device != self.device:
            raise HipacError("gather: out must be a contiguous uint8[n,224,224,3] tensor on the planes' device")
        if n:
            with torch.cuda.device(self.device):
                _check(self._lib.hipac_level_gather(self.dimg.data_ptr(), self.W, self.H, self.P, xy.data_ptr(), n,
                                                    out.data_ptr(), _stream()), "hipac_level_gather")
        return out


def window_labels(mask: torch.Tensor, xy: torch.Tensor, P: int) -> torch.Tensor:
    """mask: uint8[H,W] on cuda; returns uint8[n] (1 = tumour)."""
    _require_gpu(mask, xy)
    if mask.dtype != torch.uint8 or mask.dim() != 2:
        raise HipacError("mask must be uint8[H,W]")
    n = xy.shape[0]
    labels = torch.empty((n,), dtype=torch.uint8, device=mask.device)
    if n == 0:
        return labels
    H, W = mask.shape
    with torch.cuda.device(mask.device):
        rc = load_library().hipac_window_labels(mask.data_ptr(), W, H, W, xy.data_ptr(), n, P, labels.data_ptr(), _stream())
    _check(rc, "hipac_window_labels")
    return labels


def mask_cells(mask: torch.Tensor, width: Optional[int] = None) -> torch.Tensor:
    """uint8[H,Wp] mask (true width ``width`` <= Wp, default Wp; a row pitch that is a multiple of 16
    takes the 16-byte-load path) -> uint8[ceil(H/224), ceil(width/224)] "any pixel > 0" flags."""
    _require_gpu(mask)
    H, Wp = mask.shape
    W = Wp if width is None else int(width)
    if not 0 < W <= Wp:
        raise HipacError("mask_cells: width out of range")
    out = torch.empty(((H + 223) // 224, (W + 223) // 224), dtype=torch.uint8, device=mask.device)
    with torch.cuda.device(mask.device):
        _check(load_library().hipac_mask_cells(mask.data_ptr(), W, H, Wp, out.data_ptr(), _stream()), "hipac_mask_cells")
    return out


def window_labels_cells(cellany: torch.Tensor, W: int, H: int, xy: torch.Tensor, P: int) -> torch.Tensor:
    n = xy.shape[0]
    labels = torch.empty((n,), dtype=torch.uint8, device=xy.device)
    if n:
        with torch.cuda.device(xy.device):
            _check(load_library().hipac_window_labels_cells(cellany.data_ptr(), W, H, P, xy.data_ptr(), n,
                                                            labels.data_ptr(), _stream()), "hipac_window_labels_cells")
    return labels


def patches_normalize(patches: torch.Tensor, out_format: str = "nchw_f32", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8[n,224,224,3] on cuda -> network input (ToTensor + Normalize); ``out``: write into this tensor."""
    _require_gpu(patches)
    if patches.dtype != torch.uint8 or tuple(patches.shape[1:]) != (PATCH, PATCH, 3):
        raise HipacError("patches must be uint8[n,224,224,3]")
    n = patches.shape[0]
    if out is None:
        out = _alloc_out(n, out_format, patches.device)
    elif not out.is_contiguous() or out.device != patches.device or out.shape != _alloc_shape(n, out_format) or \
            out.dtype != _alloc_dtype(out_format):
        raise HipacError("patches_normalize: out has the wrong shape / dtype / device")
    if n == 0:
        return out
    lut = device_lut(patches.device)
    with torch.cuda.device(patches.device):
        rc = load_library().hipac_patches_normalize(patches.data_ptr(), n, lut.data_ptr(), out.data_ptr(),
                                                    _OUT_FMT[out_format], _stream())
    _check(rc, "hipac_patches_normalize")
    return out


# ----------------------------------------------------------------------------
# MIL head (hipac_mil_forward)
# ----------------------------------------------------------------------------
MIL_POOLING = {"attention": 0, "mean": 1, "max": 2}


def mil_forward(sd: Dict[str, torch.Tensor], pooling: str, feats: torch.Tensor, bag_offsets: torch.Tensor,
                want_attn: bool = True, want_pooled: bool = False):
    """Score many bags at once.  ``sd``: MILClassifier state_dict tensors (float32, on the device of
    ``feats``); ``feats`` float32[n,F] with the rows of a bag contiguous; ``bag_offsets`` int32[n_bags+1]
    (CPU or device; validated on the host).  Returns (logits[n_bags,C], attn[n] or None, pooled or None)."""
    if pooling not in MIL_POOLING:
        raise ValueError("Unknown pooling: choose from 'attention', 'mean', 'max'")
    _require_gpu(feats)
    if feats.dtype != torch.float32 or feats.dim() != 2:
        raise HipacError("feats must be float32[n, feature_dim]")
    offs_host = bag_offsets.detach().to("cpu", torch.int64)
    n, F = int(feats.shape[0]), int(feats.shape[1])
    if offs_host.dim() != 1 or offs_host.numel() < 2 or int(offs_host[0]) != 0 or int(offs_host[-1]) != n or \
            bool((offs_host[1:] <= offs_host[:-1]).any()):
        raise HipacError("bag_offsets must start at 0, end at n and increase strictly (no empty bags)")
    n_bags = offs_host.numel() - 1
    dev = feats.device
    offs = offs_host.to(torch.int32).to(dev)

    def w(key):
        t = sd[key]
        if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous():
            raise HipacError(f"MIL weight {key} must be a contiguous float32 tensor on {dev}")
        return t

    p = MilParams()
    attention = pooling == "attention"
    if attention:
        p.attn_V_w, p.attn_V_b = w("aggregator.attn_V.weight").data_ptr(), w("aggregator.attn_V.bias").data_ptr()
        p.attn_U_w, p.attn_U_b = w("aggregator.attn_U.weight").data_ptr(), w("aggregator.attn_U.bias").data_ptr()
        p.attn_dim = int(sd["aggregator.attn_V.weight"].shape[0])
        if tuple(sd["aggregator.attn_V.weight"].shape) != (p.attn_dim, F):
            raise HipacError("aggregator.attn_V.weight does not match feature_dim")
    p.fc1_w, p.fc1_b = w("classifier.0.weight").data_ptr(), w("classifier.0.bias").data_ptr()
    p.fc2_w, p.fc2_b = w("classifier.2.weight").data_ptr(), w("classifier.2.bias").data_ptr()
    p.feature_dim, p.hidden_dim = F, int(sd["classifier.0.weight"].shape[0])
    p.num_classes = int(sd["classifier.2.weight"].shape[0])
    if tuple(sd["classifier.0.weight"].shape) != (p.hidden_dim, F):
        raise HipacError("classifier.0.weight does not match feature_dim")
    logits = torch.empty((n_bags, p.num_classes), dtype=torch.float32, device=dev)
    attn = torch.empty((n,), dtype=torch.float32, device=dev) if (attention and want_attn) else None
    pooled = torch.empty((n_bags, F), dtype=torch.float32, device=dev) if want_pooled else None
    scores = torch.empty((n,), dtype=torch.float32, device=dev) if attention else None
    with torch.cuda.device(dev):
        rc = load_library().hipac_mil_forward(C.byref(p), MIL_POOLING[pooling], feats.data_ptr(), offs.data_ptr(), n,
                                              n_bags, logits.data_ptr(), _ptr(attn), _ptr(pooled), _ptr(scores),
                                              _stream())
    _check(rc, "hipac_mil_forward")
    return logits, attn, pooled


# ----------------------------------------------------------------------------
# NT-Xent (hipac_ntxent_fwd_bwd)
# ----------------------------------------------------------------------------
def ntxent_fwd_bwd(z: torch.Tensor, temperature: float, want_grad: bool = True):
    """z = cat(z_i, z_j) float32[2n,d] on cuda -> (loss float32[] , dz float32[2n,d] or None)."""
    _require_gpu(z)
    if z.dtype != torch.float32 or z.dim() != 2 or z.shape[0] % 2 or z.shape[0] == 0:
        raise HipacError("ntxent: z must be float32[2n, d]")
    n, d = z.shape[0] // 2, z.shape[1]
    lib = load_library()
    nbytes = lib.hipac_ntxent_scratch_bytes(n, d)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=z.device)
    loss = torch.empty((), dtype=torch.float32, device=z.device)
    dz = torch.empty_like(z) if want_grad else None
    with torch.cuda.device(z.device):
        rc = lib.hipac_ntxent_fwd_bwd(z.data_ptr(), n, d, float(temperature), loss.data_ptr(), _ptr(dz),
                                      scratch.data_ptr(), nbytes, _stream())
    _check(rc, "hipac_ntxent_fwd_bwd")
    return loss, dz
