"""Native training steps: the SimCLR step of ``pretrain_simclr``
(src/models/simclr.py:85-96) and the classifier step of the fine-tune loops (src/main.py:497-510,
:576-589) on the hand-written HIP kernels of csrc/train.hip.

Nothing here goes through torch autograd or torch's conv / batch-norm kernels: a step is a fixed chain of
C-ABI calls on flat device buffers --

    encoder forward (train-mode BN, per-replica statistics)      hipac_train_encoder_forward
    projector / fc                                                hipac_linear_forward
    NT-Xent over the GLOBAL batch | weighted cross-entropy        hipac_ntxent_fwd_bwd | hipac_cross_entropy_fwd_bwd
    head backward, encoder backward                               hipac_linear_backward, hipac_train_encoder_backward
    gradient all-reduce (N > 1, RCCL)                              torch.distributed (plumbing)
    Adam                                                          hipac_adam_step

PyTorch owns the device memory, the streams and the process group.  Two arithmetics: "fp32" on the exact f32
MFMA (the reference's ``pretrain_simclr`` is fp32) and "fp16" mixed precision (csrc/train_amp.hip: fp16 operands and
maps on the fp16 MFMA, fp32 accumulation and master weights, ``GradScaler``) -- what the reference's classifier loops
run under ``torch.cuda.amp.autocast()`` (src/main.py:499-508); the classifier trainer defaults to it.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Dict, List, Optional, Tuple

import torch

from . import capi

BN_MOMENTUM, BN_EPS = 0.1, 1e-5  # torch.nn.BatchNorm2d defaults (torchvision's resnet18 uses them)
_STAGES = ("layer1", "layer2", "layer3", "layer4")


def conv_table() -> List[dict]:
    """Geometry, offsets and torchvision names of the 20 convolutions, in the library's order."""
    lib = capi.load_library()
    names = [("conv1", "bn1")]
    for s, st in enumerate(_STAGES):
        for b in (0, 1):
            names.append((f"{st}.{b}.conv1", f"{st}.{b}.bn1"))
            names.append((f"{st}.{b}.conv2", f"{st}.{b}.bn2"))
            if s > 0 and b == 0:
                names.append((f"{st}.{b}.downsample.0", f"{st}.{b}.downsample.1"))
    out = []
    for i in range(lib.hipac_train_num_convs()):
        co, ci, ks, st_ = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        po, so = C.c_int64(), C.c_int64()
        capi._check(lib.hipac_train_conv_desc(i, C.byref(co), C.byref(ci), C.byref(ks), C.byref(st_), C.byref(po), C.byref(so)),
                    "hipac_train_conv_desc")
        out.append(dict(conv=names[i][0], bn=names[i][1], cout=co.value, cin=ci.value, ks=ks.value, stride=st_.value,
                        param_off=po.value, stat_off=so.value))
    assert len(out) == len(names) == 20
    return out


class FlatAdam:
    """A flat fp32 parameter buffer with its gradient and Adam state (torch.optim.Adam semantics)."""

    def __init__(self, n: int, device, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.params = torch.zeros(n, dtype=torch.float32, device=device)
        self.grads = torch.zeros_like(self.params)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.lr, self.betas, self.eps, self.t = lr, betas, eps, 0

    def step(self):
        """One Adam update from ``grads`` (callers under a GradScaler unscale and check them first)."""
        self.t += 1
        with torch.cuda.device(self.params.device):
            capi._check(capi.load_library().hipac_adam_step(
                self.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.params.numel(),
                self.lr, self.betas[0], self.betas[1], self.eps, self.t, capi._stream()), "hipac_adam_step")


class GradScaler:
    """torch.cuda.amp.GradScaler as the reference's fine-tune loops use it (src/main.py:494, :506-508; defaults:
    init_scale 65536, growth_factor 2, backoff_factor 0.5, growth_interval 2000): the loss gradient is multiplied by
    ``scale`` before the fp16 backward, ``unscale_and_check`` divides the fp32 gradients by it and tells whether one of
    them is inf / nan -- then the optimizer step is skipped and the scale halves; after ``growth_interval`` good steps in
    a row it doubles.  With several ranks the verdict is shared (MAX), as every replica must skip or step together."""

    def __init__(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000, enabled: bool = True):
        self.scale, self.growth_factor, self.backoff_factor = float(init_scale), growth_factor, backoff_factor
        self.growth_interval, self.enabled, self._good = growth_interval, enabled, 0
        self._flag: Optional[torch.Tensor] = None
        self.skipped = 0

    def get_scale(self) -> float:
        return self.scale if self.enabled else 1.0

    def unscale_and_check(self, *flats: "FlatAdam") -> bool:
        """Divide the loss scale out of every buffer's gradients; True when all of them are finite."""
        if not self.enabled:
            return True
        dev = flats[0].grads.device
        if self._flag is None or self._flag.device != dev:
            self._flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self._flag.zero_()
        lib = capi.load_library()
        with torch.cuda.device(dev):
            for f in flats:
                capi._check(lib.hipac_grads_unscale_check(f.grads.data_ptr(), f.grads.numel(), 1.0 / self.scale,
                                                          self._flag.data_ptr(), capi._stream()), "hipac_grads_unscale_check")
        bad = float(self._flag.item())
        import torch.distributed as dist

        from .dist import active
        if active():
            t = torch.tensor([bad], dtype=torch.float32, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bad = float(t.item())
        return bad == 0.0

    def update(self, ok: bool):
        if not self.enabled:
            return
        if ok:
            self._good += 1
            if self._good >= self.growth_interval:
                self.scale *= self.growth_factor
                self._good = 0
        else:
            self.scale *= self.backoff_factor
            self._good = 0
            self.skipped += 1


class NativeEncoder:
    """ResNet18 encoder (fc = Identity) as flat device buffers + the native train-mode forward / backward.
    ``precision``: "fp32" (exact f32 MFMA: the reference's pretrain_simclr arithmetic) or "fp16" (mixed precision: fp16
    operands and maps on the fp16 MFMA, fp32 accumulation, the same fp32 flat buffers -- the reference's autocast loops)."""

    def __init__(self, sd: Dict[str, torch.Tensor], device="cuda", lr: float = 1e-3, precision: str = "fp32"):
        """``sd``: bare torchvision-named tensors (``weights.canonical_state_dict`` of any reference layout)."""
        if precision not in ("fp32", "fp16"):
            raise capi.HipacError("training precision must be 'fp32' or 'fp16'")
        self.precision = precision
        self.device = torch.device(device)
        self.lib = capi.load_library()
        amp = precision == "fp16"
        self._fn_ws = self.lib.hipac_train_amp_workspace_bytes if amp else self.lib.hipac_train_workspace_bytes
        self._fn_fwd = self.lib.hipac_train_amp_encoder_forward if amp else self.lib.hipac_train_encoder_forward
        self._fn_bwd = self.lib.hipac_train_amp_encoder_backward if amp else self.lib.hipac_train_encoder_backward
        self._fn_off = self.lib.hipac_train_amp_debug_offset if amp else self.lib.hipac_train_debug_offset
        self._map_dtype, self._map_bytes = (torch.float16, 2) if amp else (torch.float32, 4)
        self.table = conv_table()
        self.opt = FlatAdam(self.lib.hipac_train_param_floats(), self.device, lr)
        self.stats = torch.zeros(self.lib.hipac_train_stat_floats(), dtype=torch.float32, device=self.device)
        self.num_batches_tracked = 0
        self._ws: Dict[int, torch.Tensor] = {}
        self._ws_batch: Dict[int, int] = {}
        self.load(sd)

    # ---- named views -------------------------------------------------------------------------
    def _views(self, flat: torch.Tensor, e: dict):
        n_w = e["cout"] * e["cin"] * e["ks"] * e["ks"]
        o = e["param_off"]
        return (flat[o:o + n_w].view(e["cout"], e["cin"], e["ks"], e["ks"]), flat[o + n_w:o + n_w + e["cout"]],
                flat[o + n_w + e["cout"]:o + n_w + 2 * e["cout"]])

    def load(self, sd: Dict[str, torch.Tensor]):
        for e in self.table:
            w, g, b = self._views(self.opt.params, e)
            w.copy_(sd[e["conv"] + ".weight"])
            g.copy_(sd[e["bn"] + ".weight"])
            b.copy_(sd[e["bn"] + ".bias"])
            so, co = e["stat_off"], e["cout"]
            self.stats[so:so + co].copy_(sd[e["bn"] + ".running_mean"])
            self.stats[so + co:so + 2 * co].copy_(sd[e["bn"] + ".running_var"])
        nbt = sd.get("bn1.num_batches_tracked")
        self.num_batches_tracked = int(nbt) if nbt is not None else 0

    def state_dict(self, grads: bool = False) -> Dict[str, torch.Tensor]:
        """Bare torchvision-named tensors (CPU clones); ``grads=True``: the gradient buffer under the same names."""
        flat = self.opt.grads if grads else self.opt.params
        out = {}
        for e in self.table:
            w, g, b = self._views(flat, e)
            out[e["conv"] + ".weight"] = w.detach().cpu().clone()
            out[e["bn"] + ".weight"] = g.detach().cpu().clone()
            out[e["bn"] + ".bias"] = b.detach().cpu().clone()
            if not grads:
                so, co = e["stat_off"], e["cout"]
                out[e["bn"] + ".running_mean"] = self.stats[so:so + co].cpu().clone()
                out[e["bn"] + ".running_var"] = self.stats[so + co:so + 2 * co].cpu().clone()
                out[e["bn"] + ".num_batches_tracked"] = torch.tensor(self.num_batches_tracked, dtype=torch.int64)
        return out

    # ---- compute -----------------------------------------------------------------------------
    def workspace(self, slot: int, batch: int) -> torch.Tensor:
        need = self._fn_ws(batch)
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws[slot] = ws
        self._ws_batch[slot] = batch
        return ws

    def forward(self, x: torch.Tensor, slot: int = 0, update_stats: bool = True) -> torch.Tensor:
        """x float32[B,3,224,224] on the encoder's device -> feats float32[B,512]; activations stay in workspace ``slot``."""
        if not x.is_cuda or x.dtype != torch.float32 or tuple(x.shape[1:]) != (3, 224, 224) or not x.is_contiguous():
            raise capi.HipacError("native encoder input must be a contiguous float32[B,3,224,224] ROCm tensor")
        B = x.shape[0]
        ws = self.workspace(slot, B)
        feats = torch.empty((B, 512), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            capi._check(self._fn_fwd(
                self.opt.params.data_ptr(), self.stats.data_ptr() if update_stats else None, x.data_ptr(), B, BN_MOMENTUM, BN_EPS,
                feats.data_ptr(), ws.data_ptr(), ws.numel(), capi._stream()), "hipac_train_encoder_forward")
        if update_stats:
            self.num_batches_tracked += 1
        return feats

    def tap(self, slot: int, kind: str, conv: int) -> torch.Tensor:
        """Test tap: a map the last forward left in workspace ``slot`` -- kind "pre" / "post" -> float32 NCHW
        [B,Cout,H,W] of conv ``conv``, "pool" -> the pooled stem map, "stats" -> (mean, rstd)."""
        B, e = self._ws_batch[slot], self.table[conv]
        off = self._fn_off(B, {"pre": 0, "post": 1, "pool": 2, "stats": 3, "pool_idx": 4}[kind], conv)
        if off < 0:
            raise capi.HipacError("train tap: bad argument")
        if kind == "pool_idx":  # uint8 NCHW [B,64,56,56]: which of the 9 window positions (dy * 3 + dx) won
            n = B * 56 * 56 * 64
            return self._ws[slot][off:off + n].view(B, 56, 56, 64).permute(0, 3, 1, 2).contiguous()
        if kind == "stats":
            v = self._ws[slot][off:off + 8 * e["cout"]].view(torch.float32)
            return v[:e["cout"]].clone(), v[e["cout"]:].clone()
        if kind == "pool":
            C_, H = 64, 56
        else:
            C_ = e["cout"]
            H = {0: 112}.get(conv, {64: 56, 128: 28, 256: 14, 512: 7}[C_])
        n = B * H * H * C_
        return (self._ws[slot][off:off + self._map_bytes * n].view(self._map_dtype).view(B, H, H, C_).permute(0, 3, 1, 2)
                .contiguous().float())

    def backward(self, dfeats: torch.Tensor, slot: int = 0, accumulate: bool = False):
        B = self._ws_batch[slot]
        if tuple(dfeats.shape) != (B, 512) or dfeats.dtype != torch.float32 or not dfeats.is_contiguous():
            raise capi.HipacError("dfeats must be a contiguous float32[B,512] tensor of the forward's batch")
        ws = self._ws[slot]
        with torch.cuda.device(self.device):
            capi._check(self._fn_bwd(
                self.opt.params.data_ptr(), dfeats.data_ptr(), B, self.opt.grads.data_ptr(), 1 if accumulate else 0,
                ws.data_ptr(), ws.numel(), capi._stream()), "hipac_train_encoder_backward")


class NativeLinear:
    """nn.Linear (+ optional ReLU) on the native GEMM; weight [N,K] and bias [N] are views of a FlatAdam buffer."""

    def __init__(self, flat: FlatAdam, offset: int, n_out: int, n_in: int, relu: bool = False):
        self.N, self.K, self.relu = n_out, n_in, relu
        self.w = flat.params[offset:offset + n_out * n_in].view(n_out, n_in)
        self.b = flat.params[offset + n_out * n_in:offset + n_out * n_in + n_out]
        self.dw = flat.grads[offset:offset + n_out * n_in].view(n_out, n_in)
        self.db = flat.grads[offset + n_out * n_in:offset + n_out * n_in + n_out]
        self.lib = capi.load_library()

    @staticmethod
    def floats(n_out: int, n_in: int) -> int:
        return n_out * n_in + n_out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = torch.empty((x.shape[0], self.N), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            capi._check(self.lib.hipac_linear_forward(x.data_ptr(), self.w.data_ptr(), self.b.data_ptr(), y.data_ptr(), x.shape[0],
                                                      self.N, self.K, 1 if self.relu else 0, capi._stream()), "hipac_linear_forward")
        return y

    def backward(self, x: torch.Tensor, y: torch.Tensor, dy: torch.Tensor, accumulate: bool, need_dx: bool = True):
        dx = torch.empty_like(x) if need_dx else None
        dym = torch.empty_like(dy) if self.relu else None
        with torch.cuda.device(x.device):
            capi._check(self.lib.hipac_linear_backward(
                x.data_ptr(), self.w.data_ptr(), dy.data_ptr(), y.data_ptr() if self.relu else None, capi._ptr(dym), capi._ptr(dx),
                self.dw.data_ptr(), self.db.data_ptr(), x.shape[0], self.N, self.K, 1 if accumulate else 0, capi._stream()),
                "hipac_linear_backward")
        return dx


def _all_reduce_sum(t: torch.Tensor):
    import torch.distributed as dist

    from .dist import active
    if active():
        if t.is_cuda and dist.get_backend() == "gloo":  # CPU rehearsal backend: stage through the host
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)
        else:
            dist.all_reduce(t)  # SUM: nn.DataParallel reduce-adds the replicas' gradients


def _all_gather_rows(t: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """(rows of every rank concatenated rank-major, first row of this rank); equal row counts per rank."""
    import torch.distributed as dist

    from .dist import active, all_gather_equal

    if not active():
        return t, 0
    return all_gather_equal(t), dist.get_rank() * t.shape[0]


class NativeSimCLRTrainer:
    """SimCLRModel (src/models/simclr.py:14-29) + nt_xent_loss (:31-54) + Adam(lr) (:79), one native step per call."""

    def __init__(self, sd: Dict[str, torch.Tensor], device="cuda", lr: float = 1e-3, temperature: float = 0.5, out_dim: int = 128,
                 precision: str = "fp32", scaler: Optional[GradScaler] = None):
        """``sd``: a SimCLRModel state_dict (``encoder.*``, ``projector.{0,2}.*``).  ``precision``: "fp32" (the
        reference's arithmetic for this loop, src/models/simclr.py:85-96) or "fp16" (opt-in mixed precision with a GradScaler)."""
        from .weights import canonical_state_dict

        self.device = torch.device(device)
        bare = canonical_state_dict({k: v for k, v in sd.items() if not k.startswith("projector.")})
        self.encoder = NativeEncoder(bare, self.device, lr, precision=precision)
        self.scaler = scaler if scaler is not None else GradScaler(enabled=precision == "fp16")
        n1, n2 = NativeLinear.floats(512, 512), NativeLinear.floats(out_dim, 512)
        self.head = FlatAdam(n1 + n2, self.device, lr)
        self.p1 = NativeLinear(self.head, 0, 512, 512, relu=True)
        self.p2 = NativeLinear(self.head, n1, out_dim, 512, relu=False)
        self.p1.w.copy_(sd["projector.0.weight"]), self.p1.b.copy_(sd["projector.0.bias"])
        self.p2.w.copy_(sd["projector.2.weight"]), self.p2.b.copy_(sd["projector.2.bias"])
        self.temperature = temperature

    def forward_backward(self, x_i: torch.Tensor, x_j: torch.Tensor) -> torch.Tensor:
        """Loss of the step (device scalar) with every gradient in place; no optimizer step."""
        enc = self.encoder
        f_i, f_j = enc.forward(x_i, slot=0), enc.forward(x_j, slot=1)  # two separate train-mode passes (:88-91)
        h_i, h_j = self.p1.forward(f_i), self.p1.forward(f_j)
        z_i, z_j = self.p2.forward(h_i), self.p2.forward(h_j)
        self.last_hidden = (h_i, h_j)  # test tap: the projector's ReLU pattern
        # the loss sees the GLOBAL batch, as nn.DataParallel's gather does; every rank evaluates the
        # same loss, so the rows of dz that belong to this rank are its complete gradient
        Zi, r0 = _all_gather_rows(z_i)
        Zj, _ = _all_gather_rows(z_j)
        loss, dZ = capi.ntxent_fwd_bwd(torch.cat([Zi, Zj], dim=0), self.temperature, want_grad=True)
        n, N = z_i.shape[0], Zi.shape[0]
        if self.scaler.enabled:
            dZ = dZ * self.scaler.get_scale()  # the loss scale rides on every gradient until unscale_and_check
        dz_i, dz_j = dZ[r0:r0 + n].contiguous(), dZ[N + r0:N + r0 + n].contiguous()
        dh_i = self.p2.backward(h_i, z_i, dz_i, accumulate=False)
        dh_j = self.p2.backward(h_j, z_j, dz_j, accumulate=True)
        df_i = self.p1.backward(f_i, h_i, dh_i, accumulate=False)
        df_j = self.p1.backward(f_j, h_j, dh_j, accumulate=True)
        enc.backward(df_i, slot=0, accumulate=False)
        enc.backward(df_j, slot=1, accumulate=True)
        _all_reduce_sum(enc.opt.grads)
        _all_reduce_sum(self.head.grads)
        return loss

    def sync_from_rank0(self):
        """N > 1: every replica starts from rank 0's parameters and running statistics, as nn.DataParallel replicates
        the module of device 0 (src/models/simclr.py:77-78)."""
        from .dist import broadcast0

        for t in (self.encoder.opt.params, self.encoder.stats, self.head.params):
            broadcast0(t)

    def step(self, x_i: torch.Tensor, x_j: torch.Tensor) -> torch.Tensor:
        loss = self.forward_backward(x_i, x_j)
        ok = self.scaler.unscale_and_check(self.encoder.opt, self.head)  # fp32: nothing to do, always True
        if ok:
            self.encoder.opt.step()
            self.head.step()
        self.scaler.update(ok)
        return loss

    def state_dict(self) -> Dict[str, torch.Tensor]:
        sd = {"encoder." + k: v for k, v in self.encoder.state_dict().items()}
        sd["projector.0.weight"], sd["projector.0.bias"] = self.p1.w.cpu().clone(), self.p1.b.cpu().clone()
        sd["projector.2.weight"], sd["projector.2.bias"] = self.p2.w.cpu().clone(), self.p2.b.cpu().clone()
        return sd

    def grad_dict(self) -> Dict[str, torch.Tensor]:
        gd = {"encoder." + k: v for k, v in self.encoder.state_dict(grads=True).items()}
        gd["projector.0.weight"], gd["projector.0.bias"] = self.p1.dw.cpu().clone(), self.p1.db.cpu().clone()
        gd["projector.2.weight"], gd["projector.2.bias"] = self.p2.dw.cpu().clone(), self.p2.db.cpu().clone()
        return gd


class NativeClassifierTrainer:
    """ResNet18Classifier (src/models/resnet.py:57-77) + CrossEntropyLoss(weight) + Adam(1e-4) (src/main.py:485-492)."""

    def __init__(self, sd: Dict[str, torch.Tensor], device="cuda", lr: float = 1e-4, class_weights: Optional[torch.Tensor] = None,
                 precision: str = "fp16", scaler: Optional[GradScaler] = None):
        """``sd``: a ResNet18Classifier state_dict (``model.*`` incl. ``model.fc``), any reference layout.
        ``precision``: "fp16" (default: the reference runs this step under autocast + GradScaler, src/main.py:499-508) or
        "fp32" (exact f32 MFMA, the wider arithmetic)."""
        from .weights import canonical_state_dict

        self.device = torch.device(device)
        bare = canonical_state_dict(sd)
        self.encoder = NativeEncoder(bare, self.device, lr, precision=precision)
        self.scaler = scaler if scaler is not None else GradScaler(enabled=precision == "fp16")
        C_ = int(bare["fc.weight"].shape[0])
        self.head = FlatAdam(NativeLinear.floats(C_, 512), self.device, lr)
        self.fc = NativeLinear(self.head, 0, C_, 512)
        self.fc.w.copy_(bare["fc.weight"]), self.fc.b.copy_(bare["fc.bias"])
        self.class_weights = None if class_weights is None else class_weights.to(self.device, torch.float32).contiguous()
        self._scratch = torch.zeros(2 + 8 * 64, dtype=torch.float32, device=self.device)  # hipac_cross_entropy_fwd_bwd: batches <= 16 384

    def forward_backward(self, x: torch.Tensor, labels: torch.Tensor):
        """(loss, logits of this rank).  N > 1: the loss is CrossEntropyLoss(weight) over the GLOBAL batch, as the
        reference evaluates it on nn.DataParallel's gathered logits (src/main.py:499-506): sum_i w_i nll_i over all
        ranks / sum_i w_i over all ranks.  Logits and labels are all-gathered (2 + 1 numbers per image), every rank
        evaluates the same loss and keeps its own rows of the gradient, so the SUM all-reduce of the parameter
        gradients below is exactly the gradient of that global loss."""
        C_ = int(self.fc.w.shape[0])
        if not labels.is_cuda and (int(labels.min()) < 0 or int(labels.max()) >= C_):
            raise capi.HipacError(f"cross_entropy: label outside [0, {C_}) (torch raises an IndexError here)")
        f = self.encoder.forward(x, slot=0)
        logits = self.fc.forward(f)
        loss = torch.empty((), dtype=torch.float32, device=self.device)
        labels = labels.to(self.device, torch.int64).contiguous()
        L, r0 = _all_gather_rows(logits)
        Y, _ = _all_gather_rows(labels)
        dL = torch.empty_like(L)
        need = 2 + 8 * ((int(L.shape[0]) + 255) // 256)  # the kernel's scratch: 2 + 8 floats per 256 rows of the GLOBAL batch
        if self._scratch.numel() < need:  # grow-only (the ABI carries no scratch size: the binding owns this contract)
            self._scratch = torch.zeros(need, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            capi._check(capi.load_library().hipac_cross_entropy_fwd_bwd(
                L.data_ptr(), Y.data_ptr(), capi._ptr(self.class_weights), L.shape[0], L.shape[1],
                loss.data_ptr(), dL.data_ptr(), self._scratch.data_ptr(), capi._stream()), "hipac_cross_entropy_fwd_bwd")
        dlogits = dL[r0:r0 + logits.shape[0]].contiguous()
        if self.scaler.enabled:
            dlogits = dlogits * self.scaler.get_scale()
        df = self.fc.backward(f, logits, dlogits, accumulate=False)
        self.encoder.backward(df, slot=0, accumulate=False)
        _all_reduce_sum(self.encoder.opt.grads)
        _all_reduce_sum(self.head.grads)
        return loss, logits

    def sync_from_rank0(self):
        """N > 1: every replica starts from rank 0's parameters and running statistics (src/main.py:481-482)."""
        from .dist import broadcast0

        for t in (self.encoder.opt.params, self.encoder.stats, self.head.params):
            broadcast0(t)

    def step(self, x: torch.Tensor, labels: torch.Tensor):
        loss, logits = self.forward_backward(x, labels)
        ok = self.scaler.unscale_and_check(self.encoder.opt, self.head)
        if ok:
            self.encoder.opt.step()
            self.head.step()
        self.scaler.update(ok)
        return loss, logits

    def state_dict(self, prefix: str = "model.") -> Dict[str, torch.Tensor]:
        sd = {prefix + k: v for k, v in self.encoder.state_dict().items()}
        sd[prefix + "fc.weight"], sd[prefix + "fc.bias"] = self.fc.w.cpu().clone(), self.fc.b.cpu().clone()
        return sd

    def grad_dict(self) -> Dict[str, torch.Tensor]:
        gd = dict(self.encoder.state_dict(grads=True))
        gd["fc.weight"], gd["fc.bias"] = self.fc.dw.cpu().clone(), self.fc.db.cpu().clone()
        return gd


# ---------------------------------------------------------------------------------------------------------
# bench (BASELINE.json configs[4])
# ---------------------------------------------------------------------------------------------------------
FWD_FLOP_PER_IMAGE = 2 * 1_813_561_344  # encoder convs, SURVEY 8(d)


def bench_simclr_step(args, rank: int, world: int, dev) -> dict:
    """One STEP = the SimCLR training step on 2 x (views / world) images per rank: two train-mode forwards,
    projector, NT-Xent over the all-gathered batch, backward, gradient all-reduce, Adam."""
    from .simclr import SimCLRModel

    torch.manual_seed(0)
    model = SimCLRModel()
    prec = getattr(args, "train_precision", "fp32")
    trainer = NativeSimCLRTrainer(model.state_dict(), device=dev, lr=1e-3, precision=prec)
    n = max(1, args.simclr_views // world)
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    x_i = torch.randn((n, 3, 224, 224), generator=g, device=dev)
    x_j = torch.randn((n, 3, 224, 224), generator=g, device=dev)
    for _ in range(args.warmup):
        trainer.step(x_i, x_j)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(x_i, x_j)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    imgs = 2 * n * world * args.steps
    # forward + data gradient + weight gradient of every conv ~ 3 x the forward FLOPs (the stem has no data gradient)
    flops = 3.0 * FWD_FLOP_PER_IMAGE * imgs
    rec = {
        "metric": "SimCLR training step, images/s (ResNet18 encoder fwd+bwd, NT-Xent, Adam)", "value": imgs / dt,
        "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32" if prec == "fp32" else "f16 (fp32 accumulate, fp32 master weights)", "data": "synthetic",
        "config": {"workload": f"simclr_step_2x{n * world}_views", "views_per_rank": n, "projector": "512-512-128",
                   "loss": "NT-Xent T=0.5 over the global batch", "optimizer": "Adam lr 1e-3",
                   "parallelism": f"data parallel x{world}: all-gather of z, all-reduce(sum) of 11.5 M fp32 gradients"
                   if world > 1 else "single GPU"},
        "final_loss": float(loss), "tflops": flops / dt / 1e12,
    }
    if rank == 0:
        # roofline of the encoder forward (the exact-f32 MFMA convolutions): HIP events on the launch stream
        enc = trainer.encoder
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        enc.forward(x_i, slot=0, update_stats=False)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(2):
            enc.forward(x_i, slot=0, update_stats=False)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 2
        tf = FWD_FLOP_PER_IMAGE * n / (ms * 1e-3) / 1e12
        peak = 157.3 if prec == "fp32" else 2500.0
        kern = ("train-mode encoder forward (conv_igemm_kernel<float> + BN passes)" if prec == "fp32" else
                "train-mode encoder forward (fp16 halo / LDS-DMA conv kernels + fp16 BN passes)")
        rec["roofline"] = {"bound": "mfma", "kernel": kern, "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                           "traffic": None, "launch_ms": ms, "flops_per_launch": FWD_FLOP_PER_IMAGE * n, "images_per_launch": n}
        rec["frac_of_mfma_peak_whole_step"] = rec["tflops"] / world / peak
        if prec == "fp16":
            rec["loss_scale"], rec["skipped_steps"] = trainer.scaler.get_scale(), trainer.scaler.skipped
    return rec


def bench_classifier_step(batch: int, steps: int, warmup: int, dev, precision: str = "fp16") -> dict:
    """The fine-tune step of the classifier loops (SURVEY a-13, src/main.py:494-510 / :575-590): train-mode ResNet18 + fc,
    weighted cross-entropy, backward, GradScaler, Adam -- on ``batch`` images (the reference's BATCH_SIZE is 512), single GPU."""
    from .resnet import ResNet18Classifier

    torch.manual_seed(0)
    trainer = NativeClassifierTrainer(ResNet18Classifier().state_dict(), device=dev, lr=1e-4, class_weights=torch.tensor([1.0, 3.5]),
                                      precision=precision)
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn((batch, 3, 224, 224), generator=g, device=dev)
    y = torch.randint(0, 2, (batch,), generator=torch.Generator().manual_seed(7))
    for _ in range(warmup):
        trainer.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = trainer.step(x, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    imgs = batch * steps
    tf = 3.0 * FWD_FLOP_PER_IMAGE * imgs / dt / 1e12
    return {"metric": "classifier fine-tune step, images/s (ResNet18 fwd+bwd, weighted CE, Adam)", "value": imgs / dt, "unit": "images/s",
            "batch": batch, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "dtype": "f16 (fp32 accumulate, fp32 master weights, GradScaler)" if precision == "fp16" else "f32", "data": "synthetic",
            "final_loss": float(loss), "tflops": tf, "frac_of_mfma_peak_whole_step": tf / (2500.0 if precision == "fp16" else 157.3)}
