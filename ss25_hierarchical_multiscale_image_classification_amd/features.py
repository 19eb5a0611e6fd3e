"""Feature extraction pipelines (mirror of src/main.py:805-894 and :897-932).

Two entry points produce the same three files the reference writes
(``patch_features_{L}.npy`` float32 (N,512), ``patch_labels_{L}.npy`` int64 (N,),
``patch_paths_{L}.txt``, src/main.py:885-893):

  * ``extract_features_from_pngs`` -- the reference's flow: a PNG patch tree read
    through ``PatchDataset``; PIL only decodes, Resize/ToTensor/Normalize and the
    ResNet18 run on the GPU.
  * ``extract_features_from_slide`` -- the fused flow: windows are cut, filtered,
    resized and scored straight from the slide in HBM; no PNG round trip (PNG is
    lossless, so the features are the same).
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import capi
from .extract import DeviceSlide, LABEL_NAMES, score_slide
from .patch_dataset import PatchDataset


def _collate_raw(items):
    imgs, labels, paths = zip(*items)
    return list(imgs), torch.tensor(labels, dtype=torch.int64), list(paths)


@torch.no_grad()
def extract_features_from_pngs(patch_dir: str, net: capi.PackedResNet18, batch_size: int = 512,
                               num_workers: int = 8, device: str = "cuda"):
    """(features float32[N,512] (cpu), labels int64[N], paths list[str])."""
    ds = PatchDataset(patch_dir, raw=True)
    loader = torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=False, num_workers=num_workers,
                                         collate_fn=_collate_raw)
    dev = torch.device(device)
    feats, labels, paths = [], [], []
    for imgs, lbls, pths in loader:
        # group by patch size: one preprocess launch per size present in the batch
        out = torch.empty((len(imgs), capi.PATCH, capi.PATCH, 3), dtype=torch.uint8, device=dev)
        sizes = sorted({int(i.shape[0]) for i in imgs})
        for P in sizes:
            sel = [k for k, i in enumerate(imgs) if int(i.shape[0]) == P]
            if any(tuple(imgs[k].shape) != (P, P, 3) for k in sel):
                raise capi.HipacError("patches must be square RGB")
            stack = torch.stack([imgs[k] for k in sel]).to(dev)  # [m,P,P,3]
            m = len(sel)
            if P % 16:
                raise capi.HipacError(f"patch size {P} is not a multiple of 16")
            xy = torch.stack([torch.zeros(m, dtype=torch.int32), torch.arange(m, dtype=torch.int32) * P], 1).to(dev)
            o, _, _ = capi.tile_preprocess(stack.view(m * P, P, 3), xy, P, "u8", want_sums=False)
            out[torch.tensor(sel, device=dev)] = o
        f, _, _ = net.forward(out)  # uint8 in: ToTensor/Normalize fused into the stem kernel
        feats.append(f.cpu())
        labels.extend(lbls.tolist())
        paths.extend(pths)
    if not feats:
        return torch.empty((0, 512)), np.zeros((0,), np.int64), []
    return torch.cat(feats), np.array(labels), paths


def extract_features_with_simclr(patch_dir: str, encoder_path: str = "simclr_encoder.pth", precision: str = "bf16",
                                 batch_size: int = 512, device: str = "cuda"):
    """src/main.py:897-932: the PNG-tree flow with ``UnifiedResNet(encoder_path, classifier=False)``, i.e. a SimCLR
    checkpoint's ``encoder.*`` tensors (fc = Identity -> [N,512]); the projector is not used."""
    from .weights import canonical_state_dict

    sd = canonical_state_dict(torch.load(encoder_path, map_location="cpu", weights_only=True))
    sd = {k: v for k, v in sd.items() if not k.startswith(("fc.", "projector."))}
    return extract_features_from_pngs(patch_dir, capi.PackedResNet18(sd, precision=precision), batch_size=batch_size,
                                      device=device)


def save_feature_files(level: int, feats, labels, paths, out_dir: str = "."):
    """src/main.py:885-893."""
    np.save(os.path.join(out_dir, f"patch_features_{level}.npy"), np.asarray(feats, dtype=np.float32))
    np.save(os.path.join(out_dir, f"patch_labels_{level}.npy"), np.asarray(labels))
    with open(os.path.join(out_dir, f"patch_paths_{level}.txt"), "w") as f:
        for p in paths:
            f.write(f"{p}\n")


def save_froc_csv(path: str, logits, meta, level_downsamples, tumor_class: int = 1) -> int:
    """Per-slide detection list in the format ``readCSVContent`` parses
    (src/utils/evaluation_FROC.py:67-88): one line ``probability,x,y`` per scored window with
    float probability and INTEGER level-0 coordinates (the evaluation indexes its mask with
    ``int(x / 2**level)``, :131-132).  ``logits`` [n,C] and ``meta`` int32[n,4] = (level, x, y, label)
    as returned by ``extract.score_slide``; the detection point is the window centre.  Returns the
    number of lines written."""
    from .extract import PATCH_SIZES

    lg = torch.as_tensor(logits).float().cpu()
    m = torch.as_tensor(meta).cpu().numpy()
    prob = torch.softmax(lg, dim=1)[:, tumor_class].numpy() if lg.numel() else np.zeros((0,), np.float32)
    with open(path, "w") as f:
        for p_, (lvl, x, y, _) in zip(prob, m):
            ds = float(level_downsamples[int(lvl)])
            half = PATCH_SIZES.get(int(lvl), 224) / 2.0
            f.write(f"{float(p_):.6f},{int((x + half) * ds)},{int((y + half) * ds)}\n")
    return len(m)


@torch.no_grad()
def extract_features_from_slide(slide: DeviceSlide, net: capi.PackedResNet18, level: int,
                                stride: Optional[int] = None, batch_windows: int = 512):
    """Fused equivalent for one slide and level; ``paths`` are the names the
    reference's extractor would have written for the kept windows."""
    feats, _, _, meta = score_slide(slide, net, levels=(level,), batch_windows=batch_windows, stride=stride,
                                    want_logits=False)
    m = meta.cpu().numpy()
    paths = [f"{slide.name}/{slide.name}_x{x}_y{y}_{LABEL_NAMES[int(l)]}.png" for _, x, y, l in m]
    return feats.cpu(), m[:, 3].astype(np.int64), paths
