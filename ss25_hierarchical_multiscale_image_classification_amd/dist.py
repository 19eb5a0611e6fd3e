"""Multi-GPU plumbing: one process per GPU, slides sharded over ranks, one exchange at
the end.

The reference's only multi-GPU construct is single-process ``nn.DataParallel``
(src/main.py:481-482, :841-842, :998-999; src/models/simclr.py:77-78), whose observable
semantics for inference are "outputs concatenated on dim 0 in input order".  Here the
units (slides, or contiguous column ranges of one slide) are independent, so the data
path needs no collective; ``gather_results`` is the single all-gather that reproduces
the DataParallel gather: rank-major concatenation == single-process order.

Backend "nccl" is RCCL on ROCm (xGMI between the 8 GPUs of a node); "gloo" is used by
the CPU tests.  Payloads are tiny (<= ~2 KB per patch), so the exchange is one
count all-gather plus one padded all-gather per tensor -- latency bound, not bandwidth
bound.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; initialises the
    process group when WORLD_SIZE > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_units(n_units: int, rank: int, world: int) -> List[int]:
    """Slide i -> rank i mod world."""
    return list(range(rank, n_units, world))


def shard_columns(n_cols: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous x-column range [c0, c1) of one slide for this rank: with the
    reference's x-outer loop order, rank-major concatenation of the ranks' results is
    exactly the single-process order."""
    base, rem = divmod(n_cols, world)
    c0 = rank * base + min(rank, rem)
    return c0, c0 + base + (1 if rank < rem else 0)


def all_gather_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather-v along dim 0 (ranks may hold different row counts): one int64 count
    exchange, then one all-gather on a max-padded buffer, trimmed and concatenated in
    rank order."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":  # gloo has no CUDA all_gather: stage through the host
        return all_gather_rows(t.cpu(), group).to(t.device)
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    if n_max == 0:
        return t
    pad = torch.zeros((n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def all_gather_equal(t: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather along dim 0 when every rank holds the SAME number of rows (the benchmark's equal patch
    shards): one collective into one preallocated tensor, no count exchange, no host synchronisation."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        return all_gather_equal(t.cpu(), group).to(t.device)
    world = dist.get_world_size(group)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out


def gather_results(feats: torch.Tensor, logits: Optional[torch.Tensor], meta: torch.Tensor, group=None):
    """Collect every rank's per-patch rows on every rank (rank-major order)."""
    return (all_gather_rows(feats, group), all_gather_rows(logits, group) if logits is not None else None,
            all_gather_rows(meta, group))


class _GatherWithGrad(torch.autograd.Function):
    """all-gather whose backward returns this rank's slice of the summed gradient --
    lets NT-Xent see the global batch under process-per-GPU training."""

    @staticmethod
    def forward(ctx, x):
        world = dist.get_world_size()
        out = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(out, x.contiguous())
        ctx.n = x.shape[0]
        return torch.cat(out, dim=0)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dist.all_reduce(g)
        r = dist.get_rank()
        return g[r * ctx.n : (r + 1) * ctx.n]


def gather_with_grad(x: torch.Tensor) -> torch.Tensor:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    return _GatherWithGrad.apply(x)
