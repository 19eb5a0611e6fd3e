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
import torch.utils.data


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


def active(group=None) -> bool:
    """True when the collectives of this module have to run: a process group with more than one rank -- or with ONE rank
    and ``HIPAC_DIST_FORCE=1`` (test hook: the one-GPU box has no second device for RCCL, so the single-rank nccl test
    drives every exchange of the product path through RCCL itself instead of returning early)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("HIPAC_DIST_FORCE") == "1"


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


def exchange_counts(n: int, device, group=None) -> List[int]:
    """Every rank's row count (one small all-gather + one host synchronisation)."""
    world = dist.get_world_size(group)
    if torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo":
        device = "cpu"
    t = torch.tensor([n], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(counts, t, group=group)
    return [int(c.item()) for c in counts]


def all_gather_rows(t: torch.Tensor, group=None, counts: Optional[List[int]] = None) -> torch.Tensor:
    """All-gather-v along dim 0 (ranks may hold different row counts): one int64 count
    exchange (skipped when the caller already holds ``counts``), then one all-gather on a max-padded
    buffer, trimmed and concatenated in rank order."""
    if not active(group):
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":  # gloo has no CUDA all_gather: stage through the host
        return all_gather_rows(t.cpu(), group, counts).to(t.device)
    world = dist.get_world_size(group)
    if counts is None:
        counts = exchange_counts(t.shape[0], t.device, group)
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
    if not active(group):
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        return all_gather_equal(t.cpu(), group).to(t.device)
    world = dist.get_world_size(group)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out


def gather_results(feats: torch.Tensor, logits: Optional[torch.Tensor], meta: torch.Tensor, group=None):
    """Collect every rank's per-patch rows on every rank (rank-major order): ONE count exchange for the three
    tensors (they have the same rows), then one padded all-gather each."""
    if not active(group):
        return feats, logits, meta
    counts = exchange_counts(feats.shape[0], feats.device, group)
    return (all_gather_rows(feats, group, counts), all_gather_rows(logits, group, counts) if logits is not None else None,
            all_gather_rows(meta, group, counts))


def rank_world() -> Tuple[int, int]:
    """(rank, world) of the initialised process group, (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def score_sharded(n_units: int, score_fn, rank: Optional[int] = None, world: Optional[int] = None, group=None):
    """BASELINE configs[3]: units (slides) sharded over the ranks (unit i -> rank i mod world), each scored by its
    owner with ``score_fn(i) -> (feats[n_i,512], logits[n_i,C] or None, meta int32[n_i,4])``, one ragged all-gather of
    every rank's rows, and the rows put back into UNIT order (a stable sort on the unit index, which travels as a fifth
    meta column): every rank returns exactly what a single process scoring units 0, 1, 2, ... would have produced --
    the order nn.DataParallel's gather gives the reference (src/main.py:841-842, :870).
    Returns (feats, logits or None, meta int32[N,5] = level, x, y, label, unit)."""
    if rank is None or world is None:
        rank, world = rank_world()
    fs, ls, ms = [], [], []
    dev, n_cls = None, None
    for i in shard_units(n_units, rank, world):
        f, l, m = score_fn(i)
        dev = f.device
        fs.append(f)
        if l is not None:
            ls.append(l)
            n_cls = l.shape[1]
        ms.append(torch.cat([m.to(torch.int32), torch.full((m.shape[0], 1), i, dtype=torch.int32, device=m.device)], dim=1))
    if dev is None:  # a rank without units still takes part in the exchange (as many columns as the others, zero rows)
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    has_logits = torch.tensor([0 if n_cls is None else n_cls], dtype=torch.int64)
    if active(group):
        lst = [torch.zeros_like(has_logits) for _ in range(world)]
        dist.all_gather(lst, has_logits.to(dev) if dist.get_backend(group) == "nccl" else has_logits, group=group)
        n_cls = max(int(t.item()) for t in lst) or None
    feats = torch.cat(fs) if fs else torch.zeros((0, 512), dtype=torch.float32, device=dev)
    logits = (torch.cat(ls) if ls else torch.zeros((0, n_cls), dtype=torch.float32, device=dev)) if n_cls else None
    meta = torch.cat(ms) if ms else torch.zeros((0, 5), dtype=torch.int32, device=dev)
    if active(group):
        feats, logits, meta = gather_results(feats, logits, meta, group)
        order = torch.argsort(meta[:, 4], stable=True)
        feats, meta = feats[order], meta[order]
        logits = logits[order] if logits is not None else None
    return feats, logits, meta


def broadcast0(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place broadcast of rank 0's tensor (nn.DataParallel replicates the module of device 0 before every forward)."""
    if active(group):
        if t.is_cuda and dist.get_backend(group) == "gloo":
            h = t.cpu()
            dist.broadcast(h, 0, group=group)
            t.copy_(h)
        else:
            dist.broadcast(t, 0, group=group)
    return t


def all_reduce_sum_scalars(values: Sequence[float], device=None) -> List[float]:
    """Sum of a few host numbers over the ranks (loss and accuracy counters of the training loops)."""
    if not active():
        return list(values)
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor(list(values), dtype=torch.float64, device=device if on_gpu else "cpu")
    dist.all_reduce(t)
    return [float(v) for v in t.cpu()]


class RankBatchSampler(torch.utils.data.Sampler):
    """Batch sampler of process-per-GPU training: every rank draws the SAME (seeded) permutation, cuts it into global
    batches of ``batch_size`` and keeps its own contiguous share of each -- the chunk nn.DataParallel's scatter would
    hand to replica ``rank`` (src/main.py:481-482).  Shares are equal on all ranks (the collectives need that), so up to
    ``world - 1`` samples of a batch that does not divide are left out of that batch.  world == 1: plain batches."""

    def __init__(self, n: int, batch_size: int, rank: int = 0, world: int = 1, shuffle: bool = True, seed: int = 0):
        self.n, self.batch_size, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, batch_size, rank, world, shuffle, seed, 0

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def _batches(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            perm = torch.randperm(self.n, generator=g).tolist()
        else:
            perm = list(range(self.n))
        for b0 in range(0, self.n, self.batch_size):
            idx = perm[b0:b0 + self.batch_size]
            per = len(idx) // self.world
            if per > 0:
                yield idx[self.rank * per:(self.rank + 1) * per]

    def __iter__(self):
        # the permutation of THIS epoch is drawn now (a generator would read self.epoch only at its first next(), i.e.
        # after the increment below): set_epoch(e) yields epoch e's order, as the device loaders do
        batches = list(self._batches())
        self.epoch += 1  # a new permutation per pass, the same one on every rank
        return iter(batches)

    def __len__(self):
        full, rem = divmod(self.n, self.batch_size)
        return full + (1 if rem // self.world > 0 else 0)


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
