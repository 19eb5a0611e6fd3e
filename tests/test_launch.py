"""CPU tests of the product launcher (launch.py), the rank batch sampler and the sharded scoring entry point."""
import os
import subprocess
import sys
import time

import pytest
import torch

from ss25_hierarchical_multiscale_image_classification_amd import dist as hdist, launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cmds(code, n, extra=None):
    return launch.child_commands(["-c", code], [], n, launch.free_port(), extra_env=extra)


def test_child_commands_carry_torchruns_environment():
    cmds = launch.child_commands(["-m", "pkg.main"], ["--patch", "--world_size", "4", "--_child"], 4, 29555, python="py", base_env={"A": "1"})
    assert len(cmds) == 4
    for r, (cmd, env) in enumerate(cmds):
        assert cmd == ["py", "-m", "pkg.main", "--patch", "--world_size", "4", "--_child"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"], env["MASTER_PORT"]) == (str(r), str(r), "4", "127.0.0.1", "29555")
        assert env["A"] == "1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_launcher_relays_rank0_and_returns_zero(capsys):
    code = "import os, sys; print('hello from', os.environ['RANK']); sys.exit(0)"
    assert launch.launch_ranks(_cmds(code, 3)) == 0
    assert capsys.readouterr().out.strip() == "hello from 0"


def test_launcher_terminates_siblings_of_a_failed_rank(capsys):
    """Rank 1 dies at once with code 7; ranks 0 and 2 would sleep for a minute (a rendezvous that never completes): the
    launcher returns 7 within seconds and names the rank."""
    code = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\nprint('alive', flush=True); time.sleep(60)"
    t0 = time.time()
    rc = launch.launch_ranks(_cmds(code, 3))
    assert rc == 7 and time.time() - t0 < 20
    cap = capsys.readouterr()
    assert "rank 1 exited with code 7" in cap.err and "alive" in cap.out


def test_launcher_rank_timeout():
    code = "import time; time.sleep(60)"
    t0 = time.time()
    assert launch.launch_ranks(_cmds(code, 2), rank_timeout=1.0) == 124
    assert time.time() - t0 < 20


def test_rank_batch_sampler_shares_every_global_batch():
    n, B, world = 103, 16, 4
    samplers = [hdist.RankBatchSampler(n, B, r, world, shuffle=True, seed=3) for r in range(world)]
    for epoch in range(2):
        per_rank = [list(s) for s in samplers]
        assert len({len(b) for b in per_rank}) == 1 and len(per_rank[0]) == len(samplers[0])
        seen = []
        for batches in zip(*per_rank):
            assert len({len(b) for b in batches}) == 1  # equal shares (the collectives need that)
            seen += [i for b in batches for i in b]
        assert len(seen) == len(set(seen)) and set(seen) <= set(range(n))
        assert len(seen) >= n - (world - 1)  # only the remainder of the last batch is left out
        if epoch == 0:
            first = seen
    assert seen != first  # a new permutation per epoch, the same one on every rank
    one = hdist.RankBatchSampler(10, 4, 0, 1, shuffle=False)
    assert list(one) == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]


def _score_sharded_rank(out_dir):
    import torch.distributed as dist

    rank, world, _ = hdist.init_from_env("gloo")
    counts = [3, 0, 5, 2, 4]  # unit 1 is empty; with 3 ranks, rank 2 owns unit 2 only

    def score(i):
        n = counts[i]
        f = torch.full((n, 512), float(i)) + torch.arange(n, dtype=torch.float32)[:, None] / 100
        meta = torch.stack([torch.full((n,), 3), torch.arange(n) * 224, torch.full((n,), i), torch.arange(n) % 2], dim=1).to(torch.int32)
        return f, f[:, :2].clone(), meta

    f, l, m = hdist.score_sharded(len(counts), score, rank, world)
    torch.save((f, l, m), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_score_sharded_returns_the_single_process_order(tmp_path, world):
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); "
            f"import test_launch; test_launch._score_sharded_rank({str(tmp_path)!r})")
    assert launch.launch_ranks(_cmds(code, world), rank_timeout=120) == 0
    counts = [3, 0, 5, 2, 4]
    for r in range(world):
        f, l, m = torch.load(tmp_path / f"rank{r}.pt", weights_only=True)
        assert m[:, 4].tolist() == sum(([i] * c for i, c in enumerate(counts)), [])
        assert torch.equal(f[:, 0], torch.cat([float(i) + torch.arange(c, dtype=torch.float32) / 100 for i, c in enumerate(counts)]))
        assert torch.equal(l, f[:, :2]) and m.dtype == torch.int32 and m.shape == (sum(counts), 5)
