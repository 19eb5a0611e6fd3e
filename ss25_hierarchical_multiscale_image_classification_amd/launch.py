"""One process per GPU, started by the product itself.

The reference wraps its model in ``nn.DataParallel`` whenever more than one GPU is visible
(src/main.py:481-482, :563-564, :841-842, :998-999; src/models/simclr.py:77-78): one process, N replicas.
Here ``main.py --world_size N`` (and ``bench.py --gpus N``) starts N fresh processes -- one rank per GPU,
``torch.distributed`` over RCCL -- BEFORE the parent has touched the GPU: a process that has initialised HIP
is never re-executed, the parent only supervises.

``launch_ranks`` polls every child: when one exits non-zero (a failed RCCL init, an out-of-memory kill) the
others -- which would sit in the rendezvous or in a collective until torch's own timeout -- are terminated at
once and that rank's exit code is returned with its number.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence, Tuple

Cmd = Tuple[List[str], Dict[str, str]]


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child_commands(entry: Sequence[str], argv: Sequence[str], n: int, port: int, python: str = sys.executable,
                   base_env: Optional[Dict[str, str]] = None, extra_env: Optional[Dict[str, str]] = None) -> List[Cmd]:
    """(argv, env) of every rank's process: ``python <entry...> <argv> --_child`` with torchrun's environment
    variables and a rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    base_env = dict(os.environ if base_env is None else base_env)
    cmds = []
    for r in range(n):
        env = dict(base_env)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL fails without it on this driver
        if extra_env:
            env.update(extra_env)
        cmds.append(([python, *entry, *[a for a in argv if a != "--_child"], "--_child"], env))
    return cmds


def _terminate(procs, grace_s: float = 5.0) -> None:
    for p in procs:
        if p.poll() is None:
            p.terminate()
    t_end = time.time() + grace_s
    for p in procs:
        while p.poll() is None and time.time() < t_end:
            time.sleep(0.05)
        if p.poll() is None:
            p.kill()


def launch_ranks(cmds: Sequence[Cmd], rank_timeout: Optional[float] = None, poll_s: float = 0.1, name: str = "launcher",
                 out=None) -> int:
    """Start every rank as a fresh child process, relay rank 0's stdout to ``out`` (default sys.stdout) when all are
    done, and return 0 or the exit code of the FIRST rank seen to fail (its siblings are terminated right away).
    ``rank_timeout`` (seconds, None = unlimited): a run that takes longer is terminated and reported as 124."""
    out = sys.stdout if out is None else out
    procs = []
    for r, (cmd, env) in enumerate(cmds):
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks: List[bytes] = []
    drain = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)  # a full pipe would block rank 0
    drain.start()
    t0 = time.time()
    failed: Optional[Tuple[int, int]] = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if rank_timeout is not None and time.time() - t0 > rank_timeout:
            failed = (-1, 124)
            break
        time.sleep(poll_s)
    if failed is not None:
        _terminate(procs)
    drain.join(timeout=10)
    out.write(b"".join(chunks).decode(errors="replace"))
    out.flush()
    if failed is not None:
        r, c = failed
        what = f"rank {r} exited with code {c}" if r >= 0 else f"no result after {rank_timeout:.0f} s"
        print(f"{name}: {what}; the other ranks were terminated (exit codes {[p.poll() for p in procs]})", file=sys.stderr)
        return c if c > 0 else 1
    return 0
