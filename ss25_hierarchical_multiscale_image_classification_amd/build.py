"""Build libhipac_hip.so (gfx950) in-tree with hipcc.

Cross-compiles without a GPU.  The .so stays inside the package directory so it
travels with the tree (it is git-ignored, not gpurun-ignored).

    python -m ss25_hierarchical_multiscale_image_classification_amd.build [--force] [--asan]

``--asan`` builds ``libhipac_hip_asan.so``: the HOST side of every translation unit (argument checks, workspace
plans, weight packing, Pillow coefficient tables, launch bookkeeping) under AddressSanitizer
(``-fsanitize=address -fno-gpu-sanitize``; device code is compiled as usual -- GPU ASan is not available on
this pool).  tests/test_host_asan.py drives its host-only entry points under the ASan runtime.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = CSRC / os.environ.get("HIPAC_OBJ_DIR", "build")
LIB = PKG / os.environ.get("HIPAC_LIB_NAME", "libhipac_hip.so")
SOURCES = ["hipac_capi.hip", "preprocess.hip", "level_planes.hip", "mil.hip", "ntxent.hip", "conv_bf16.hip", "conv_f16.hip",
           "conv_f32.hip", "conv_f16x3.hip", "conv_f16q8.hip", "train.hip", "train_amp.hip", "augment.hip", "jpeg_decode.hip"]
HEADERS = ["common.h", "conv_igemm.h", "block_c64.h", "halo16.h", "halo16x2.h", "e4m3.h", "band16.h", "block16_c64.h", "train_common.h", "../../include/hipac.h"]
ARCH = "gfx950"
# -ffp-contract=off: the host-side Pillow coefficient restatement must round every
# double operation separately (no fused multiply-add), see preprocess.hip.
EXTRA = os.environ.get("HIPAC_EXTRA_FLAGS", "").split()
FLAGS = [*EXTRA, "-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> Path:
    OBJ.mkdir(exist_ok=True)
    hipcc = _hipcc()
    hdrs = [CSRC / h for h in HEADERS] + [Path(__file__)]
    jobs = []
    for src in SOURCES:
        s = CSRC / src
        o = OBJ / (Path(src).stem + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc, *FLAGS, "-c", str(s), "-o", str(o)]
        if verbose:
            print("[hipac build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s.name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(compile_one, jobs))
    objs = [OBJ / (Path(src).stem + ".o") for src in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print("[hipac build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def asan_runtime() -> Path:
    """The clang ASan runtime a non-instrumented python must preload to dlopen the --asan build."""
    cands = sorted(Path("/opt/rocm/lib/llvm/lib/clang").glob("*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not cands:
        raise RuntimeError("libclang_rt.asan-x86_64.so not found under /opt/rocm/lib/llvm")
    return cands[-1]


def build_asan(verbose: bool = False) -> Path:
    """Host-ASan variant, in its own object directory and library name (a child process: the flags are module state)."""
    env = dict(os.environ, HIPAC_LIB_NAME="libhipac_hip_asan.so", HIPAC_OBJ_DIR="build_asan",
               HIPAC_EXTRA_FLAGS="-fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -g")
    r = subprocess.run([sys.executable, str(Path(__file__)), *([] if verbose else ["--quiet"])], env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"asan build failed:\n{r.stdout}\n{r.stderr}")
    return PKG / "libhipac_hip_asan.so"


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_asan(verbose=True))
    else:
        print(build_library(force="--force" in sys.argv, verbose="--quiet" not in sys.argv))
