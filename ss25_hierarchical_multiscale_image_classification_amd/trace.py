"""Optional roctx ranges around the phases of the hot path (off unless HIPAC_ROCTX=1).

`rocprofv3 --marker-trace --kernel-trace -- python3 ...` then shows which launches belong to the window decisions of a
level, to the gather and to the ResNet forward of a slide scan.  The ranges are host-side markers only: nothing on the
device path changes, and without the environment variable `span()` is a no-op that never loads the library.
"""
from __future__ import annotations

import contextlib
import ctypes
import os

_LIB = None
_TRIED = False


def _lib():
    global _LIB, _TRIED
    if not _TRIED:
        _TRIED = True
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _LIB = lib
                break
            except (OSError, AttributeError):
                continue
    return _LIB


def enabled() -> bool:
    return os.environ.get("HIPAC_ROCTX", "0") == "1"


@contextlib.contextmanager
def span(name: str):
    """roctxRangePush(name) ... roctxRangePop() when HIPAC_ROCTX=1 and a roctx library is present; else nothing."""
    lib = _lib() if enabled() else None
    if lib is None:
        yield
        return
    lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        lib.roctxRangePop()
