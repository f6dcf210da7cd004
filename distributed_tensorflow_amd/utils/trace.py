"""Optional roctx ranges for rocprofv3 timelines (``--marker-trace``).

The reference has no tracing at all (``time`` is imported and never used, reference
.idea/MNISTDist.py:8).  Here the host-side phases of a step -- native executor calls,
collective enqueues, display, checkpoint, evaluation -- can be bracketed with roctx ranges so that
a rocprofv3 marker trace lines them up with the kernels they enqueue:

    DTF_ROCTX=1 rocprofv3 --marker-trace --kernel-trace -d out -- python3 mnistdist.py --trace ...

Disabled (the default) every entry point returns after one global read; the roctx library is
only loaded when tracing is switched on, and a missing library downgrades to disabled with a
single warning instead of failing the run.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import sys

_LIBS = ("librocprofiler-sdk-roctx.so", "libroctx64.so")   # rocprofv3's own roctx first, then roctracer's
_enabled = os.environ.get("DTF_ROCTX", "0") not in ("", "0")
_lib = None
_depth = 0


def _load():
    global _lib, _enabled
    if _lib is not None:
        return _lib
    for name in _LIBS:
        for cand in (name, os.path.join("/opt/rocm/lib", name)):
            try:
                lib = ctypes.CDLL(cand)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                lib.roctxMarkA.argtypes = [ctypes.c_char_p]
                _lib = lib
                return lib
            except (OSError, AttributeError):
                continue
    print("[trace] no roctx library found (looked for %s): tracing disabled" % ", ".join(_LIBS), file=sys.stderr)
    _enabled = False
    return None


def set_enabled(on: bool = True) -> bool:
    """Switch tracing on/off; returns whether it is on (False if the roctx library is missing)."""
    global _enabled
    _enabled = bool(on)
    if _enabled:
        _load()
    return _enabled


def enabled() -> bool:
    return _enabled and _load() is not None


def push(name: str) -> None:
    global _depth
    if _enabled:
        lib = _load()
        if lib is not None:
            lib.roctxRangePushA(name.encode())
            _depth += 1


def pop() -> None:
    global _depth
    if _enabled and _lib is not None and _depth > 0:
        _lib.roctxRangePop()
        _depth -= 1


def mark(name: str) -> None:
    if _enabled:
        lib = _load()
        if lib is not None:
            lib.roctxMarkA(name.encode())


@contextlib.contextmanager
def range(name: str):   # noqa: A001 - mirrors roctx's vocabulary
    push(name)
    try:
        yield
    finally:
        pop()
