"""ctypes binding of libdtf_kernels.so (hand-written gfx950 HIP kernels).

The library is built in-tree by tools/build_native.py (``hipcc
--offload-arch=gfx950``) and exposes a plain C ABI that takes raw device
pointers plus the HIP stream to launch on, so a launch costs one foreign call
and kernels enqueue on whatever stream PyTorch has current (HIP-graph capture
included).  There is deliberately no eager/PyTorch fallback for these ops: on a
GPU box a missing library is an error, not a silent slow path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_long, c_uint, c_uint64, c_void_p
from pathlib import Path
from typing import Optional

_LIB_DIR = Path(__file__).resolve().parent.parent / "_lib"
_LIB_NAME = "libdtf_kernels.so"
_lib: Optional[ctypes.CDLL] = None


class NativeLibraryMissing(RuntimeError):
    pass


_SIGS = {
    # name: argtypes
    "dtf_conv1_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_conv1_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_conv2_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_conv2_dgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_conv2_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "dtf_gemm_nt": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                    c_float, c_uint64, c_void_p, c_int, c_void_p],
    "dtf_gemm_tn": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "dtf_transpose_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_head_fwd_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                         c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_int, c_void_p],
    "dtf_head_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_sgd_update": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_long, c_long, c_void_p],
    "dtf_adam_update": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_long,
                        c_long, c_void_p],
    "dtf_advance_state": [c_void_p, c_long, c_long, c_void_p],
    "dtf_cast_bf16": [c_void_p, c_void_p, c_long, c_void_p],
    "dtf_reduce_segments": [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_int),
                            ctypes.POINTER(c_int), ctypes.POINTER(c_long), c_void_p],
    "dtf_sample_indices": [c_void_p, c_void_p, c_int, c_uint, c_uint, c_uint, c_int, c_void_p],
    "dtf_gen_synthetic": [c_void_p, c_void_p, c_long, c_uint, c_void_p],
    "dtf_init_truncated_normal": [c_void_p, c_long, c_float, c_uint64, c_uint64, c_void_p],
    "dtf_fill_f32": [c_void_p, c_long, c_float, c_void_p],
    "dtf_u8_to_bf16_unit": [c_void_p, c_void_p, c_long, c_void_p],
}


def lib_path() -> Path:
    return _LIB_DIR / _LIB_NAME


def available() -> bool:
    return lib_path().exists()


def load() -> ctypes.CDLL:
    """Load (once) and return the kernel library; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not p.exists():
        raise NativeLibraryMissing(
            f"{p} not found: run `python tools/build_native.py` (hipcc --offload-arch=gfx950) first")
    lib = ctypes.CDLL(str(p), mode=os.RTLD_GLOBAL if hasattr(os, "RTLD_GLOBAL") else ctypes.DEFAULT_MODE)
    for name, argtypes in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    _lib = lib
    return lib


def ptr(t) -> Optional[int]:
    """Raw device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: code {rc}"
                           + (" (unsupported shape/config)" if rc == -1 else
                              " (bad argument)" if rc in (-2, -3) else " (HIP launch error)"))
