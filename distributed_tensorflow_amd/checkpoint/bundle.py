"""ctypes binding of libdtf_ckpt.so: read/write TensorFlow checkpoints -- V2 ("tensor bundle",
TF >= 0.12) and Saver V1 (the single-file format of the TF 0.x the reference script targets)."""
from __future__ import annotations

import ctypes
from ctypes import c_char_p, c_int, c_int64, c_size_t, c_uint32, c_void_p
from pathlib import Path
from typing import Dict

import numpy as np

_LIB = Path(__file__).resolve().parent.parent / "_lib" / "libdtf_ckpt.so"
_lib = None

# tensorflow.DataType enum values (types.proto)
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_INT64, DT_BFLOAT16 = 1, 2, 3, 4, 9, 14
_NP_TO_DT = {np.dtype(np.float32): DT_FLOAT, np.dtype(np.float64): DT_DOUBLE, np.dtype(np.int32): DT_INT32,
             np.dtype(np.uint8): DT_UINT8, np.dtype(np.int64): DT_INT64}
_DT_TO_NP = {v: k for k, v in _NP_TO_DT.items()}
_DT_TO_NP[DT_BFLOAT16] = np.dtype(np.uint16)  # raw bits


class CheckpointError(RuntimeError):
    pass


def _load():
    global _lib
    if _lib is None:
        if not _LIB.exists():
            raise CheckpointError(f"{_LIB} not built: run `python tools/build_native.py`")
        lib = ctypes.CDLL(str(_LIB))
        lib.dtf_crc32c.argtypes = [c_void_p, c_size_t]
        lib.dtf_crc32c.restype = c_uint32
        lib.dtf_crc32c_mask.argtypes = [c_uint32]
        lib.dtf_crc32c_mask.restype = c_uint32
        lib.dtf_crc32c_unmask.argtypes = [c_uint32]
        lib.dtf_crc32c_unmask.restype = c_uint32
        lib.dtf_ckpt_writer_new.argtypes = [c_char_p]
        lib.dtf_ckpt_writer_new.restype = c_void_p
        lib.dtf_ckpt_writer_add.argtypes = [c_void_p, c_char_p, c_int, c_int, ctypes.POINTER(c_int64), c_void_p, c_int64]
        lib.dtf_ckpt_writer_finish.argtypes = [c_void_p]
        lib.dtf_ckpt_writer_finish_v1.argtypes = [c_void_p]
        lib.dtf_ckpt_reader_is_v1.argtypes = [c_void_p]
        lib.dtf_ckpt_writer_error.argtypes = [c_void_p]
        lib.dtf_ckpt_writer_error.restype = c_char_p
        lib.dtf_ckpt_writer_free.argtypes = [c_void_p]
        lib.dtf_ckpt_writer_free.restype = None
        lib.dtf_ckpt_reader_open.argtypes = [c_char_p]
        lib.dtf_ckpt_reader_open.restype = c_void_p
        lib.dtf_ckpt_reader_error.argtypes = [c_void_p]
        lib.dtf_ckpt_reader_error.restype = c_char_p
        lib.dtf_ckpt_reader_num.argtypes = [c_void_p]
        lib.dtf_ckpt_reader_num_shards.argtypes = [c_void_p]
        lib.dtf_ckpt_reader_entry.argtypes = [c_void_p, c_int, c_char_p, c_int, ctypes.POINTER(c_int),
                                              ctypes.POINTER(c_int), ctypes.POINTER(c_int64),
                                              ctypes.POINTER(c_int64), ctypes.POINTER(c_int)]
        lib.dtf_ckpt_reader_read.argtypes = [c_void_p, c_int, c_void_p, c_int64]
        lib.dtf_ckpt_reader_free.argtypes = [c_void_p]
        lib.dtf_ckpt_reader_free.restype = None
        _lib = lib
    return _lib


def crc32c(data: bytes) -> int:
    return int(_load().dtf_crc32c(data, len(data)))


def crc32c_masked(data: bytes) -> int:
    lib = _load()
    return int(lib.dtf_crc32c_mask(lib.dtf_crc32c(data, len(data))))


def write_v1(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """Write the single Saver-V1 file ``<prefix>`` (what ``tf.train.Saver`` produced before TF 0.12)."""
    write_bundle(prefix, tensors, version=1)


def checkpoint_format(prefix: str) -> int:
    """1 or 2: the on-disk format of the checkpoint at ``prefix``."""
    lib = _load()
    r = lib.dtf_ckpt_reader_open(str(prefix).encode())
    try:
        err = lib.dtf_ckpt_reader_error(r).decode()
        if err:
            raise CheckpointError(f"{prefix}: {err}")
        return 1 if lib.dtf_ckpt_reader_is_v1(r) else 2
    finally:
        lib.dtf_ckpt_reader_free(r)


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray], version: int = 2) -> None:
    """Write ``<prefix>.index`` + ``<prefix>.data-00000-of-00001`` (version 2) or the V1 file ``<prefix>``."""
    if version not in (1, 2):
        raise CheckpointError(f"write version {version}: TensorFlow has Saver formats 1 and 2")
    lib = _load()
    w = lib.dtf_ckpt_writer_new(str(prefix).encode())
    try:
        for name, arr in tensors.items():
            a = np.asarray(arr)
            if not a.flags.c_contiguous:  # (np.ascontiguousarray would promote 0-d scalars to 1-d)
                a = a.copy(order="C")
            if a.dtype not in _NP_TO_DT:
                raise CheckpointError(f"unsupported dtype {a.dtype} for {name}")
            dims = (c_int64 * max(1, a.ndim))(*a.shape)
            if lib.dtf_ckpt_writer_add(w, name.encode(), _NP_TO_DT[a.dtype], a.ndim, dims,
                                       a.ctypes.data_as(c_void_p), a.nbytes) != 0:
                raise CheckpointError(lib.dtf_ckpt_writer_error(w).decode())
        if (lib.dtf_ckpt_writer_finish if version == 2 else lib.dtf_ckpt_writer_finish_v1)(w) != 0:
            raise CheckpointError(lib.dtf_ckpt_writer_error(w).decode())
    finally:
        lib.dtf_ckpt_writer_free(w)


def list_bundle(prefix: str) -> Dict[str, tuple]:
    """name -> (numpy dtype, shape) without reading tensor data (tf.train.list_variables)."""
    return {k: (v[0], v[1]) for k, v in _read(prefix, data=False).items()}


def read_bundle(prefix: str) -> Dict[str, np.ndarray]:
    """Read every tensor of a checkpoint (V2 bundle or Saver V1, detected from the files; crc-verified)."""
    return _read(prefix, data=True)


def _read(prefix: str, data: bool):
    lib = _load()
    r = lib.dtf_ckpt_reader_open(str(prefix).encode())
    try:
        err = lib.dtf_ckpt_reader_error(r).decode()
        if err:
            raise CheckpointError(f"{prefix}: {err}")
        out = {}
        name = ctypes.create_string_buffer(1024)
        dtype, ndim, sliced = c_int(), c_int(), c_int()
        dims = (c_int64 * 8)()
        nbytes = c_int64()
        for i in range(lib.dtf_ckpt_reader_num(r)):
            rc = lib.dtf_ckpt_reader_entry(r, i, name, 1024, ctypes.byref(dtype), ctypes.byref(ndim), dims,
                                           ctypes.byref(nbytes), ctypes.byref(sliced))
            key = name.value.decode(errors="replace")
            if rc != 0:
                raise CheckpointError(f"{prefix}: cannot describe entry {i} ({key!r}): "
                                      + ("more than 8 dimensions" if rc == -2 else f"code {rc}"))
            if sliced.value:
                raise CheckpointError(f"{key}: partitioned (sliced) variables are not supported")
            if dtype.value not in _DT_TO_NP:
                raise CheckpointError(f"{key}: unsupported TF dtype enum {dtype.value}")
            npdt = _DT_TO_NP[dtype.value]
            shape = tuple(int(dims[k]) for k in range(ndim.value))
            if not data:
                out[key] = (npdt, shape)
                continue
            try:
                arr = np.empty(shape, dtype=npdt)
            except (MemoryError, ValueError) as e:
                raise CheckpointError(f"{key}: cannot allocate shape {shape}: {e}") from None
            if arr.nbytes != nbytes.value:
                raise CheckpointError(f"{key}: entry size {nbytes.value} != shape bytes {arr.nbytes}")
            rc = lib.dtf_ckpt_reader_read(r, i, arr.ctypes.data_as(c_void_p), arr.nbytes)
            if rc != 0:
                raise CheckpointError(f"{prefix}: {lib.dtf_ckpt_reader_error(r).decode()}")
            out[key] = arr
        return out
    finally:
        lib.dtf_ckpt_reader_free(r)
