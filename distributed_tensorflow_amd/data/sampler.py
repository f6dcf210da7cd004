"""Host mirror of the device batch sampler (csrc/kernels/common.hpp::feistel_permute).

The reference draws batches with ``mnist.train.next_batch(B)`` -- a host-side
epoch shuffle (reference .idea/MNISTDist.py:178).  On MI355X the dataset never
leaves HBM, so a batch is just a vector of row indices produced on device by a
keyed 4-round Feistel bijection over [0, n) (cycle walking).  This module
computes the identical indices with numpy so CPU workers, tests and the oracle
see exactly the batches the GPU trains on.
"""
from __future__ import annotations

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def half_bits(n: int) -> int:
    hb = 1
    while (1 << (2 * hb)) < n:
        hb += 1
    return hb


def feistel_permute(i, n: int, key: int) -> np.ndarray:
    """Bijection on [0, n): position -> dataset row."""
    hb = np.uint64(half_bits(n))
    mask = (np.uint64(1) << hb) - np.uint64(1)
    i = np.atleast_1d(np.asarray(i, dtype=np.uint64)).copy()
    key = np.uint64(key & 0xFFFFFFFF)
    todo = np.ones(i.shape, dtype=bool)
    while todo.any():
        cur = i[todo]
        L, R = cur >> hb, cur & mask
        for r in range(4):
            rk = np.uint64((r * 0x9E3779B9) & 0xFFFFFFFF)
            t = L ^ (_mix32(R ^ key ^ rk) & mask)
            L, R = R, t
        cur = (L << hb) | R
        i[todo] = cur
        todo[todo] = cur >= np.uint64(n)
    return i.astype(np.int64)


def epoch_key(seed: int, epoch: int) -> int:
    return int(_mix32(np.asarray([(seed + epoch * 0x9E3779B9) & 0xFFFFFFFF], dtype=np.uint64))[0])


def batch_indices(cursor: int, batch: int, n: int, seed: int, shuffle: bool = True) -> np.ndarray:
    """Dataset rows of the batch starting at global sample number ``cursor``."""
    s = cursor + np.arange(batch, dtype=np.int64)
    epoch, pos = s // n, s % n
    if not shuffle:
        return pos
    out = np.empty(batch, dtype=np.int64)
    for e in np.unique(epoch):
        m = epoch == e
        out[m] = feistel_permute(pos[m], n, epoch_key(seed & 0xFFFFFFFF, int(e)))
    return out
