"""Synthetic MNIST-shaped data (28x28x1 in [0,1], 10 classes).

Integer-hash only, so it is bit-identical to the device generator
(csrc/kernels/data.hip::gen_synthetic) and every pixel is exactly representable
in bf16.  Each class has a fixed 7x7 binary template (4x4-pixel cells); "on"
cells are bright noise, "off" cells dim noise, so the task is learnable and
loss/accuracy curves are meaningful without any dataset download.
"""
from __future__ import annotations

import numpy as np

from .sampler import _mix32

_M32 = np.uint64(0xFFFFFFFF)


def synthetic_mnist(n: int, seed: int = 1234):
    """-> (images float32 [n,784], labels int32 [n])"""
    seed = np.uint64(seed & 0xFFFFFFFF)
    i = np.arange(n, dtype=np.uint64)
    labels = (_mix32(seed ^ ((i * np.uint64(0x85EBCA6B)) & _M32)) % np.uint64(10)).astype(np.uint64)
    p = np.arange(784, dtype=np.uint64)
    cell = (p // np.uint64(28) >> np.uint64(2)) * np.uint64(7) + (p % np.uint64(28) >> np.uint64(2))
    tb = _mix32(np.uint64(0xC0FFEE) + labels[:, None] * np.uint64(49) + cell[None, :]) & np.uint64(1)
    t = (i[:, None] * np.uint64(784) + p[None, :]) & _M32
    u6 = _mix32(((seed * np.uint64(0x9E3779B9)) & _M32) + ((t * np.uint64(0x27D4EB2F)) & _M32)
                + np.uint64(0x165667B1)) >> np.uint64(26)
    u6 = u6.astype(np.float32)
    images = np.where(tb == 1, (64.0 + u6) / 128.0, u6 / 256.0).astype(np.float32)
    return images, labels.astype(np.int32)
