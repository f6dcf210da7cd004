"""Plain-PyTorch reference ("oracle") implementations of every op, in TF layouts.

These are the numerics ground truth for the HIP kernels (tests compare against
them in fp32) and the compute path of the CPU backend used by the ps/worker
plumbing configuration.  Layouts are the reference's (reference
.idea/MNISTDist.py:52-90): activations NHWC, filters HWIO, dense weights
[in, out], flatten in HWC order.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def conv2d_same(x_nhwc: torch.Tensor, w_hwio: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """tf.nn.conv2d(strides 1, SAME) + bias_add + relu  (reference M:52-56)."""
    x = x_nhwc.permute(0, 3, 1, 2)
    w = w_hwio.permute(3, 2, 0, 1)
    pad = w_hwio.shape[0] // 2
    y = F.conv2d(x, w, b, stride=1, padding=pad)
    return F.relu(y).permute(0, 2, 3, 1)


def maxpool2d(x_nhwc: torch.Tensor, k: int = 2) -> torch.Tensor:
    """tf.nn.max_pool(ksize k, stride k, SAME) on even-sized maps  (reference M:59-62)."""
    return F.max_pool2d(x_nhwc.permute(0, 3, 1, 2), k, k).permute(0, 2, 3, 1)


def dropout(x: torch.Tensor, keep_prob: float, generator=None) -> torch.Tensor:
    """tf.nn.dropout: x / keep * floor(keep + U[0,1))  (reference M:86)."""
    if keep_prob >= 1.0:
        return x
    u = torch.rand(x.shape, generator=generator, device=x.device, dtype=x.dtype)
    return x / keep_prob * torch.floor(keep_prob + u)


def conv_net(x: torch.Tensor, p: dict, keep_prob: float = 1.0, generator=None, mask: torch.Tensor = None) -> torch.Tensor:
    """The reference model function (reference M:66-90).  x: [B,784] -> logits [B,classes].
    ``mask``: an explicit [B, hidden] 0/1 dropout keep mask instead of drawing one (tests feed the engine's)."""
    x = x.reshape(-1, 28, 28, 1)
    c1 = maxpool2d(conv2d_same(x, p["wc1"], p["bc1"]))
    c2 = maxpool2d(conv2d_same(c1, p["wc2"], p["bc2"]))
    flat = c2.reshape(c2.shape[0], -1)  # NHWC contiguous -> HWC flatten, as tf.reshape does
    h1 = F.relu(flat @ p["wd1"] + p["bd1"])
    if mask is not None and keep_prob < 1.0:
        h1 = h1 * (mask.to(h1.dtype) * (1.0 / keep_prob))
    else:
        h1 = dropout(h1, keep_prob, generator)
    return h1 @ p["out_w"] + p["out_b"]


def softmax_cross_entropy_with_logits(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """Per-row loss.  labels: dense [B,C] distribution (one-hot in the reference) or int [B]."""
    logp = F.log_softmax(logits.float(), dim=-1)
    if labels.dtype in (torch.int32, torch.int64):
        return -logp.gather(1, labels.long().unsqueeze(1)).squeeze(1)
    return -(labels.float() * logp).sum(-1)


def accuracy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """mean(argmax(pred) == argmax(y))  (reference M:152-153)."""
    if labels.dtype not in (torch.int32, torch.int64):
        labels = labels.argmax(-1)
    return (logits.argmax(-1) == labels.long()).float().mean()


def truncated_normal(shape, stddev: float, generator=None, device="cpu") -> torch.Tensor:
    """tf.truncated_normal: N(0, stddev) with |z| > 2 sigma redrawn  (reference M:42-44)."""
    out = torch.empty(shape, device=device)
    torch.nn.init.trunc_normal_(out, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=generator)
    return out * stddev


def pool_argmax(x_nhwc: torch.Tensor):
    """2x2/2 max pool returning (pooled, argmax q = 2*dy+dx, first max wins)."""
    B, H, W, C = x_nhwc.shape
    win = x_nhwc.reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(B, H // 2, W // 2, C, 4)
    pooled, arg = win.max(dim=-1)
    # torch.max returns the first maximal index on CPU; enforce it explicitly for ties
    first = (win == pooled.unsqueeze(-1)).float().argmax(dim=-1)
    return pooled, first.to(torch.uint8)


def conv2_pack_reference(w):
    """Fragment-ordered packs of a [25, cin, cout] (HWIO) conv2 weight, as csrc/kernels/conv2.hip
    (pack_conv2_weights) lays them out: returns (fwd_pack, dgrad_pack), each flat with 25*cin*cout
    elements.  fwd: [tap][ci/32][co/16][(ci%32)/8][co%16][ci%8]; dgrad: [tap][co/32][ci/16][(co%32)/8][ci%16][co%8]."""
    t, cin, cout = w.shape
    f = w.reshape(t, cin // 32, 4, 8, cout // 16, 16).permute(0, 1, 4, 2, 5, 3).contiguous().reshape(-1)
    d = w.reshape(t, cin // 16, 16, cout // 32, 4, 8).permute(0, 3, 1, 4, 2, 5).contiguous().reshape(-1)
    return f, d


def conv2_pack_reference_f32(w):
    """Fragment-ordered fp32 packs of a [25, cin, cout] conv2 weight as csrc/kernels/conv2_f32.hip
    (pack_conv2_weights_f32) lays them out: (fwd_pack, dgrad_pack), each flat with 25*cin*cout elements.
    fwd: [tap][ci/16][co/16][(ci%16)/4][co%16][ci%4]; dgrad: [tap][co/16][ci/16][(co%16)/4][ci%16][co%4]."""
    t, cin, cout = w.shape
    f = w.reshape(t, cin // 16, 4, 4, cout // 16, 16).permute(0, 1, 4, 2, 5, 3).contiguous().reshape(-1)
    d = w.reshape(t, cin // 16, 16, cout // 16, 4, 4).permute(0, 3, 1, 4, 2, 5).contiguous().reshape(-1)
    return f, d


# ---------------------------------------------------------------------------------------------
# bf16-emulated oracle: the fp32 model with a round-to-bf16 at exactly the points where the mixed-precision engine
# stores a bf16 tensor -- p1, p2, h1 on the way forward; dp1, dflat (= d p2), da (= d h1) on the way back -- and the
# bf16 compute copies of wc1 / wc2 / wd1 (conv1 converts wc1 when it loads its MFMA fragments; out_w and all biases
# are read in fp32 by the kernels).  What is left between
# this and the engine is accumulation order (fp32) and the odd value that lands on the other side of a bf16 rounding
# boundary because of it.  Tests use it to DERIVE the engine-vs-fp32 tolerance instead of picking one.
# ---------------------------------------------------------------------------------------------
class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)      # the engine stores this gradient as bf16 too


def round_bf16(x: torch.Tensor) -> torch.Tensor:
    """bf16 rounding of an activation, forward and backward (see _RoundBF16)."""
    return _RoundBF16.apply(x)


def bf16_compute_copy(w: torch.Tensor) -> torch.Tensor:
    """The value of bf16(w) with the gradient of w (the kernels read the bf16 copy, the optimizer owns fp32 w)."""
    return w + (w.detach().to(torch.bfloat16).to(torch.float32) - w.detach())


def conv_net_bf16_emulated(x: torch.Tensor, p: dict, keep_prob: float = 1.0, mask: torch.Tensor = None) -> torch.Tensor:
    """conv_net with the engine's bf16 storage points.  ``mask``: [B, hidden] 0/1 dropout keep mask (the engine's
    Philox mask, see dropout_mask_reference); required when keep_prob < 1."""
    x = x.reshape(-1, 28, 28, 1)
    c1 = round_bf16(maxpool2d(conv2d_same(x, bf16_compute_copy(p["wc1"]), p["bc1"])))
    c2 = round_bf16(maxpool2d(conv2d_same(c1, bf16_compute_copy(p["wc2"]), p["bc2"])))
    flat = c2.reshape(c2.shape[0], -1)
    h1 = F.relu(flat @ bf16_compute_copy(p["wd1"]) + p["bd1"])
    if keep_prob < 1.0:
        h1 = h1 * (mask.to(h1.dtype) * (1.0 / keep_prob))
    h1 = round_bf16(h1)
    return h1 @ p["out_w"] + p["out_b"]


def philox4x32_10(key: int, ctr_lo, ctr_hi: int):
    """Host mirror of csrc/kernels/common.hpp Philox::run for a vector of low counters: returns word 0 (uint32 array)."""
    import numpy as np
    ctr_lo = np.asarray(ctr_lo, dtype=np.uint64)
    M32 = np.uint64(0xFFFFFFFF)
    c0, c1 = ctr_lo & M32, ctr_lo >> np.uint64(32)
    c2 = np.full_like(c0, np.uint64(ctr_hi & 0xFFFFFFFF))
    c3 = np.full_like(c0, np.uint64((ctr_hi >> 32) & 0xFFFFFFFF))
    k0, k1 = key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & M32, p1 >> np.uint64(32), p1 & M32
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0.astype(np.uint32)


def dropout_mask_reference(seed: int, offset: int, rows: int, hidden: int, keep_prob: float) -> torch.Tensor:
    """The engine's dropout keep mask of one step as a [rows, hidden] uint8 tensor: element (b, j) is kept iff
    u32_to_unit_float(Philox(seed, b * hidden + j, offset)[0]) < keep_prob  (csrc/kernels/common.hpp dropout_scale)."""
    import numpy as np
    r0 = philox4x32_10(int(seed), np.arange(rows * hidden, dtype=np.uint64), int(offset))
    u = (r0 >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return torch.from_numpy((u < np.float32(keep_prob)).astype(np.uint8).reshape(rows, hidden))
