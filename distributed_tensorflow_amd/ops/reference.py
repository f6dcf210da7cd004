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


def conv_net(x: torch.Tensor, p: dict, keep_prob: float = 1.0, generator=None) -> torch.Tensor:
    """The reference model function (reference M:66-90).  x: [B,784] -> logits [B,classes]."""
    x = x.reshape(-1, 28, 28, 1)
    c1 = maxpool2d(conv2d_same(x, p["wc1"], p["bc1"]))
    c2 = maxpool2d(conv2d_same(c1, p["wc2"], p["bc2"]))
    flat = c2.reshape(c2.shape[0], -1)  # NHWC contiguous -> HWC flatten, as tf.reshape does
    h1 = F.relu(flat @ p["wd1"] + p["bd1"])
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
