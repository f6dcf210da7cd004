"""Process-group bring-up: RCCL over xGMI on GPUs, gloo for the CPU plumbing config.

Replaces the reference's ``tf.train.ClusterSpec`` + in-process gRPC
``tf.train.Server`` (reference .idea/MNISTDist.py:94-103).  One process per
GPU; ``torch.distributed`` backend "nccl" IS RCCL on ROCm.
"""
from __future__ import annotations

import datetime
import os
from typing import Optional

import torch
import torch.distributed as dist


def init_process_group(backend: str, rank: int, world_size: int, master_addr: Optional[str] = None,
                       master_port: Optional[int] = None, device: Optional[torch.device] = None,
                       timeout_s: float = 600.0):
    """Rendezvous over a TCPStore at master_addr:master_port (env MASTER_ADDR/PORT by default)."""
    if dist.is_initialized():
        return
    # (HSA_ENABLE_IPC_MODE_LEGACY=0 -- dmabuf IPC, which peer mappings need on this driver -- is set at package import,
    # before the HIP runtime initialises: distributed_tensorflow_amd/__init__.py)
    if master_addr is not None:
        os.environ["MASTER_ADDR"] = master_addr
    if master_port is not None:
        os.environ["MASTER_PORT"] = str(master_port)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kwargs = dict(backend=backend, rank=rank, world_size=world_size,
                  timeout=datetime.timedelta(seconds=timeout_s))
    if backend == "nccl" and device is not None:
        kwargs["device_id"] = device
    dist.init_process_group(**kwargs)


# ---------------------------------------------------------------------------------------------
# Collectives that also work for GPU tensors over gloo (--comm_backend=gloo: several tasks sharing ONE GPU,
# which RCCL refuses -- how the GPU suite runs the multi-process modes on a one-GPU box).  ProcessGroupGloo's
# own device-tensor path is not usable on this torch/ROCm build: with kernels in flight on the caller's stream a
# collective stalls for seconds to forever, and even fenced with a device synchronize it stalled intermittently
# under the data-parallel schedule.  So a device tensor is staged through host memory HERE and gloo only ever
# sees CPU tensors.  RCCL ("nccl") and CPU tensors go straight through.
# ---------------------------------------------------------------------------------------------
def staged(t: torch.Tensor, group=None) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum(t: torch.Tensor, group=None):
    if staged(t, group):
        h = t.cpu()                      # blocking copy: also drains the producing kernels of the current stream
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def broadcast(t: torch.Tensor, src: int, group=None):
    if staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        if dist.get_rank() != src:
            t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


_scratch = {}


def _scratch_like(t: torch.Tensor, numel: int, tag: str) -> torch.Tensor:
    """A persistent staging buffer per (tag, size, dtype, device): the slice collectives below hand RCCL
    NON-overlapping send / receive buffers (in-place reduce-scatter / all-gather are legal in NCCL but are the road less
    travelled; one small device copy per call buys the common path)."""
    key = (tag, numel, t.dtype, t.device)
    buf = _scratch.get(key)
    if buf is None:
        buf = _scratch[key] = torch.empty(numel, dtype=t.dtype, device=t.device)
    return buf


def reduce_scatter_sum_inplace(t: torch.Tensor, rank: int, world: int, group=None):
    """``t`` is ``world`` equal slices; on return slice ``rank`` holds the sum over the ranks of that slice (the other
    slices are unspecified).  gloo has no reduce-scatter: there (tests, one-GPU rehearsals) the whole range is
    all-reduced, which gives the same slice."""
    per = t.numel() // world
    if dist.get_backend(group) == "gloo":
        all_reduce_sum(t, group=group)
    else:
        out = _scratch_like(t, per, "rs")
        dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM, group=group)
        t[rank * per:(rank + 1) * per].copy_(out)


def all_gather_inplace(t: torch.Tensor, rank: int, world: int, group=None):
    """``t`` is ``world`` equal slices and slice ``rank`` is valid here: afterwards every slice is, on every rank."""
    per = t.numel() // world
    if dist.get_backend(group) == "gloo":
        h = t[rank * per:(rank + 1) * per].cpu() if t.is_cuda else t[rank * per:(rank + 1) * per].clone()
        if h.dtype == torch.bfloat16:          # gloo moves bytes; bf16 is not among its types
            h = h.view(torch.uint8)
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h, group=group)
        out = torch.cat(parts)
        t.copy_(out.view(t.dtype) if out.dtype != t.dtype else out)
    else:
        send = _scratch_like(t, per, "ag")
        send.copy_(t[rank * per:(rank + 1) * per])
        dist.all_gather_into_tensor(t, send, group=group)


def reduce_sum(t: torch.Tensor, dst: int, group=None):
    if staged(t, group):
        h = t.cpu()
        dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if dist.get_rank() == dst:
            t.copy_(h)
    else:
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)


def send(t: torch.Tensor, dst: int):
    dist.send(t.cpu() if staged(t) else t, dst=dst)


def recv(t: torch.Tensor, src: int):
    if staged(t):
        h = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(h, src=src)
        t.copy_(h)
    else:
        dist.recv(t, src=src)


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def allreduce_latency_us(numel: int, group=None, device: Optional[torch.device] = None, iters: int = 20,
                         warmup: int = 5, dtype: torch.dtype = torch.float32) -> float:
    """Mean wall time (us) of one stand-alone SUM all-reduce of ``numel`` elements, max over ranks.

    Every rank of ``group`` must call it with the same arguments.  bench.py reports it (outside the
    timed region) for the two messages of a data-parallel step, so a scaling run also records the two
    terms of the step model in docs/ARCHITECTURE.md: the exposed tail latency and the dense transfer
    time that has to hide under backward.
    """
    import time

    device = device or torch.device("cpu")
    buf = torch.zeros(int(numel), dtype=dtype, device=device)

    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize(device)

    sync()
    for _ in range(warmup):
        all_reduce_sum(buf, group=group)
    sync()
    dist.barrier(group=group)
    sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        all_reduce_sum(buf, group=group)
    sync()
    us = torch.tensor([(time.perf_counter() - t0) * 1e6 / iters], dtype=torch.float64)
    if dist.get_backend(group) != "gloo":
        us = us.to(device)
    dist.all_reduce(us, op=dist.ReduceOp.MAX, group=group)
    return float(us.item())
