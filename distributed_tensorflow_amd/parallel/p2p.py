"""One-shot peer-to-peer all-reduce for the small tail message of a data-parallel step.

The reference moves every tensor through gRPC to a parameter server; the
data-parallel replacement here moves two messages per step, and the second one -- head + conv
gradients, 0.25 MB -- is needed by the very next kernel, so its LATENCY is exposed.  xGMI is a full
point-to-point mesh: every rank pushes its vector into a slot on every peer with plain stores, raises
a flag, and sums the slots it received, in rank order (csrc/kernels/p2p.hip).  One hop instead of a
ring's 2(N-1), bit-identical results on all ranks, no float atomics.

Bring-up is collective (every rank of ``group``): allocate an uncached buffer, exchange hipIpc handles
through the process group, map the peers.  ``self_check`` compares against the process group's own
all-reduce on random data and takes a unanimous vote, so a platform where peer mappings do not work
falls back to RCCL instead of training on wrong sums; every in-kernel wait is bounded and reports a
timeout through ``status()`` instead of hanging the GPU.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch
import torch.distributed as dist

from ..ops import native
from . import comm


class P2PAllReduce:
    MAX_WORLD = 8

    def __init__(self, n_max: int, device: torch.device, group=None, timeout_s: float = 2.0, two_shot: bool = False):
        """two_shot=False: one-shot (every rank receives and sums all vectors -- latency-optimal, for small messages);
        two_shot=True: reduce-scatter + all-gather as peer pushes (bandwidth-optimal, for the 12.8 MB dense message)."""
        self.lib = native.load()
        self.two_shot = bool(two_shot)
        self.group, self.device = group, torch.device(device)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if not 1 <= self.world <= self.MAX_WORLD:
            raise ValueError(f"P2PAllReduce supports up to {self.MAX_WORLD} ranks, got {self.world}")
        self.quantum = 4 * self.world if self.two_shot else 4      # vector lengths must be multiples of this
        self.n_max = -(-int(n_max) // self.quantum) * self.quantum
        self._bytes_fn = self.lib.dtf_p2p2_bytes if self.two_shot else self.lib.dtf_p2p_bytes
        self.blocks = 64            # two-shot only: workgroups per launch (<= 256), the same on every rank
        self.timeout_s = float(timeout_s)
        self.seq = 0
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._mapped: List[int] = []
        self._own = ctypes.c_void_p()
        err: Optional[str] = None
        handle = ctypes.create_string_buffer(64)
        with torch.cuda.device(self.device):
            try:
                nbytes = self._bytes_fn(self.world, self.n_max)
                native.check(self.lib.dtf_p2p_alloc(nbytes, ctypes.byref(self._own)), "p2p alloc")
                native.check(self.lib.dtf_p2p_export(self._own, handle), "hipIpcGetMemHandle")
            except Exception as e:     # keep making the collective calls below: the peers are waiting in them
                err = repr(e)
            handles: List[Optional[bytes]] = [None] * self.world
            dist.all_gather_object(handles, None if err else handle.raw, group=group)
            bases = (ctypes.c_void_p * self.world)()
            if err is None and any(h is None for h in handles):
                err = "a peer could not allocate/export its buffer"
            if err is None:
                try:
                    for r, h in enumerate(handles):
                        if r == self.rank:
                            bases[r] = self._own.value
                        else:
                            p = ctypes.c_void_p()
                            native.check(self.lib.dtf_p2p_import(h, ctypes.byref(p)), f"hipIpcOpenMemHandle(rank {r})")
                            bases[r] = p.value
                            self._mapped.append(p.value)
                except Exception as e:
                    err = repr(e)
            self._bases = bases
        oks: List[Optional[str]] = [None] * self.world
        dist.all_gather_object(oks, err, group=group)     # unanimous: either every rank has every mapping or nobody goes on
        if any(o is not None for o in oks):
            self.close()
            raise RuntimeError("P2PAllReduce bring-up failed: " + "; ".join(f"rank {r}: {o}" for r, o in enumerate(oks) if o))
        dist.barrier(group=group)     # nobody pushes before everybody has mapped (and zeroed) everything

    def allreduce(self, t: torch.Tensor, stream: Optional[int] = None, gather: bool = True) -> Optional[int]:
        """SUM-all-reduce the fp32 vector ``t`` in place, on ``stream`` (raw handle; default: the current one).
        two-shot with gather=False: ``t`` is left alone and the address of the result in this rank's (uncached)
        staging buffer is returned -- valid until the all-reduce after the next one; the consumer reads it there."""
        n = t.numel()
        if t.dtype != torch.float32 or not t.is_contiguous() or n > self.n_max or n % self.quantum:
            raise ValueError(f"P2PAllReduce: fp32, contiguous, a multiple of {self.quantum} and at most n_max elements")
        self.seq += 1
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        if self.two_shot:
            rc = self.lib.dtf_p2p2_allreduce(t.data_ptr(), self._bases, self.world, self.rank, n, self.n_max, self.seq,
                                             self._status.data_ptr(), self.timeout_s, self.blocks, int(gather), s)
        else:
            rc = self.lib.dtf_p2p_allreduce(t.data_ptr(), self._bases, self.world, self.rank, n, self.n_max, self.seq,
                                            self._status.data_ptr(), self.timeout_s, s)
        native.check(rc, "p2p allreduce")
        if self.two_shot and not gather:     # stageB(parity) of the own allocation: after stageA = [2][n_max] floats
            return self._own.value + 4 * (2 * self.n_max + (self.seq & 1) * self.n_max)
        return None

    def status(self) -> int:
        """0 = fine; 1 = some wait timed out since the last call (synchronises the device)."""
        v = int(self._status.item())
        if v:
            self._status.zero_()
        return v

    def self_check(self, n: int, iters: int = 8, rtol: float = 1e-5) -> bool:
        """Collective: random vectors through this path and through the process group; True only if EVERY rank saw
        matching sums and no timeout, on every iteration (both parities, reused slots)."""
        ok = True
        g = torch.Generator(device=self.device)
        g.manual_seed(1234 + self.rank)
        for _ in range(iters):
            x = torch.randn(n, device=self.device, generator=g) * (1.0 + self.rank)
            ref = x.clone()
            comm.all_reduce_sum(ref, group=self.group)
            self.allreduce(x)
            torch.cuda.synchronize(self.device)
            tol = rtol * float(ref.abs().max().item() + 1.0) * self.world
            ok = ok and self.status() == 0 and bool((x - ref).abs().max().item() <= tol)
        vote = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        if comm.staged(vote, self.group):
            vote = vote.cpu()
        dist.all_reduce(vote, op=dist.ReduceOp.MIN, group=self.group)
        return bool(vote.item())

    def latency_us(self, n: int, iters: int = 20, warmup: int = 5) -> float:
        """Collective: mean stand-alone time of one all-reduce of n elements, max over ranks."""
        import time
        x = torch.zeros(n, device=self.device)
        for _ in range(warmup):
            self.allreduce(x)
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)
        t0 = time.perf_counter()
        for _ in range(iters):
            self.allreduce(x)
        torch.cuda.synchronize(self.device)
        us = torch.tensor([(time.perf_counter() - t0) * 1e6 / iters], dtype=torch.float64)
        if dist.get_backend(self.group) != "gloo":
            us = us.to(self.device)
        dist.all_reduce(us, op=dist.ReduceOp.MAX, group=self.group)
        return float(us.item())

    def close(self):
        for p in self._mapped:
            self.lib.dtf_p2p_unmap(p)
        self._mapped = []
        if self._own.value:
            self.lib.dtf_p2p_free(self._own)
            self._own = ctypes.c_void_p()
