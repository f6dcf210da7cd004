"""Synchronous data parallelism: bucketed all-reduce of the flat gradient arena,
overlapped with backward.

This is what replaces the reference's worker<->ps gRPC traffic (pull 13.1 MB of
parameters + push 13.1 MB of gradients per worker step) and
the ``replica_device_setter`` placement (reference .idea/MNISTDist.py:110-111):
parameters are replicated, and each step moves ONE gradient arena through RCCL.

Design for xGMI (point-to-point links, ring collectives are per-link bound):
  * the arena starts with wd1 -- 98 % of the bytes -- whose gradient is final
    right after the fc1 wgrad GEMM.  It is all-reduced on a
    dedicated comm stream while the conv backward kernels still run on the
    compute stream.  One producer kernel finalises the whole prefix at once, so
    the default is ONE large collective (RCCL pipelines it internally over the
    xGMI links; every extra collective costs launch latency on both host and
    device); ``bucket_mb`` can split it for experiments;
  * everything else (head + conv tensors, 0.3 MB, contiguous behind wd1) goes out as
    one last bucket after a single slab reduction, and is updated by one fused launch;
  * buckets are byte ranges of the flat arena (no per-tensor collectives, no
    copies: RCCL reduces in place);
  * the optimizer kernel applies the 1/world scaling (hyper[4]), so no extra
    pass touches the gradients;
  * the update of the dense (wd1) range runs on the comm stream right behind
    its all-reduce and is only waited for by the NEXT step's fc1 forward, so the
    big all-reduce also hides behind the next step's sampler + conv forward;
  * ``shard_dense`` (default at world > 1 over RCCL): the dense message is a REDUCE-SCATTER, each rank applies the
    optimizer to its 1/world shard of wd1 only (fp32 master + moments of that shard), and the updated weights come
    back as an ALL-GATHER of what the kernels actually read -- the bf16 copy (fp32 configuration: the fp32 weights).
    Per rank that is (N-1)/N x (12.8 + 6.4) MB on the links instead of 2 (N-1)/N x 12.8 MB, and the dense update
    kernel touches 1/N of the arena.  The fp32 master weights (and Adam moments) of the other ranks' shards go stale in
    between; ``gather_master()`` (a collective: every rank calls it) brings them up to date before anything reads
    them -- checkpoints, parameter dumps, the end of training (parallel/strategies.py arranges the agreement).
  * ``grad_comm="bf16"`` (opt-in, mixed-precision configuration, RCCL / process-group paths): the fc1 weight gradient
    is produced, communicated and consumed in bf16 -- half the bytes of the dominant message ((N-1)/N x (6.4 + 6.4) MB
    per rank when sharded).  The sum over ranks is then a bf16 sum (each hop rounds to 8 bits of mantissa), which is
    NOT the arithmetic of the reference's fp32 gradients: tests/test_engine_numerics_gpu.py holds the convergence
    comparison, and benchmark numbers taken with it say so in their ``config``.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist

from ..engine import HipEngine
from ..ops.native import check
from ..utils import trace
from . import comm as _pg


def make_buckets(lo: int, hi: int, bucket_elems: int) -> List[Tuple[int, int]]:
    """Split [lo, hi) into contiguous chunks of at most bucket_elems elements."""
    out = []
    s = lo
    while s < hi:
        e = min(hi, s + bucket_elems)
        out.append((s, e))
        s = e
    return out


class SyncDataParallel:
    """Overlapped synchronous DP step for the HIP engine.

    Timeline of step t (compute stream C, comm streams A and B):

      C: sample, conv1, conv2 fwd            | needs only the conv params (updated at the end of t-1)
      C: wait(wd1 of t-1 updated)            | the dominant all-reduce of step t-1 has had the whole
      C: fc1 fwd, head fwd/bwd               |   conv backward of t-1 AND the conv forward of t to finish
      C: head wgrad (slabs), fc1 wgrad       -> event dense_ready
      A:   all-reduce(wd1 buckets) ... then (after fc1 dgrad has read the old weights)
      A:   optimizer on the dense range (fp32 master + bf16 copy) -> event dense_updated
      C: fc1 dgrad, conv2 dgrad, conv2 wgrad, conv1 wgrad, ONE slab reduce (head + conv) -> event tail_ready
      C:   all-reduce(head + conv tail)   (second communicator: runs concurrently with A inside RCCL)
      C: ONE fused launch: optimizer on the tail, state advance, wc2 repack, out_w transpose

    Semantics are exactly synchronous SGD/Adam (no stale gradients): only the *placement* of the
    dense update moves, hidden behind work that does not read those weights.
    """

    DENSE_SLOT, TAIL_SLOT = 3, 4   # per-range Adam step counters in the device state

    def __init__(self, engine: HipEngine, bucket_mb: float = 64.0, group=None, tail_group=None,
                 defer_dense_update: bool = True, tail_p2p=None, dense_p2p=None, p2p_tail_update: bool = True,
                 shard_dense=None, grad_comm: str = "fp32"):
        if grad_comm not in ("fp32", "bf16"):
            raise ValueError(f"grad_comm must be fp32 or bf16, not {grad_comm!r}")
        if grad_comm == "bf16" and engine.dtype != "bf16":
            raise ValueError("grad_comm=bf16 needs the bf16 configuration (the fp32 configuration is exact end to end)")
        if grad_comm == "bf16" and dense_p2p is not None:
            raise ValueError("grad_comm=bf16 uses the process-group collectives; the peer-to-peer dense kernel sums fp32")
        self.grad_comm = grad_comm
        self.eng = engine
        self.group = group
        self.tail_group = tail_group if tail_group is not None else group
        self.world = dist.get_world_size(group)
        self.comm_stream = torch.cuda.Stream(engine.device)
        # The tail collective runs on the COMPUTE stream by default: it sits at the very end of backward and
        # everything after it depends on it, so a stream of its own buys no overlap and costs four event calls
        # (5-8 us of host time each on this stack, and the step is host-bound).  What it does need is its own
        # communicator (tail_group), so that it is not ordered behind the long dense all-reduce inside RCCL.
        # Without a second communicator it has to queue on the comm stream behind the dense collective.
        # tail_p2p (parallel/p2p.py): the tail goes through the one-shot peer-to-peer kernel instead of RCCL, also on
        # the compute stream.
        self.tail_p2p = tail_p2p
        # (A variant with the exchange INSIDE the fused tail kernel -- ~1000 workgroups each doing their own peer
        # round trip -- was removed: +22 us per step at world 1 from the per-workgroup system-scope fences, and it
        # needs every one of those spinning workgroups co-resident next to the overlapped dense kernel, which held
        # only by accident of the stream -> hardware-queue mapping.)
        # p2p_tail_update (default with tail_p2p): the exchange kernel also applies the tail update to what it has just
        # summed (csrc/kernels/optim.hip p2p_tail_allreduce_update: 64 fat workgroups, 64 fences), so the step ends
        # {slab reduce, exchange + update} instead of {slab reduce, exchange, update}.
        self.p2p_tail_update = bool(p2p_tail_update) and tail_p2p is not None
        # dense_p2p: the 12.8 MB message through the two-shot peer-to-peer kernel (on the comm stream, like RCCL's)
        self.dense_p2p = dense_p2p
        self.tail_stream = None if (tail_group is not None or tail_p2p is not None) else self.comm_stream
        self.defer = defer_dense_update
        lay = engine.layout
        self.n_dense = lay.segment("wd1", "wd1")[1]     # the dense collective covers wd1 only; head + conv form the tail
        be = max(1, int(bucket_mb * (1 << 20) / 4)) & ~3
        self.dense_buckets = make_buckets(0, self.n_dense, be)
        self.tail_bucket = (self.n_dense, lay.total)
        engine.set_grad_scale(1.0 / self.world)
        self._have_pending_dense = False
        # sharded dense update (module docstring): needs equal, 32-byte aligned shards and the process group path
        self.rank = dist.get_rank(group)
        # what the all-gather carries: the tensor the fc1 kernels read
        self._dense_weights = (engine.params if engine.dtype == "fp32" else engine.shadow)[0:self.n_dense]
        # what the dense collective carries: the fp32 gradient arena's wd1 range, or its 16-bit stand-in
        if grad_comm == "bf16":
            self._dense_grad = engine.enable_dense_g16()[0:self.n_dense]
        else:
            engine.disable_dense_g16()
            self._dense_grad = engine.grads[0:self.n_dense]
        self.master_stale = False      # True while other ranks' shards of the fp32 master / moments are out of date
        self.shard = False
        self.shard_lo, self.shard_hi = 0, self.n_dense
        self.set_shard(self.world > 1 if shard_dense is None else bool(shard_dense))

    def can_shard(self) -> bool:
        return self.n_dense % (8 * self.world) == 0

    def set_shard(self, on: bool):
        """Switch the dense message between all-reduce and reduce-scatter / sharded update / all-gather.  Collective
        when it switches sharding OFF after sharded steps (the master weights are gathered first)."""
        on = bool(on) and self.can_shard()
        if not on and self.shard:
            self.gather_master()
        self.shard = on
        per = self.n_dense // self.world
        self.shard_lo, self.shard_hi = (self.rank * per, (self.rank + 1) * per) if on else (0, self.n_dense)

    def broadcast_parameters(self, src: int = 0):
        """Chief initialises, everybody else receives (reference: chief runs init_op)."""
        _pg.broadcast(self.eng.params, src, group=self.group)
        _pg.broadcast(self.eng.state, src, group=self.group)
        self.eng.state[self.eng.STATE_CURSOR] = 0
        self.eng.refresh_shadows()
        torch.cuda.synchronize()

    def _allreduce(self, lo: int, hi: int, group):
        # RCCL reduces in place on the current stream; over gloo (ranks sharing one GPU in tests and rehearsals)
        # parallel/comm.py stages the range through the host -- correct, nothing overlaps
        # (a dense bucket lives in the 16-bit gradient buffer when grad_comm == "bf16")
        _pg.all_reduce_sum(self._dense_grad[lo:hi] if hi <= self.n_dense else self.eng.grads[lo:hi], group=group)

    def _reduce_scatter(self):
        _pg.reduce_scatter_sum_inplace(self._dense_grad, self.rank, self.world, group=self.group)

    def _all_gather(self):
        _pg.all_gather_inplace(self._dense_weights, self.rank, self.world, group=self.group)

    def train_step(self):
        """One synchronous step.  All launches, stream waits and event records of the schedule above
        are issued natively (csrc/runtime/engine.hip: dtf_engine_dp_phase1..3); Python only enqueues
        the two collectives on their streams, so the host cost of a step is 3 foreign calls + 2
        all-reduce enqueues (the pure-Python choreography measured 245 us/step at world 1, more than
        the 180 us of GPU work it was feeding)."""
        eng = self.eng
        lib, d = eng.lib, eng._d()
        tr = trace if trace.enabled() else None     # roctx ranges (utils/trace.py), off by default
        cur = torch.cuda.current_stream(eng.device).cuda_stream
        comm = self.comm_stream.cuda_stream
        tail = cur if self.tail_stream is None else self.tail_stream.cuda_stream
        if tr: tr.push("dp_phase1: fwd + fc1 wgrad")
        check(lib.dtf_engine_dp_phase1(d, int(self._have_pending_dense), cur, comm), "dp_phase1")
        if tr: tr.pop(); tr.push("allreduce dense (enqueue)")
        dense_src = None     # where the deferred dense update reads the summed gradient (None: the gradient arena)
        if self.master_stale and not (self.shard and self.dense_p2p is None):
            raise RuntimeError("sharded dense updates were followed by an unsharded step without gather_master(): "
                               "use set_shard(False)")
        if self.shard and self.dense_p2p is None:
            with torch.cuda.stream(self.comm_stream):
                self._reduce_scatter()
        elif self.dense_p2p is not None:
            # the result stays in the kernel's staging buffer: no copy back into the arena (a third of its local traffic)
            dense_src = self.dense_p2p.allreduce(eng.grads[0:self.n_dense], comm, gather=False)
        else:
            with torch.cuda.stream(self.comm_stream):
                for lo, hi in self.dense_buckets:
                    self._allreduce(lo, hi, self.group)
        if tr: tr.pop(); tr.push("dp_phase2: conv backward")
        shard = self.shard and self.dense_p2p is None
        lo, hi = (self.shard_lo, self.shard_hi) if shard else (0, self.n_dense)
        check(lib.dtf_engine_dp_phase2(d, lo, hi, self.DENSE_SLOT, 1, dense_src, int(not shard), cur, comm, tail),
              "dp_phase2")
        if shard:
            if tr: tr.pop(); tr.push("all-gather dense weights (enqueue)")
            with torch.cuda.stream(self.comm_stream):
                self._all_gather()
            check(lib.dtf_engine_dp_record_updated(d, comm), "dp_record_updated")
            self.master_stale = eng.dtype != "fp32" and self.world > 1
        if self.tail_p2p is not None and self.p2p_tail_update:
            if tr: tr.pop(); tr.push("dp_phase3: p2p all-reduce + update (one launch)")
            p = self.tail_p2p
            p.seq += 1
            check(lib.dtf_engine_dp_phase3_p2p(d, self.TAIL_SLOT, eng.step_inc, int(not self.defer), cur, p._bases, p.world,
                                               p.rank, p.n_max, p.seq, p._status.data_ptr(), p.timeout_s), "dp_phase3_p2p")
            if tr: tr.pop()
            self._have_pending_dense = self.defer
            return
        if tr: tr.pop(); tr.push("allreduce tail (enqueue)")
        if self.tail_p2p is not None:
            self.tail_p2p.allreduce(eng.grads[self.tail_bucket[0]:self.tail_bucket[1]], cur)
        elif self.tail_stream is None:
            self._allreduce(*self.tail_bucket, self.tail_group)
        else:
            with torch.cuda.stream(self.tail_stream):
                self._allreduce(*self.tail_bucket, self.tail_group)
        if tr: tr.pop(); tr.push("dp_phase3: update")
        check(lib.dtf_engine_dp_phase3(d, self.n_dense, self.TAIL_SLOT, eng.step_inc, int(not self.defer), cur, tail),
              "dp_phase3")
        if tr: tr.pop()
        self._have_pending_dense = self.defer

    def gather_master(self):
        """Collective (every rank of the group): bring the fp32 master weights -- and the Adam moments -- of the dense
        range up to date on every rank after sharded updates.  No-op when nothing is stale."""
        if not self.master_stale:
            return
        self.flush()
        eng = self.eng
        arenas = [eng.params] + ([eng.adam_m, eng.adam_v] if eng.optimizer == "adam" else [])
        for a in arenas:
            _pg.all_gather_inplace(a[0:self.n_dense], self.rank, self.world, group=self.group)
        self.master_stale = False

    def flush(self):
        """Make every pending update visible to the current stream (call before reading parameters)."""
        if self._have_pending_dense:
            check(self.eng.lib.dtf_engine_dp_flush(self.eng._d(), torch.cuda.current_stream(self.eng.device).cuda_stream),
                  "dp_flush")
            self._have_pending_dense = False
