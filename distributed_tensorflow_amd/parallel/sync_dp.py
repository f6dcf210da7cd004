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
  * ``fabric`` (parallel/dp_fabric.py, csrc/kernels/dp.hip): BOTH messages as peer-to-peer stores over the xGMI mesh
    instead of RCCL -- the dense one sharded the same way (reduce-scatter / shard update / all-gather of the compute
    copy), the tail as a one-shot exchange whose kernel is also the step's commit point -- with every per-step counter
    in device memory: the step is ONE native call with constant arguments, captured once into a HIP graph
    (``capture()``), and a step then costs the host one graph launch.  Timed-out waits are all-or-nothing (a poisoned
    rank applies nothing any more, and says so in its flags).
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
                 defer_dense_update: bool = True, fabric=None, collapse_world1: bool = True,
                 shard_dense=None, grad_comm: str = "fp32", fabric_single_stream: bool = False):
        if grad_comm not in ("fp32", "bf16"):
            raise ValueError(f"grad_comm must be fp32 or bf16, not {grad_comm!r}")
        if grad_comm == "bf16" and engine.dtype != "bf16":
            raise ValueError("grad_comm=bf16 needs the bf16 configuration (the fp32 configuration is exact end to end)")
        if grad_comm == "bf16" and fabric is not None:
            raise ValueError("grad_comm=bf16 uses the process-group collectives; the exchange fabric sums fp32")
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
        self.tail_stream = None if tail_group is not None else self.comm_stream
        # fabric (parallel/dp_fabric.py): both messages peer to peer, the step one graph-capturable native call.
        # collapse_world1: a one-rank job applies the dense SGD update in the wgrad epilogue like the single-GPU step
        # (False keeps the N > 1 kernels in the step -- what a one-GPU box can measure of their local cost).
        self.fabric = fabric
        self.collapse_world1 = bool(collapse_world1)
        # fabric_single_stream: the dense exchange runs on the compute stream itself instead of a branch next to the
        # conv backward (no overlap, no second hardware queue: for many ranks that share ONE GPU in rehearsals)
        self.fabric_single_stream = bool(fabric_single_stream)
        self._graph = None
        self.defer = defer_dense_update
        lay = engine.layout
        self.n_dense = lay.segment("wd1", "wd1")[1]     # the dense collective covers wd1 only; head + conv form the tail
        if fabric is not None and (fabric.desc.n_dense != self.n_dense or fabric.desc.n_tail != lay.total - self.n_dense
                                   or fabric.world != self.world):
            raise ValueError("the exchange fabric was built for another model / group")
        be = max(1, int(bucket_mb * (1 << 20) / 4)) & ~3
        self.dense_buckets = make_buckets(0, self.n_dense, be)
        self.tail_bucket = (self.n_dense, lay.total)
        engine.set_grad_scale(1.0 / self.world)
        self._have_pending_dense = False
        # sharded dense update (module docstring): needs equal, 32-byte aligned shards
        self.rank = dist.get_rank(group)
        # what the all-gather carries: the tensor the fc1 kernels read
        self._dense_weights = (engine.params if engine.dtype == "fp32" else engine.shadow)[0:self.n_dense]
        # what the dense collective carries: the fp32 gradient arena's wd1 range, or its 16-bit stand-in
        if grad_comm == "bf16":
            self._dense_grad = engine.enable_dense_g16()[0:self.n_dense]
        else:
            engine.disable_dense_g16()
            self._dense_grad = engine.grads[0:self.n_dense]
        # Sharded updates leave the OTHER ranks' shards of the fp32 master weights stale (mixed precision: the all-gather
        # carries the bf16 copy) and -- with Adam, in either precision -- of the moments, which never travel.
        self.weights_stale = False
        self.moments_stale = False
        self.shard = False
        self.shard_lo, self.shard_hi = 0, self.n_dense
        self.set_shard(self.world > 1 if (shard_dense is None or fabric is not None) else bool(shard_dense))

    @property
    def master_stale(self) -> bool:
        """True while other ranks' shards of the fp32 master weights / optimizer moments are out of date here."""
        return self.weights_stale or self.moments_stale

    @property
    def needs_gather(self) -> bool:
        """Checkpoints of this job need a job-wide gather_master() first."""
        return self.shard and self.world > 1 and (self.eng.dtype != "fp32" or self.eng.optimizer == "adam")

    def can_shard(self) -> bool:
        return self.n_dense % (8 * self.world) == 0

    def set_shard(self, on: bool):
        """Switch the dense message between all-reduce and reduce-scatter / sharded update / all-gather.  Collective
        when it switches sharding OFF after sharded steps (the master weights are gathered first)."""
        on = bool(on) and self.can_shard()
        if self.fabric is not None and on != (self.world > 1):
            raise ValueError("the exchange fabric always shards the dense update over the ranks")
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
        if self._graph is not None:
            self._graph.replay()
            self._mark_stale()
            return
        if self.fabric is not None:
            self._fabric_step()
            self._mark_stale()
            return
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
        if self.master_stale and not self.shard:
            raise RuntimeError("sharded dense updates were followed by an unsharded step without gather_master(): "
                               "use set_shard(False)")
        if self.shard:
            with torch.cuda.stream(self.comm_stream):
                self._reduce_scatter()
        else:
            with torch.cuda.stream(self.comm_stream):
                for lo, hi in self.dense_buckets:
                    self._allreduce(lo, hi, self.group)
        if tr: tr.pop(); tr.push("dp_phase2: conv backward")
        shard = self.shard
        lo, hi = (self.shard_lo, self.shard_hi) if shard else (0, self.n_dense)
        check(lib.dtf_engine_dp_phase2(d, lo, hi, self.DENSE_SLOT, 1, dense_src, int(not shard), cur, comm, tail),
              "dp_phase2")
        if shard:
            if tr: tr.pop(); tr.push("all-gather dense weights (enqueue)")
            with torch.cuda.stream(self.comm_stream):
                self._all_gather()
            check(lib.dtf_engine_dp_record_updated(d, comm), "dp_record_updated")
            self._mark_stale()
        if tr: tr.pop(); tr.push("allreduce tail (enqueue)")
        if self.tail_stream is None:
            self._allreduce(*self.tail_bucket, self.tail_group)
        else:
            with torch.cuda.stream(self.tail_stream):
                self._allreduce(*self.tail_bucket, self.tail_group)
        if tr: tr.pop(); tr.push("dp_phase3: update")
        check(lib.dtf_engine_dp_phase3(d, self.n_dense, self.TAIL_SLOT, eng.step_inc, int(not self.defer), cur, tail),
              "dp_phase3")
        if tr: tr.pop()
        self._have_pending_dense = self.defer

    def _mark_stale(self):
        if self.shard and self.world > 1:
            self.weights_stale = self.eng.dtype != "fp32"
            self.moments_stale = self.eng.optimizer == "adam"

    def use_fabric(self, fabric):
        """Switch between the exchange fabric and the process-group path (collective when sharded masters are stale:
        they are gathered before the fabric is dropped).  A captured graph is released."""
        self._graph = None
        if fabric is None:
            if self.fabric is not None:
                self.flush()
                self.gather_master()
            self.fabric = None
            return
        if self.grad_comm == "bf16":
            raise ValueError("grad_comm=bf16 uses the process-group collectives; the exchange fabric sums fp32")
        lay = self.eng.layout
        if fabric.desc.n_dense != self.n_dense or fabric.desc.n_tail != lay.total - self.n_dense or fabric.world != self.world:
            raise ValueError("the exchange fabric was built for another model / group")
        self.flush()
        self.fabric = fabric
        self.set_shard(self.world > 1)

    def fabric_self_check(self, steps: int = 2) -> bool:
        """Collective: run real steps through the fabric (both parities) and verify on EVERY rank that
          * nothing timed out and nobody is poisoned;
          * the tail exchange delivered everybody's vector: a control word published as rank + 1 sums to W (W + 1) / 2;
          * the summed gradient of this rank's dense shard agrees with the process group's all-reduce of the same data;
          * the reduced tail is bit-identical on all ranks (rank-ordered sums).
        True only on a unanimous vote.  Parameters, moments and step state are restored: invisible to training."""
        eng, fab, W = self.eng, self.fabric, self.world
        keep = [eng.params, eng.state, eng.shadow, eng.wc2T, eng.out_wT, eng.hyper]
        if eng.optimizer == "adam":
            keep += [eng.adam_m, eng.adam_v]
        snap = [t.clone() for t in keep]
        ok = True
        ctrl = eng.layout.total - 4
        try:
            eng.hyper[11] = float(self.rank + 1)
            for _ in range(max(2, steps)):
                self._fabric_step()
            torch.cuda.synchronize(eng.device)
            ok = fab.poisoned() == 0
            ok = ok and abs(float(eng.grads[ctrl + 3].item()) - W * (W + 1) / 2) < 1e-3
            per = self.n_dense // W
            ref = eng.grads[0:self.n_dense].clone()          # this rank's own dense gradient is still in the arena
            _pg.all_reduce_sum(ref, group=self.group)
            mine = ref[self.rank * per:(self.rank + 1) * per]
            tol = 1e-5 * W * float(mine.abs().max().item() + 1e-6)
            if W > 1 or not self.collapse_world1:      # (a collapsed one-rank step runs no dense exchange)
                ok = ok and bool((fab.gsum - mine).abs().max().item() <= tol)
            tails = [torch.empty(eng.layout.total - self.n_dense) for _ in range(W)]
            dist.all_gather(tails, eng.grads[self.n_dense:].cpu(), group=self.group)
            ok = ok and all(torch.equal(tails[0], t) for t in tails[1:])
        except Exception as e:          # a failed launch on one rank is a "no" vote, not a stranded peer
            import sys
            print(f"[dp] fabric self-check: {type(e).__name__}: {e}", file=sys.stderr)
            ok = False
        for t, s0 in zip(keep, snap):
            t.copy_(s0)
        self.weights_stale = self.moments_stale = False     # the snapshot restored every shard
        torch.cuda.synchronize(eng.device)
        vote = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if dist.get_backend(self.group) != "gloo":
            vote = vote.to(eng.device)
        dist.all_reduce(vote, op=dist.ReduceOp.MIN, group=self.group)
        return bool(vote.item())

    def _fabric_step(self):
        eng = self.eng
        # bit 0: collapse a one-rank job; bit 1: every rank has a GPU to itself (kernel order of the backward, engine.hip)
        flags = int(self.collapse_world1) | (2 if getattr(self.fabric, "exclusive_devices", False) else 0)
        check(eng.lib.dtf_engine_dp_step(eng._d(), self.fabric.byref(), flags, eng.step_inc,
                                         torch.cuda.current_stream(eng.device).cuda_stream,
                                         torch.cuda.current_stream(eng.device).cuda_stream if self.fabric_single_stream
                                         else self.comm_stream.cuda_stream),
              "dp_step")

    def capture(self, warmup: int = 2):
        """Capture the fabric step into ONE HIP graph (collective: every rank calls it at the same point).  The warm-up
        launches are real steps -- lazy kernel set-up must not happen under capture, and both parities of the exchange
        buffers get used once -- so parameters, moments and step state are snapshotted and restored: capturing is
        invisible to training.  Afterwards train_step() is a graph launch: no per-step host work besides it."""
        if self.fabric is None:
            raise RuntimeError("only the exchange-fabric step has constant launch arguments (the RCCL path is issued by "
                               "torch.distributed per step)")
        eng = self.eng
        keep = [eng.params, eng.state, eng.shadow, eng.wc2T, eng.out_wT]
        if eng.optimizer == "adam":
            keep += [eng.adam_m, eng.adam_v]
        snap = [t.clone() for t in keep]
        st = torch.cuda.Stream(eng.device)
        st.wait_stream(torch.cuda.current_stream(eng.device))
        with torch.cuda.stream(st):
            for _ in range(max(2, warmup)):
                self._fabric_step()
        torch.cuda.current_stream(eng.device).wait_stream(st)
        torch.cuda.synchronize(eng.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            self._fabric_step()
        for t, s0 in zip(keep, snap):
            t.copy_(s0)
        torch.cuda.synchronize(eng.device)
        self._graph = g
        return g

    def release_graph(self):
        self._graph = None

    def gather_master(self):
        """Collective (every rank of the group): bring the fp32 master weights -- and the Adam moments -- of the dense
        range up to date on every rank after sharded updates.  No-op when nothing is stale."""
        if not self.master_stale:
            return
        self.flush()
        eng = self.eng
        arenas = ([eng.params] if self.weights_stale else []) + ([eng.adam_m, eng.adam_v] if self.moments_stale else [])
        for a in arenas:
            _pg.all_gather_inplace(a[0:self.n_dense], self.rank, self.world, group=self.group)
        self.weights_stale = self.moments_stale = False

    def flush(self):
        """Make every pending update visible to the current stream (call before reading parameters)."""
        if self._have_pending_dense:
            check(self.eng.lib.dtf_engine_dp_flush(self.eng._d(), torch.cuda.current_stream(self.eng.device).cuda_stream),
                  "dp_flush")
            self._have_pending_dense = False
