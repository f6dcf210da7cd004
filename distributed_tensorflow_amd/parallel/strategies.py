"""Distribution strategies: how one optimisation step crosses process boundaries.

Reference behaviour being replaced/emulated (reference .idea/MNISTDist.py:98-111, :173-188):
between-graph replication where every worker pulls all variables from the ps tasks and pushes
all gradients back each step, the ps applying them with no locking (asynchronous, stale
gradients); variables are sharded whole-tensor round-robin over ps tasks; the shared
``global_step`` counts applied worker gradients.

Strategies (all work on the backend's flat fp32 param / grad tensors):
  LocalStrategy        single process.
  SyncAllReduce        synchronous DP: all-reduce(mean) of the gradient arena among workers
                       (RCCL; overlapped with backward on the GPU backend, see sync_dp.py).
                       ps tasks, if any, only idle until the workers finish.
  AsyncP2P             true asynchronous parameter server over point-to-point send/recv:
                       each ps task owns its round-robin shard of the variables, applies every
                       worker gradient the moment it arrives (arrival order, no barrier) and
                       returns fresh parameters; tolerates dying workers.
  AsyncRounds          parameter-server semantics on collectives (reduce to the ps rank, apply,
                       RCCL broadcast of the parameters), with a staleness bound of 0 or 1 round:
                       gradients are applied un-averaged and global_step advances by the number
                       of workers per round.  This is the GPU-friendly emulation (few large
                       collectives instead of many p2p messages).
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from ..backends import Backend
from ..cluster import TaskInfo, replica_device_setter
from ..models.spec import TF_VARIABLE_NAMES
from . import comm

KIND_GRAD, KIND_PULL, KIND_BYE = 0.0, 1.0, 2.0


def _creation_order() -> List[str]:
    return [k for k in TF_VARIABLE_NAMES]  # wc1, wc2, wd1, out_w, bc1, bc2, bd1, out_b, global_step


def ps_shards(task: TaskInfo, backend: Backend) -> Tuple[Dict[int, List[Tuple[int, int]]], int]:
    """ps task -> list of [lo, hi) arena ranges it owns; and the ps task owning global_step."""
    placement = replica_device_setter(task.cluster, _creation_order())
    shards: Dict[int, List[Tuple[int, int]]] = {p: [] for p in range(max(1, task.num_ps))}
    for name, p in placement.items():
        if name == "global_step":
            continue
        lo = backend.layout.offsets[name]
        shards[p].append((lo, lo + backend.layout.sizes[name]))
    for p in shards:
        shards[p].sort()
    return shards, placement["global_step"]


def _gather(flat: torch.Tensor, ranges, out: torch.Tensor):
    o = 0
    for lo, hi in ranges:
        out[o:o + hi - lo].copy_(flat[lo:hi])
        o += hi - lo


def _scatter(src: torch.Tensor, ranges, flat: torch.Tensor):
    o = 0
    for lo, hi in ranges:
        flat[lo:hi].copy_(src[o:o + hi - lo])
        o += hi - lo


def _set_int(t: torch.Tensor, i: int, v: int):
    t[i:i + 1].view(torch.int32)[0] = int(v)


def _get_int(t: torch.Tensor, i: int) -> int:
    return int(t[i:i + 1].view(torch.int32)[0].item())


class LocalStrategy:
    name = "local"
    coordinated_stop = False      # a stop request takes effect at once, locally

    def __init__(self, backend: Backend):
        self.b = backend
        self.step_count = backend.get_global_step()

    def broadcast_initial(self):
        self.step_count = self.b.get_global_step()   # the supervisor may just have restored a checkpoint

    def step(self) -> int:
        if hasattr(self.b, "train_step_local"):
            self.b.train_step_local()
        else:
            self.b.compute_gradients()
            self.b.apply_gradients()
        self.step_count += 1
        return self.step_count

    def finish(self):
        pass


class SyncAllReduce:
    """Synchronous data parallelism among the worker ranks."""

    name = "sync"
    # Stop requests are COORDINATED: a synchronous worker that simply left the loop would strand its peers in the
    # next collective.  request_stop() raises this worker's flag in the control block of the gradient arena, which
    # rides in every step's (tail) all-reduce; poll_stop() reads the sum, which is the same number on every worker
    # after the same step, so all of them leave the loop together (reference: sv.should_stop(), M:173).
    coordinated_stop = True

    def __init__(self, backend: Backend, task: TaskInfo, worker_group, chief_rank: int, tail_group=None,
                 p2p: bool = False, log=None, stop_check_steps: int = 20, shard_dense=None, grad_comm: str = "fp32",
                 hip_graph: bool = True, p2p_timeout_s: float = 60.0):
        if grad_comm not in ("fp32", "bf16"):
            raise ValueError(f"--grad_comm must be fp32 or bf16, not {grad_comm!r}")
        if grad_comm == "bf16" and (backend.name != "hip" or backend.engine.dtype != "bf16"):
            raise ValueError("--grad_comm=bf16 needs --backend=hip with --dtype=bf16")
        self.b, self.task, self.group, self.chief_rank = backend, task, worker_group, chief_rank
        self.world = task.num_workers
        self.step_count = backend.get_global_step()
        self._dp = None
        self._fabric = None
        self._stop_requested = False
        self.stop = False
        # Sharded dense update (parallel/sync_dp.py): the chief's timed checkpoints need every worker to join a
        # gather of the master weights first.  The request rides in the control block like the stop flag; every worker
        # sees it after the same step, gathers, and the chief saves right there (Supervisor.should_stop()).
        self._gather_requested = False
        self.gather_done = False
        # the GPU path reads the flag (a device sync) only every stop_check_steps global steps; the CPU path every step
        self.stop_check_steps = max(1, int(stop_check_steps))
        # The overlapped schedule (parallel/sync_dp.py) needs collectives that run on streams: RCCL, or the
        # peer-to-peer kernels (--p2p_allreduce; these also work between tasks that share one GPU over gloo).
        # (an explicit --dense_update=sharded also selects it over gloo, staged through the host: tests, rehearsals)
        if backend.name == "hip" and (dist.get_backend(worker_group) != "gloo" or p2p or shard_dense
                                      or grad_comm == "bf16"):
            from .sync_dp import SyncDataParallel
            fabric = None
            if p2p and grad_comm != "bf16":
                fabric = self._bring_up_fabric(backend.engine, worker_group, log or (lambda *a: None), p2p_timeout_s)
            self._fabric = fabric
            self._dp = SyncDataParallel(backend.engine, group=worker_group, tail_group=tail_group, fabric=fabric,
                                        shard_dense=shard_dense, grad_comm=grad_comm)
            if fabric is not None and not self._dp.fabric_self_check():
                # (collective, unanimous) peer stores or their visibility do not work on this node: process group
                (log or print)(f"worker/{task.task_index}: exchange fabric failed its self-check; collectives go through "
                               "the process group")
                self._dp.use_fabric(None)
                fabric.close()
                fabric = self._fabric = None
                self._dp.set_shard(self.world > 1 if shard_dense is None else bool(shard_dense))
            if fabric is not None and hip_graph:
                self._dp.capture()       # the whole step becomes one graph launch (collective: every worker is here)
            if grad_comm == "bf16" and log:
                log(f"worker/{task.task_index}: fc1 weight gradient communicated in bf16 (--grad_comm=bf16)")
            if self._dp.shard and log:
                log(f"worker/{task.task_index}: dense update sharded over {self.world} workers "
                    f"(reduce-scatter -> update of 1/{self.world} -> all-gather of the compute copy)"
                    + (", peer-to-peer exchange fabric" + (", one HIP graph per step" if hip_graph else "") if fabric else ""))

    def _bring_up_fabric(self, eng, group, log, timeout_s: float):
        """Both messages through the exchange fabric (csrc/kernels/dp.hip) if every worker can map every peer
        (collective and unanimous: all workers end up with the fabric or all with the process group)."""
        from .dp_fabric import DpFabric
        n_dense = eng.layout.segment("wd1", "wd1")[1]
        try:
            # generous bound: the chief's evaluations stall its peers inside the exchange kernels
            fab = DpFabric(n_dense, eng.layout.total - n_dense, eng.device, dtype=eng.dtype, group=group, timeout_s=timeout_s)
        except RuntimeError as e:
            log(f"worker/{self.task.task_index}: exchange fabric unavailable ({e}); collectives go through the process group")
            return None
        log(f"worker/{self.task.task_index}: gradient exchange: peer-to-peer fabric")
        return fab

    def check_health(self):
        """A timed-out peer-to-peer wait (or a peer that reports one) has stopped every update on this worker; the run
        must end (parallel/dp_fabric.py P2PHealthError -> exit status 3 in the trainer)."""
        if getattr(self, "_fabric", None) is not None:
            self._fabric.health()

    def broadcast_initial(self):
        broadcast_state(self.b, self.chief_rank)
        self.step_count = self.b.get_global_step()

    def request_stop(self):
        self._stop_requested = True

    @property
    def sharded(self) -> bool:
        """True when checkpoints need a job-wide gather first (the dense update is sharded over the workers)."""
        return self._dp is not None and self._dp.needs_gather

    def request_gather(self):
        """Chief: ask every worker to join a gather of the master weights at the next control poll."""
        self._gather_requested = True

    def poll_stop(self) -> bool:
        """True on every worker after the same step once any worker has requested a stop.  The same read serves the
        gather request: when its sum is up, every worker gathers here, after the same step (``gather_done``)."""
        if self.stop:
            return True
        if self._dp is None or self.step_count % self.stop_check_steps == 0:
            sums = self.b.control_sums()
            self.stop = sums[Backend.CTRL_STOP] > 0.0
            if self._dp is not None and sums[Backend.CTRL_GATHER] > 0.0:
                self._dp.gather_master()
                self.gather_done = True
                if self._gather_requested:
                    self._gather_requested = False
                    self.b.set_control(Backend.CTRL_GATHER, 0.0)
        return self.stop

    def step(self) -> int:
        # this rank's flags go out with the step's (tail) all-reduce
        if self._stop_requested:
            self.b.set_control(Backend.CTRL_STOP, 1.0)
        if self._gather_requested:
            self.b.set_control(Backend.CTRL_GATHER, 1.0)
        if self._dp is not None:
            self._dp.train_step()              # the control flags are published by the step's slab reduction
        else:
            self.b.compute_gradients()
            self.b.publish_control()
            comm.all_reduce_sum(self.b.grads, group=self.group)
            self.b.apply_gradients(scale=1.0 / self.world)
        self.step_count += 1
        return self.step_count

    def set_learning_rate(self, lr: float):
        """Change the learning rate between two steps.  The previous step's dense update may still be pending on the
        comm stream and reads the rate from device memory: the compute stream (where the new value is written) first
        waits for it, otherwise that update takes the old or the new rate depending on timing -- on each replica
        separately (GPU-side wait, no host synchronisation; only the steps of a warm-up pay it)."""
        if self._dp is not None:
            self._dp.flush()
        self.b.set_learning_rate(lr)

    def finish(self):
        """Every worker calls this after the same last step."""
        if self._dp is not None:
            self._dp.flush()
            self.check_health()                # before any collective: a poisoned worker must not gather
            self._dp.gather_master()           # collective; no-op unless the dense update is sharded
        if self._fabric is not None:
            dist.barrier(group=self.group)     # nobody unmaps a buffer a peer's last step may still store into
            self._fabric.close()
            self._fabric = None


def broadcast_state(backend: Backend, src_rank: int, group=None):
    """Chief -> everyone: parameters, global_step AND the optimizer slots (reference: chief runs init_op / restore
    while the other tasks wait for the variables to be initialised).  Without the slots a restored Adam
    run would continue with the chief's moments on the chief and zeros everywhere else: every replica would then
    apply a different update to the same all-reduced gradient and the replicas would drift apart unnoticed."""
    comm.broadcast(backend.params, src_rank, group=group)
    dev = backend.params.device
    st = torch.tensor([backend.get_global_step(), backend.get_adam_t()], dtype=torch.int64, device=dev)
    comm.broadcast(st, src_rank, group=group)
    backend.set_global_step(int(st[0].item()))
    slots = backend.adam_arenas()
    if slots is not None:
        for t in slots:
            comm.broadcast(t, src_rank, group=group)
        backend.set_adam_t(int(st[1].item()))
    backend.params_changed()


# ----------------------------------------------------------------------------- async, p2p
class AsyncP2PWorker:
    name = "async-p2p"
    coordinated_stop = False      # asynchronous workers come and go on their own (reference M:174-176)

    def __init__(self, backend: Backend, task: TaskInfo, chief_rank: int):
        self.b, self.task, self.chief_rank = backend, task, chief_rank
        self.shards, self.step_owner = ps_shards(task, backend)
        dev = backend.params.device
        self.msg = {p: torch.zeros(sum(h - l for l, h in r) + 2, dtype=torch.float32, device=dev)
                    for p, r in self.shards.items()}
        self.resp = {p: torch.zeros_like(m) for p, m in self.msg.items()}
        self.step_count = backend.get_global_step()
        self.stop = False

    def broadcast_initial(self):
        broadcast_state(self.b, self.chief_rank)
        self.step_count = self.b.get_global_step()

    def _exchange(self, kind: float):
        b = self.b
        for p, ranges in self.shards.items():          # push to every ps task ...
            m = self.msg[p]
            n = m.numel() - 2
            if kind == KIND_GRAD:
                _gather(b.grads, ranges, m)
            m[n] = kind
            comm.send(m, p)
        if kind == KIND_BYE:
            return
        for p, ranges in self.shards.items():          # ... and pull the fresh shard back
            r = self.resp[p]
            n = r.numel() - 2
            comm.recv(r, p)
            _scatter(r, ranges, b.params)
            if p == self.step_owner:
                self.step_count = _get_int(r, n)
                self.stop = bool(_get_int(r, n + 1))
        b.params_changed()
        b.set_global_step(self.step_count)

    def step(self) -> int:
        self.b.compute_gradients()
        self.b.advance_input()             # no local update follows: move on to the next batch
        self._exchange(KIND_GRAD)
        return self.step_count

    def finish(self):
        self._exchange(KIND_BYE)


class AsyncP2PServer:
    """One ps task: owns a shard of the variables, applies gradients in arrival order."""

    def __init__(self, backend: Backend, task: TaskInfo, chief_rank: int, training_iter: int, log=None):
        self.b, self.task, self.chief_rank, self.training_iter = backend, task, chief_rank, training_iter
        self.shards, self.step_owner = ps_shards(task, backend)
        self.ranges = self.shards[task.task_index]
        self.n = sum(h - l for l, h in self.ranges)
        self.log = log or (lambda *a: None)
        self.applied = 0
        self.dropped: List[int] = []

    def broadcast_initial(self):
        broadcast_state(self.b, self.chief_rank)

    def serve(self):
        """One service thread per worker (blocking recv -> apply under a lock -> send), so a slow
        or dead worker never stalls the others; gradients are applied in arrival order."""
        import threading
        b, n = self.b, self.n
        dev = b.params.device
        workers = self.task.cluster.worker_ranks()
        lock = threading.Lock()
        self._step = b.get_global_step()
        b.grads.zero_()

        def service(w: int):
            buf = torch.zeros(n + 2, dtype=torch.float32, device=dev)
            resp = torch.zeros(n + 2, dtype=torch.float32, device=dev)
            while True:
                try:
                    comm.recv(buf, w)
                    kind = float(buf[n].item())
                    if kind == KIND_BYE:
                        return
                    with lock:
                        if kind == KIND_GRAD and self._step < self.training_iter:
                            _scatter(buf, self.ranges, b.grads)      # un-averaged apply, arrival order
                            b.apply_gradients(scale=1.0, step_inc=1)
                            self._step += 1
                            self.applied += 1
                        _gather(b.params, self.ranges, resp)
                        _set_int(resp, n, self._step)
                        _set_int(resp, n + 1, 1 if self._step >= self.training_iter else 0)
                    comm.send(resp, w)
                except (RuntimeError, ValueError) as e:     # worker died: keep serving the others
                    self.log(f"ps/{self.task.task_index}: worker rank {w} lost ({type(e).__name__}); continuing")
                    self.dropped.append(w)
                    return

        threads = [threading.Thread(target=service, args=(w,), daemon=True) for w in workers]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        b.set_global_step(self._step)
        return self._step


# ----------------------------------------------------------------------------- async, collectives
class AsyncRounds:
    """Parameter-server emulation with reduce + broadcast collectives (ps rank 0 owns the weights)."""

    name = "async-rounds"
    coordinated_stop = False

    def __init__(self, backend: Backend, task: TaskInfo, chief_rank: int, staleness: int = 1, optimizer: str = "sgd"):
        if staleness not in (0, 1):
            raise ValueError("staleness must be 0 or 1 rounds")
        if optimizer != "sgd":
            # one application of the SUM of W gradients equals W sequential un-averaged applications for SGD only;
            # with Adam it would silently be a different algorithm (and global_step would run W times ahead of t)
            raise ValueError("--async_impl=broadcast applies the summed gradient of a round in one update, which is "
                             "the parameter-server semantics for SGD only: use --async_impl=p2p with --optimizer adam")
        self.b, self.task, self.chief_rank, self.staleness = backend, task, chief_rank, staleness
        self.ps_rank = 0
        self.W = task.num_workers
        self.step_count = backend.get_global_step()
        # staleness 1 receives the next parameters in the background: into host memory when gloo carries a GPU model
        self._recv = torch.zeros_like(backend.params, device="cpu" if comm.staged(backend.params) else None)
        self._send = torch.zeros_like(backend.grads)
        self._pending = None

    def broadcast_initial(self):
        broadcast_state(self.b, self.chief_rank)
        self.step_count = self.b.get_global_step()

    # worker side -------------------------------------------------------------
    def step(self) -> int:
        b = self.b
        b.compute_gradients()
        b.advance_input()                  # no local update follows: move on to the next batch
        self._send.copy_(b.grads)
        if self._pending is not None:                 # parameters of the previous round land now
            self._pending.wait()
            b.params.copy_(self._recv)
            b.params_changed()
            self._pending = None
        comm.reduce_sum(self._send, self.ps_rank)
        if self.staleness == 0:
            comm.broadcast(b.params, self.ps_rank)
            b.params_changed()
        else:
            self._pending = dist.broadcast(self._recv, src=self.ps_rank, async_op=True)
        self.step_count += self.W
        b.set_global_step(self.step_count)
        return self.step_count

    def finish(self):
        if self._pending is not None:
            self._pending.wait()
            self.b.params.copy_(self._recv)
            self.b.params_changed()
            self._pending = None

    # ps side -----------------------------------------------------------------
    def serve(self, training_iter: int) -> int:
        b = self.b
        step = b.get_global_step()
        is_owner = self.task.task_index == 0
        while step < training_iter:
            b.grads.zero_()
            comm.reduce_sum(b.grads, self.ps_rank)
            if is_owner:
                # W sequential un-averaged SGD applications == one application of the sum
                b.apply_gradients(scale=1.0, step_inc=self.W)
            comm.broadcast(b.params, self.ps_rank)
            step += self.W
        b.set_global_step(step)
        return step
