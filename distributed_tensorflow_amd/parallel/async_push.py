"""Asynchronous parameter server as a native GPU path: peer-to-peer mailboxes instead of collectives.

The reference trains ONLY this way (reference .idea/MNISTDist.py:98-111, :174-188): every worker pushes its gradients
to the ps tasks, which apply them un-averaged in arrival order with no locking, and reads the variables back; the
shared ``global_step`` counts applied gradients (M:188).  Here the transport is csrc/kernels/ps.hip: every task maps
the mailboxes of the tasks it talks to (hipIpc over xGMI, or the same device when tasks share a GPU) and a message is
plain stores + a system-scope fence + sequence flags.

  * a worker never waits for another worker; the ps applies whatever has arrived, in the order its poll finds it;
  * the ps's answer carries what the worker's kernels READ -- the bf16 compute copy of wd1 (6.4 MB) and the fp32 tail
    (0.25 MB) -- not the 13.1 MB fp32 arena, and the worker installs it with one copy kernel + the tail's derived
    copies: no full-arena recast / repack per step;
  * ``staleness`` k: the worker computes gradient n + 1 on the answer to its gradient n - k.  k = 0 is the reference's
    pull-after-push; k >= 1 overlaps the round trip with compute.  k + 1 slots per mailbox make buffer reuse safe
    without further handshakes (csrc/kernels/ps.hip header);
  * variables are sharded over the ps tasks exactly as ``replica_device_setter`` places them (whole tensors,
    round-robin in creation order; cluster.py), each ps applies and answers for its own ranges;
  * the fp32 master weights -- and Adam's moments and update count -- live on the ps.  The chief fetches them with
    "master" requests before a checkpoint (``prepare_save``), the native form of the Saver reading the variables from
    the ps, so a restarted job resumes warm: the restored state reaches the ps in the initial broadcast;
  * every applied gradient's staleness (global_step now - global_step of the parameters it was computed with) is
    counted on the ps and logged as a histogram at the end.

Bring-up is collective and unanimous: if any task cannot allocate / export / map, every task raises and the trainer
falls back to the collective emulation (AsyncRounds).
"""
from __future__ import annotations

import ctypes
import time
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from ..backends import Backend
from ..cluster import TaskInfo
from ..ops import native
from .strategies import broadcast_state, ps_shards

KIND_GRAD, KIND_BYE, KIND_MASTER, KIND_ADAM_M, KIND_ADAM_V = 1, 2, 3, 4, 5


class PsLost(RuntimeError):
    """A parameter-server task did not answer within the bound (lost or stalled)."""
MAX_STALENESS = 8


def _align(n: int, a: int = 256) -> int:
    return (n + a - 1) // a * a


class _Layout:
    """Byte offsets inside a ps mailbox and inside a worker box (the same arithmetic on both sides of a mapping)."""

    def __init__(self, W: int, P: int, NB: int, n_total: int, n_dense: int, blocks: int):
        self.W, self.P, self.NB, self.n_total, self.n_dense, self.blocks = W, P, NB, n_total, n_dense, blocks
        # mailbox (owned by a ps): flags [W][NB][blocks] u32 | hdr [W][NB][4] u32 | data [W][NB][n_total] f32
        self.mb_flags = 0
        self.mb_hdr = _align(W * NB * blocks * 4)
        self.mb_data = self.mb_hdr + _align(W * NB * 16)
        self.mb_bytes = self.mb_data + W * NB * n_total * 4
        # box (owned by a worker): flags [P][NB][blocks] | hdr [P][NB][4] | full_flags [P][blocks] |
        #                          f32 [NB][n_total] | bf16 [NB][n_dense] | full [n_total] f32
        self.bx_flags = 0
        self.bx_hdr = _align(P * NB * blocks * 4)
        self.bx_full_flags = self.bx_hdr + _align(P * NB * 16)
        self.bx_f32 = self.bx_full_flags + _align(P * blocks * 4)
        self.bx_bf16 = self.bx_f32 + _align(NB * n_total * 4)
        self.bx_full = self.bx_bf16 + _align(NB * n_dense * 2)
        self.bx_bytes = self.bx_full + n_total * 4

    # mailbox addresses for worker w, slot s
    def mb(self, base: int, w: int, s: int) -> Tuple[int, int, int]:
        i = w * self.NB + s
        return (base + self.mb_flags + i * self.blocks * 4, base + self.mb_hdr + i * 16,
                base + self.mb_data + i * self.n_total * 4)

    # box addresses for ps p, slot s: flags, hdr, f32 slot, bf16 slot
    def bx(self, base: int, p: int, s: int) -> Tuple[int, int, int, int]:
        i = p * self.NB + s
        return (base + self.bx_flags + i * self.blocks * 4, base + self.bx_hdr + i * 16,
                base + self.bx_f32 + s * self.n_total * 4, base + self.bx_bf16 + s * self.n_dense * 2)

    def bx_master(self, base: int, p: int) -> Tuple[int, int]:
        return base + self.bx_full, base + self.bx_full_flags + p * self.blocks * 4


class PushFabric:
    """Allocation + hipIpc exchange of the mailboxes: ps tasks own mailboxes, workers own boxes; a ps maps every
    worker's box, a worker maps every ps's mailbox.  Collective over the default process group."""

    def __init__(self, backend: Backend, task: TaskInfo, staleness: int):
        if backend.name != "hip":
            raise ValueError("the peer-to-peer parameter server needs the hip backend")
        if not 0 <= staleness <= MAX_STALENESS:
            raise ValueError(f"--staleness must be in [0, {MAX_STALENESS}] for --async_impl=push")
        self.lib = native.load()
        self.b, self.task = backend, task
        self.eng = backend.engine
        self.k, self.NB = int(staleness), int(staleness) + 1
        self.W, self.P = task.num_workers, task.num_ps
        if self.W > self.lib.dtf_ps_max_workers():
            raise ValueError(f"at most {self.lib.dtf_ps_max_workers()} workers per ps mailbox")
        lay = self.eng.layout
        self.n_total = lay.total
        self.n_dense = lay.segment("wd1", "wd1")[1] if self.eng.dtype == "bf16" else 0
        self.blocks = self.lib.dtf_ps_blocks()
        self.L = _Layout(self.W, self.P, self.NB, self.n_total, self.n_dense, self.blocks)
        shards, self.step_owner = ps_shards(task, backend)
        # 16-byte granules: a tensor's padded extent belongs to the tensor (out_b has 10 floats, its range covers 12)
        self.ranges: Dict[int, List[Tuple[int, int]]] = {
            p: [(lo, lo + (hi - lo + 3) // 4 * 4) for lo, hi in r] for p, r in shards.items() if r}
        if self.step_owner not in self.ranges:      # more ps tasks than variables: the counter's ps holds no tensor
            self.step_owner = min(self.ranges)
        self._own = ctypes.c_void_p()
        self._mapped: List[int] = []
        self.bases: Dict[int, int] = {}        # rank -> base address as mapped here
        err: Optional[str] = None
        handle = ctypes.create_string_buffer(64)
        nbytes = self.L.mb_bytes if task.is_ps else self.L.bx_bytes
        with torch.cuda.device(self.eng.device):
            try:
                native.check(self.lib.dtf_p2p_alloc(nbytes, ctypes.byref(self._own)), "mailbox alloc")
                native.check(self.lib.dtf_p2p_export(self._own, handle), "hipIpcGetMemHandle")
            except Exception as e:          # keep making the collective calls: the peers are waiting in them
                err = repr(e)
            world = task.world_size
            handles: List[Optional[bytes]] = [None] * world
            dist.all_gather_object(handles, None if err else handle.raw)
            if err is None and any(h is None for h in handles):
                err = "a peer could not allocate/export its mailbox"
            if err is None:
                peers = task.cluster.worker_ranks() if task.is_ps else task.cluster.ps_ranks()
                try:
                    self.bases[task.rank] = self._own.value
                    for r in peers:
                        p = ctypes.c_void_p()
                        native.check(self.lib.dtf_p2p_import(handles[r], ctypes.byref(p)), f"hipIpcOpenMemHandle(rank {r})")
                        self.bases[r] = p.value
                        self._mapped.append(p.value)
                except Exception as e:
                    err = repr(e)
            oks: List[Optional[str]] = [None] * world
            dist.all_gather_object(oks, err)
        if any(o is not None for o in oks):
            self.close()
            raise RuntimeError("mailbox bring-up failed: " + "; ".join(f"rank {r}: {o}" for r, o in enumerate(oks) if o))
        dist.barrier()          # nobody pushes before everybody has mapped (and zeroed) everything

    def carr(self, p: int):
        r = self.ranges[p]
        lo = (ctypes.c_long * len(r))(*[a for a, _ in r])
        hi = (ctypes.c_long * len(r))(*[b for _, b in r])
        return len(r), lo, hi

    def close(self):
        for p in self._mapped:
            self.lib.dtf_p2p_unmap(p)
        self._mapped = []
        if self._own.value:
            self.lib.dtf_p2p_free(self._own)
            self._own = ctypes.c_void_p()


class AsyncPushWorker:
    name = "async-push"
    coordinated_stop = False      # asynchronous workers come and go on their own (reference M:174-176)

    def __init__(self, backend: Backend, task: TaskInfo, chief_rank: int, staleness: int = 0, timeout_s: float = 30.0,
                 look_ahead: bool = True):
        self.b, self.task, self.chief_rank = backend, task, chief_rank
        # staleness 0 (the reference's pull-after-push) has to tell the caller the global step that includes its own
        # update, i.e. it waits for the answer every step.  look_ahead: the kernels of the NEXT gradient are enqueued
        # before that wait (they follow the install in stream order, so they run on exactly the parameters they should),
        # and the wait is for the install's event only -- the GPU works through the host's launch latency instead of
        # idling in it.  A gradient enqueued this way after the last step is simply never pushed.
        self.look_ahead = bool(look_ahead) and int(staleness) == 0
        self._ahead = False
        self._installed_ev = None
        self.fab = PushFabric(backend, task, staleness)
        self.k = self.fab.k
        self.timeout_s = float(timeout_s)
        self.w = task.task_index
        self.seq = 0              # last message sent
        self.installed = 0        # last answer installed
        self.step_count = backend.get_global_step()
        self.stop = False
        self._closed = False
        eng = self.fab.eng
        dev = eng.device
        self._version = torch.zeros(1, dtype=torch.int64, device=dev)       # global_step of the installed parameters
        self._status = torch.zeros(1, dtype=torch.int32, device=dev)
        self._hdr = torch.zeros(8, dtype=torch.int32).pin_memory()   # {step lo, hi, stop, seq, Adam updates} from the owner ps

    def broadcast_initial(self):
        broadcast_state(self.b, self.chief_rank)
        self.step_count = self.b.get_global_step()
        self._version.fill_(self.step_count)

    # ------------------------------------------------------------------ messages
    def _push(self, n: int, kind: int):
        f, eng, L = self.fab, self.fab.eng, self.fab.L
        s = torch.cuda.current_stream(eng.device).cuda_stream
        for p in f.ranges:
            flags, hdr, data = L.mb(f.bases[self.task.cluster.ps_ranks()[p]], self.w, n % f.NB)
            nr, lo, hi = f.carr(p)
            native.check(f.lib.dtf_ps_push(eng.grads.data_ptr(), data, flags, hdr, self._version.data_ptr(), nr, lo, hi,
                                           f.n_total, n, kind, s), "ps_push")

    def _install(self, r: int, master: bool = False, master_dst: Optional[torch.Tensor] = None):
        """Wait for the answer to message r (bounded, on the GPU) and copy it into the arenas.  ``master``: the answer
        also carries an fp32 arena in the box's "full" area -- the master weights, or Adam's m / v into ``master_dst``."""
        f, eng, L = self.fab, self.fab.eng, self.fab.L
        s = torch.cuda.current_stream(eng.device).cuda_stream
        own = f.bases[self.task.rank]
        bf16 = f.n_dense > 0
        for p in f.ranges:
            flags, hdr, f32, b16 = L.bx(own, p, r % f.NB)
            full, full_flags = L.bx_master(own, p) if master else (None, None)
            nr, lo, hi = f.carr(p)
            native.check(f.lib.dtf_ps_install(flags, hdr, f32, b16 if bf16 else None, f.n_dense, nr, lo, hi, f.n_total, r,
                                              eng.params.data_ptr(), eng.shadow.data_ptr() if bf16 else None, full,
                                              full_flags, master_dst.data_ptr() if master_dst is not None else None,
                                              eng.state.data_ptr(), self._version.data_ptr(),
                                              self._hdr.data_ptr(), int(p == f.step_owner), self._status.data_ptr(),
                                              self.timeout_s, s), "ps_install")
        # derived copies of the (small) tail only: bf16 copy, conv2 fragment packs, transposed output weight
        if bf16:
            lo = f.n_dense
            native.check(f.lib.dtf_cast_bf16(eng.params[lo:].data_ptr(), eng.shadow[lo:].data_ptr(), f.n_total - lo, s),
                         "cast tail")
        eng.refresh_transposes()
        self.installed = r

    def _catch_up(self, upto: int):
        for r in range(self.installed + 1, upto + 1):
            self._install(r)

    def _read_header(self, sync: bool):
        if sync:
            torch.cuda.current_stream(self.fab.eng.device).synchronize()
        h = self._hdr
        # h[5]: an install kernel's wait ran out (mirrored by the kernel itself: with staleness k > 0 this thread never
        # synchronises, and must not go on pushing gradients computed on parameters that are no longer refreshed)
        if int(h[5]) or (sync and int(self._status.item())):
            raise PsLost("parameter server did not answer in time (lost or stalled ps task)")
        # the GPU may be rewriting the header right now (k > 0): the sequence word is written last, so the other words
        # are taken only between two equal, non-zero readings of it
        for _ in range(4):
            seq = int(h[3])
            if seq <= 0:
                return
            step, stop = (int(h[0]) & 0xFFFFFFFF) | ((int(h[1]) & 0xFFFFFFFF) << 32), bool(int(h[2]))
            if int(h[3]) == seq:
                self.step_count, self.stop = step, stop
                return

    # ------------------------------------------------------------------ strategy interface
    def step(self) -> int:
        n = self.seq + 1
        if not self._ahead:
            self.b.compute_gradients()             # on the parameters of answer n - 1 - k
            self.b.advance_input()                 # the update happens on the ps: move on to the next batch here
        self._ahead = False
        self._push(n, KIND_GRAD)
        self.seq = n
        self._catch_up(n - self.k)                 # answer n - k must be in before gradient n + 1 is computed
        # k = 0: the reference's semantics, the step returns the global step that includes this worker's update
        # (one wait per step, inherent in pull-after-push); k > 0: the host runs ahead and reads the newest header
        # that has landed
        if self.look_ahead and not self.stop:
            eng = self.fab.eng
            eng.freeze_metrics()                   # display lines report the batch of the gradient just pushed
            if self._installed_ev is None:
                self._installed_ev = torch.cuda.Event()
            self._installed_ev.record()
            self.b.compute_gradients()             # gradient n + 1, behind the install of answer n in stream order
            self.b.advance_input()
            self._ahead = True
            self._installed_ev.synchronize()       # the header of answer n is in pinned memory
            self._read_header(sync=False)
        else:
            self._read_header(sync=(self.k == 0))
        return self.step_count

    def prepare_save(self):
        """Chief, before a checkpoint: fetch the fp32 master weights from the ps tasks."""
        if self._closed:           # after finish(): the master is already here and the ps tasks have been told goodbye
            return
        self._catch_up(self.seq)
        eng = self.fab.eng
        requests = [(KIND_MASTER, None)]
        if eng.optimizer == "adam":        # the moments live on the ps too: without them a restart would resume cold
            requests += [(KIND_ADAM_M, eng.adam_m), (KIND_ADAM_V, eng.adam_v)]
        for kind, dst in requests:
            n = self.seq + 1
            self._push(n, kind)
            self.seq = n
            self._install(n, master=True, master_dst=dst)
        self._read_header(sync=True)
        if eng.optimizer == "adam":
            self.b.set_adam_t(int(self._hdr[4]) & 0xFFFFFFFF)

    def finish(self):
        self._catch_up(self.seq)
        self._read_header(sync=True)
        if self.task.is_chief:
            self.prepare_save()                    # the final checkpoint / parameter dump wants the fp32 master
        self._push(self.seq + 1, KIND_BYE)
        self.seq += 1
        self._closed = True
        torch.cuda.current_stream(self.fab.eng.device).synchronize()
        self.b.set_global_step(self.step_count)


class AsyncPushServer:
    """One ps task: polls its mailbox, applies gradients in the order the poll finds them, answers each worker."""

    def __init__(self, backend: Backend, task: TaskInfo, chief_rank: int, training_iter: int, staleness: int = 0,
                 log=None, peer_lost: Optional[Callable[[int], bool]] = None, budget_s: float = 0.02, idle_s: float = 0.002):
        self.b, self.task, self.chief_rank, self.training_iter = backend, task, chief_rank, training_iter
        # one launch of the service kernel runs at most budget_s and leaves early after idle_s without a message
        import os
        self.budget_s = float(os.environ.get("DTF_PS_BUDGET_S", budget_s))      # (tuning overrides for tools/async_path_cost.py)
        self.idle_s = float(os.environ.get("DTF_PS_IDLE_S", idle_s))
        self.fab = PushFabric(backend, task, staleness)
        self.log = log or (lambda *a: None)
        self.peer_lost = peer_lost
        self.applied = 0
        self.dropped: List[int] = []
        self.staleness_hist: Dict[int, int] = {}

    def broadcast_initial(self):
        broadcast_state(self.b, self.chief_rank)

    def serve(self) -> int:
        """The service loop lives on the GPU (csrc/kernels/ps.hip ps_serve): one launch polls the mailboxes, applies
        whatever has arrived in arrival order, answers and counts, for as long as messages keep coming (at most
        ``budget`` seconds, or until nothing has arrived for ``idle``).  This thread only relaunches it, reads the
        counters it leaves in pinned memory, and looks after what is not the GPU's business: lost workers."""
        f, eng, L = self.fab, self.fab.eng, self.fab.L
        p = self.task.task_index
        if p not in f.ranges:          # more ps tasks than variables: nothing to own, nobody writes to this task
            self.log(f"ps/{p}: owns no variable; nothing to serve")
            return self.b.get_global_step()
        lib = f.lib
        W, NB = f.W, f.NB
        own = f.bases[self.task.rank]
        wr = self.task.cluster.worker_ranks()
        stream = torch.cuda.current_stream(eng.device)
        s = stream.cuda_stream
        adam = eng.optimizer == "adam"
        adam_t0 = self.b.get_adam_t() if adam else 0
        nr, lo, hi = f.carr(p)
        ctl = ctypes.c_void_p()
        native.check(lib.dtf_p2p_alloc(int(lib.dtf_ps_ctl_bytes()), ctypes.byref(ctl)), "ps control block")
        out = torch.zeros(int(lib.dtf_ps_out_words()), dtype=torch.int32).pin_memory()
        boxes = (ctypes.c_void_p * W)(*[f.bases[wr[w]] for w in range(W)])
        layout9 = (ctypes.c_long * 9)(L.mb_flags, L.mb_hdr, L.mb_data, L.bx_flags, L.bx_hdr, L.bx_full_flags, L.bx_f32,
                                      L.bx_bf16, L.bx_full)
        native.check(lib.dtf_ps_serve_init(ctl, W, self.b.get_global_step(), s), "ps_serve_init")
        round0, dropped, last_check = 0, 0, time.monotonic()
        step = self.b.get_global_step()
        try:
            while True:
                native.check(lib.dtf_ps_serve(int(adam), ctl, own, boxes, W, NB, p, f.n_total, f.n_dense, layout9, nr, lo, hi,
                                              eng.params.data_ptr(), eng.adam_m.data_ptr() if adam else None,
                                              eng.adam_v.data_ptr() if adam else None, eng.hyper.data_ptr(), adam_t0,
                                              self.training_iter, round0, self.budget_s, self.idle_s, out.data_ptr(), s),
                             "ps_serve")
                stream.synchronize()
                o = [int(x) & 0xFFFFFFFF for x in out.tolist()]
                round0, step, self.applied = o[0], o[1] | (o[2] << 32), o[3]
                if o[5]:
                    raise RuntimeError(f"ps/{p}: the service kernel gave up on a stuck workgroup")
                if o[4]:
                    break
                now = time.monotonic()
                if self.peer_lost is not None and now - last_check > 1.0:
                    last_check = now
                    for w in range(W):
                        if o[8 + w] and not (dropped >> w) & 1 and self.peer_lost(wr[w]):
                            self.log(f"ps/{p}: worker rank {wr[w]} lost; continuing")     # reference: workers come and go
                            self.dropped.append(wr[w])
                            dropped |= 1 << w
                    native.check(lib.dtf_ps_ctl_set_dropped(ctl, dropped, s), "ps_ctl_set")
            hist = o[24:24 + 64]
            self.staleness_hist = {d: n for d, n in enumerate(hist) if n}
        finally:
            stream.synchronize()
            lib.dtf_p2p_free(ctl)
        stream.synchronize()
        self.b.set_global_step(step)
        if adam:
            self.b.set_adam_t(adam_t0 + self.applied)
        if self.staleness_hist:
            tot = sum(self.staleness_hist.values())
            mean = sum(k * v for k, v in self.staleness_hist.items()) / tot
            self.log(f"ps/{p}: applied {self.applied} gradients; staleness (global steps between the parameters a "
                     f"gradient was computed with and its application): mean {mean:.2f}, histogram "
                     + " ".join(f"{k}:{v}" for k, v in sorted(self.staleness_hist.items())))
        return step
