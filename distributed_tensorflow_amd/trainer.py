"""Training driver: the reference's ``main`` (reference .idea/MNISTDist.py:93-193) re-designed.

One process per task.  ``run(flags)`` resolves the role from (job_name, task_index), brings up
the process group, builds the backend (MI355X engine or CPU oracle), the distribution strategy
(sync all-reduce / async parameter server / local) and the supervisor, then runs

    while not sv.should_stop() and step < training_iter:
        step = strategy.step()            # sampler + fwd + bwd + collective + update
        every display_step: print the reference's console line (+ throughput to JSONL)
        sv.maybe_save()

``training_iter`` bounds the GLOBAL step as in the reference (M:173, M:188): with W async
workers each performs about training_iter / W updates; in sync mode one global step is one
collective update over the global batch.
"""
from __future__ import annotations

import os
import sys
import time
from typing import Optional

import torch
import torch.distributed as dist

from .backends import make_backend
from .cluster import ClusterSpec, TaskInfo, single_process_cluster
from .data.mnist import read_data_sets
from .models.spec import ModelSpec
from .parallel import comm
from .parallel.liveness import EXIT_PEER_LOST, Liveness, PeerLost
from .parallel.strategies import (AsyncP2PServer, AsyncP2PWorker, AsyncRounds, LocalStrategy, SyncAllReduce)
from .checkpoint import Saver
from .supervisor import Supervisor
from .utils import trace
from .utils.metrics import MetricsLogger, StepTimer, format_display_line


class FaultInjected(RuntimeError):
    pass


def resolve_task(F) -> TaskInfo:
    if not F.job_name and not F.worker_hosts and not F.ps_hosts:
        return single_process_cluster()
    cluster = ClusterSpec.from_flags(F.ps_hosts, F.worker_hosts)
    if F.job_name not in ("ps", "worker"):
        # the reference hits a NameError on sv.stop() here; fail clearly instead
        raise SystemExit(f"--job_name must be 'ps' or 'worker', got {F.job_name!r}")
    return TaskInfo(cluster, F.job_name, F.task_index)


def _pick_backend(F) -> str:
    """--backend: hip (hand-written gfx950 kernels) | cpu (PyTorch oracle + gloo plumbing) | auto.
    --dtype selects the precision of the HIP engine's own kernel sets -- bf16 MFMA with fp32 master weights, or
    the exact-fp32 MFMA set (the reference's float32, M:113-115); it never selects another backend."""
    dtype = getattr(F, "dtype", "bf16")
    if dtype not in ("bf16", "fp32"):
        raise SystemExit(f"--dtype must be bf16 or fp32, got {dtype!r}")
    if F.backend not in ("auto", "hip", "cpu"):
        raise SystemExit(f"--backend must be auto, hip or cpu, got {F.backend!r} "
                         "(the ATen/MIOpen reference lives in bench/torch_reference.py only)")
    if F.backend != "auto":
        return F.backend
    return "hip" if torch.cuda.is_available() else "cpu"


def run(F, log=print) -> dict:
    task = resolve_task(F)
    kind = _pick_backend(F)
    world = task.world_size
    mode = F.mode
    if world == 1:
        mode = "local"
    if mode == "async" and task.num_ps == 0:
        raise SystemExit("--mode=async needs at least one ps task (--ps_hosts)")
    grad_comm = getattr(F, "grad_comm", "fp32")
    if grad_comm not in ("fp32", "bf16"):
        raise SystemExit(f"--grad_comm must be fp32 or bf16, got {grad_comm!r}")
    if grad_comm == "bf16" and (kind != "hip" or getattr(F, "dtype", "bf16") != "bf16" or mode != "sync"):
        # every task checks this before the rendezvous, so a refused job ends everywhere at once
        raise SystemExit("--grad_comm=bf16 needs --mode=sync with --backend=hip and --dtype=bf16 (more than one task)")

    if kind == "hip":
        local_rank = int(os.environ.get("LOCAL_RANK", F.local_rank if F.local_rank >= 0 else task.rank))
        local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")

    live = None
    if world > 1:
        host, port = task.cluster.master()
        cb = getattr(F, "comm_backend", "auto")
        if cb not in ("auto", "gloo", "nccl"):
            raise SystemExit(f"--comm_backend={cb!r}: expected auto, gloo or nccl")
        # gloo moves GPU tensors through the host: slow, but it lets several tasks share one GPU (RCCL refuses
        # duplicate devices), which is how the GPU suite runs the multi-process modes on a one-GPU box
        backend_name = "gloo" if (kind == "cpu" or cb == "gloo") else ("nccl" if cb == "nccl" else "cpu:gloo,cuda:nccl")
        comm.init_process_group(backend_name, task.rank, world, master_addr=host, master_port=port,
                                device=device if (kind != "cpu" and backend_name != "gloo") else None, timeout_s=F.comm_timeout)
        # Failure detection on a control plane of its own.  Synchronous jobs: every task depends on
        # every worker (a lost worker can never be waited out) and on the task hosting the store.  Asynchronous jobs
        # keep the reference's contract: workers come and go, only the ps tasks are indispensable.
        if mode == "sync":
            watch = task.cluster.worker_ranks() + [0]
        else:
            watch = task.cluster.ps_ranks() if task.is_worker else []
        live = Liveness(host, port, task.rank, watch, name=f"{task.job_name}/{task.task_index}",
                        timeout_s=F.heartbeat_timeout, grace_s=min(10.0, F.heartbeat_timeout), log=log)

    spec = ModelSpec.from_name(getattr(F, "width", "") or F.model)
    data = read_data_sets(F.data_dir, one_hot=(kind == "cpu"), synthetic=F.synthetic,
                          seed=F.seed + 1000 * task.rank)   # every worker shuffles independently (ref. item 9)
    backend = make_backend(kind, spec, F.batch_size, data, optimizer=F.optimizer, learning_rate=F.learning_rate,
                           keep_prob=F.keep_prob, seed=F.seed + task.rank, device=device, use_graph=F.hip_graph,
                           dtype=getattr(F, "dtype", "bf16"))

    chief_rank = task.cluster.rank_of("worker", 0)
    worker_group = tail_group = None
    if world > 1:
        # collective calls: every rank (ps included) makes them, in the same order
        worker_group = dist.new_group(task.cluster.worker_ranks())
        tail_group = dist.new_group(task.cluster.worker_ranks())     # 2nd communicator: concurrent small all-reduce

    async_impl = F.async_impl
    if async_impl not in ("auto", "push", "p2p", "broadcast"):
        raise SystemExit(f"--async_impl={async_impl!r}: expected auto, push, p2p or broadcast")
    if async_impl == "auto":
        async_impl = "p2p" if kind == "cpu" else "push"
    staleness = int(getattr(F, "staleness", -1))
    if mode == "async":
        if async_impl == "push" and kind != "hip":
            raise SystemExit("--async_impl=push moves gradients with GPU kernels: it needs --backend=hip")
        if async_impl == "p2p" and dist.is_initialized() and dist.get_backend() != "gloo":
            # one Python thread per worker in blocking send/recv: not something RCCL's stream-ordered send/recv supports
            raise SystemExit("--async_impl=p2p serves every worker from a thread of its own with blocking send/recv, which "
                             "works over gloo only: use --async_impl=push (peer-to-peer kernels) or --comm_backend=gloo")
        if staleness < 0:
            staleness = 0 if async_impl == "push" else 1
    push_fallback = None
    if mode == "async" and async_impl == "push":
        # Bring-up is collective and unanimous (every task raises or none does), so every task falls back together.
        from .parallel.async_push import AsyncPushServer, AsyncPushWorker
        try:
            if task.is_ps:
                push_obj = AsyncPushServer(backend, task, chief_rank, F.training_iter, staleness=staleness, log=log,
                                           peer_lost=live.peer_lost if live is not None else None)
            else:
                push_obj = AsyncPushWorker(backend, task, chief_rank, staleness=staleness,
                                           timeout_s=min(float(F.comm_timeout), 30.0))
        except RuntimeError as e:
            push_fallback = str(e)
            log(f"{task.job_name}/{task.task_index}: peer-to-peer parameter server unavailable ({push_fallback[:200]}); "
                "falling back to --async_impl=broadcast")
            async_impl, staleness = "broadcast", min(staleness, 1)

    metrics = MetricsLogger(F.metrics_file or None)
    result = dict(role=f"{task.job_name}/{task.task_index}", mode=mode, backend=kind)

    # ------------------------------------------------------------------ ps role
    if task.is_ps:
        if mode == "async" and getattr(F, "lr_scaling", "none") != "none":
            # the ps applies the updates: the rate for one worker's batch, no warm-up (updates arrive unordered)
            from .utils.lr_schedule import scaled_rate
            backend.set_learning_rate(scaled_rate(F.learning_rate, F.batch_size, getattr(F, "lr_base_batch", 128), F.lr_scaling))
        log(f"ps/{task.task_index}: serving ({mode}{'/' + async_impl if mode == 'async' else ''}) "
            f"for {task.num_workers} worker(s)")
        if mode == "async" and async_impl == "push":
            push_obj.broadcast_initial()
            result["global_step"] = push_obj.serve()
            result["applied"], result["dropped_workers"] = push_obj.applied, push_obj.dropped
            result["staleness_hist"] = dict(push_obj.staleness_hist)
            push_obj.fab.close()
        elif mode == "async" and async_impl == "p2p":
            srv = AsyncP2PServer(backend, task, chief_rank, F.training_iter, log=log)
            srv.broadcast_initial()
            result["global_step"] = srv.serve()
            result["applied"], result["dropped_workers"] = srv.applied, srv.dropped
        elif mode == "async":
            srv = AsyncRounds(backend, task, chief_rank, staleness=staleness, optimizer=F.optimizer)
            srv.broadcast_initial()
            result["global_step"] = srv.serve(F.training_iter)
        else:
            # sync mode: parameters are replicated on the workers; the ps task has nothing to
            # serve.  It takes part in the initial broadcast and then blocks like server.join()
            # (reference M:105-106) until the workers are done.
            # (on a store key, not in a collective: a barrier parked for the whole run would hit the collective
            # timeout of a long job and turn a good run into a failed one)
            from .parallel.strategies import broadcast_state
            broadcast_state(backend, chief_rank)
            try:
                live.wait_flag("workers_done")
            except PeerLost as e:
                log(f"ps/{task.task_index}: {e}; the synchronous job cannot continue -- exiting")
                live.close(orderly=False)
                raise SystemExit(EXIT_PEER_LOST)
        log(f"ps/{task.task_index}: done")
        _finish_control_plane(live, task, world)
        comm.shutdown()
        return result

    # -------------------------------------------------------------- worker role
    if mode == "local":
        strategy = LocalStrategy(backend)
    elif mode == "sync":
        strategy = SyncAllReduce(backend, task, worker_group, chief_rank, tail_group=tail_group,
                                 p2p=bool(getattr(F, "p2p_allreduce", False)), log=log,
                                 stop_check_steps=F.health_check_steps,
                                 shard_dense={"auto": None, "sharded": True, "replicated": False}[
                                     getattr(F, "dense_update", "auto")],
                                 grad_comm=getattr(F, "grad_comm", "fp32"), hip_graph=bool(F.hip_graph),
                                 p2p_timeout_s=float(getattr(F, "p2p_timeout", 60.0)))
    elif async_impl == "push":
        strategy = push_obj
    elif async_impl == "p2p":
        strategy = AsyncP2PWorker(backend, task, chief_rank)
    else:
        strategy = AsyncRounds(backend, task, chief_rank, staleness=staleness, optimizer=F.optimizer)

    if mode == "async" and F.optimizer == "adam" and task.is_chief and async_impl != "push":
        log("note: in asynchronous mode the Adam moments live on the ps tasks and are not part of the chief's "
            "checkpoints; a restarted job resumes the weights and global_step with fresh moments")
    sv = Supervisor(backend, is_chief=task.is_chief, logdir=F.logdir or None, save_model_secs=F.save_model_secs,
                    saver=Saver(write_version=getattr(F, "saver_write_version", 2)), init_seed=F.seed, log=log,
                    # asynchronous modes: the moments live on the ps; only the push path fetches them for checkpoints
                    save_optimizer_state=(mode != "async" or async_impl == "push"))
    health = getattr(strategy, "check_health", None)
    sv.pre_save_check = health       # never checkpoint parameters that a timed-out peer-to-peer sum may have touched
    _install_signal_handlers(sv, log, task)
    timer = StepTimer()
    eval_every = int(getattr(F, "eval_every", 0) or 0)
    tracing = trace.set_enabled(True) if getattr(F, "trace", False) else trace.enabled()
    images_per_step = F.batch_size * (task.num_workers if mode == "sync" else 1)
    # large-batch rule (utils/lr_schedule.py): the global batch of a synchronous job is batch_size x workers
    from .utils.lr_schedule import LrSchedule
    lr_sched = LrSchedule(F.learning_rate, images_per_step, getattr(F, "lr_base_batch", 128),
                          getattr(F, "lr_scaling", "none"), getattr(F, "lr_warmup_steps", 0))
    if lr_sched.peak_lr != F.learning_rate:
        log(f"{task.job_name}/{task.task_index}: learning rate {F.learning_rate:g} x ({images_per_step}/"
            f"{getattr(F, 'lr_base_batch', 128)}, {lr_sched.rule}) = {lr_sched.peak_lr:g}"
            + (f", warm-up over {lr_sched.warmup_steps} steps" if lr_sched.warmup_steps else ""))
    if mode == "async" and lr_sched.peak_lr != F.learning_rate:
        log("note: asynchronous mode applies every worker's gradient separately (the global batch is one worker's batch)")
    t_start = time.perf_counter()
    try:
        _train_loop(F, task, mode, backend, strategy, sv, live, timer, metrics, result, images_per_step, eval_every,
                    tracing, health, log, lr_sched)
    except SystemExit:
        raise
    except BaseException as e:
        # Tell the peers now (they would otherwise find out after the heartbeat timeout), say what happened in one
        # line, and leave with a status of its own.  A lost peer surfaces here either as PeerLost (the monitor saw
        # it first) or as the collective's own error (gloo: connection reset; RCCL: watchdog timeout).
        if live is not None:
            live.announce_death(f"{type(e).__name__}: {e}")
            lost = _is_peer_loss(e, mode)
            if lost:
                log(f"{task.job_name}/{task.task_index}: lost contact with a peer in {mode} mode ({type(e).__name__}: "
                    f"{str(e).splitlines()[0][:200]}); aborting.  Restart the job: it resumes from the last checkpoint "
                    f"in {F.logdir or '<no logdir>'}.")
                live.close(orderly=False)
                sys.stdout.flush()
                os._exit(EXIT_PEER_LOST)      # a wedged communicator must not be waited for in destructors
            live.close(orderly=False)
        raise
    result["saved"] = list(sv.saved)
    result["restored_from"] = sv.restored_from
    if hasattr(strategy, "fab"):
        strategy.fab.close()         # every answer has been installed and the ps tasks have this worker's goodbye
    if world > 1:
        if mode == "sync" and task.is_chief:
            live.set_flag("workers_done")            # releases the ps tasks blocked in their join
        _finish_control_plane(live, task, world)
        comm.shutdown()
    metrics.close()
    log("Optimization Finished!")    # reference M:193
    return result


def _is_peer_loss(e: BaseException, mode: str) -> bool:
    """Only what a lost or stalled peer produces ends the job with the "lost contact" message and exit status 3:
    the heartbeat monitor's PeerLost, the exchange fabric's health error, and the process group's own failures
    (gloo: connection reset / closed by peer; RCCL: DistBackendError, watchdog timeouts).  Everything else -- a failed
    checkpoint write, a launch error, a bug -- propagates with its traceback: a restart would only loop on it."""
    from .parallel.dp_fabric import P2PHealthError
    if isinstance(e, (PeerLost, P2PHealthError)):
        return True
    if mode != "sync":
        return False
    dist_errors = tuple(t for t in (getattr(dist, "DistBackendError", None), getattr(dist, "DistNetworkError", None),
                                    getattr(dist, "DistError", None)) if isinstance(t, type))
    if dist_errors and isinstance(e, dist_errors):
        return True
    if isinstance(e, (ConnectionError, TimeoutError)):
        return True
    msg = str(e)
    return isinstance(e, RuntimeError) and any(k in msg for k in (
        "Connection reset", "Connection closed", "closed by peer", "Socket Timeout", "Timed out waiting", "timed out",
        "Broken pipe", "NCCL", "RCCL", "ProcessGroup", "connection refused", "Connection refused"))


def _finish_control_plane(live, task: TaskInfo, world: int):
    """Orderly exit: mark this task done; the task hosting the store (rank 0) leaves last."""
    if live is None:
        return
    live.close(orderly=True)
    if task.rank == 0:
        t0 = time.monotonic()
        while time.monotonic() - t0 < 20.0:
            try:
                if all(live.store.check([f"dtf/done/{r}"]) for r in range(world)):
                    break
            except Exception:
                break
            time.sleep(0.05)


def _install_signal_handlers(sv, log, task: TaskInfo):
    """SIGTERM / SIGINT ask for an orderly stop (checkpoint, then exit) instead of killing the job mid-step: in
    synchronous mode the request is job-wide (every worker stops after the same step)."""
    import signal
    import threading
    if threading.current_thread() is not threading.main_thread():
        return

    def handler(signum, _frame):
        log(f"{task.job_name}/{task.task_index}: signal {signum}: requesting stop")
        sv.request_stop()

    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(sig, handler)
        except (ValueError, OSError):
            pass


def _train_loop(F, task, mode, backend, strategy, sv, live, timer, metrics, result, images_per_step, eval_every,
                tracing, health, log, lr_sched=None):
    t_start = time.perf_counter()
    with sv.managed_session(strategy):
        step = strategy.step_count
        result["start_step"] = step
        local_steps = 0
        cur_lr = None
        while not sv.should_stop() and step < F.training_iter:
            if live is not None:
                live.raise_if_failed()
            prev = step
            if lr_sched is not None:
                lr = lr_sched(step)            # a function of the global step: the same on every synchronous worker
                if lr != cur_lr:
                    # (a strategy with updates in flight on other streams orders them first)
                    getattr(strategy, "set_learning_rate", backend.set_learning_rate)(lr)
                    cur_lr = lr
            if getattr(F, "fault_mute_step", -1) >= 0 and local_steps == F.fault_mute_step and \
                    (F.fault_inject_task < 0 or F.fault_inject_task == task.task_index) and \
                    getattr(strategy, "_fabric", None) is not None:
                log(f"worker/{task.task_index}: injected stall: the next step raises no peer-to-peer flags")
                strategy._fabric.mute_step(strategy._fabric.next_seq())
            if tracing: trace.push(f"step {prev}")
            step = strategy.step()
            if tracing: trace.pop()
            local_steps += 1
            timer.add(images_per_step)
            if F.fault_inject_step >= 0 and local_steps == F.fault_inject_step and \
                    (F.fault_inject_task < 0 or F.fault_inject_task == task.task_index):
                log(f"worker/{task.task_index}: injected fault at local step {local_steps}")
                os._exit(17)
            if F.request_stop_step >= 0 and local_steps == F.request_stop_step and \
                    (F.request_stop_task < 0 or F.request_stop_task == task.task_index):
                log(f"worker/{task.task_index}: requesting stop at local step {local_steps}")
                sv.request_stop()
            if not getattr(strategy, "coordinated_stop", False) and getattr(strategy, "stop", False):
                sv.stop()                                  # async: the ps says the step budget is used up
            if health is not None and local_steps % F.health_check_steps == 0:
                health()                                   # a timed-out peer-to-peer wait ends the run, not the epoch
            if prev % F.display_step == 0:
                loss, acc = backend.last_metrics()       # batch `prev`, pre-update weights (= ref. M:182)
                log(format_display_line(task.job_name, task.task_index, prev, loss, acc))
                lap = timer.lap()
                metrics.log(step=prev, loss=loss, accuracy=acc, role=result["role"], **lap)
                if health is not None:
                    health()
            if eval_every > 0 and task.is_chief and step // eval_every != prev // eval_every \
                    and not (F.eval and step >= F.training_iter):      # the final evaluation covers the last step
                backend.synchronize()    # a deferred dense update (parallel/sync_dp.py) may still be on the comm stream
                with trace.range("evaluate test"):
                    tl, ta = backend.evaluate("test")    # other workers keep training in async mode, wait in sync mode
                log(f"test accuracy: {ta:.4f}  test loss: {tl:.4f}  (global_step {step})")
                metrics.log(step=step, test_loss=tl, test_accuracy=ta, role=result["role"])
            if tracing: trace.push("supervisor.maybe_save")
            sv.maybe_save()
            if tracing: trace.pop()
        strategy.finish()
        backend.synchronize()
        if getattr(F, "dump_params", ""):
            import numpy as np
            np.save(F.dump_params, backend.params.detach().cpu().numpy()[:backend.layout.ctrl])
        result["global_step"] = step
        result["local_steps"] = local_steps
        result["train_seconds"] = time.perf_counter() - t_start
        loss, acc = backend.last_metrics() if local_steps else (float("nan"), float("nan"))
        result["final_loss"], result["final_train_accuracy"] = loss, acc
        if F.eval and task.is_chief:
            with trace.range("evaluate test"):
                tl, ta = backend.evaluate("test")
            result["test_loss"], result["test_accuracy"] = tl, ta
            log(f"test accuracy: {ta:.4f}  test loss: {tl:.4f}  (global_step {step})")
            metrics.log(step=step, test_loss=tl, test_accuracy=ta, role=result["role"])
