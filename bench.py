#!/usr/bin/env python3
"""Headline benchmark: training throughput (images/s, whole job) of the deep-MNIST
CNN on N MI355X GPUs, synchronous data parallel over RCCL.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

Metric / config follow BASELINE.json ("train images/sec (whole node) at 1/2/4/8
MI355X", sync DP, global batch 8192 at 8 GPUs): weak scaling with a fixed
per-GPU batch of 1024, bf16 compute with fp32 master weights, the reference's
optimizer (plain SGD, lr 1e-3, dropout keep_prob 1.0; reference
.idea/MNISTDist.py:30,149,179), HBM-resident synthetic 28x28x1 data, random
TF-style init.  Every step runs sampler + forward + loss + backward +
gradient all-reduce + update + global_step increment; nothing is skipped.
"""
import argparse
import json
import os
import sys
import time

# The contract is ONE line on stdout.  Native libraries write there too (RCCL prints a version banner on stdout when
# its first communicator comes up), so file descriptor 1 is pointed at stderr for the life of the process and the
# result line goes to a private duplicate of the real stdout.
_RESULT_FD = 1
if __name__ == "__main__":
    _RESULT_FD = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--per_gpu_batch", type=int, default=1024)
    ap.add_argument("--model", default="deep_mnist")
    ap.add_argument("--optimizer", default="sgd")
    ap.add_argument("--keep_prob", type=float, default=1.0)
    ap.add_argument("--dataset_size", type=int, default=60000)
    ap.add_argument("--prewarm_s", type=float, default=0.3,
                    help="time-based pre-warm before --warmup/--steps, outside the timed region (reported as prewarm_steps)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"],
                    help="bf16: bf16 MFMA compute copies, fp32 accumulation and fp32 master weights (default; BASELINE.json "
                         "config 2 names fp32/bf16).  fp32: the reference's precision end to end -- exact-fp32 MFMA kernels, "
                         "fp32 activations, no shadow copies (reference .idea/MNISTDist.py:113-115)")
    ap.add_argument("--single_stream", action="store_true", help="disable the fork/join side streams in backward")
    ap.add_argument("--multi_stream", action="store_true", help="force the fork/join side streams (default: by batch size)")
    ap.add_argument("--no_graph", action="store_true", help="eager launches instead of HIP-graph replay (1 GPU)")
    ap.add_argument("--graph", action="store_true", help="force HIP-graph replay (default: only above 1024 images/GPU)")
    ap.add_argument("--dense", choices=["auto", "allreduce", "sharded"], default="auto",
                    help="N > 1, dense (fc1) gradient over RCCL: all-reduce + replicated update, or reduce-scatter + "
                         "update of 1/N + all-gather of the compute copy; auto times one 50-step trial of each")
    ap.add_argument("--force_dp", action="store_true",
                    help="diagnostic: run the data-parallel schedule (process group, two communicators, comm_diag) "
                         "even at --gpus 1, to exercise the N>1 code path on a one-GPU box")
    ap.add_argument("--no_p2p_tail", action="store_true",
                    help="N>1: keep the small tail all-reduce on RCCL instead of the one-shot peer-to-peer kernel")
    ap.add_argument("--grad_comm", choices=["fp32", "bf16"], default="fp32",
                    help="N > 1 / --force_dp, bf16 configuration: bf16 = opt-in 16-bit collective for the fc1 weight gradient "
                         "(NOT the reference's fp32 gradient sum; recorded in config.grad_comm)")
    ap.add_argument("--no_p2p_dense", action="store_true",
                    help="N>1: keep the dense all-reduce on RCCL even if the two-shot peer-to-peer kernel measures faster")
    ap.add_argument("--rehearse_one_gpu", action="store_true",
                    help="diagnostic: every rank on cuda:0 over gloo -- rehearses the torchrun N>1 launch of this file "
                         "on a one-GPU box (RCCL refuses ranks that share a device); the number means nothing")
    args = ap.parse_args()

    from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
    from distributed_tensorflow_amd.models.spec import ModelSpec
    from distributed_tensorflow_amd.parallel.sync_dp import SyncDataParallel
    from distributed_tensorflow_amd.parallel import comm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    if args.rehearse_one_gpu:
        import faulthandler
        faulthandler.dump_traceback_later(120, exit=True)   # a rehearsal that stalls says where
        local_rank = 0
    ndev = torch.cuda.device_count()
    if ndev and local_rank >= ndev:      # launcher restricted this rank's visible devices (e.g. one GPU per process)
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dp_path = world > 1 or args.force_dp
    if dp_path:
        comm.init_process_group(backend="gloo" if args.rehearse_one_gpu else "nccl", rank=rank, world_size=world,
                                device=None if args.rehearse_one_gpu else torch.device("cuda", local_rank))

    spec = ModelSpec.from_name(args.model)
    B = args.per_gpu_batch
    ds = DeviceDataset.synthetic(args.dataset_size, seed=1234, dtype=args.dtype)
    eng = HipEngine(spec, B, ds, optimizer=args.optimizer, learning_rate=1e-3, keep_prob=args.keep_prob,
                    seed=1000 + rank,
                    # None = engine default (by batch size); the data-parallel path measured faster without the fork/join
                    multi_stream=False if (args.single_stream or dp_path) else (True if args.multi_stream else None),
                    fuse_dense_update=not dp_path, dtype=args.dtype)
    eng.init_params(seed=7)

    use_graph = False
    tail_p2p, tail_how = None, "n/a"
    dense_p2p, dense_how = None, "n/a"
    p2p_blocks_us = dense_cand = tail_trials = None
    if dp_path:
        try:   # second communicator: the small tail all-reduce must not queue behind the 12.8 MB one inside RCCL
            tail_group = torch.distributed.new_group()
        except Exception as e:   # still correct without it, only slower: the tail then rides the comm stream
            print(f"[bench] rank {rank}: no second communicator ({e}); tail all-reduce shares the first", file=sys.stderr)
            tail_group = None
        # The latency-bound tail message (0.25 MB) goes through the one-shot peer-to-peer kernel (parallel/p2p.py) if
        # its bring-up works on this node AND it reproduces the process group's sums on every rank; otherwise RCCL.
        tail_p2p, tail_how = None, "rccl"
        dense_p2p, dense_how = None, "rccl"
        dev = torch.device("cuda", local_rank)
        n_dense_arena = eng.layout.segment("wd1", "wd1")[1]
        n_tail_arena = eng.layout.total - n_dense_arena

        def bring_up(n, two_shot):
            """-> (P2PAllReduce or None, why not)"""
            try:
                from distributed_tensorflow_amd.parallel.p2p import P2PAllReduce
                # bounded waits: a dead path costs seconds.  (A rehearsal's ranks share ONE GPU, where a spinning
                # exchange kernel can keep a peer's kernel off the device for many seconds: a longer bound there.)
                cand = P2PAllReduce(n, dev, timeout_s=30.0 if args.rehearse_one_gpu else 2.0, two_shot=two_shot)
                if n % cand.quantum:
                    cand.close()
                    return None, "rccl (length not a multiple of 4 x world)"
                if cand.self_check(n, iters=4):
                    return cand, ""
                cand.close()
                return None, "rccl (p2p self-check failed)"
            except Exception as e:      # bring-up votes are unanimous, so every rank lands here together
                print(f"[bench] rank {rank}: {e}", file=sys.stderr)
                return None, f"rccl (p2p bring-up failed: {type(e).__name__})"

        if not args.no_p2p_tail and world <= 8:
            tail_p2p, why = bring_up(n_tail_arena, False)
            tail_how = "p2p one-shot + update in one launch" if tail_p2p is not None else why
        # The dense message (12.8 MB, hidden under backward if it is fast enough) goes peer-to-peer only if the
        # two-shot kernel is correct here AND clearly faster than RCCL stand-alone on this node.
        # The dense message (12.8 MB, overlapped with backward) has a peer-to-peer candidate too (two-shot kernel); whether
        # it is used is decided further down by timing the STEP with either path: stand-alone latency says little about
        # a collective whose cost is what it takes from the kernels it overlaps with (world 1, 64 workgroups: 50 us
        # stand-alone, +30 us on the step).  DTF_BENCH_DENSE_P2P_WORLD1 lets a one-rank --force_dp run try it as well.
        dense_cand = None
        if (not args.no_p2p_dense and args.grad_comm == "fp32" and world <= 8
                and (world > 1 or os.environ.get("DTF_BENCH_DENSE_P2P_WORLD1"))):
            dense_cand, why = bring_up(n_dense_arena, True)
            if dense_cand is None:
                dense_how = why
        dp = SyncDataParallel(eng, tail_group=tail_group, tail_p2p=tail_p2p, dense_p2p=dense_p2p,
                              shard_dense={"auto": None, "allreduce": False, "sharded": True}[args.dense],
                              grad_comm=args.grad_comm)
        dense_how = "rccl reduce-scatter + sharded update + all-gather" if dp.shard else "rccl all-reduce"
        dp.broadcast_parameters()
        step = dp.train_step
    else:
        # The whole step is one foreign call into the native executor either way.  Measured A/B: with the
        # single-stream 8-kernel step, eager launches run 2.5-5 % faster than replaying the captured graph up
        # to 1024 images/GPU; from 2048 up the graph (with its backward fork/join) wins by ~6 %.
        use_graph = args.graph or (not args.no_graph and B > 1024 and args.dtype == "bf16")
        if use_graph:
            eng.capture()
        step = eng.train_step

    def barrier():
        if dp_path:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def step_ms(n=200):
        for _ in range(10):
            step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(n):
            step()
        barrier()
        return max_over_ranks((time.perf_counter() - t1) * 1e3 / n)

    def max_over_ranks(x):       # the same number on every rank, so that every rank takes the same decision
        if not dp_path:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64)
        if torch.distributed.get_backend() != "gloo":
            t = t.cuda()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def probe_ok(p2p):           # five steps first: a path whose waits time out (2 s each) must not get a 200-step trial
        for _ in range(5):
            step()
        barrier()
        return max_over_ranks(p2p.status()) == 0

    # ---- N > 1: which path carries each message.  ONE 50-step trial per candidate, at most three candidates per
    # message, and the whole selection is guarded: anything unexpected sends both messages to RCCL (the verified
    # fallback) and the benchmark still runs and prints its line. ----
    TRIAL_STEPS = 50

    def select_paths():
        nonlocal dense_p2p, dense_how, tail_p2p, tail_how, p2p_blocks_us, tail_trials
        if world > 1 and args.dense == "auto" and dp.can_shard():
            # the dense message over RCCL: sharded (reduce-scatter / update of 1/N / all-gather) or plain all-reduce
            dp.set_shard(True)
            t_sh = step_ms(TRIAL_STEPS)
            dp.set_shard(False)            # gathers the master weights first (collective)
            t_ar = step_ms(TRIAL_STEPS)
            dp.set_shard(t_sh <= t_ar)
            dense_how = ("rccl reduce-scatter + sharded update + all-gather" if dp.shard else "rccl all-reduce") + \
                f" ({t_sh:.4f} sharded vs {t_ar:.4f} all-reduce ms/step)"
        if dense_cand is not None:
            rccl_how, want_shard = dense_how, dp.shard
            trials = {"rccl": step_ms(TRIAL_STEPS)}
            dp.set_shard(False)            # the peer-to-peer kernel all-reduces: full masters needed on every rank
            dp.dense_p2p = dense_cand
            for nb in (64,):
                dense_cand.blocks = nb
                if not probe_ok(dense_cand):
                    dense_cand._status.fill_(1)      # disqualified below (and the ranks re-aligned)
                    break
                trials[f"p2p two-shot, {nb} workgroups"] = step_ms(TRIAL_STEPS)
            p2p_blocks_us = {k: round(v, 4) for k, v in trials.items()}
            best = min(trials, key=trials.get)
            st = max_over_ranks(dense_cand.status())      # a timed-out wait on ANY rank disqualifies it
            if best != "rccl" and trials[best] < 0.98 * trials["rccl"] and st == 0:
                dense_cand.blocks = int(best.split(",")[1].split()[0])
                dense_p2p, dense_how = dense_cand, f"{best} ({trials[best]:.4f} vs rccl {trials['rccl']:.4f} ms/step)"
            else:
                dp.dense_p2p = None
                if st:
                    dp.broadcast_parameters()      # sums may have been wrong on the way: re-align the ranks
                dp.set_shard(want_shard)
                others = [v for k, v in trials.items() if k != "rccl"]
                dense_how = rccl_how + f"; {trials['rccl']:.4f} ms/step" + (f" vs p2p two-shot {min(others):.4f}" if others else
                                                                             "; p2p two-shot waits timed out")
                barrier()
                dense_cand.close()
        if tail_p2p is not None:
            # the same question for the tail message: exchange + update kernel vs {RCCL all-reduce, update}, by step time
            t_p2p = step_ms(TRIAL_STEPS) if probe_ok(tail_p2p) else float("inf")
            if t_p2p == float("inf"):
                tail_p2p._status.fill_(1)
            dp.tail_p2p = None
            t_pg = step_ms(TRIAL_STEPS)
            st = max_over_ranks(tail_p2p.status())
            tail_trials = {"p2p": round(t_p2p, 4) if t_p2p != float("inf") else None, "rccl": round(t_pg, 4)}
            if t_p2p <= t_pg and st == 0:
                dp.tail_p2p = tail_p2p
                tail_how += f" ({t_p2p:.4f} vs rccl {t_pg:.4f} ms/step)"
            else:
                if st:
                    dp.broadcast_parameters()
                barrier()
                tail_p2p.close()
                tail_p2p = None
                tail_how = f"rccl ({t_pg:.4f} ms/step" + (", p2p waits timed out)" if st else f" vs p2p {t_p2p:.4f})")

    def prewarm():
        """Time-based warm-up, outside the timed region and reported: the requested --warmup / --steps may be a few
        milliseconds of GPU time in total (the driver runs 5 + 20 steps), which would otherwise be measured on a cold,
        low-clocked device with lazily set kernel attributes.  Runs before the path-selection trials as well: the first
        candidate must not be the one that is timed while the clocks ramp."""
        nonlocal prewarm_steps
        t_pw = time.perf_counter()
        while args.prewarm_s > 0:
            for _ in range(10):
                step()
            prewarm_steps += 10
            torch.cuda.synchronize()
            more = time.perf_counter() - t_pw < args.prewarm_s
            if dp_path:      # every rank must run the same number of steps: rank 0's clock decides
                more = max_over_ranks(1.0 if (rank == 0 and more) else 0.0) > 0
            if not more:
                break

    prewarm_steps = 0
    if dp_path:
        try:
            prewarm()
            select_paths()
        except Exception as e:      # collectives inside are unanimous (max over ranks), so every rank lands here together
            print(f"[bench] rank {rank}: path selection failed ({type(e).__name__}: {e}); both messages on RCCL", file=sys.stderr)
            dp.tail_p2p = dp.dense_p2p = None
            dp.shard, dp.master_stale = False, False      # plain all-reduce; broadcast_parameters below re-aligns everything
            dp.shard_lo, dp.shard_hi = 0, dp.n_dense
            tail_p2p = dense_p2p = None
            tail_how = dense_how = f"rccl (path selection failed: {type(e).__name__})"
            try:
                dp.broadcast_parameters()
            except Exception:
                pass

    prewarm()
    for _ in range(args.warmup):
        step()
    barrier()
    if tail_p2p is not None or dense_p2p is not None:
        # Last line of defence before the clock starts: a peer-to-peer wait that timed out during the warm-up steps
        # (on any rank) sends both messages back to RCCL, and the warm-up is repeated on that path.
        if os.environ.get("DTF_BENCH_FAKE_P2P_TIMEOUT") and rank == world - 1 and tail_p2p is not None:
            tail_p2p._status.fill_(1)          # test hook for this fallback (tests/test_distributed_gpu.py)
        bad = torch.tensor([sum(p.status() for p in (tail_p2p, dense_p2p) if p is not None)], dtype=torch.int32)
        if torch.distributed.get_backend() != "gloo":
            bad = bad.cuda()
        torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX)
        if int(bad.item()):
            dp.tail_p2p = dp.dense_p2p = None
            for p in (tail_p2p, dense_p2p):
                if p is not None:
                    p.close()
            tail_p2p = dense_p2p = None
            tail_how = dense_how = "rccl (p2p wait timed out during warm-up)"
            dp.broadcast_parameters()          # the ranks may have drifted apart on the bad sums
            for _ in range(args.warmup):
                step()
            barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0

    dt = max_over_ranks(dt)
    loss, acc = eng.last_metrics()
    # a timed-out wait on ANY rank means wrong sums on that rank: reduce the status words over the ranks
    p2p_timeout = False
    if tail_p2p is not None and max_over_ranks(tail_p2p.status()):
        tail_how += " (a wait TIMED OUT inside the timed region: sums invalid)"
        p2p_timeout = True
    if dense_p2p is not None and max_over_ranks(dense_p2p.status()):
        dense_how += " (a wait TIMED OUT inside the timed region: sums invalid)"
        p2p_timeout = True
    def emit(comm_diag):
        if rank != 0:
            return
        global_batch = B * world
        value = global_batch * args.steps / dt
        _, fb = spec.flops_per_image()
        line = json.dumps({
            "metric": "train images/sec (whole node)",
            "value": round(value, 1),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_steps": prewarm_steps,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (HBM-resident 28x28x1, random TF-style init)",
            "config": {"model": spec.name, "global_batch": global_batch, "per_gpu_batch": B, "seq_len": None,
                       "parallelism": f"dp{world}" + ("(forced dp schedule)" if args.force_dp and world == 1 else "") + ("(rehearsal: gloo, one GPU)" if args.rehearse_one_gpu else ""), "optimizer": args.optimizer, "keep_prob": args.keep_prob,
                       "graph": bool(use_graph),
                       # HIP streams of this process: compute (+ comm) + the engine's backward side streams (RCCL's own not counted)
                       "streams": 1 + (1 if dp_path else 0) + eng.streams_in_use(),
                       **({"tail_allreduce": tail_how, "dense_allreduce": dense_how, "grad_comm": args.grad_comm}
                          if dp_path else {})},
            "model_tflops": round(value * fb / 1e12, 2),
            "final_loss": round(loss, 4),
            **({"comm_diag": comm_diag} if comm_diag is not None else {}),
        })
        os.write(_RESULT_FD, (line + "\n").encode())
    # The diagnostics below run collectives after the measurement is complete: if they stall (they never have), every
    # rank gives up after 90 s, rank 0 still prints the headline line, and the processes exit.
    def give_up():
        emit({"error": "comm_diag stalled for 90 s and was abandoned"})
        os._exit(3)      # the headline line is out; the exit status still says that something stalled

    import threading
    watchdog = threading.Timer(90.0, give_up)
    watchdog.daemon = True
    comm_diag = None
    if dp_path:
        watchdog.start()
        # Outside the timed region: stand-alone latency of the step's two all-reduces (the dense fc1 gradient
        # and the small tail), i.e. the two communication terms of the scaling model in docs/ARCHITECTURE.md.
        try:
            dev = torch.device("cuda", local_rank)
            n_dense = spec.flat * spec.hidden
            n_tail = spec.num_params() - n_dense
            comm_diag = {
                "dense_allreduce_bytes": 4 * n_dense, "tail_allreduce_bytes": 4 * n_tail,
                "dense_allreduce_us": round(comm.allreduce_latency_us(n_dense, device=dev), 1),
                "tail_allreduce_us": round(comm.allreduce_latency_us(n_tail, group=tail_group, device=dev, iters=50), 1),
            }
            # ... and the data-parallel step with one or both collectives left out (ranks then drift apart, which
            # no longer matters here): which of the two is exposed, and what the schedule costs without either
            real, tail_bucket = dp._allreduce, tuple(dp.tail_bucket)
            real_rs, real_ag = dp._reduce_scatter, dp._all_gather

            def timed(skip_dense, skip_tail, n=100):
                def ar(lo, hi, group):
                    is_tail = (lo, hi) == tail_bucket
                    if not (skip_tail if is_tail else skip_dense):
                        real(lo, hi, group)
                dp._allreduce = ar
                if skip_dense:
                    dp._reduce_scatter = dp._all_gather = lambda: None
                p2p_real = (dp.tail_p2p, dp.dense_p2p)     # a skipped message must not go peer-to-peer either
                if skip_tail:
                    dp.tail_p2p = None      # -> process-group path, where ar() drops the message
                if skip_dense:
                    dp.dense_p2p = None
                for _ in range(20):
                    step()
                barrier()
                t1 = time.perf_counter()
                for _ in range(n):
                    step()
                barrier()
                ms = max_over_ranks((time.perf_counter() - t1) * 1e3 / n)
                dp._allreduce, (dp.tail_p2p, dp.dense_p2p) = real, p2p_real
                dp._reduce_scatter, dp._all_gather = real_rs, real_ag
                return round(ms, 4)

            if tail_p2p is not None:
                xt = torch.zeros(n_tail_arena, device=dev)
                for _ in range(10):
                    tail_p2p.allreduce(xt)
                barrier()
                t1 = time.perf_counter()
                for _ in range(100):
                    tail_p2p.allreduce(xt)
                torch.cuda.synchronize()
                comm_diag["tail_p2p_us"] = round((time.perf_counter() - t1) * 1e4, 1)
                comm_diag["tail_p2p_timeouts"] = tail_p2p.status()
            if p2p_blocks_us is not None:
                comm_diag["step_ms_by_dense_allreduce"] = p2p_blocks_us
            if tail_trials is not None:
                comm_diag["step_ms_by_tail_allreduce"] = tail_trials
            comm_diag["step_ms_no_allreduce"] = timed(True, True)
            comm_diag["step_ms_dense_allreduce_only"] = timed(False, True)
            comm_diag["step_ms_tail_allreduce_only"] = timed(True, False)
        except Exception as e:   # a diagnostic must never cost the headline line
            comm_diag = {"error": repr(e)}
        watchdog.cancel()
    emit(comm_diag)
    if p2p_timeout:
        print("[bench] a peer-to-peer wait timed out inside the timed region: the value above is INVALID", file=sys.stderr)
    if dp_path:
        barrier()
        for p2p in (tail_p2p, dense_p2p):
            if p2p is not None:
                p2p.close()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
