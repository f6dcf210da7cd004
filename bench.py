#!/usr/bin/env python3
"""Headline benchmark: training throughput (images/s, whole job) of the deep-MNIST
CNN on N MI355X GPUs, synchronous data parallel.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

Metric / config follow BASELINE.json ("train images/sec (whole node) at 1/2/4/8
MI355X", sync DP, global batch 8192 at 8 GPUs): weak scaling with a fixed
per-GPU batch of 1024, the reference's optimizer (plain SGD, lr 1e-3, dropout
keep_prob 1.0; reference .idea/MNISTDist.py:30,149,179), HBM-resident synthetic
28x28x1 data, random TF-style init.  Every step runs sampler + forward + loss +
backward + gradient exchange + update + global_step increment; nothing is skipped.

ONE JSON line: the headline (``value``, ``ms_per_step``, ``dtype``) is the bf16 configuration (bf16 MFMA compute copies,
fp32 accumulation, fp32 master weights; BASELINE.json names fp32/bf16), and the same line carries the reference's own
precision -- the exact-fp32 kernel set (M:113-115), timed with the same --warmup / --steps right afterwards -- under
``"fp32"``.  ``config`` is self-describing: kernel library, build flavour, every DTF_* override in the environment,
and at N > 1 which path carried each gradient message and the trial times behind that choice.
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

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import distributed_tensorflow_amd  # noqa: E402,F401  (sets the IPC mode the peer mappings need BEFORE HIP initialises)
import torch  # noqa: E402

TRIAL_STEPS = 50


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
                    help="precision of the HEADLINE numbers.  bf16 (default): bf16 MFMA compute copies, fp32 accumulation and "
                         "fp32 master weights, and the exact-fp32 configuration is timed as well and reported under \"fp32\".  "
                         "fp32: the reference's precision end to end as the headline (reference .idea/MNISTDist.py:113-115)")
    ap.add_argument("--no_fp32", action="store_true", help="do not time the fp32 configuration after the bf16 one")
    ap.add_argument("--single_stream", action="store_true", help="disable the fork/join side streams in backward")
    ap.add_argument("--multi_stream", action="store_true", help="force the fork/join side streams (default: by batch size)")
    ap.add_argument("--no_graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--graph", action="store_true", help="force HIP-graph replay (1 GPU default: only above 1024 images/GPU)")
    ap.add_argument("--dense", choices=["auto", "allreduce", "sharded"], default="auto",
                    help="N > 1, dense (fc1) gradient over RCCL: all-reduce + replicated update, or reduce-scatter + "
                         "update of 1/N + all-gather of the compute copy; auto times one trial of each")
    ap.add_argument("--force_dp", action="store_true",
                    help="diagnostic: run the data-parallel step (process group, exchange fabric) even at --gpus 1, to "
                         "exercise the N>1 code path on a one-GPU box")
    ap.add_argument("--no_collapse", action="store_true",
                    help="--force_dp at one rank: keep the N>1 kernels of the dense exchange in the step (a one-rank job "
                         "otherwise applies the dense SGD update in the wgrad epilogue like the single-GPU step)")
    ap.add_argument("--no_p2p", action="store_true",
                    help="N>1: keep both gradient messages on RCCL instead of the peer-to-peer exchange fabric")
    ap.add_argument("--grad_comm", choices=["fp32", "bf16"], default="fp32",
                    help="N > 1 / --force_dp, bf16 configuration: bf16 = opt-in 16-bit collective for the fc1 weight gradient "
                         "(NOT the reference's fp32 gradient sum; recorded in config.grad_comm; RCCL path only)")
    ap.add_argument("--rehearse_one_gpu", action="store_true",
                    help="diagnostic: every rank on cuda:0 over gloo -- rehearses the torchrun N>1 launch of this file "
                         "on a one-GPU box (RCCL refuses ranks that share a device); the number means nothing")
    ap.add_argument("--comm_diag", action="store_true",
                    help="N>1: after the measurement, time the step with one or both messages left out (comm_diag)")
    args = ap.parse_args()

    from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine, Tuning
    from distributed_tensorflow_amd.models.spec import ModelSpec
    from distributed_tensorflow_amd.ops import native
    from distributed_tensorflow_amd.parallel.sync_dp import SyncDataParallel
    from distributed_tensorflow_amd.parallel import comm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    if args.rehearse_one_gpu:
        import faulthandler
        faulthandler.dump_traceback_later(240, exit=True)   # a rehearsal that stalls says where
        local_rank = 0
    ndev = torch.cuda.device_count()
    if ndev and local_rank >= ndev:      # launcher restricted this rank's visible devices (e.g. one GPU per process)
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dp_path = world > 1 or args.force_dp
    gloo = bool(args.rehearse_one_gpu)
    tail_group = None
    if dp_path:
        comm.init_process_group(backend="gloo" if gloo else "nccl", rank=rank, world_size=world,
                                device=None if gloo else dev)
        try:   # second communicator: the small tail all-reduce must not queue behind the 12.8 MB one inside RCCL
            tail_group = torch.distributed.new_group()
        except Exception as e:   # still correct without it, only slower: the tail then rides the comm stream
            print(f"[bench] rank {rank}: no second communicator ({e}); tail all-reduce shares the first", file=sys.stderr)

    spec = ModelSpec.from_name(args.model)
    B = args.per_gpu_batch
    _, flops_fb = spec.flops_per_image()

    def barrier():
        if dp_path:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):       # the same number on every rank, so that every rank takes the same decision
        if not dp_path:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64)
        if not gloo:
            t = t.cuda()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def measure(dtype):
        """Build the engine for one precision, choose the gradient paths (N > 1), warm up and time --steps steps.
        -> dict of results (the same on every rank)."""
        ds = DeviceDataset.synthetic(args.dataset_size, seed=1234, dtype=dtype)
        eng = HipEngine(spec, B, ds, optimizer=args.optimizer, learning_rate=1e-3, keep_prob=args.keep_prob,
                        seed=1000 + rank,
                        # None = engine default (by batch size); the data-parallel path measured faster without the fork/join
                        multi_stream=False if (args.single_stream or dp_path) else (True if args.multi_stream else None),
                        fuse_dense_update=not dp_path, dtype=dtype)
        eng.init_params(seed=7)
        info = {"graph": False}
        fabric = None
        dp = None
        if dp_path:
            grad_comm = args.grad_comm if dtype == "bf16" else "fp32"
            dp = SyncDataParallel(eng, tail_group=tail_group, collapse_world1=not args.no_collapse,
                                  shard_dense={"auto": None, "allreduce": False, "sharded": True}[args.dense],
                                  grad_comm=grad_comm,
                                  # more than four ranks rehearsing on ONE GPU: one hardware queue per rank (see below)
                                  fabric_single_stream=bool(args.rehearse_one_gpu and world > 4))
            dp.broadcast_parameters()
            step = dp.train_step
            info.update(grad_comm=grad_comm, gradient_exchange="rccl", dense="rccl " + ("reduce-scatter + sharded update + "
                        "all-gather" if dp.shard else "all-reduce"), tail="rccl all-reduce + fused update")
        else:
            # The whole step is one foreign call into the native executor either way.  Measured A/B: with the
            # single-stream 8-kernel step, eager launches run 2.5-5 % faster than replaying the captured graph up
            # to 1024 images/GPU; from 2048 up the graph (with its backward fork/join) wins by ~6 %.
            info["graph"] = bool(args.graph or (not args.no_graph and B > 1024 and dtype == "bf16"))
            if info["graph"]:
                eng.capture()
            step = eng.train_step

        def step_ms(n):
            for _ in range(10):
                step()
            barrier()
            t1 = time.perf_counter()
            for _ in range(n):
                step()
            barrier()
            return max_over_ranks((time.perf_counter() - t1) * 1e3 / n)

        prewarm_steps = 0

        def prewarm():
            """Time-based warm-up, outside the timed region and reported: the requested --warmup / --steps may be a few
            milliseconds of GPU time in total (the driver runs 5 + 20 steps), which would otherwise be measured on a
            cold, low-clocked device with lazily set kernel attributes."""
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

        def fabric_graph_or_eager(trials):
            """The fabric step as eager launches (one native call; the host runs far ahead of the GPU) or as ONE HIP graph
            (a step is a graph launch): whichever replays faster here -- a dependent launch costs more inside a graph
            than from the stream on this stack, the host less.  -> ms/step of the choice."""
            t_eager = float("inf") if args.graph else step_ms(TRIAL_STEPS)
            t_graph = float("inf")
            # (a rehearsal with more than four ranks on ONE GPU skips the graph trial: eight processes' graph branches
            #  oversubscribe the device's hardware queues, the scheduler starts time-slicing them at millisecond
            #  quanta, and every spin-wait for a peer then costs a quantum -- 22 ms/step, observed, and it stays that
            #  way after the graph is released.  One process per GPU has the queues to itself.)
            if not args.no_graph and not (args.rehearse_one_gpu and world > 4):
                dp.capture()
                t_graph = step_ms(TRIAL_STEPS)
                if t_eager < t_graph:
                    dp.release_graph()
            info["graph"] = t_graph <= t_eager
            trials["p2p fabric"] = round(min(t_eager, t_graph), 4)
            trials["p2p fabric, eager"], trials["p2p fabric, one graph"] = t_eager, t_graph
            return min(t_eager, t_graph)

        def back_to_rccl(why):
            """Both messages through the process group's all-reduce: the verified fallback (collective)."""
            nonlocal fabric
            dp.use_fabric(None)
            dp.shard, dp.weights_stale, dp.moments_stale = False, False, False
            dp.shard_lo, dp.shard_hi = 0, dp.n_dense
            if fabric is not None:
                try:
                    fabric.close()
                except Exception:
                    pass
                fabric = None
            info.update(gradient_exchange=f"rccl ({why})", dense="rccl all-reduce", tail="rccl all-reduce + fused update",
                        graph=False)
            dp.broadcast_parameters()          # the ranks may have drifted apart on the way: re-align everything

        def select_paths():
            """N > 1: ONE trial per candidate.  RCCL: sharded vs all-reduce dense message.  Then the exchange fabric
            (both messages peer to peer, the step one HIP graph) if it comes up on this node, passes its self-check
            on every rank and is not slower than the best RCCL step."""
            nonlocal fabric
            trials = {}
            if world > 1 and args.dense == "auto" and dp.can_shard():
                dp.set_shard(True)
                trials["rccl sharded"] = step_ms(TRIAL_STEPS)
                dp.set_shard(False)            # gathers the master weights first (collective)
                trials["rccl all-reduce"] = step_ms(TRIAL_STEPS)
                dp.set_shard(trials["rccl sharded"] <= trials["rccl all-reduce"])
            else:
                trials["rccl " + ("sharded" if dp.shard else "all-reduce")] = step_ms(TRIAL_STEPS)
            best_rccl = min(trials.values())
            info["dense"] = "rccl " + ("reduce-scatter + sharded update + all-gather" if dp.shard else "all-reduce")
            want_shard = dp.shard
            if not args.no_p2p and info["grad_comm"] == "fp32" and world <= 8:
                from distributed_tensorflow_amd.parallel.dp_fabric import DpFabric
                try:
                    n_dense = eng.layout.segment("wd1", "wd1")[1]
                    # bounded waits: a dead path costs seconds.  (A rehearsal's ranks share ONE GPU, where a spinning
                    # exchange kernel can keep a peer's kernel off the device for many seconds: a longer bound there.)
                    fabric = DpFabric(n_dense, eng.layout.total - n_dense, dev, dtype=dtype,
                                      timeout_s=30.0 if args.rehearse_one_gpu else 2.0)
                except RuntimeError as e:      # unanimous: every rank lands here together
                    print(f"[bench] rank {rank}: {e}", file=sys.stderr)
                    info["gradient_exchange"] = f"rccl (fabric bring-up failed: {str(e)[:80]})"
                if fabric is not None:
                    dp.set_shard(False)        # (collective gather if the RCCL trials left sharded masters)
                    dp.use_fabric(fabric)
                    ok = dp.fabric_self_check()
                    if ok and os.environ.get("DTF_BENCH_FAKE_P2P_TIMEOUT") and rank == world - 1:
                        fabric.ctl[0] = 1      # test hook for the fallback below (tests/test_distributed_gpu.py)
                    t_fab = float("inf")
                    if ok:
                        t_fab = fabric_graph_or_eager(trials)
                        ok = max_over_ranks(fabric.poisoned()) == 0
                    if not ok:
                        trials["p2p fabric"] = None
                    if ok and t_fab <= 1.02 * best_rccl:
                        info.update(gradient_exchange="p2p fabric (csrc/kernels/dp.hip)",
                                    dense="p2p reduce-scatter + sharded update + all-gather of the compute copy",
                                    tail="p2p one-shot exchange + commit in one launch")
                    else:
                        why = "fabric self-check / trial failed" if not ok else f"fabric {t_fab:.4f} vs rccl {best_rccl:.4f} ms/step"
                        dp.release_graph()
                        if not ok:
                            back_to_rccl(why)
                        else:
                            dp.use_fabric(None)        # (collective: gathers the sharded masters)
                            fabric.close()
                            fabric = None
                            info.update(gradient_exchange=f"rccl ({why})", graph=False)
                        dp.set_shard(want_shard)
                        info["dense"] = "rccl " + ("reduce-scatter + sharded update + all-gather" if dp.shard else "all-reduce")
            info["step_ms_by_path"] = {k: (round(v, 4) if v is not None else None) for k, v in trials.items()
                                       if v != float("inf")}

        if dp_path:
            try:
                prewarm()
                if world > 1:
                    select_paths()
                else:
                    # one rank (--force_dp): the fabric step, so that a one-GPU box measures what the N > 1 step costs
                    # locally -- the exchange kernels run with a world of one
                    from distributed_tensorflow_amd.parallel.dp_fabric import DpFabric
                    if not args.no_p2p and info["grad_comm"] == "fp32":
                        n_dense = eng.layout.segment("wd1", "wd1")[1]
                        fabric = DpFabric(n_dense, eng.layout.total - n_dense, dev, dtype=dtype, timeout_s=2.0)
                        dp.use_fabric(fabric)
                        trials = {}
                        fabric_graph_or_eager(trials)
                        info["step_ms_by_path"] = {k: round(v, 4) for k, v in trials.items() if v != float("inf")}
                        info.update(gradient_exchange="p2p fabric (csrc/kernels/dp.hip)",
                                    dense=("applied in the wgrad epilogue (one rank)" if args.optimizer == "sgd" else
                                           "applied by the tail kernel's commit workgroups (one rank)") if dp.collapse_world1
                                    else "p2p reduce-scatter + sharded update + all-gather of the compute copy",
                                    tail="p2p one-shot exchange + commit in one launch")
            except Exception as e:      # collectives inside are unanimous (max over ranks), so every rank lands here together
                print(f"[bench] rank {rank}: path selection failed ({type(e).__name__}: {e}); both messages on RCCL", file=sys.stderr)
                try:
                    dp.release_graph()
                    back_to_rccl(f"path selection failed: {type(e).__name__}")
                except Exception:
                    pass

        prewarm()
        for _ in range(args.warmup):
            step()
        barrier()
        if fabric is not None:
            # Last line of defence before the clock starts: a poisoned fabric on ANY rank (a wait timed out during the
            # warm-up) sends both messages back to RCCL, and the warm-up is repeated on that path.
            why = int(max_over_ranks(fabric.poisoned()))
            if why:
                dp.release_graph()
                back_to_rccl("a p2p wait timed out during warm-up" if why & 1 else f"fabric poisoned during warm-up (reason bits {why})")
                for _ in range(args.warmup):
                    step()
                barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        loss, _acc = eng.last_metrics()
        invalid = False
        if fabric is not None and max_over_ranks(fabric.poisoned()):
            info["gradient_exchange"] += " (a wait TIMED OUT inside the timed region: steps were not applied)"
            invalid = True
        value = B * world * args.steps / dt
        res = {"value": round(value, 1), "ms_per_step": round(1e3 * dt / args.steps, 4), "prewarm_steps": prewarm_steps,
               "model_tflops": round(value * flops_fb / 1e12, 2), "final_loss": round(loss, 4), "info": info,
               "streams": 1 + (1 if dp_path else 0) + eng.streams_in_use(), "invalid": invalid}
        if dp_path and args.comm_diag and world > 1:
            res["comm_diag"] = comm_diag(eng, dp, step, fabric)
        if dp_path:
            barrier()
            dp.release_graph()
            if fabric is not None:
                fabric.close()
        return res

    def comm_diag(eng, dp, step, fabric):
        """Outside the timed region: stand-alone latency of the process group's two all-reduces and the RCCL step with
        one or both messages left out (ranks then drift apart, which no longer matters here)."""
        try:
            n_dense = spec.flat * spec.hidden
            n_tail = spec.num_params() - n_dense
            out = {"dense_allreduce_bytes": 4 * n_dense, "tail_allreduce_bytes": 4 * n_tail,
                   "dense_allreduce_us": round(comm.allreduce_latency_us(n_dense, device=dev), 1),
                   "tail_allreduce_us": round(comm.allreduce_latency_us(n_tail, group=tail_group, device=dev, iters=50), 1)}
            if fabric is not None:
                return out
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
                for _ in range(20):
                    step()
                barrier()
                t1 = time.perf_counter()
                for _ in range(n):
                    step()
                barrier()
                ms = max_over_ranks((time.perf_counter() - t1) * 1e3 / n)
                dp._allreduce, dp._reduce_scatter, dp._all_gather = real, real_rs, real_ag
                return round(ms, 4)

            out["step_ms_no_allreduce"] = timed(True, True)
            out["step_ms_dense_allreduce_only"] = timed(False, True)
            out["step_ms_tail_allreduce_only"] = timed(True, False)
            return out
        except Exception as e:   # a diagnostic must never cost the headline line
            return {"error": repr(e)}

    head = measure(args.dtype)
    other = None
    if args.dtype == "bf16" and not args.no_fp32:
        try:
            other = measure("fp32")
        except Exception as e:      # the headline must not depend on the second configuration
            print(f"[bench] rank {rank}: fp32 configuration failed: {type(e).__name__}: {e}", file=sys.stderr)
            other = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        global_batch = B * world
        info = head["info"]
        par = f"dp{world}" + ("(forced dp step)" if args.force_dp and world == 1 else "") + \
            (f"(TIMING ONLY: dense shard of a {os.environ['DTF_DP_FAKE_WORLD']}-rank job, results invalid)"
             if os.environ.get("DTF_DP_FAKE_WORLD") and world == 1 and args.force_dp else "") + \
            ("(rehearsal: gloo, one GPU)" if args.rehearse_one_gpu else "")
        config = {"model": spec.name, "global_batch": global_batch, "per_gpu_batch": B, "seq_len": None, "parallelism": par,
                  "optimizer": args.optimizer, "keep_prob": args.keep_prob, "graph": bool(info.get("graph")),
                  # HIP streams of this process: compute (+ comm) + the engine's backward side streams (RCCL's own not counted)
                  "streams": head["streams"],
                  "kernels_lib": str(native.lib_path()), "build_flavour": native.build_flavour(),
                  "env_overrides": Tuning.env_overrides()}
        if dp_path:
            config.update({k: info[k] for k in ("gradient_exchange", "dense", "tail", "grad_comm", "step_ms_by_path") if k in info})
        line = {
            "metric": "train images/sec (whole node)", "value": head["value"], "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "prewarm_steps": head["prewarm_steps"],
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic (HBM-resident 28x28x1, random TF-style init)", "config": config,
            "model_tflops": head["model_tflops"], "final_loss": head["final_loss"],
        }
        if other is not None:
            if "error" in other:
                line["fp32"] = other
            else:
                line["fp32"] = {k: other[k] for k in ("value", "ms_per_step", "model_tflops", "final_loss", "prewarm_steps")}
                line["fp32"]["graph"] = bool(other["info"].get("graph"))
                if dp_path:
                    line["fp32"]["gradient_exchange"] = other["info"].get("gradient_exchange")
                    line["fp32"]["step_ms_by_path"] = other["info"].get("step_ms_by_path")
        if "comm_diag" in head:
            line["comm_diag"] = head["comm_diag"]
        os.write(_RESULT_FD, (json.dumps(line) + "\n").encode())
    if head.get("invalid"):
        print("[bench] a peer-to-peer wait timed out inside the timed region: the value above is INVALID", file=sys.stderr)
    if dp_path:
        barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
