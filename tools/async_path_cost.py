#!/usr/bin/env python3
"""Per-step cost of the asynchronous parameter-server path on ONE GPU (BASELINE.json config 4 has no multi-GPU number
from this harness): 1 ps + W workers share cuda:0, mailboxes mapped through hipIpc on the same device.

    python tools/async_path_cost.py [--workers 1] [--staleness 0,1,2] [--batch 1024] [--steps 200]

Prints, per staleness bound, the worker's ms/step and images/s next to the local fused step of the same batch size.
All tasks compete for the one GPU (the ps's apply kernels and the workers' steps are serialised by the hardware
scheduler), so this is the cost of the PATH -- push, poll, apply + answer, install -- not a scaling number.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def role(args):
    import torch
    from distributed_tensorflow_amd.backends import HipBackend
    from distributed_tensorflow_amd.cluster import ClusterSpec, TaskInfo
    from distributed_tensorflow_amd.data.mnist import read_data_sets
    from distributed_tensorflow_amd.models.spec import ModelSpec
    from distributed_tensorflow_amd.parallel import comm
    from distributed_tensorflow_amd.parallel.async_push import AsyncPushServer, AsyncPushWorker
    rank, W = args.rank, args.workers
    torch.cuda.set_device(0)
    comm.init_process_group("gloo", rank, 1 + W, "127.0.0.1", args.port, timeout_s=120)
    cluster = ClusterSpec({"ps": [f"127.0.0.1:{args.port}"], "worker": [f"127.0.0.1:{args.port + 1 + i}" for i in range(W)]})
    task = TaskInfo(cluster, "ps" if rank == 0 else "worker", max(0, rank - 1))
    data = read_data_sets("", synthetic=True, synthetic_sizes=(16384, 256))
    backend = HipBackend(ModelSpec(), args.batch, data, optimizer="sgd", learning_rate=1e-3, seed=3 + rank, dtype=args.dtype)
    backend.init_params(1)
    total = args.steps * W + 40 * W
    if rank == 0:
        t0 = time.monotonic()
        srv = AsyncPushServer(backend, task, chief_rank=1, training_iter=10 ** 9, staleness=args.k, log=lambda *a: None,
                              peer_lost=lambda r: time.monotonic() - t0 > 120.0)
        srv.broadcast_initial()
        srv.serve()
        hist = srv.staleness_hist
        tot = max(1, sum(hist.values()))
        print("PS " + json.dumps({"applied": srv.applied, "mean_staleness": sum(k * v for k, v in hist.items()) / tot}), flush=True)
        srv.fab.close()
    else:
        wk = AsyncPushWorker(backend, task, chief_rank=1, staleness=args.k, timeout_s=30.0)
        wk.broadcast_initial()
        for _ in range(40):
            wk.step()
        torch.cuda.synchronize()
        torch.distributed.barrier(group=None) if False else None
        t0 = time.perf_counter()
        for _ in range(args.steps):
            wk.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        wk.finish()
        print("WK " + json.dumps({"rank": rank, "ms_per_step": 1e3 * dt / args.steps}), flush=True)
        wk.fab.close()
    comm.shutdown()


def local_step_ms(batch, dtype, steps):
    import torch
    from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
    from distributed_tensorflow_amd.models.spec import ModelSpec
    ds = DeviceDataset.synthetic(16384, seed=1, dtype=dtype)
    out = {}
    for name, fuse in (("fused", True), ("unfused (gradient arena + optimizer kernel)", False)):
        eng = HipEngine(ModelSpec(), batch, ds, optimizer="sgd", learning_rate=1e-3, seed=3, fuse_dense_update=fuse, dtype=dtype)
        eng.init_params(1)
        for _ in range(40):
            eng.train_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.train_step()
        torch.cuda.synchronize()
        out[name] = 1e3 * (time.perf_counter() - t0) / steps
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=1)
    ap.add_argument("--staleness", default="0,1,2")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--port", type=int, default=0)
    ap.add_argument("--k", type=int, default=0)
    args = ap.parse_args()
    if args.rank >= 0:
        return role(args)
    loc = local_step_ms(args.batch, args.dtype, args.steps)
    print(f"batch {args.batch}, {args.dtype}: local step " + ", ".join(f"{k} {v:.4f} ms" for k, v in loc.items()))
    rows = []
    for k in [int(x) for x in args.staleness.split(",")]:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        procs = [subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), "--rank", str(r), "--port", str(port),
                                   "--k", str(k), "--workers", str(args.workers), "--batch", str(args.batch), "--steps",
                                   str(args.steps), "--dtype", args.dtype], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                  text=True) for r in range(1 + args.workers)]
        ms, ps = [], {}
        for p in procs:
            try:
                o, _ = p.communicate(timeout=240)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise SystemExit("timed out:\n" + p.communicate()[0][-2000:])
            if p.returncode != 0:
                raise SystemExit(o[-3000:])
            for line in o.splitlines():
                if line.startswith("WK "):
                    ms.append(json.loads(line[3:])["ms_per_step"])
                if line.startswith("PS "):
                    ps = json.loads(line[3:])
        row = dict(staleness=k, workers=args.workers, worker_ms_per_step=max(ms), images_per_s=args.workers * args.batch / (max(ms) * 1e-3),
                   mean_applied_staleness=ps.get("mean_staleness"))
        rows.append(row)
        print(json.dumps(row), flush=True)
    print(f"\n| staleness bound | workers | worker ms/step | images/s (all workers) | mean staleness of applied gradients | vs local fused step |")
    print("|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['staleness']} | {r['workers']} | {r['worker_ms_per_step']:.4f} | {r['images_per_s'] / 1e6:.2f} M | "
              f"{r['mean_applied_staleness']:.2f} | {r['worker_ms_per_step'] / loc['fused']:.2f}x |")


if __name__ == "__main__":
    main()
