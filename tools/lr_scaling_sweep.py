#!/usr/bin/env python3
"""LR-scaling sweep on ONE MI355X (BASELINE.json config 5: widened CNN, global batch 256 ... 8192).

Synchronous data parallelism computes exactly the large-batch gradient, so the optimisation question -- which rate
rule keeps a growing global batch converging -- is answered on one GPU by running the global batch there.  For every
(batch, rule) the model trains for a fixed SAMPLE budget (the same number of epochs, so fewer steps at larger batches)
and is evaluated on the held-out split:

    python tools/lr_scaling_sweep.py --data digits --model wide --optimizer sgd --base_lr 0.001 --epochs 60
    python tools/lr_scaling_sweep.py --data synthetic --model wide --optimizer sgd --base_lr 0.001 --epochs 4

Prints a markdown table and writes gpurun_out/lr_sweep_<data>_<model>_<optimizer>.json.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", choices=["digits", "synthetic"], default="digits")
    ap.add_argument("--model", default="wide")
    ap.add_argument("--optimizer", default="sgd")
    ap.add_argument("--base_lr", type=float, default=0.001)
    ap.add_argument("--base_batch", type=int, default=128)
    ap.add_argument("--epochs", type=float, default=60)
    ap.add_argument("--batches", default="128,256,512,1024,2048,4096,8192")
    ap.add_argument("--rules", default="none,linear,sqrt")
    ap.add_argument("--warmup_frac", type=float, default=0.05)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    import torch
    from distributed_tensorflow_amd.data.mnist import read_data_sets
    from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
    from distributed_tensorflow_amd.models.spec import ModelSpec
    from distributed_tensorflow_amd.utils.lr_schedule import LrSchedule

    if args.data == "digits":
        d = tempfile.mkdtemp(prefix="digits_")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "digits_to_idx.py"), d], check=True, capture_output=True)
        data = read_data_sets(d, one_hot=False, validation_size=100)
    else:
        data = read_data_sets("", one_hot=False, synthetic=True)
    tr, te = data.train, data.test
    ds = DeviceDataset.from_float(torch.from_numpy(tr.images), torch.from_numpy(tr.int_labels), "cuda", dtype=args.dtype)
    dt = DeviceDataset.from_float(torch.from_numpy(te.images), torch.from_numpy(te.int_labels), "cuda", dtype=args.dtype)
    spec = ModelSpec.from_name(args.model)
    budget = int(args.epochs * ds.n)
    rows = []
    for B in [int(b) for b in args.batches.split(",")]:
        steps = max(1, budget // B)
        for rule in args.rules.split(","):
            if B == args.base_batch and rule != "none":
                continue
            sched = LrSchedule(args.base_lr, B, args.base_batch, rule, int(round(args.warmup_frac * steps)) if rule != "none" else 0)
            eng = HipEngine(spec, B, ds, optimizer=args.optimizer, learning_rate=args.base_lr, seed=args.seed, dtype=args.dtype)
            eng.init_params(seed=args.seed + 1)
            cur, tail, diverged = None, [], False
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(steps):
                lr = sched(s)
                if lr != cur:
                    eng.set_learning_rate(lr)
                    cur = lr
                eng.train_step()
                if s >= steps - 10 or s % 50 == 49:
                    l = eng.last_metrics()[0]
                    if not math.isfinite(l):
                        diverged = True
                        break
                    if s >= steps - 10:
                        tail.append(l)
            torch.cuda.synchronize()
            dt_s = time.perf_counter() - t0
            if diverged:
                tl = ta = float("nan")
            else:
                tl, ta = eng.evaluate(dt.images, dt.labels)
            row = dict(batch=B, rule=rule, peak_lr=sched.peak_lr, warmup=sched.warmup_steps, steps=steps,
                       train_loss=(sum(tail) / len(tail)) if tail and not diverged else float("nan"),
                       test_loss=tl, test_acc=ta, diverged=diverged, images_per_s=(s + 1) * B / dt_s)
            rows.append(row)
            print(json.dumps(row), flush=True)
            del eng
            torch.cuda.empty_cache()
    out = os.path.join(ROOT, "gpurun_out", f"lr_sweep_{args.data}_{args.model}_{args.optimizer}.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(dict(args=vars(args), n_train=ds.n, n_test=dt.n, sample_budget=budget, rows=rows), f, indent=1)
    rules = args.rules.split(",")
    print(f"\n{args.data}, {spec.name}, {args.optimizer}, base lr {args.base_lr:g} @ batch {args.base_batch}, "
          f"{args.epochs:g} epochs ({budget} samples), {dt.n} test images; cell = test accuracy / test loss (peak lr)")
    print("| global batch | steps | " + " | ".join(rules) + " |")
    print("|---|---|" + "---|" * len(rules))
    for B in sorted({r["batch"] for r in rows}):
        cells = []
        for rule in rules:
            m = [r for r in rows if r["batch"] == B and r["rule"] == rule]
            if not m:
                cells.append("= none")
            elif m[0]["diverged"]:
                cells.append(f"diverged ({m[0]['peak_lr']:g})")
            else:
                cells.append(f"{m[0]['test_acc']:.4f} / {m[0]['test_loss']:.3f} ({m[0]['peak_lr']:g})")
        st = [r for r in rows if r["batch"] == B][0]["steps"]
        print(f"| {B} | {st} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main()
