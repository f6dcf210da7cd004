#!/usr/bin/env python3
"""LR-scaling sweep on ONE MI355X (BASELINE.json config 5: widened CNN, global batch 256 ... 8192).

Synchronous data parallelism computes exactly the large-batch gradient, so the optimisation question -- which rate
rule keeps a growing global batch converging -- is answered on one GPU by running the global batch there.  For every
(batch, rule) the model trains for a fixed SAMPLE budget (the same number of epochs, so fewer steps at larger batches)
and is evaluated on the held-out split:

    python tools/lr_scaling_sweep.py --data digits --model wide --optimizer sgd --base_lr 0.001 --epochs 60
    python tools/lr_scaling_sweep.py --data synthetic --model wide --optimizer sgd --base_lr 0.001 --epochs 4

Real images (--data digits): 5-fold cross-validation over all 1797 UCI optdigits images (every image is a test image
once; --cv_seeds repetitions), the protocol of tools/convergence_cv.py: a cell is the POOLED test accuracy over
1797 x seeds predictions with its binomial standard error, so that differences between rules can be told from noise
(rounds 1-2 read a single 297-image split to the third digit: sigma ~ 0.7 % there).

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
    ap.add_argument("--cv_folds", type=int, default=5, help="--data digits: folds of the cross-validation")
    ap.add_argument("--cv_seeds", type=int, default=1, help="--data digits: repetitions with another split / initialisation")
    args = ap.parse_args()

    import torch
    from distributed_tensorflow_amd.data.mnist import read_data_sets
    from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
    from distributed_tensorflow_amd.models.spec import ModelSpec
    from distributed_tensorflow_amd.utils.lr_schedule import LrSchedule

    import numpy as np
    spec = ModelSpec.from_name(args.model)

    def train_eval(ds, dt, B, rule, steps, seed):
        sched = LrSchedule(args.base_lr, B, args.base_batch, rule, int(round(args.warmup_frac * steps)) if rule != "none" else 0)
        eng = HipEngine(spec, B, ds, optimizer=args.optimizer, learning_rate=args.base_lr, seed=seed, dtype=args.dtype)
        eng.init_params(seed=seed + 1)
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
        tl, ta = (float("nan"), float("nan")) if diverged else eng.evaluate(dt.images, dt.labels)
        out = dict(peak_lr=sched.peak_lr, warmup=sched.warmup_steps, train_loss=(sum(tail) / len(tail)) if tail and not diverged
                   else float("nan"), test_loss=tl, test_acc=ta, diverged=diverged, images_per_s=(s + 1) * B / dt_s, n_test=dt.n)
        del eng
        torch.cuda.empty_cache()
        return out

    splits = []          # (train DeviceDataset, test DeviceDataset, seed)
    if args.data == "digits":
        from sklearn.datasets import load_digits
        dg = load_digits()
        img = np.pad(np.kron(dg.images, np.ones((3, 3))), ((0, 0), (2, 2), (2, 2)))
        X = np.clip(img / 16.0, 0.0, 1.0).astype(np.float32).reshape(len(img), 784)
        Y = dg.target.astype(np.int64)
        for cs in range(args.cv_seeds):
            folds = np.array_split(np.random.RandomState(100 + cs).permutation(len(Y)), args.cv_folds)
            for k in range(args.cv_folds):
                tr = np.concatenate([folds[j] for j in range(args.cv_folds) if j != k])
                splits.append((DeviceDataset.from_float(torch.from_numpy(X[tr]), torch.from_numpy(Y[tr]), "cuda", dtype=args.dtype),
                               DeviceDataset.from_float(torch.from_numpy(X[folds[k]]), torch.from_numpy(Y[folds[k]]), "cuda",
                                                        dtype=args.dtype), args.seed + 1000 * cs + k))
    else:
        data = read_data_sets("", one_hot=False, synthetic=True)
        tr, te = data.train, data.test
        splits.append((DeviceDataset.from_float(torch.from_numpy(tr.images), torch.from_numpy(tr.int_labels), "cuda", dtype=args.dtype),
                       DeviceDataset.from_float(torch.from_numpy(te.images), torch.from_numpy(te.int_labels), "cuda", dtype=args.dtype),
                       args.seed))
    ds, dt = splits[0][0], splits[0][1]
    budget = int(args.epochs * ds.n)
    rows = []
    for B in [int(b) for b in args.batches.split(",")]:
        steps = max(1, budget // B)
        for rule in args.rules.split(","):
            if B == args.base_batch and rule != "none":
                continue
            cells = [train_eval(dtr, dte, B, rule, steps, seed) for dtr, dte, seed in splits]
            n_all = sum(c["n_test"] for c in cells)
            div = sum(1 for c in cells if c["diverged"])
            ok = [c for c in cells if not c["diverged"]]
            acc = sum(c["test_acc"] * c["n_test"] for c in ok) / max(1, sum(c["n_test"] for c in ok)) if ok else float("nan")
            n_ok = sum(c["n_test"] for c in ok)
            row = dict(batch=B, rule=rule, peak_lr=cells[0]["peak_lr"], warmup=cells[0]["warmup"], steps=steps,
                       train_loss=float(np.mean([c["train_loss"] for c in ok])) if ok else float("nan"),
                       test_loss=float(np.mean([c["test_loss"] for c in ok])) if ok else float("nan"), test_acc=acc,
                       sigma=math.sqrt(max(acc * (1 - acc), 0.0) / n_ok) if ok and n_ok else float("nan"),
                       diverged=div == len(cells), diverged_cells=div, cells=len(cells), n_test=n_all,
                       images_per_s=float(np.mean([c["images_per_s"] for c in cells])))
            rows.append(row)
            print(json.dumps(row), flush=True)
    out = os.path.join(ROOT, "gpurun_out", f"lr_sweep_{args.data}_{args.model}_{args.optimizer}.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(dict(args=vars(args), n_train=ds.n, n_test=dt.n, sample_budget=budget, rows=rows), f, indent=1)
    rules = args.rules.split(",")
    print(f"\n{args.data}, {spec.name}, {args.optimizer}, base lr {args.base_lr:g} @ batch {args.base_batch}, "
          f"{args.epochs:g} epochs ({budget} samples), " + (f"{args.cv_folds}-fold cross-validation x {args.cv_seeds} seed(s): "
          f"{rows[0]['n_test']} test predictions per cell; cell = pooled test accuracy +- binomial sigma / mean test loss (peak lr)"
          if args.data == "digits" else f"{dt.n} test images; cell = test accuracy / test loss (peak lr)"))
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
                cells.append(f"{100 * m[0]['test_acc']:.2f} +- {100 * m[0]['sigma']:.2f} % / {m[0]['test_loss']:.3f} ({m[0]['peak_lr']:g})"
                             + (f" [{m[0]['diverged_cells']} of {m[0]['cells']} folds diverged]" if m[0]["diverged_cells"] else ""))
        st = [r for r in rows if r["batch"] == B][0]["steps"]
        print(f"| {B} | {st} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main()
