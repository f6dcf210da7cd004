#!/usr/bin/env python3
"""Convergence evidence with a resolution that can be defended, from what is importable here (VERDICT r2 Next #8).

There is no MNIST in this environment (no network), so "steps to 99.2 % on MNIST" stays PARITY UNPINNED.  What can be
measured: scikit-learn's bundled UCI optdigits (1797 real handwritten digits, 8x8 upsampled to 28x28) under 5-fold
cross-validation -- every image is a test image exactly once per seed -- times 3 seeds, i.e. 5391 test predictions per
configuration (binomial sigma ~ 0.15 % at 98.5 %; a single 297-image split, as in rounds 1-2, has sigma ~ 0.7 %).

  engines   fp32 HIP engine | bf16 HIP engine | bf16 + 16-bit dense gradient (--grad_comm bf16's rounding) | fp32 PyTorch oracle
  recipes   tutorial   Adam 1e-3, keep_prob 0.5, batch 50, 1500 steps  (the TF tutorial's optimizer / dropout / batch;
                       its 1e-4 x 20000 steps is sized for 55 000 images, not 1438)
            reference  SGD 1e-3, keep_prob 1.0, batch 128, 10000 steps (MNISTDist.py:28-30,149,179 exactly)

Every engine of a (fold, seed) cell starts from the SAME initial weights (the oracle's TF-style init).

    python tools/convergence_cv.py [--seeds 3] [--folds 5] [--out gpurun_out/convergence_cv.md]
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
import distributed_tensorflow_amd  # noqa: E402,F401
import numpy as np  # noqa: E402
import torch  # noqa: E402

RECIPES = {
    "tutorial (Adam 1e-3, keep 0.5, batch 50, 1500 steps)": dict(optimizer="adam", lr=1e-3, keep=0.5, batch=50, steps=1500),
    "reference (SGD 1e-3, keep 1.0, batch 128, 10000 steps; M:28-30,149,179)": dict(optimizer="sgd", lr=1e-3, keep=1.0, batch=128,
                                                                                  steps=10000),
}
ENGINES = ["hip fp32", "hip bf16", "hip bf16 + 16-bit dense gradient", "oracle fp32 (PyTorch)"]


def digits():
    from sklearn.datasets import load_digits
    d = load_digits()
    img = np.kron(d.images, np.ones((3, 3)))
    img = np.pad(img, ((0, 0), (2, 2), (2, 2)))
    img = np.clip(img / 16.0, 0.0, 1.0).astype(np.float32).reshape(len(img), 784)
    return img, d.target.astype(np.int64)


def run_hip(dtype, g16, init, xtr, ytr, xte, yte, r, seed):
    from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
    from distributed_tensorflow_amd.models.spec import ModelSpec
    ds = DeviceDataset.from_float(torch.from_numpy(xtr), torch.from_numpy(ytr), "cuda", dtype=dtype)
    dt = DeviceDataset.from_float(torch.from_numpy(xte), torch.from_numpy(yte), "cuda", dtype=dtype)
    eng = HipEngine(ModelSpec(), r["batch"], ds, optimizer=r["optimizer"], learning_rate=r["lr"], keep_prob=r["keep"], seed=seed,
                    multi_stream=False, fuse_dense_update=not g16, dtype=dtype)
    eng.load_params(init)
    if g16:
        from distributed_tensorflow_amd.parallel.sync_dp import SyncDataParallel
        dp = SyncDataParallel(eng, grad_comm="bf16")
        step = dp.train_step
    else:
        step = eng.train_step
    for _ in range(r["steps"]):
        step()
    if g16:
        dp.flush()
    torch.cuda.synchronize()
    loss, acc = eng.evaluate(dt.images, dt.labels)
    return acc, loss


def run_oracle(init, xtr, ytr, xte, yte, r, seed):
    from distributed_tensorflow_amd.models.oracle import OracleModel
    m = OracleModel(device="cuda", seed=seed, optimizer=r["optimizer"], learning_rate=r["lr"], keep_prob=r["keep"])
    m.load_params(init)
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    x, y = torch.from_numpy(xtr).cuda(), torch.from_numpy(ytr).cuda()
    n, B = len(ytr), r["batch"]
    perm, pos = torch.randperm(n, generator=g), 0
    for _ in range(r["steps"]):
        if pos + B > n:
            perm, pos = torch.randperm(n, generator=g), 0
        idx = perm[pos:pos + B].cuda()
        pos += B
        m.train_step(x[idx], y[idx])
    loss, acc = m.loss_and_accuracy(torch.from_numpy(xte).cuda(), torch.from_numpy(yte).cuda())
    return float(acc), float(loss)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--folds", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "convergence_cv.md"))
    ap.add_argument("--quick", action="store_true", help="a tenth of the steps (smoke run of this script)")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    from distributed_tensorflow_amd.models.oracle import OracleModel
    from distributed_tensorflow_amd.parallel import comm
    comm.init_process_group(backend="nccl", rank=0, world_size=1, device=torch.device("cuda", 0))   # for the 16-bit gradient path
    X, Y = digits()
    N = len(Y)
    results = {}
    t0 = time.time()
    for rname, r in RECIPES.items():
        r = dict(r)
        if args.quick:
            r["steps"] //= 10
        for seed in range(args.seeds):
            perm = np.random.RandomState(100 + seed).permutation(N)
            folds = np.array_split(perm, args.folds)
            for k in range(args.folds):
                te = folds[k]
                tr = np.concatenate([folds[j] for j in range(args.folds) if j != k])
                init = OracleModel(seed=1000 * seed + k).export_params()
                cell = dict(xtr=X[tr], ytr=Y[tr], xte=X[te], yte=Y[te], r=r, seed=1000 * seed + k)
                for ename in ENGINES:
                    if ename.startswith("oracle"):
                        acc, loss = run_oracle(init, **cell)
                    else:
                        acc, loss = run_hip("fp32" if "fp32" in ename else "bf16", "16-bit" in ename, init, **cell)
                    results.setdefault((rname, ename), []).append(dict(seed=seed, fold=k, n=len(te), acc=acc, loss=loss))
                print(f"[{time.time() - t0:6.0f}s] {rname.split(' ')[0]} seed {seed} fold {k}: " +
                      "  ".join(f"{e.split(' ')[0]} {e.split(' ')[1]}{'+g16' if '16-bit' in e else ''} {results[(rname, e)][-1]['acc']:.4f}"
                                for e in ENGINES), flush=True)
    lines = ["# 5-fold cross-validation on the 1797 UCI optdigits images (28x28), %d seeds: test accuracy" % args.seeds, "",
             "Every image is a test image once per seed; `pooled` = correct / all test predictions, `sigma` = its binomial",
             "standard error; `folds` = mean +- standard deviation over the (seed, fold) cells; `vs oracle` = mean and standard",
             "error of the PAIRED per-cell accuracy difference to the fp32 PyTorch oracle (same split, same initial weights).",
             "NOT MNIST: the 99.2 % figure of the TF tutorial stays parity unpinned (no MNIST files, no network).", ""]
    for rname in RECIPES:
        lines += [f"## {rname}", "", "| engine | pooled accuracy | sigma | folds mean +- sd | mean test loss | vs oracle (paired) |", "|---|---|---|---|---|---|"]
        base = results[(rname, ENGINES[-1])]
        for ename in ENGINES:
            cells = results[(rname, ename)]
            tot = sum(c["n"] for c in cells)
            correct = sum(c["acc"] * c["n"] for c in cells)
            p = correct / tot
            sig = math.sqrt(p * (1 - p) / tot)
            accs = np.array([c["acc"] for c in cells])
            diff = accs - np.array([c["acc"] for c in base])
            se = diff.std(ddof=1) / math.sqrt(len(diff)) if len(diff) > 1 else float("nan")
            lines.append(f"| {ename} | {100 * p:.2f} % ({int(round(correct))}/{tot}) | {100 * sig:.2f} % | {100 * accs.mean():.2f} +- "
                         f"{100 * accs.std(ddof=1):.2f} % | {np.mean([c['loss'] for c in cells]):.4f} | "
                         + ("-" if ename == ENGINES[-1] else f"{100 * diff.mean():+.2f} +- {100 * se:.2f} %") + " |")
        lines.append("")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(args.out.replace(".md", ".json"), "w") as f:
        json.dump({f"{a} | {b}": v for (a, b), v in results.items()}, f)
    print("\n".join(lines))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
