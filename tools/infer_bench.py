#!/usr/bin/env python3
"""Forward-only (evaluation / serving) throughput of the engine on one MI355X: the reference's test-set evaluation
(`M:190-195`, accuracy over mnist.test) is `HipEngine.evaluate` -- per chunk the fused conv1+conv2 forward, the fc1
split-K GEMM, the head kernel (no backward) and the K9 metric reduction, one host synchronisation per call.

  python tools/infer_bench.py [--dtype bf16|fp32] [--batch 1024] [--images 65536] [--model deep_mnist]
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine  # noqa: E402
from distributed_tensorflow_amd.models.spec import ModelSpec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--images", type=int, default=65536)
ap.add_argument("--model", default="deep_mnist")
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
spec = ModelSpec.from_name(a.model)
ds = DeviceDataset.synthetic(a.images, seed=1234, dtype=a.dtype)
eng = HipEngine(spec, a.batch, ds, seed=1, dtype=a.dtype)
eng.init_params(seed=1)
for _ in range(3):
    loss, acc = eng.evaluate(ds.images, ds.labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    loss, acc = eng.evaluate(ds.images, ds.labels)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
fwd, _ = spec.flops_per_image()
print(json.dumps({"metric": "evaluation images/sec (forward only, 1 GPU)", "value": round(a.images / dt, 1), "dtype": a.dtype,
                  "model": spec.name, "chunk": a.batch, "images_per_call": a.images, "ms_per_call": round(1e3 * dt, 3),
                  "us_per_chunk": round(1e6 * dt / (a.images / a.batch), 2), "model_tflops": round(a.images / dt * fwd / 1e12, 2),
                  "loss": round(loss, 4), "accuracy": round(acc, 4)}))
