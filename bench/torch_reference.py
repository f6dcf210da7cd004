#!/usr/bin/env python3
"""On-device baseline: the same model / batch / optimizer in plain PyTorch-ROCm (ATen ->
MIOpen + hipBLASLt, autograd), eager and under torch.compile-free CUDA-graph-free conditions.
The reference repo publishes no numbers and TensorFlow is not installable here, so this is the
"what you get without the hand-written kernels" comparison point quoted in BASELINE.md.

  python bench/torch_reference.py --batch 1024 --steps 50 [--bf16] [--channels_last]
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class DeepMnist(nn.Module):
    def __init__(self, c1=32, c2=64, hidden=1024):
        super().__init__()
        self.c1 = nn.Conv2d(1, c1, 5, padding=2)
        self.c2 = nn.Conv2d(c1, c2, 5, padding=2)
        self.f1 = nn.Linear(49 * c2, hidden)
        self.f2 = nn.Linear(hidden, 10)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.c1(x)), 2)
        x = F.max_pool2d(F.relu(self.c2(x)), 2)
        x = x.flatten(1)
        return self.f2(F.relu(self.f1(x)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--channels_last", action="store_true")
    ap.add_argument("--wide", action="store_true")
    ap.add_argument("--device", default="cuda")
    a = ap.parse_args()
    dev = torch.device(a.device)
    m = (DeepMnist(128, 256, 4096) if a.wide else DeepMnist()).to(dev)
    if a.channels_last:
        m = m.to(memory_format=torch.channels_last)
    opt = torch.optim.SGD(m.parameters(), lr=1e-3)
    N = 60000
    xs = torch.rand(N, 1, 28, 28, device=dev)
    ys = torch.randint(0, 10, (N,), device=dev)

    def step(i):
        idx = torch.randint(0, N, (a.batch,), device=dev)
        x, y = xs[idx], ys[idx]
        if a.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        with torch.autocast(dev.type, dtype=torch.bfloat16, enabled=a.bf16):
            loss = F.cross_entropy(m(x), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    for i in range(a.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    sync()
    dt = time.perf_counter() - t0
    print(json.dumps({"engine": "pytorch-rocm eager (ATen/MIOpen/hipBLASLt)", "device": str(dev),
                      "dtype": "bf16-autocast" if a.bf16 else "fp32", "channels_last": a.channels_last,
                      "model": "wide" if a.wide else "deep_mnist", "batch": a.batch,
                      "images_per_sec": round(a.batch * a.steps / dt, 1), "ms_per_step": round(1e3 * dt / a.steps, 4)}))


if __name__ == "__main__":
    main()
