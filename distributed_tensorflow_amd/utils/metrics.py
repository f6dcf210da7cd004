"""Console + JSONL observability.

The reference's only observable output is a print every ``display_step`` steps
(reference .idea/MNISTDist.py:181-186) and "Optimization Finished!" (M:193).  The
console line keeps that exact format (scripts grep it); everything else (images/s,
step time, eval accuracy) goes to a JSONL metrics file and optional extra lines.
"""
from __future__ import annotations

import json
import os
import time
from typing import Optional


def format_display_line(job_name: str, task_index: int, step: int, loss: float, acc: float) -> str:
    """Byte-for-byte what the reference's Python-3 ``print(a, b, c, ...)`` call emits."""
    parts = ("job: %s/%s" % (job_name, task_index), "step: ", step, "mini_batch loss: ", loss,
             "training accuracy: ", acc)
    return " ".join(str(p) for p in parts)


class StepTimer:
    """Wall-clock throughput between display points (device work is async: call sync first)."""

    def __init__(self):
        self.t0 = time.perf_counter()
        self.images = 0
        self.steps = 0

    def add(self, images: int, steps: int = 1):
        self.images += images
        self.steps += steps

    def lap(self):
        t1 = time.perf_counter()
        dt = max(t1 - self.t0, 1e-9)
        out = dict(images_per_sec=self.images / dt, steps_per_sec=self.steps / dt,
                   ms_per_step=1e3 * dt / max(1, self.steps), seconds=dt)
        self.t0, self.images, self.steps = t1, 0, 0
        return out


class MetricsLogger:
    def __init__(self, path: Optional[str] = None, also_print: bool = False):
        self.path = path
        self.also_print = also_print
        self._f = None
        if path:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            self._f = open(path, "a", buffering=1)

    def log(self, **kv):
        kv.setdefault("time", time.time())
        line = json.dumps(kv, sort_keys=True)
        if self._f:
            self._f.write(line + "\n")
        if self.also_print:
            print(line, flush=True)

    def close(self):
        if self._f:
            self._f.close()
            self._f = None
