"""Large-batch learning-rate rule (BASELINE.json config 5: "widened CNN for LR-scaling experiments").

The reference fixes ``learning_rate`` (0.001, reference .idea/MNISTDist.py:30) for its batch of 128 (M:29).  Growing the
global batch -- more workers in synchronous mode, or a larger per-worker batch -- keeps the gradient's mean but shrinks
its noise, and the rate has to follow:

    none     lr = base                                   (the reference's behaviour)
    linear   lr = base * (global_batch / base_batch)     (Goyal et al., "Accurate, Large Minibatch SGD")
    sqrt     lr = base * sqrt(global_batch / base_batch) (Krizhevsky, "One weird trick ..."; the usual choice for Adam)

with a linear warm-up from ``base`` to the scaled rate over ``warmup_steps`` global steps (large rates are unstable
from a random start).  ``LrSchedule(step)`` is pure; the trainer writes its value into the optimizer's device-side
hyper-parameters only on steps where it changes.
"""
from __future__ import annotations

import math

RULES = ("none", "linear", "sqrt")


def scaled_rate(base_lr: float, global_batch: int, base_batch: int = 128, rule: str = "none") -> float:
    if rule not in RULES:
        raise ValueError(f"--lr_scaling must be one of {RULES}, got {rule!r}")
    k = float(global_batch) / float(max(1, base_batch))
    if rule == "linear":
        return base_lr * k
    if rule == "sqrt":
        return base_lr * math.sqrt(k)
    return base_lr


class LrSchedule:
    def __init__(self, base_lr: float, global_batch: int, base_batch: int = 128, rule: str = "none",
                 warmup_steps: int = 0):
        self.base_lr = float(base_lr)
        self.peak_lr = scaled_rate(base_lr, global_batch, base_batch, rule)
        self.warmup_steps = max(0, int(warmup_steps))
        self.rule = rule

    def __call__(self, step: int) -> float:
        """Rate for the update that takes global_step from ``step`` to ``step + 1``."""
        if self.warmup_steps and step < self.warmup_steps and self.peak_lr != self.base_lr:
            return self.base_lr + (self.peak_lr - self.base_lr) * (step + 1) / self.warmup_steps
        return self.peak_lr

    @property
    def constant(self) -> bool:
        return self.warmup_steps == 0 or self.peak_lr == self.base_lr
