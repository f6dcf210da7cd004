"""Plain-PyTorch fp32 oracle of the deep-MNIST model, TF layouts throughout.

Two jobs:
  * numerics ground truth for the HIP engine (same parameter tensors, same
    batches -> loss / gradients must agree within bf16 tolerance);
  * the compute backend of CPU workers (the "1 ps + 2 workers on localhost
    CPU" plumbing configuration of BASELINE.json runs on this + gloo).

It mirrors the reference graph (reference .idea/MNISTDist.py:113-156):
conv_net -> reduce_mean(softmax_cross_entropy) -> GradientDescentOptimizer
(or Adam) with a global_step counter, TF-faithful initialisation
(truncated_normal stddev 0.1 weights, constant 0.1 biases).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from ..ops import reference as R
from .spec import ArenaLayout, ModelSpec


class OracleModel:
    def __init__(self, spec: Optional[ModelSpec] = None, device="cpu", seed: int = 0, optimizer: str = "sgd",
                 learning_rate: float = 1e-3, keep_prob: float = 1.0, adam_betas=(0.9, 0.999),
                 adam_eps: float = 1e-8):
        self.spec = spec or ModelSpec()
        self.layout = ArenaLayout(self.spec)
        self.device = torch.device(device)
        self.keep_prob = keep_prob
        self.optimizer = optimizer
        self.lr = learning_rate
        self.betas, self.eps = adam_betas, adam_eps
        self.gen = torch.Generator(device="cpu").manual_seed(seed)
        # one flat arena (same layout as the GPU engine) with per-tensor views
        self.arena = torch.zeros(self.layout.total, dtype=torch.float32, device=self.device)
        self.grad_arena = torch.zeros_like(self.arena)
        self.params: Dict[str, torch.Tensor] = {n: self.layout.view(self.arena, n) for n in self.layout.names()}
        self.global_step = 0
        self._adam_m = self._adam_v = None
        self._adam_t = 0
        self.dropout_mask = None      # explicit [B, hidden] keep mask for the next training forward (tests)
        self.init_params()

    # ------------------------------------------------------------------ init
    def init_params(self):
        for name, shape in self.spec.shapes().items():
            if name.startswith("w") or name == "out_w":
                v = R.truncated_normal(shape, 0.1, generator=self.gen)
            else:
                v = torch.full(shape, 0.1)
            self.params[name].copy_(v.to(self.device))

    def load_params(self, tensors: Dict[str, torch.Tensor]):
        for n in self.layout.names():
            self.params[n].copy_(tensors[n].to(self.device, torch.float32).reshape(self.layout.shape(n)))

    def export_params(self) -> Dict[str, torch.Tensor]:
        return {n: self.params[n].detach().cpu().clone() for n in self.layout.names()}

    # --------------------------------------------------------------- compute
    def logits(self, x: torch.Tensor, train: bool = False, params=None) -> torch.Tensor:
        p = self.params if params is None else params
        keep = self.keep_prob if train else 1.0
        mask = self.dropout_mask.to(self.device) if (train and self.dropout_mask is not None) else None
        return self._net(x.to(self.device, torch.float32), p, keep, mask)

    def _net(self, x, p, keep, mask):
        return R.conv_net(x, p, keep, mask=mask)

    def loss_and_accuracy(self, x, labels, train: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        with torch.no_grad():
            z = self.logits(x, train)
            labels = labels.to(self.device)
            return R.softmax_cross_entropy_with_logits(z, labels).mean(), R.accuracy(z, labels)

    def compute_gradients(self, x, labels) -> Tuple[float, float]:
        """Fill grad_arena with d(mean loss)/d(params); returns (loss, accuracy)."""
        leaf = self.arena.detach().clone().requires_grad_(True)
        p = {n: self.layout.view(leaf, n) for n in self.layout.names()}
        z = self.logits(x, train=True, params=p)
        labels = labels.to(self.device)
        loss = R.softmax_cross_entropy_with_logits(z, labels).mean()
        loss.backward()
        self.grad_arena.copy_(leaf.grad)
        return float(loss.item()), float(R.accuracy(z.detach(), labels).item())

    def grads(self) -> Dict[str, torch.Tensor]:
        return {n: self.layout.view(self.grad_arena, n).detach().cpu().clone() for n in self.layout.names()}

    def apply_gradients(self, grad_arena: Optional[torch.Tensor] = None, scale: float = 1.0, step_inc: int = 1):
        """ApplyGradientDescent / Adam over the flat arena + global_step increment."""
        g = (self.grad_arena if grad_arena is None else grad_arena.to(self.device)) * scale
        if self.optimizer == "sgd":
            self.arena.add_(g, alpha=-self.lr)
        elif self.optimizer == "adam":
            if self._adam_m is None:
                self._adam_m = torch.zeros_like(self.arena)
                self._adam_v = torch.zeros_like(self.arena)
            self._adam_t += 1
            b1, b2 = self.betas
            self._adam_m.mul_(b1).add_(g, alpha=1 - b1)
            self._adam_v.mul_(b2).addcmul_(g, g, value=1 - b2)
            lr_t = self.lr * (1 - b2 ** self._adam_t) ** 0.5 / (1 - b1 ** self._adam_t)
            self.arena.addcdiv_(self._adam_m, self._adam_v.sqrt().add_(self.eps), value=-lr_t)
        else:
            raise ValueError(self.optimizer)
        self.global_step += step_inc

    def train_step(self, x, labels) -> Tuple[float, float]:
        loss, acc = self.compute_gradients(x, labels)
        self.apply_gradients()
        return loss, acc


class Bf16EmulatedOracle(OracleModel):
    """The fp32 oracle with a bf16 rounding at every point where the mixed-precision engine stores a bf16 tensor
    (ops/reference.py conv_net_bf16_emulated).  Engine-vs-fp32 tolerances in the tests are derived from the distance
    between this model and the plain fp32 oracle."""

    def _net(self, x, p, keep, mask):
        return R.conv_net_bf16_emulated(x, p, keep, mask=mask)
