"""Compute backends behind one small interface, so the trainer / supervisor / parallel
strategies are identical for the MI355X engine and the CPU oracle.

  HipBackend  hand-written gfx950 kernels on an HBM-resident dataset (engine.HipEngine), bf16 or exact fp32
  CpuBackend  fp32 PyTorch oracle + TF-style ``next_batch`` input pipeline; this is what the
              "1 ps + 2 workers on localhost CPU" plumbing configuration runs on (gloo)

Both expose the model state as ONE flat fp32 parameter tensor and ONE flat gradient tensor
(models.spec.ArenaLayout), which is what the collectives move.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .data.mnist import Datasets
from .models.oracle import OracleModel
from .models.spec import ArenaLayout, ModelSpec


class Backend:
    name = "abstract"
    spec: ModelSpec
    layout: ArenaLayout
    batch: int

    # flat state -----------------------------------------------------------
    @property
    def params(self) -> torch.Tensor: raise NotImplementedError
    @property
    def grads(self) -> torch.Tensor: raise NotImplementedError
    def params_changed(self): pass
    # step pieces ----------------------------------------------------------
    def compute_gradients(self): raise NotImplementedError
    def apply_gradients(self, scale: float = 1.0, step_inc: int = 1): raise NotImplementedError
    def advance_input(self):
        """Move on to the next batch (and dropout draw) after compute_gradients() when NO local apply_gradients()
        follows -- asynchronous workers, whose updates happen on the ps.  Backends whose input pipeline advances by
        itself (next_batch) need nothing."""
    def last_metrics(self) -> Tuple[float, float]: raise NotImplementedError
    def evaluate(self, split: str = "test") -> Tuple[float, float]: raise NotImplementedError
    def synchronize(self): pass
    # state I/O ------------------------------------------------------------
    def get_global_step(self) -> int: raise NotImplementedError
    def set_learning_rate(self, lr: float): raise NotImplementedError
    def set_global_step(self, v: int): raise NotImplementedError
    def init_params(self, seed: int): raise NotImplementedError

    def export_params(self) -> Dict[str, np.ndarray]:
        flat = self.params.detach().cpu()
        return {n: self.layout.view(flat, n).numpy().copy() for n in self.layout.names()}

    def load_params(self, tensors: Dict[str, np.ndarray]):
        flat = torch.zeros(self.layout.total, dtype=torch.float32)
        for n in self.layout.names():
            self.layout.view(flat, n).copy_(torch.as_tensor(np.asarray(tensors[n], dtype=np.float32)))
        self.params.copy_(flat.to(self.params.device))
        self.params_changed()

    def export_optimizer_state(self) -> Optional[dict]: return None
    def load_optimizer_state(self, state: dict): pass

    def snapshot(self, with_optimizer: bool = True):
        """A consistent copy of (parameters, global_step, optimizer slots) taken BETWEEN two steps without waiting
        for it: returns ``finish() -> (params by name, global_step, optimizer state or None)``, to be called from
        the checkpoint writer thread (it blocks there until the copy has landed in host memory)."""
        params, step = self.export_params(), self.get_global_step()
        opt = self.export_optimizer_state() if with_optimizer else None
        return lambda: (params, step, opt)

    # Job-wide control flags: the last 4 floats of the gradient arena travel in every synchronous step's
    # (tail) all-reduce.  Each worker publishes ITS flags there every step -- slot 0: stop request, slot 1: "gather the
    # sharded master weights" request -- and every worker reads the same sums after the same step.
    CTRL_STOP, CTRL_GATHER = 0, 1

    @property
    def ctrl_src(self) -> torch.Tensor:
        """This rank's own 4 control flags (the source that is published every step)."""
        if getattr(self, "_ctrl_src", None) is None:
            self._ctrl_src = torch.zeros(4, dtype=torch.float32, device=self.params.device)
        return self._ctrl_src

    def set_control(self, slot: int, value: float):
        self.ctrl_src[slot:slot + 1].fill_(float(value))

    def publish_control(self):
        """Copy this rank's flags into the control block of the gradient arena (before the all-reduce).  The native
        data-parallel step does this inside its slab reduction; the generic path calls it."""
        self.grads[self.layout.ctrl:self.layout.ctrl + 4].copy_(self.ctrl_src)

    def control_sums(self):
        """The 4 summed flags of the last step's all-reduce (reads device memory: synchronises the stream on the GPU)."""
        return self.grads[self.layout.ctrl:self.layout.ctrl + 4].tolist()

    def raise_stop_flag(self):
        self.set_control(self.CTRL_STOP, 1.0)

    def stop_flag_sum(self) -> float:
        """> 0 once any worker has raised its flag."""
        return float(self.control_sums()[self.CTRL_STOP])

    # optimizer slots as flat arenas, for the chief -> workers broadcast after a restore (None: no slots)
    def adam_arenas(self) -> Optional[Tuple[torch.Tensor, torch.Tensor]]: return None
    def get_adam_t(self) -> int: return 0
    def set_adam_t(self, t: int): pass


class CpuBackend(Backend):
    name = "cpu"

    def __init__(self, spec: ModelSpec, batch: int, data: Datasets, optimizer="sgd", learning_rate=1e-3,
                 keep_prob=1.0, seed=0):
        self.spec, self.batch, self.data = spec, batch, data
        self.model = OracleModel(spec, device="cpu", seed=seed, optimizer=optimizer, learning_rate=learning_rate,
                                 keep_prob=keep_prob)
        self.layout = self.model.layout
        self._metrics = (float("nan"), float("nan"))

    params = property(lambda self: self.model.arena)
    grads = property(lambda self: self.model.grad_arena)

    def init_params(self, seed: int):
        self.model.gen.manual_seed(seed)
        self.model.init_params()

    def compute_gradients(self):
        x, y = self.data.train.next_batch(self.batch)     # reference M:178
        self._metrics = self.model.compute_gradients(torch.from_numpy(np.ascontiguousarray(x)),
                                                     torch.from_numpy(np.ascontiguousarray(y)))

    def apply_gradients(self, scale: float = 1.0, step_inc: int = 1):
        self.model.apply_gradients(scale=scale, step_inc=step_inc)

    def last_metrics(self):
        return self._metrics

    def evaluate(self, split: str = "test"):
        ds = getattr(self.data, split)
        tot_l = tot_a = 0.0
        n = ds.num_examples
        for lo in range(0, n, 1000):
            x = torch.from_numpy(ds.images[lo:lo + 1000])
            y = torch.from_numpy(np.ascontiguousarray(ds.labels[lo:lo + 1000]))
            l, a = self.model.loss_and_accuracy(x, y)
            tot_l += float(l) * x.shape[0]
            tot_a += float(a) * x.shape[0]
        return tot_l / n, tot_a / n

    def get_global_step(self): return self.model.global_step
    def set_global_step(self, v): self.model.global_step = int(v)
    def set_learning_rate(self, lr: float): self.model.lr = float(lr)

    def adam_arenas(self):
        m = self.model
        if m.optimizer != "adam":
            return None
        if m._adam_m is None:
            m._adam_m = torch.zeros_like(m.arena)
            m._adam_v = torch.zeros_like(m.arena)
        return m._adam_m, m._adam_v

    def get_adam_t(self): return int(self.model._adam_t)
    def set_adam_t(self, t): self.model._adam_t = int(t)

    def export_optimizer_state(self):
        m = self.model
        if m.optimizer != "adam" or m._adam_m is None:
            return None
        return {"m": {n: self.layout.view(m._adam_m, n).numpy().copy() for n in self.layout.names()},
                "v": {n: self.layout.view(m._adam_v, n).numpy().copy() for n in self.layout.names()},
                "t": m._adam_t}

    def load_optimizer_state(self, state):
        m = self.model
        if m.optimizer != "adam" or not state:
            return
        m._adam_m = torch.zeros_like(m.arena)
        m._adam_v = torch.zeros_like(m.arena)
        for n in self.layout.names():
            self.layout.view(m._adam_m, n).copy_(torch.as_tensor(state["m"][n]))
            self.layout.view(m._adam_v, n).copy_(torch.as_tensor(state["v"][n]))
        m._adam_t = int(state.get("t", 0))


class HipBackend(Backend):
    name = "hip"

    def __init__(self, spec: ModelSpec, batch: int, data: Datasets, optimizer="sgd", learning_rate=1e-3,
                 keep_prob=1.0, seed=0, device="cuda", use_graph=True, dtype="bf16"):
        from .engine import DeviceDataset, HipEngine
        self.spec, self.batch, self.data = spec, batch, data
        self.dtype = dtype
        tr = data.train
        self.train_ds = DeviceDataset.from_float(torch.from_numpy(tr.images), torch.from_numpy(tr.int_labels), device,
                                                 dtype=dtype)
        self._eval_cache = {}
        self.engine = HipEngine(spec, batch, self.train_ds, optimizer=optimizer, learning_rate=learning_rate,
                                keep_prob=keep_prob, seed=seed, device=device, fuse_dense_update=True, dtype=dtype)
        self.layout = self.engine.layout
        # measured on MI355X: eager launches of the 8-kernel step beat graph replay up to 1024 images/step
        self.use_graph = bool(use_graph) and batch > 1024 and dtype == "bf16"
        self._host_step = 0

    params = property(lambda self: self.engine.params)
    grads = property(lambda self: self.engine.grads)
    ctrl_src = property(lambda self: self.engine.ctrl_src)

    def init_params(self, seed: int):
        self.engine.init_params(seed)

    def params_changed(self):
        self.engine.refresh_shadows()

    def compute_gradients(self):
        e, ds = self.engine, self.train_ds
        e.launch_sample()
        e.launch_forward(ds.images, ds.labels, e.idx, backward=True)
        e.launch_backward_dense()
        e.launch_backward_conv(ds.images, e.idx)

    def advance_input(self):
        # the sample cursor and the dropout offset live in the device step state and are normally advanced by the
        # optimizer launch; without one, every step would draw the same batch and the same mask
        e = self.engine
        from .ops.native import check
        check(e.lib.dtf_advance_state(e.state.data_ptr(), e.B, 0, 1, torch.cuda.current_stream(e.device).cuda_stream),
              "advance_state")

    def apply_gradients(self, scale: float = 1.0, step_inc: int = 1):
        e = self.engine
        if scale != getattr(self, "_scale", None):
            e.set_grad_scale(scale)
            self._scale = scale
        e.step_inc = step_inc
        e.launch_update()
        self._host_step += step_inc

    def train_step_local(self):
        """Fused single-GPU step (HIP-graph replay when enabled)."""
        e = self.engine
        if self.use_graph and e._graph is None:
            e.capture()
        e.train_step()
        self._host_step += e.step_inc

    def last_metrics(self):
        return self.engine.last_metrics()

    def evaluate(self, split: str = "test"):
        from .engine import DeviceDataset
        if split not in self._eval_cache:
            ds = getattr(self.data, split)
            self._eval_cache[split] = DeviceDataset.from_float(torch.from_numpy(ds.images),
                                                               torch.from_numpy(ds.int_labels), self.engine.device,
                                                               dtype=self.dtype)
        d = self._eval_cache[split]
        return self.engine.evaluate(d.images, d.labels)

    def synchronize(self):
        torch.cuda.synchronize(self.engine.device)

    def snapshot(self, with_optimizer: bool = True):
        """Device-side snapshot on a side stream: D2D copy of the arenas (the training stream only waits for THAT,
        a few microseconds), then an asynchronous D2H copy into pinned memory that overlaps the following steps."""
        e = self.engine
        dev = e.device
        adam = with_optimizer and e.optimizer == "adam"
        if getattr(self, "_snap", None) is None:
            n = self.layout.total
            k = 3 if e.optimizer == "adam" else 1
            self._snap = dict(stream=torch.cuda.Stream(dev), dev=torch.empty((k, n), dtype=torch.float32, device=dev),
                              host=torch.empty((k, n), dtype=torch.float32, pin_memory=True),
                              st_dev=torch.empty_like(e.state),
                              st_host=torch.empty(e.state.shape, dtype=e.state.dtype, pin_memory=True),
                              copied=torch.cuda.Event(), landed=torch.cuda.Event())
        sn = self._snap
        cur = torch.cuda.current_stream(dev)
        sn["stream"].wait_stream(cur)
        with torch.cuda.stream(sn["stream"]):
            sn["dev"][0].copy_(e.params, non_blocking=True)
            if adam:
                sn["dev"][1].copy_(e.adam_m, non_blocking=True)
                sn["dev"][2].copy_(e.adam_v, non_blocking=True)
            sn["st_dev"].copy_(e.state, non_blocking=True)
            sn["copied"].record()
            sn["host"].copy_(sn["dev"], non_blocking=True)
            sn["st_host"].copy_(sn["st_dev"], non_blocking=True)
            sn["landed"].record()
        cur.wait_event(sn["copied"])          # the next step may overwrite the arenas once the D2D copy is done
        lay = self.layout

        def finish():
            sn["landed"].synchronize()        # on the writer thread
            host, st = sn["host"], sn["st_host"]
            params = {nm: lay.view(host[0], nm).numpy().copy() for nm in lay.names()}
            step = int(st[e.STATE_STEP])
            opt = None
            if adam:
                opt = {"m": {nm: lay.view(host[1], nm).numpy().copy() for nm in lay.names()},
                       "v": {nm: lay.view(host[2], nm).numpy().copy() for nm in lay.names()},
                       "t": int(st[e.STATE_ADAM_T])}
            return params, step, opt
        return finish

    def set_learning_rate(self, lr: float): self.engine.set_learning_rate(float(lr))
    def get_global_step(self): return self.engine.global_step
    def set_global_step(self, v):
        self.engine.set_global_step(v)
        self._host_step = int(v)

    def adam_arenas(self):
        e = self.engine
        return (e.adam_m, e.adam_v) if e.optimizer == "adam" else None

    def get_adam_t(self): return int(self.engine.state[self.engine.STATE_ADAM_T].item())

    def set_adam_t(self, t):
        e = self.engine
        e.state[e.STATE_ADAM_T] = int(t)
        e.state[e.STATE_ADAM_T2] = int(t)

    def export_optimizer_state(self):
        e = self.engine
        if e.optimizer != "adam":
            return None
        m, v = e.adam_m.detach().cpu(), e.adam_v.detach().cpu()
        return {"m": {n: self.layout.view(m, n).numpy().copy() for n in self.layout.names()},
                "v": {n: self.layout.view(v, n).numpy().copy() for n in self.layout.names()},
                "t": int(e.state[e.STATE_ADAM_T].item())}

    def load_optimizer_state(self, state):
        e = self.engine
        if e.optimizer != "adam" or not state:
            return
        for n in self.layout.names():
            self.layout.view(e.adam_m, n).copy_(torch.as_tensor(state["m"][n]).to(e.device))
            self.layout.view(e.adam_v, n).copy_(torch.as_tensor(state["v"][n]).to(e.device))
        e.state[e.STATE_ADAM_T] = int(state.get("t", 0))
        e.state[e.STATE_ADAM_T2] = int(state.get("t", 0))


def make_backend(kind: str, spec: ModelSpec, batch: int, data: Datasets, **kw) -> Backend:
    if kind == "auto":
        kind = "hip" if torch.cuda.is_available() else "cpu"
    if kind == "hip":
        return HipBackend(spec, batch, data, **kw)
    if kind == "cpu":
        kw.pop("device", None)
        kw.pop("use_graph", None)
        kw.pop("dtype", None)      # the CPU oracle is fp32
        return CpuBackend(spec, batch, data, **kw)
    raise ValueError(f"unknown backend {kind!r}")
