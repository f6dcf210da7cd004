"""HipEngine: the MI355X training/eval engine for the deep-MNIST CNN.

This is the GPU replacement for everything ``sess.run([optimizer, global_step],
feed_dict)`` does in the reference (reference .idea/MNISTDist.py:178-188): batch
selection, forward, loss/accuracy, backward, parameter update and the
global_step increment all run as ~15 hand-written gfx950 kernels over
pre-allocated arenas, with no host input pipeline, no autograd and no
ATen/MIOpen/hipBLASLt call on the hot path.  The whole step can be captured in a
HIP graph and replayed (hyper-parameters, the sample cursor, the dropout offset
and global_step all live in device memory, so replays are not frozen).

Memory plan (all resident, sized once for ``batch``):
  params / grads      flat fp32 arenas (ArenaLayout order = backward order)
  shadow              bf16 copy of the arena (+ K-contiguous transposes of wd1/wc2)
  activations         bf16 NHWC, pooled only (pre-pool tensors never exist)
  split-K slabs       deterministic partial sums for the wgrad reductions
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from .models.spec import ArenaLayout, ModelSpec
from .ops import native
from .ops.native import check, ptr


def _half_bits(n: int) -> int:
    hb = 1
    while (1 << (2 * hb)) < n:
        hb += 1
    return hb


@dataclass
class DeviceDataset:
    """An HBM-resident dataset: bf16 images [N,784] in [0,1] and int32 labels [N]."""

    images: torch.Tensor
    labels: torch.Tensor

    @property
    def n(self) -> int:
        return int(self.images.shape[0])

    @staticmethod
    def synthetic(n: int, seed: int = 1234, device="cuda") -> "DeviceDataset":
        """K12: generate MNIST-shaped synthetic data directly in HBM."""
        lib = native.load()
        images = torch.empty((n, 784), dtype=torch.bfloat16, device=device)
        labels = torch.empty((n,), dtype=torch.int32, device=device)
        s = torch.cuda.current_stream().cuda_stream
        check(lib.dtf_gen_synthetic(ptr(images), ptr(labels), n, seed & 0xFFFFFFFF, s), "gen_synthetic")
        return DeviceDataset(images, labels)

    @staticmethod
    def from_uint8(images_u8: torch.Tensor, labels: torch.Tensor, device="cuda") -> "DeviceDataset":
        """Upload raw idx pixels once and convert to bf16 [0,1] on device (TF: float32(x)/255)."""
        lib = native.load()
        src = images_u8.reshape(images_u8.shape[0], -1).contiguous().to(device)
        dst = torch.empty(src.shape, dtype=torch.bfloat16, device=device)
        s = torch.cuda.current_stream().cuda_stream
        check(lib.dtf_u8_to_bf16_unit(ptr(src), ptr(dst), src.numel(), s), "u8_to_bf16")
        return DeviceDataset(dst, labels.to(device=device, dtype=torch.int32).contiguous())

    @staticmethod
    def from_float(images: torch.Tensor, labels: torch.Tensor, device="cuda") -> "DeviceDataset":
        if labels.dim() == 2:  # one-hot, as the reference feeds them
            labels = labels.argmax(-1)
        return DeviceDataset(images.reshape(images.shape[0], -1).to(device=device, dtype=torch.bfloat16).contiguous(),
                             labels.to(device=device, dtype=torch.int32).contiguous())


class HipEngine:
    STATE_CURSOR, STATE_STEP, STATE_DROPOUT, STATE_ADAM_T = 0, 1, 2, 3

    def __init__(self, spec: ModelSpec, batch: int, dataset: Optional[DeviceDataset] = None,
                 optimizer: str = "sgd", learning_rate: float = 1e-3, keep_prob: float = 1.0,
                 seed: int = 0, device="cuda", grad_scale: float = 1.0, shuffle: bool = True,
                 adam_betas=(0.9, 0.999), adam_eps: float = 1e-8, step_inc: int = 1, multi_stream: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError("HipEngine needs a GPU (use backend='cpu' / models.oracle on CPU)")
        self.lib = native.load()
        self.spec, self.B = spec, int(batch)
        self.layout = ArenaLayout(spec)
        self.device = torch.device(device)
        self.optimizer = optimizer
        self.keep_prob = float(keep_prob)
        self.seed = int(seed)
        self.shuffle = bool(shuffle)
        self.step_inc = int(step_inc)
        self.dataset = dataset
        dev, B, H, C1, C2, FLAT = self.device, self.B, spec.hidden, spec.c1, spec.c2, spec.flat
        n = self.layout.total

        f32 = dict(dtype=torch.float32, device=dev)
        bf = dict(dtype=torch.bfloat16, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        self.params = torch.zeros(n, **f32)
        self.grads = torch.zeros(n, **f32)
        self.shadow = torch.zeros(n, **bf)
        self.wd1T = torch.zeros((H, FLAT), **bf)
        self.wc2T = torch.zeros((25, C2, C1), **bf)
        if optimizer == "adam":
            self.adam_m = torch.zeros(n, **f32)
            self.adam_v = torch.zeros(n, **f32)
        elif optimizer != "sgd":
            raise ValueError(f"unknown optimizer {optimizer!r}")
        self.state = torch.zeros(8, dtype=torch.int64, device=dev)
        self.hyper = torch.tensor([learning_rate, adam_betas[0], adam_betas[1], adam_eps, grad_scale, 0.0, 0.0, 0.0],
                                  **f32)

        self.idx = torch.zeros(B, dtype=torch.int32, device=dev)
        self.p1 = torch.empty((B, 14, 14, C1), **bf)
        self.arg1 = torch.empty((B, 14, 14, C1), **u8)
        self.p2 = torch.empty((B, 7, 7, C2), **bf)
        self.arg2 = torch.empty((B, 7, 7, C2), **u8)
        self.h1 = torch.empty((B, H), **bf)
        self.logits = torch.empty((B, spec.classes), **f32)
        self.loss_rows = torch.empty(B, **f32)
        self.correct_rows = torch.empty(B, **f32)
        self.dlogits = torch.empty((B, spec.classes), **f32)
        self.da = torch.empty((B, H), **bf)
        self.dflat = torch.empty((B, FLAT), **bf)
        self.dp1 = torch.empty((B, 14, 14, C1), **bf)

        # split-K geometry
        self.head_ks = max(1, min(16, B // 16))
        self.head_stride = H * spec.classes + 12 + H
        self.head_part = torch.empty((self.head_ks, self.head_stride), **f32)
        self.c2_g = max(1, min((B + 1) // 2, 102))
        self.c2_stride = 25 * C1 * C2 + C2
        self.c2_part = torch.empty((self.c2_g, self.c2_stride), **f32)
        self.c1_g = max(1, min(B, 512))
        self.c1_stride = 26 * C1
        self.c1_part = torch.empty((self.c1_g, self.c1_stride), **f32)

        lay = self.layout
        self._head_seg = lay.segment("out_w", "bd1")
        assert self._head_seg[1] - self._head_seg[0] == self.head_stride
        self._c2_seg = (lay.offsets["wc2"], lay.offsets["bc2"] + C2)
        assert self._c2_seg[1] - self._c2_seg[0] == self.c2_stride
        self._c1_seg = (lay.offsets["wc1"], lay.offsets["bc1"] + C1)
        assert self._c1_seg[1] - self._c1_seg[0] == self.c1_stride

        self._graph = None
        self._graph_stream = None
        # fork/join side streams for the independent branches of backward
        self.multi_stream = bool(multi_stream)
        self._side1 = torch.cuda.Stream(dev)
        self._side2 = torch.cuda.Stream(dev)
        self._ev_head, self._ev_fork = torch.cuda.Event(), torch.cuda.Event()
        self._ev_join1, self._ev_join2 = torch.cuda.Event(), torch.cuda.Event()

    # ------------------------------------------------------------------ utils
    def _s(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def pview(self, name: str, arena: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.layout.view(self.params if arena is None else arena, name)

    def _pptr(self, arena: torch.Tensor, name: str) -> int:
        return arena.data_ptr() + self.layout.offsets[name] * arena.element_size()

    def set_learning_rate(self, lr: float):
        self.hyper[0] = lr

    def set_grad_scale(self, s: float):
        self.hyper[4] = s

    @property
    def global_step(self) -> int:
        return int(self.state[self.STATE_STEP].item())

    def set_global_step(self, v: int):
        self.state[self.STATE_STEP] = int(v)

    # ------------------------------------------------------------- parameters
    def init_params(self, seed: Optional[int] = None):
        """K10: weights ~ truncated_normal(stddev 0.1), biases = 0.1 (reference M:42-49)."""
        seed = self.seed if seed is None else seed
        s = self._s()
        self.params.zero_()
        for k, name in enumerate(("wc1", "wc2", "wd1", "out_w")):
            check(self.lib.dtf_init_truncated_normal(self._pptr(self.params, name), self.layout.sizes[name], 0.1,
                                                     seed, k, s), "init_truncated_normal")
        for name in ("bc1", "bc2", "bd1", "out_b"):
            check(self.lib.dtf_fill_f32(self._pptr(self.params, name), self.layout.sizes[name], 0.1, s), "fill")
        self.refresh_shadows()

    def load_params(self, tensors: Dict[str, torch.Tensor]):
        for name in self.layout.names():
            self.pview(name).copy_(tensors[name].to(self.device, torch.float32).reshape(self.layout.shape(name)))
        self.refresh_shadows()

    def export_params(self) -> Dict[str, torch.Tensor]:
        return {name: self.pview(name).detach().cpu().clone() for name in self.layout.names()}

    def export_grads(self) -> Dict[str, torch.Tensor]:
        return {name: self.layout.view(self.grads, name).detach().cpu().clone() for name in self.layout.names()}

    def refresh_shadows(self, cast: bool = True):
        """bf16 compute copies: the arena itself (+ K-contiguous transposes for the NT kernels)."""
        s = self._s()
        sp = self.spec
        if cast:
            check(self.lib.dtf_cast_bf16(ptr(self.params), ptr(self.shadow), self.layout.total, s), "cast_bf16")
        check(self.lib.dtf_transpose_bf16(self._pptr(self.shadow, "wd1"), ptr(self.wd1T), 1, sp.flat, sp.hidden, s),
              "transpose wd1")
        check(self.lib.dtf_transpose_bf16(self._pptr(self.shadow, "wc2"), ptr(self.wc2T), 25, sp.c1, sp.c2, s),
              "transpose wc2")

    # ----------------------------------------------------------------- phases
    def launch_sample(self):
        ds = self.dataset
        check(self.lib.dtf_sample_indices(ptr(self.state), ptr(self.idx), self.B, ds.n, _half_bits(ds.n),
                                          self.seed & 0xFFFFFFFF, 1 if self.shuffle else 0, self._s()), "sample")

    def launch_forward(self, images: torch.Tensor, labels: torch.Tensor, idx: Optional[torch.Tensor],
                       backward: bool, n: Optional[int] = None, train_mode: bool = True):
        """conv1 -> conv2 -> fc1 -> head (loss/acc [+ dlogits, da])."""
        lib, sp, s = self.lib, self.spec, self._s()
        B = self.B if n is None else n
        keep = self.keep_prob if train_mode else 1.0
        check(lib.dtf_conv1_fwd(ptr(images), ptr(idx), self._pptr(self.params, "wc1"), self._pptr(self.params, "bc1"),
                                ptr(self.p1), ptr(self.arg1), B, sp.c1, 0, s), "conv1_fwd")
        check(lib.dtf_conv2_fwd(ptr(self.p1), ptr(self.wc2T), self._pptr(self.params, "bc2"), ptr(self.p2),
                                ptr(self.arg2), B, sp.c1, sp.c2, s), "conv2_fwd")
        check(lib.dtf_gemm_nt(ptr(self.p2), ptr(self.wd1T), ptr(self.h1), self._pptr(self.params, "bd1"), B,
                              sp.hidden, sp.flat, sp.flat, sp.flat, sp.hidden, 1, keep, self.seed, ptr(self.state),
                              0, s), "fc1_fwd")
        check(lib.dtf_head_fwd_bwd(ptr(self.h1), self._pptr(self.params, "out_w"), self._pptr(self.params, "out_b"),
                                   ptr(labels), ptr(idx), ptr(self.logits), ptr(self.loss_rows),
                                   ptr(self.correct_rows), ptr(self.dlogits), ptr(self.da), B, sp.hidden,
                                   1.0 / B, 1.0 / keep, 1 if backward else 0, s), "head")

    def _reduce(self, segs):
        k = len(segs)
        parts = (ctypes.c_void_p * k)(*[p.data_ptr() for p, _, _, _ in segs])
        outs = (ctypes.c_void_p * k)(*[self.grads.data_ptr() + 4 * lo for _, _, (lo, hi), _ in segs])
        G = (ctypes.c_int * k)(*[g for _, g, _, _ in segs])
        n = (ctypes.c_int * k)(*[hi - lo for _, _, (lo, hi), _ in segs])
        st = (ctypes.c_long * k)(*[stride for _, _, _, stride in segs])
        check(self.lib.dtf_reduce_segments(k, parts, outs, G, n, st, self._s()), "reduce_segments")

    def launch_backward_dense(self):
        """head wgrad + fc1 wgrad: after this the first (head|wd1) part of the grad arena is final."""
        lib, sp, s, B = self.lib, self.spec, self._s(), self.B
        check(lib.dtf_head_wgrad(ptr(self.h1), ptr(self.dlogits), ptr(self.da), ptr(self.head_part), B, sp.hidden,
                                 self.head_ks, s), "head_wgrad")
        self._reduce([(self.head_part, self.head_ks, self._head_seg, self.head_stride)])
        check(lib.dtf_gemm_tn(ptr(self.p2), ptr(self.da), self._pptr(self.grads, "wd1"), sp.flat, sp.hidden, B,
                              sp.flat, sp.hidden, sp.hidden, 0, s), "fc1_wgrad")

    def launch_backward_conv(self, images: torch.Tensor, idx: Optional[torch.Tensor]):
        """fc1 dgrad -> {conv2 dgrad -> conv1 wgrad} || conv2 wgrad -> slab reduce.

        conv2's two gradient kernels only share inputs, so the filter gradient runs on a side
        stream next to the data gradient (a fork/join that HIP-graph capture preserves): both are
        latency-bound at this size and co-scheduling them fills the machine."""
        lib, sp, s, B = self.lib, self.spec, self._s(), self.B
        cur = torch.cuda.current_stream(self.device)
        check(lib.dtf_gemm_nt(ptr(self.da), self._pptr(self.shadow, "wd1"), ptr(self.dflat), None, B, sp.flat,
                              sp.hidden, sp.hidden, sp.hidden, sp.flat, 0, 1.0, 0, None, 0, s), "fc1_dgrad")
        if self.multi_stream:
            self._ev_fork.record(cur)
            self._side2.wait_event(self._ev_fork)
            s2 = self._side2.cuda_stream
        else:
            s2 = s
        check(lib.dtf_conv2_wgrad(ptr(self.p1), ptr(self.dflat), ptr(self.arg2), ptr(self.p2), ptr(self.c2_part), B,
                                  sp.c1, sp.c2, self.c2_g, s2), "conv2_wgrad")
        if self.multi_stream:
            self._ev_join2.record(self._side2)
        check(lib.dtf_conv2_dgrad(ptr(self.dflat), ptr(self.arg2), ptr(self.p2), self._pptr(self.shadow, "wc2"),
                                  ptr(self.dp1), B, sp.c1, sp.c2, s), "conv2_dgrad")
        check(lib.dtf_conv1_wgrad(ptr(images), ptr(idx), ptr(self.dp1), ptr(self.arg1), ptr(self.p1),
                                  ptr(self.c1_part), B, sp.c1, self.c1_g, s), "conv1_wgrad")
        if self.multi_stream:
            cur.wait_event(self._ev_join2)
        self._reduce([(self.c2_part, self.c2_g, self._c2_seg, self.c2_stride),
                      (self.c1_part, self.c1_g, self._c1_seg, self.c1_stride)])

    def launch_backward(self, images: torch.Tensor, idx: Optional[torch.Tensor]):
        """Whole backward with the dense-layer weight gradients (head + fc1 wgrad) on a side stream
        concurrent with the conv chain.  Returns after joining: the grad arena is final."""
        if not self.multi_stream:
            self.launch_backward_dense()
            self.launch_backward_conv(images, idx)
            return
        cur = torch.cuda.current_stream(self.device)
        self._ev_head.record(cur)
        self._side1.wait_event(self._ev_head)
        with torch.cuda.stream(self._side1):
            self.launch_backward_dense()
            self._ev_join1.record(self._side1)
        self.launch_backward_conv(images, idx)
        cur.wait_event(self._ev_join1)

    def launch_update(self, advance_cursor: bool = True):
        lib, s = self.lib, self._s()
        n = self.layout.total
        cinc = self.B if advance_cursor else 0
        if self.optimizer == "sgd":
            check(lib.dtf_sgd_update(ptr(self.params), ptr(self.grads), ptr(self.shadow), ptr(self.hyper), n,
                                     ptr(self.state), cinc, self.step_inc, s), "sgd_update")
        else:
            check(lib.dtf_adam_update(ptr(self.params), ptr(self.grads), ptr(self.adam_m), ptr(self.adam_v),
                                      ptr(self.shadow), ptr(self.hyper), n, ptr(self.state), cinc, self.step_inc, s),
                  "adam_update")
        self.refresh_shadows(cast=False)

    # ------------------------------------------------------------------ steps
    def launch_train_step(self):
        """One full optimisation step on the resident dataset (all launches, no host sync)."""
        ds = self.dataset
        self.launch_sample()
        self.launch_forward(ds.images, ds.labels, self.idx, backward=True)
        self.launch_backward(ds.images, self.idx)
        self.launch_update()

    def train_step_on(self, images: torch.Tensor, labels: torch.Tensor):
        """One step on an explicit batch (bf16 [B,784], int32 [B]) -- the feed_dict path."""
        assert images.shape[0] == self.B
        self.launch_forward(images, labels, None, backward=True)
        self.launch_backward(images, None)
        self.launch_update(advance_cursor=False)

    def capture(self, warmup: int = 1):
        """Capture launch_train_step into a HIP graph (hipGraph via torch.cuda.CUDAGraph).

        The warm-up launches (lazy kernel attribute set-up must not happen under capture) run real
        steps, so parameters / optimizer moments / step state are snapshotted and restored:
        capturing is invisible to training."""
        keep = [self.params, self.state, self.shadow, self.wd1T, self.wc2T]
        if self.optimizer == "adam":
            keep += [self.adam_m, self.adam_v]
        snap = [t.clone() for t in keep]
        st = torch.cuda.Stream(self.device)
        st.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(st):
            for _ in range(max(1, warmup)):
                self.launch_train_step()
        torch.cuda.current_stream(self.device).wait_stream(st)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            self.launch_train_step()
        for t, s0 in zip(keep, snap):
            t.copy_(s0)
        torch.cuda.synchronize(self.device)
        self._graph, self._graph_stream = g, st
        return g

    def train_step(self):
        if self._graph is not None:
            self._graph.replay()
        else:
            self.launch_train_step()

    def last_metrics(self):
        """(mean loss, accuracy) of the most recent forward; forces a device sync."""
        return float(self.loss_rows.mean().item()), float(self.correct_rows.mean().item())

    @torch.no_grad()
    def evaluate(self, images: torch.Tensor, labels: torch.Tensor):
        """Mean loss / accuracy over an arbitrary number of rows (chunks of the engine batch)."""
        n = images.shape[0]
        tot_loss, tot_ok = 0.0, 0.0
        for lo in range(0, n, self.B):
            hi = min(n, lo + self.B)
            self.launch_forward(images[lo:hi], labels[lo:hi], None, backward=False, n=hi - lo, train_mode=False)
            tot_loss += float(self.loss_rows[:hi - lo].sum().item())
            tot_ok += float(self.correct_rows[:hi - lo].sum().item())
        return tot_loss / n, tot_ok / n

    @torch.no_grad()
    def predict_logits(self, images: torch.Tensor) -> torch.Tensor:
        n = images.shape[0]
        assert n <= self.B
        dummy = torch.zeros(n, dtype=torch.int32, device=self.device)
        self.launch_forward(images, dummy, None, backward=False, n=n, train_mode=False)
        return self.logits[:n].clone()
