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
  shadow              bf16 copy of the arena (+ a K-contiguous transpose of the 100 KB conv2 filter)
  activations         bf16 NHWC, pooled only (pre-pool tensors never exist)
  split-K slabs       deterministic partial sums for the wgrad reductions
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from .models.spec import ArenaLayout, ModelSpec
from .ops import native
from .ops.native import check, ptr


@dataclass(frozen=True)
class Tuning:
    """Every launch-geometry override the engine takes from the environment, read in ONE place (A/B switches for the
    tools under tools/; unset = the measured defaults).  ``env_overrides()`` lists every ``DTF_*`` variable that is
    set -- these and the ones the kernels read themselves (DTF_GEMM_DMA, DTF_IMGS1_MAX_B, DTF_F32_MSPLIT, ...) -- and
    bench.py writes that list into its result line, so a number taken under a non-default configuration says so."""

    dgrad_splits: Optional[int] = None    # DTF_DGRAD_SPLITS  fp32 fc1 dgrad: K ranges
    head_ks: Optional[int] = None         # DTF_HEAD_KS       output-layer wgrad: batch splits
    c2w_nt: Optional[int] = None          # DTF_C2W_NT        bf16 conv2 wgrad: co tiles per wave (2 | 4)
    c2_g: Optional[int] = None            # DTF_C2_G          conv2 wgrad: image slabs
    fused_c1w: bool = True                # DTF_FUSED_C1W     conv2 dgrad + conv1 wgrad as one kernel

    @staticmethod
    def from_env(env=None) -> "Tuning":
        env = os.environ if env is None else env

        def num(name):
            v = env.get(name)
            return int(v) if v else None

        nt = env.get("DTF_C2W_NT")
        return Tuning(dgrad_splits=num("DTF_DGRAD_SPLITS"), head_ks=num("DTF_HEAD_KS"),
                      c2w_nt=int(nt) if nt in ("2", "4") else None, c2_g=num("DTF_C2_G"),
                      fused_c1w=env.get("DTF_FUSED_C1W", "1") == "1")

    @staticmethod
    def env_overrides(env=None) -> Dict[str, str]:
        env = os.environ if env is None else env
        return {k: env[k] for k in sorted(env) if k.startswith("DTF_")}


def _half_bits(n: int) -> int:
    hb = 1
    while (1 << (2 * hb)) < n:
        hb += 1
    return hb


@dataclass
class DeviceDataset:
    """An HBM-resident dataset: images [N,784] in [0,1] and int32 labels [N].  Images are bf16 for the
    mixed-precision engine and fp32 for the fp32 engine (the reference's float32 placeholders, M:113)."""

    images: torch.Tensor
    labels: torch.Tensor

    @property
    def n(self) -> int:
        return int(self.images.shape[0])

    @property
    def dtype(self) -> str:
        return "fp32" if self.images.dtype == torch.float32 else "bf16"

    @staticmethod
    def synthetic(n: int, seed: int = 1234, device="cuda", dtype: str = "bf16") -> "DeviceDataset":
        """K12: generate MNIST-shaped synthetic data directly in HBM.  The generator's pixel values are exactly
        representable in bf16, so the fp32 dataset holds the same numbers."""
        lib = native.load()
        images = torch.empty((n, 784), dtype=torch.bfloat16, device=device)
        labels = torch.empty((n,), dtype=torch.int32, device=device)
        s = torch.cuda.current_stream().cuda_stream
        check(lib.dtf_gen_synthetic(ptr(images), ptr(labels), n, seed & 0xFFFFFFFF, s), "gen_synthetic")
        if dtype == "fp32":
            images = images.float()     # one-time set-up, not a step op
        return DeviceDataset(images, labels)

    @staticmethod
    def from_uint8(images_u8: torch.Tensor, labels: torch.Tensor, device="cuda", dtype: str = "bf16") -> "DeviceDataset":
        """Upload raw idx pixels once and convert to [0,1] on device (TF: float32(x)/255)."""
        lib = native.load()
        src = images_u8.reshape(images_u8.shape[0], -1).contiguous().to(device)
        s = torch.cuda.current_stream().cuda_stream
        if dtype == "fp32":
            dst = torch.empty(src.shape, dtype=torch.float32, device=device)
            check(lib.dtf_u8_to_f32_unit(ptr(src), ptr(dst), src.numel(), s), "u8_to_f32")
        else:
            dst = torch.empty(src.shape, dtype=torch.bfloat16, device=device)
            check(lib.dtf_u8_to_bf16_unit(ptr(src), ptr(dst), src.numel(), s), "u8_to_bf16")
        return DeviceDataset(dst, labels.to(device=device, dtype=torch.int32).contiguous())

    @staticmethod
    def from_float(images: torch.Tensor, labels: torch.Tensor, device="cuda", dtype: str = "bf16") -> "DeviceDataset":
        if labels.dim() == 2:  # one-hot, as the reference feeds them
            labels = labels.argmax(-1)
        tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
        return DeviceDataset(images.reshape(images.shape[0], -1).to(device=device, dtype=tdt).contiguous(),
                             labels.to(device=device, dtype=torch.int32).contiguous())


class HipEngine:
    STATE_CURSOR, STATE_STEP, STATE_DROPOUT, STATE_ADAM_T, STATE_ADAM_T2 = 0, 1, 2, 3, 4

    def __init__(self, spec: ModelSpec, batch: int, dataset: Optional[DeviceDataset] = None,
                 optimizer: str = "sgd", learning_rate: float = 1e-3, keep_prob: float = 1.0,
                 seed: int = 0, device="cuda", grad_scale: float = 1.0, shuffle: bool = True,
                 adam_betas=(0.9, 0.999), adam_eps: float = 1e-8, step_inc: int = 1, multi_stream: Optional[bool] = None,
                 fuse_dense_update: bool = False, dtype: str = "bf16", tuning: Optional[Tuning] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("HipEngine needs a GPU (use backend='cpu' / models.oracle on CPU)")
        self.lib = native.load()
        self.tuning = tun = Tuning.from_env() if tuning is None else tuning
        if dtype not in ("bf16", "fp32"):
            raise ValueError(f"dtype must be 'bf16' or 'fp32', got {dtype!r}")
        # "bf16": bf16 MFMA compute copies, fp32 accumulation and fp32 master weights.
        # "fp32": the reference's precision (M:113-115 float32 everywhere): fp32 activations, weights read directly,
        #         exact-fp32 MFMA kernels (csrc/kernels/*_f32.hip); no bf16 copy of anything exists.
        self.dtype = dtype
        if dataset is not None and dataset.dtype != dtype:
            raise ValueError(f"dataset holds {dataset.dtype} images but the engine computes in {dtype}")
        self.spec, self.B = spec, int(batch)
        self.layout = ArenaLayout(spec)
        self.device = torch.device(device)
        self.optimizer = optimizer
        self.keep_prob = float(keep_prob)
        self.seed = int(seed)
        self.shuffle = bool(shuffle)
        self.step_inc = int(step_inc)
        self.dataset = dataset
        # fork/join side streams in backward (conv2 wgrad || dgrad chain || fc1 wgrad).  Measured A/B on the
        # 8..9-kernel step: -5 % at batch 1024 (fork/join costs more than the overlap returns), +3..4 % from 2048 up.
        # (the fp32 kernels fill the chip on their own: single stream)
        self.multi_stream = (batch > 1024 and dtype == "bf16") if multi_stream is None else bool(multi_stream)
        self.fuse_dense_update = bool(fuse_dense_update)
        dev, B, H, C1, C2, FLAT = self.device, self.B, spec.hidden, spec.c1, spec.c2, spec.flat
        n = self.layout.total

        f32 = dict(dtype=torch.float32, device=dev)
        bf = dict(dtype=torch.bfloat16, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        act = f32 if dtype == "fp32" else bf               # activation storage type
        self.params = torch.zeros(n, **f32)
        self.grads = torch.zeros(n, **f32)
        self.shadow = torch.zeros(n if dtype == "bf16" else 4, **bf)   # fp32 engine: no compute copies
        # [fwd pack | dgrad pack] of wc2 in MFMA fragment order (csrc/kernels/conv2.hip / conv2_f32.hip)
        self.wc2T = torch.zeros((2, 25 * C1 * C2), **act)
        self.out_wT = torch.zeros((10, H), **f32)          # output weight transposed, for the head kernel
        if optimizer == "adam":
            self.adam_m = torch.zeros(n, **f32)
            self.adam_v = torch.zeros(n, **f32)
        elif optimizer != "sgd":
            raise ValueError(f"unknown optimizer {optimizer!r}")
        self.state = torch.zeros(8, dtype=torch.int64, device=dev)
        # [lr, beta1, beta2, eps, gradient scale, -, -, - | 4 control flags of this rank: stop request, gather request, -, -]
        # The control flags are published into the control block of the gradient arena by the step's slab reduction
        # and summed over the ranks by the tail collective (csrc/runtime/engine.hip backward_conv_impl).
        self.hyper = torch.tensor([learning_rate, adam_betas[0], adam_betas[1], adam_eps, grad_scale, 0.0, 0.0, 0.0,
                                   0.0, 0.0, 0.0, 0.0], **f32)
        self.ctrl_src = self.hyper[8:12]

        self.idx = torch.zeros(B, dtype=torch.int32, device=dev)
        self.p1 = torch.empty((B, 14, 14, C1), **act)
        self.arg1 = torch.empty((B, 14, 14, C1), **u8)
        self.p2 = torch.empty((B, 7, 7, C2), **act)
        self.arg2 = torch.empty((B, 7, 7, C2), **u8)
        self.h1 = torch.empty((B, H), **act)
        tiles = ((B + 127) // 128) * ((H + 127) // 128)
        # split-K of the fc1 forward GEMM (csrc/runtime/engine.hip auto_splits: ~256 workgroups; the fp32 kernels go to
        # 32 splits, which is what fills the chip at the reference's batch 100)
        self.fc1_splits = max(1, min(32 if dtype == "fp32" else (16 if tiles <= 8 else 8), -(-256 // tiles)))
        self.fc1_part = torch.empty((self.fc1_splits, B, H), **f32)
        # fp32 fc1 dgrad at a small batch: ceil(B / 128) * ceil(FLAT / 128) output tiles do not fill 256 CUs (batch
        # 100: 25) -> split-K partial products, summed by the slab-reduction kernel.  (The bf16 dgrad uses 64-row
        # tiles and H / 32 = 32 K steps; it is not K-bound the same way.)
        d_tiles = ((B + 127) // 128) * ((FLAT + 127) // 128)
        self.dgrad_splits = max(1, min(8, H // 128, 256 // d_tiles)) if dtype == "fp32" else 1
        if tun.dgrad_splits:                                    # tuning override
            self.dgrad_splits = max(1, min(tun.dgrad_splits, H // 32)) if dtype == "fp32" else 1
        self.dgrad_part = torch.empty((self.dgrad_splits, B, FLAT) if self.dgrad_splits > 1 else (4,), **f32)
        self.logits = torch.empty((B, spec.classes), **f32)
        self.loss_rows = torch.empty(B, **f32)
        self.correct_rows = torch.empty(B, **f32)
        self.dlogits = torch.empty((B, spec.classes), **f32)
        self._metrics_acc = torch.zeros(2, dtype=torch.float64, device=dev)          # K9 accumulator [sum loss, #correct]
        self._metrics_host = torch.zeros(2, dtype=torch.float64).pin_memory()
        self.da = torch.empty((B, H), **act)
        self.dflat = torch.empty((B, FLAT), **act)
        self.dp1 = torch.empty((B, 14, 14, C1), **act)

        # split-K geometry
        # output-layer wgrad: 256-column workgroups x head_ks batch splits (slabs summed by the fused tail)
        self.head_ks = max(1, min(128, max(B // 64, min(8, B // 8))))   # 64 rows per split (measured: 8..16 splits best at 1024)
        if tun.head_ks:                                         # tuning override
            self.head_ks = max(1, min(tun.head_ks, B))
        self.head_stride = H * spec.classes + 12 + H
        self.head_part = torch.empty((self.head_ks, self.head_stride), **f32)
        # conv2 wgrad split-K over images: G slabs x (C1/16 * C2/32) output tiles workgroups.  64 slabs fill the
        # chip for the 32/64 model (4 tiles); the widened model has 64 tiles and 3.3 MB slabs -> 8 slabs
        # 0 = the library's default launch for 16 ci x 32 co tiles, the producer / consumer kernel (conv5x5_wgrad_pc),
        # at every batch size and for both models: step -0.6 % at 100 images, -3.8 % at 512, -7.2 % at 1024, -2.1 % at
        # 2048, -1.2 % at 4096, -1.8 % at 8192, -0.8 % at 16384 against the self-staging kernel (16 ci x 32 co tile up to
        # 4096 images, 16 ci x 64 co with twice the slabs above); the widened model 1.442 -> 1.375 ms at 1024 images
        # (64 tiles x 8 slabs against 32 tiles of 16 ci x 64 co x 8 slabs).  tools/scratch/ab_pc_batches.sh, ab_pc_big.sh,
        # ab_pc_wide.sh: same device, alternating.  DTF_C2W_NT=2|4 selects the self-staging kernel (tests, A/B).
        self.c2_nt = 0
        if tun.c2w_nt:                                          # A/B switch for tools
            self.c2_nt = tun.c2w_nt
        c2_tiles = (C1 // 16) * (C2 // (16 * (self.c2_nt or 2)))
        default_g = 8 if C1 >= 128 else (64 if c2_tiles >= 4 else 128)
        self.c2_g = max(1, min((B + 1) // 2, tun.c2_g or default_g))
        if dtype == "fp32":
            # conv5x5_wgrad_f32: one workgroup per (image range, 32 ci x 64 co tile), one workgroup per CU
            f32_tiles = (C1 // 32) * (C2 // 64)
            self.c2_g = max(1, min(B, tun.c2_g or max(1, 256 // f32_tiles)))
        self.c2_stride = 25 * C1 * C2 + C2
        self.c2_part = torch.empty((self.c2_g, self.c2_stride), **f32)
        # conv1 wgrad slabs: the fused conv2-dgrad -> conv1-wgrad kernel of the 32/64 model writes one per
        # workgroup (2 images each); the stand-alone kernel takes any grid
        # Measured (bench.py A/B): +2.7 % at batch 256, neutral at 1024, -10 % at 4096+ (lower occupancy of the
        # dgrad phase, one slab per two images) -> fused up to 1024 images per GPU.  Env var: A/B switch for tools.
        fused_c1w = (C1, C2) == (32, 64) and B <= 1024 and dtype == "bf16" and tun.fused_c1w
        # up to dtf_conv2_small_batch() images the fused kernels take ONE image per workgroup (twice the workgroups,
        # half the serial work of each: at batch 100 two images per workgroup are 50 workgroups on 256 CUs)
        self.c1w_imgs = 0 if not fused_c1w else (1 if 1 < B <= int(self.lib.dtf_conv2_small_batch()) else 2)
        self.c1_g = (B + self.c1w_imgs - 1) // self.c1w_imgs if fused_c1w else max(1, min(B, 512))
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
        self.dense_g16 = None      # see enable_dense_g16()
        self._build_desc()

    def _build_desc(self):
        """Fill the native executor's descriptor (csrc/runtime/engine.hip) once; every phase below is
        then a single foreign call that enqueues all of its kernels / stream forks from C++."""
        d = native.EngineDesc()
        sp, lay = self.spec, self.layout
        d.B, d.c1, d.c2, d.hidden, d.flat, d.classes = self.B, sp.c1, sp.c2, sp.hidden, sp.flat, sp.classes
        d.head_ks, d.c2_g, d.c1_g = self.head_ks, self.c2_g, self.c1_g
        d.fc1_splits, d.dgrad_splits, d.c2_nt, d.c1w_imgs = self.fc1_splits, self.dgrad_splits, self.c2_nt, self.c1w_imgs
        d.optimizer = 0 if self.optimizer == "sgd" else 1
        d.shuffle, d.multi_stream = int(self.shuffle), int(self.multi_stream)
        # fc1 SGD update inside its wgrad GEMM (single-GPU train_step only; the wd1 gradient is then never
        # materialised in the gradient arena)
        d.fuse_dense_update = int(self.fuse_dense_update and self.optimizer == "sgd")
        d.sample_seed, d.seed, d.keep_prob = self.seed & 0xFFFFFFFF, self.seed & (2 ** 64 - 1), self.keep_prob
        d.dtype = 1 if self.dtype == "fp32" else 0
        d.n_total = lay.total
        for name in lay.names():
            setattr(d, "off_" + name, lay.offsets[name])
        d.head_stride, d.c2_stride, d.c1_stride = self.head_stride, self.c2_stride, self.c1_stride
        for name in ("params", "grads", "shadow", "fc1_part", "wc2T", "out_wT", "state", "hyper", "idx", "p1", "arg1", "p2", "arg2",
                     "h1", "logits", "loss_rows", "correct_rows", "dlogits", "da", "dflat", "dp1", "head_part",
                     "c2_part", "c1_part", "dgrad_part"):
            setattr(d, name, getattr(self, name).data_ptr())
        if self.optimizer == "adam":
            d.adam_m, d.adam_v = self.adam_m.data_ptr(), self.adam_v.data_ptr()
        self._desc = d
        self._set_dataset(self.dataset)
        check(self.lib.dtf_engine_init(ctypes.byref(d)), "engine_init")

    def _set_dataset(self, ds: Optional[DeviceDataset]):
        if ds is not None and ds.dtype != self.dtype:
            raise ValueError(f"dataset holds {ds.dtype} images but the engine computes in {self.dtype}")
        self.dataset = ds
        if ds is not None:
            self._desc.ds_images, self._desc.ds_labels = ds.images.data_ptr(), ds.labels.data_ptr()
            self._desc.ds_n, self._desc.ds_hb = ds.n, _half_bits(ds.n)

    def streams_in_use(self) -> int:
        """HIP streams created by the native executor (side streams of the backward fork/join, 0 or 2)."""
        return int(self.lib.dtf_engine_stream_count(ctypes.byref(self._desc)))

    def __del__(self):
        try:
            self.lib.dtf_engine_destroy(ctypes.byref(self._desc))
        except Exception:
            pass

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

    def export_dropout_mask(self, offset: Optional[int] = None, rows: Optional[int] = None) -> torch.Tensor:
        """Debug launch: the [rows, hidden] uint8 keep mask the fc1 epilogue applies at dropout offset ``offset``
        (default: the offset the NEXT step will use, i.e. the device counter now).  Read it BEFORE the step."""
        rows = self.B if rows is None else rows
        if offset is None:
            offset = int(self.state[self.STATE_DROPOUT].item())
        out = torch.empty((rows, self.spec.hidden), dtype=torch.uint8, device=self.device)
        check(self.lib.dtf_dropout_mask(self.seed & (2 ** 64 - 1), int(offset), float(self.keep_prob), ptr(out),
                                        out.numel(), self._s()), "dropout_mask")
        return out

    def refresh_shadows(self, cast: bool = True):
        """bf16 compute copies: the arena itself (+ K-contiguous transposes for the NT kernels)."""
        s = self._s()
        sp = self.spec
        if cast and self.dtype == "bf16":
            check(self.lib.dtf_cast_bf16(ptr(self.params), ptr(self.shadow), self.layout.total, s), "cast_bf16")
        self.refresh_transposes()

    # ----------------------------------------------------------------- phases
    # Each phase is ONE call into the native executor, which enqueues its kernels on the current
    # stream (and forks/joins its own side streams).  No host synchronisation anywhere.
    def _d(self):
        return ctypes.byref(self._desc)

    def enable_dense_g16(self) -> torch.Tensor:
        """16-bit dense gradient for the data-parallel step (parallel/sync_dp.py ``grad_comm="bf16"``): the fc1 weight
        gradient is written as bf16 into this buffer by its GEMM, travels in bf16, and the dense update reads it
        there.  Mixed-precision configuration only.  Returns the buffer (flat, wd1's element count)."""
        if self.dtype != "bf16":
            raise ValueError("the 16-bit gradient collective belongs to the bf16 configuration")
        if self.dense_g16 is None:
            lo, hi = self.layout.segment("wd1", "wd1")
            self.dense_g16 = torch.zeros(hi - lo, dtype=torch.bfloat16, device=self.device)
            self._desc.dense_g16 = self.dense_g16.data_ptr()
        return self.dense_g16

    def disable_dense_g16(self):
        self.dense_g16 = None
        self._desc.dense_g16 = None

    def launch_sample(self):
        check(self.lib.dtf_engine_sample(self._d(), self._s()), "sample")

    def launch_forward_conv(self, images: torch.Tensor, idx: Optional[torch.Tensor], n: Optional[int] = None):
        """conv1 -> conv2 (reads only the conv parameters)."""
        check(self.lib.dtf_engine_forward_conv(self._d(), ptr(images), ptr(idx), self.B if n is None else n,
                                               self._s()), "forward_conv")

    def launch_forward_dense(self, labels: torch.Tensor, idx: Optional[torch.Tensor], backward: bool,
                             n: Optional[int] = None, train_mode: bool = True):
        """fc1 -> head (loss/acc [+ dlogits, da]); reads the fc1 / head parameters."""
        check(self.lib.dtf_engine_forward_dense(self._d(), ptr(labels), ptr(idx), int(backward),
                                                self.B if n is None else n, int(train_mode), self._s()),
              "forward_dense")

    def launch_forward(self, images: torch.Tensor, labels: torch.Tensor, idx: Optional[torch.Tensor],
                       backward: bool, n: Optional[int] = None, train_mode: bool = True):
        """conv1 -> conv2 -> fc1 -> head (loss/acc [+ dlogits, da])."""
        self.launch_forward_conv(images, idx, n)
        self.launch_forward_dense(labels, idx, backward, n, train_mode)

    def launch_backward_dense(self):
        """head wgrad + fc1 wgrad: after this the first (head|wd1) part of the grad arena is final."""
        check(self.lib.dtf_engine_backward_dense(self._d(), self._s()), "backward_dense")

    def launch_backward_conv(self, images: torch.Tensor, idx: Optional[torch.Tensor], record_dgrad: bool = False):
        """fc1 dgrad -> {conv2 dgrad -> conv1 wgrad} || conv2 wgrad (side stream) -> slab reduce.
        ``record_dgrad`` records the native event ``ev_dgrad`` right after fc1 dgrad, the last reader
        of the bf16 wd1 copy (see wait_dgrad_event)."""
        check(self.lib.dtf_engine_backward_conv(self._d(), ptr(images), ptr(idx), int(record_dgrad), self._s()),
              "backward_conv")

    def wait_dgrad_event(self):
        """Make the current stream wait for the event recorded by launch_backward_conv(record_dgrad=True)."""
        check(self.lib.dtf_stream_wait_event(self._s(), self._desc.ev_dgrad), "wait ev_dgrad")

    def launch_backward(self, images: torch.Tensor, idx: Optional[torch.Tensor]):
        """Whole backward; with multi_stream the dense-layer weight gradients run on a side stream
        concurrent with the conv chain (fork/join, preserved by HIP-graph capture)."""
        check(self.lib.dtf_engine_backward(self._d(), ptr(images), ptr(idx), self._s()), "backward")

    def launch_update_range(self, lo: int, hi: int, advance: bool, advance_cursor: bool = True,
                            adam_slot: int = 3):
        """Optimizer over arena floats [lo, hi) (+ bf16 shadow).  ``advance`` bumps the sample cursor,
        global_step and dropout offset (exactly one range per step does that); every range keeps its
        own Adam step counter (state[adam_slot]) so ranges may be updated from different streams."""
        check(self.lib.dtf_engine_update_range(self._d(), lo, hi, int(advance), int(advance_cursor), self.step_inc,
                                               adam_slot, self._s()), "update_range")

    def refresh_transposes(self, wc2: bool = True):
        """K-contiguous copy of the conv2 filter for the forward kernel (wd1 is used in place)."""
        check(self.lib.dtf_engine_transposes(self._d(), int(wc2), self._s()), "transposes")

    def launch_update(self, advance_cursor: bool = True):
        self.launch_update_range(0, self.layout.total, True, advance_cursor)
        self.refresh_transposes()

    # ------------------------------------------------------------------ steps
    def launch_train_step(self):
        """One full optimisation step on the resident dataset: a single native call."""
        check(self.lib.dtf_engine_train_step(self._d(), self.step_inc, self._s()), "train_step")

    @property
    def act_dtype(self) -> torch.dtype:
        return torch.float32 if self.dtype == "fp32" else torch.bfloat16

    def train_step_on(self, images: torch.Tensor, labels: torch.Tensor):
        """One step on an explicit batch ([B,784] in the engine's dtype, int32 [B]) -- the feed_dict path."""
        assert images.shape[0] == self.B and images.dtype == self.act_dtype
        self.launch_forward(images, labels, None, backward=True)
        self.launch_backward(images, None)
        self.launch_update(advance_cursor=False)

    def capture(self, warmup: int = 1):
        """Capture launch_train_step into a HIP graph (hipGraph via torch.cuda.CUDAGraph).

        The warm-up launches (lazy kernel attribute set-up must not happen under capture) run real
        steps, so parameters / optimizer moments / step state are snapshotted and restored:
        capturing is invisible to training."""
        keep = [self.params, self.state, self.shadow, self.wc2T, self.out_wT]
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

    def _metrics(self, n: int, accumulate: bool):
        """K9: reduce loss_rows / correct_rows[0..n) into the device accumulator (one tiny launch, no ATen)."""
        check(self.lib.dtf_metrics_reduce(ptr(self.loss_rows), ptr(self.correct_rows), n, ptr(self._metrics_acc),
                                          int(accumulate), self._s()), "metrics_reduce")

    def _read_metrics(self):
        """One 16-byte device-to-host copy into pinned memory + one stream synchronisation."""
        self._metrics_host.copy_(self._metrics_acc, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return float(self._metrics_host[0]), float(self._metrics_host[1])

    def freeze_metrics(self):
        """Reduce the metrics of the most recent forward NOW (one tiny launch) and make last_metrics() report that
        value until the next freeze: a caller that has already enqueued the next step's forward (the asynchronous
        worker's look-ahead, parallel/async_push.py) still reports the batch its last completed step trained on."""
        self._metrics(self.B, accumulate=False)
        self._metrics_frozen = True

    def last_metrics(self):
        """(mean loss, accuracy) of the most recent forward; forces a device sync."""
        if not getattr(self, "_metrics_frozen", False):
            self._metrics(self.B, accumulate=False)
        sl, sc = self._read_metrics()
        return sl / self.B, sc / self.B

    @torch.no_grad()
    def evaluate(self, images: torch.Tensor, labels: torch.Tensor):
        """Mean loss / accuracy over an arbitrary number of rows (chunks of the engine batch): every chunk's forward
        and reduction are enqueued back to back, the host synchronises ONCE at the end."""
        n = images.shape[0]
        for lo in range(0, n, self.B):
            hi = min(n, lo + self.B)
            self.launch_forward(images[lo:hi], labels[lo:hi], None, backward=False, n=hi - lo, train_mode=False)
            self._metrics(hi - lo, accumulate=lo > 0)
        sl, sc = self._read_metrics()
        return sl / n, sc / n

    @torch.no_grad()
    def predict_logits(self, images: torch.Tensor) -> torch.Tensor:
        n = images.shape[0]
        assert n <= self.B
        dummy = torch.zeros(n, dtype=torch.int32, device=self.device)
        self.launch_forward(images, dummy, None, backward=False, n=n, train_mode=False)
        return self.logits[:n].clone()
