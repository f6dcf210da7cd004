"""ctypes binding of libdtf_kernels.so (hand-written gfx950 HIP kernels).

The library is built in-tree by tools/build_native.py (``hipcc
--offload-arch=gfx950``) and exposes a plain C ABI that takes raw device
pointers plus the HIP stream to launch on, so a launch costs one foreign call
and kernels enqueue on whatever stream PyTorch has current (HIP-graph capture
included).  There is deliberately no eager/PyTorch fallback for these ops: on a
GPU box a missing library is an error, not a silent slow path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_uint, c_uint64, c_void_p
from pathlib import Path
from typing import Optional

_LIB_DIR = Path(__file__).resolve().parent.parent / "_lib"
_LIB_NAME = "libdtf_kernels.so"
_lib: Optional[ctypes.CDLL] = None


class NativeLibraryMissing(RuntimeError):
    pass


_SIGS = {
    # name: argtypes
    "dtf_conv1_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_conv1_fwd_sampled": [c_void_p, c_void_p, c_void_p, c_uint, c_uint, c_uint, c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_conv1_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_conv2_pack_weights": [c_void_p, c_void_p, c_int, c_int, c_void_p],
    "dtf_conv2_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_conv12_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_uint, c_uint, c_uint, c_int, c_void_p, c_void_p, c_void_p,
                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "dtf_conv2_dgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_conv2_dgrad_conv1_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "dtf_conv2_dgrad_conv1_wgrad_g": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_void_p],
    "dtf_conv2_small_batch": [],
    "dtf_conv2_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "dtf_conv2_wgrad_nt": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "dtf_gemm_nt": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                    c_float, c_uint64, c_void_p, c_int, c_void_p],
    "dtf_gemm_tn_sgd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "dtf_gemm_tn": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "dtf_gemm_nn_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                        c_float, c_uint64, c_void_p, c_void_p],
    "dtf_transpose_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_head_fwd_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                         c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_int, c_void_p],
    "dtf_head_fused_fwd_bwd": [c_void_p, c_int, c_void_p, c_float, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                               c_float, c_float, c_int, c_void_p],
    "dtf_transpose_head_w": [c_void_p, c_void_p, c_int, c_void_p],
    "dtf_head_set_onepass": [c_int],
    "dtf_dropout_mask": [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_float, c_void_p, c_long, c_void_p],
    "dtf_metrics_reduce": [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p],
    "dtf_gemm_nn_partial": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "dtf_head_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_gemm_nt_head_wgrad": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_sgd_update": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_long, c_long, c_long, c_void_p],
    "dtf_adam_update": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_int,
                        c_void_p, c_long, c_long, c_long, c_void_p],
    "dtf_sgd_update_g16": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_long, c_long, c_long, c_void_p],
    "dtf_adam_update_g16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_int,
                            c_void_p, c_long, c_long, c_long, c_void_p],
    "dtf_gemm_tn_g16_ev": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "dtf_advance_state": [c_void_p, c_long, c_long, c_long, c_void_p],
    "dtf_cast_bf16": [c_void_p, c_void_p, c_long, c_void_p],
    "dtf_reduce_segments": [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_int),
                            ctypes.POINTER(c_int), ctypes.POINTER(c_long), c_void_p],
    "dtf_fused_reduce_sgd": [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                             ctypes.POINTER(c_long), ctypes.POINTER(c_long), c_void_p, c_void_p, c_void_p, c_void_p,
                             c_long, c_long, c_void_p, c_long, c_int, c_int, c_void_p, c_long, c_int, c_void_p, c_long, c_long,
                             c_long, c_void_p],
    "dtf_fused_reduce_adam": [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                              ctypes.POINTER(c_long), ctypes.POINTER(c_long), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_long, c_long, c_void_p, c_long, c_int, c_int, c_void_p, c_long, c_int, c_void_p,
                              c_int, c_long, c_long, c_long, c_void_p],
    "dtf_p2p_bytes": [c_int, c_long],
    "dtf_ps_blocks": [],
    "dtf_ps_max_workers": [],
    "dtf_ps_push": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, ctypes.POINTER(c_long), ctypes.POINTER(c_long),
                    c_long, c_uint, c_uint, c_void_p],
    "dtf_ps_install": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, ctypes.POINTER(c_long), ctypes.POINTER(c_long),
                       c_long, c_uint, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                       c_void_p, c_double, c_void_p],
    "dtf_ps_ctl_bytes": [],
    "dtf_ps_out_words": [],
    "dtf_ps_serve_init": [c_void_p, c_int, ctypes.c_ulonglong, c_void_p],
    "dtf_ps_ctl_set_dropped": [c_void_p, c_uint, c_void_p],
    "dtf_ps_serve": [c_int, c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_long, c_long,
                     ctypes.POINTER(c_long), c_int, ctypes.POINTER(c_long), ctypes.POINTER(c_long), c_void_p, c_void_p, c_void_p,
                     c_void_p, c_uint, ctypes.c_ulonglong, c_uint, c_double, c_double, c_void_p, c_void_p],
    "dtf_p2p_alloc": [c_long, ctypes.POINTER(c_void_p)],
    "dtf_p2p_free": [c_void_p],
    "dtf_p2p_export": [c_void_p, c_char_p],
    "dtf_p2p_import": [c_char_p, ctypes.POINTER(c_void_p)],
    "dtf_p2p_unmap": [c_void_p],
    "dtf_p2p_allreduce": [c_void_p, ctypes.POINTER(c_void_p), c_int, c_int, c_long, c_long, c_uint, c_void_p, c_double,
                          c_void_p],
    "dtf_p2p2_bytes": [c_int, c_long],
    "dtf_p2p2_allreduce": [c_void_p, ctypes.POINTER(c_void_p), c_int, c_int, c_long, c_long, c_uint, c_void_p, c_double,
                           c_int, c_int, c_void_p],
    "dtf_sample_indices": [c_void_p, c_void_p, c_int, c_uint, c_uint, c_uint, c_int, c_void_p],
    "dtf_gen_synthetic": [c_void_p, c_void_p, c_long, c_uint, c_void_p],
    "dtf_init_truncated_normal": [c_void_p, c_long, c_float, c_uint64, c_uint64, c_void_p],
    "dtf_fill_f32": [c_void_p, c_long, c_float, c_void_p],
    "dtf_u8_to_bf16_unit": [c_void_p, c_void_p, c_long, c_void_p],
    # ---- exact-fp32 kernel set (csrc/kernels/*_f32.hip) ----
    "dtf_conv1_f32_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_conv1_f32_fwd_sampled": [c_void_p, c_void_p, c_void_p, c_uint, c_uint, c_uint, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_conv1_f32_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_u8_to_f32_unit": [c_void_p, c_void_p, c_long, c_void_p],
    "dtf_conv2_f32_pack_weights": [c_void_p, c_void_p, c_int, c_int, c_void_p],
    "dtf_conv2_f32_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_conv2_f32_dgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_conv2_f32_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "dtf_gemm_f32_nn_partial": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "dtf_gemm_f32_force_t112": [c_int],
    "dtf_gemm_f32_nt": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "dtf_gemm_f32_nt_head_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_gemm_f32_nt_splitk": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                               c_void_p],
    "dtf_gemm_f32_tn": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "dtf_gemm_f32_tn_sk": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                           c_void_p, c_void_p],
    "dtf_gemm_f32_sk_workspace_bytes": [],
    "dtf_gemm_f32_sk_workspace_alloc": [],
    "dtf_gemm_f32_sk_workspace_free": [c_void_p],
    "dtf_head_f32_fwd_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_int, c_void_p],
    "dtf_head_f32_fused_fwd_bwd": [c_void_p, c_int, c_void_p, c_float, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_float, c_float, c_int, c_void_p],
    "dtf_head_f32_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
}


class EngineDesc(ctypes.Structure):
    """Mirror of ``struct DtfEngineDesc`` (csrc/runtime/engine.hip)."""

    _fields_ = (
        [(n, c_int) for n in ("B", "c1", "c2", "hidden", "flat", "classes", "head_ks", "c2_g", "c1_g", "optimizer",
                              "shuffle", "multi_stream", "fc1_splits", "c2_nt", "dgrad_splits", "c1w_imgs")]
        + [(n, c_uint) for n in ("ds_n", "ds_hb", "sample_seed", "fuse_dense_update")]
        + [("seed", c_uint64), ("keep_prob", c_float), ("dtype", c_int), ("n_total", c_long)]
        + [(n, c_long) for n in ("off_out_w", "off_out_b", "off_bd1", "off_wd1", "off_wc2", "off_bc2", "off_wc1",
                                 "off_bc1", "head_stride", "c2_stride", "c1_stride")]
        + [(n, c_void_p) for n in ("params", "grads", "shadow", "fc1_part", "wc2T", "adam_m", "adam_v", "state", "hyper",
                                   "idx", "p1", "arg1", "p2", "arg2", "h1", "logits", "loss_rows", "correct_rows",
                                   "dlogits", "da", "dflat", "dp1", "head_part", "c2_part", "c1_part", "dgrad_part", "dense_g16", "out_wT",
                                   "ds_images",
                                   "ds_labels")]
        + [(n, c_void_p) for n in ("side1", "side2", "ev_head", "ev_fork", "ev_join1", "ev_join2", "ev_dgrad",
                                   "ev_dense_ready", "ev_dense_updated", "ev_tail_ready", "ev_tail_reduced", "sk_ws")]
    )


_EP = ctypes.POINTER(EngineDesc)
_ENGINE_SIGS = {
    "dtf_engine_init": [_EP],
    "dtf_engine_destroy": [_EP],
    "dtf_engine_stream_count": [_EP],
    "dtf_stream_wait_event": [c_void_p, c_void_p],
    "dtf_event_record": [c_void_p, c_void_p],
    "dtf_engine_sample": [_EP, c_void_p],
    "dtf_engine_forward_conv": [_EP, c_void_p, c_void_p, c_int, c_void_p],
    "dtf_engine_forward_dense": [_EP, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtf_engine_backward_dense": [_EP, c_void_p],
    "dtf_engine_backward_conv": [_EP, c_void_p, c_void_p, c_int, c_void_p],
    "dtf_engine_backward": [_EP, c_void_p, c_void_p, c_void_p],
    "dtf_engine_update_range": [_EP, c_long, c_long, c_int, c_int, c_long, c_int, c_void_p],
    "dtf_engine_transposes": [_EP, c_int, c_void_p],
    "dtf_engine_train_step": [_EP, c_long, c_void_p],
    "dtf_engine_dp_phase1": [_EP, c_int, c_void_p, c_void_p],
    "dtf_engine_dp_phase2": [_EP, c_long, c_long, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    "dtf_engine_dp_record_updated": [_EP, c_void_p],
    # (the fabric descriptor -- parallel/dp_fabric.py FabricDesc -- is passed by reference as an opaque pointer)
    "dtf_engine_dp_step": [_EP, c_void_p, c_int, c_long, c_void_p, c_void_p],
    "dtf_dp_fabric_layout": [c_void_p],
    "dtf_engine_dp_phase3": [_EP, c_long, c_int, c_long, c_int, c_void_p, c_void_p],
    "dtf_engine_dp_flush": [_EP, c_void_p],
}


def lib_path() -> Path:
    # DTF_KERNELS_LIB: load a diagnostic build of the kernel library instead (A/B measurements)
    override = os.environ.get("DTF_KERNELS_LIB")
    return Path(override) if override else _LIB_DIR / _LIB_NAME


def available() -> bool:
    return lib_path().exists()


def load() -> ctypes.CDLL:
    """Load (once) and return the kernel library; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not p.exists():
        raise NativeLibraryMissing(
            f"{p} not found: run `python tools/build_native.py` (hipcc --offload-arch=gfx950) first")
    lib = ctypes.CDLL(str(p), mode=os.RTLD_GLOBAL if hasattr(os, "RTLD_GLOBAL") else ctypes.DEFAULT_MODE)
    for name, argtypes in list(_SIGS.items()) + list(_ENGINE_SIGS.items()):
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = (c_long if name in ("dtf_p2p_bytes", "dtf_p2p2_bytes", "dtf_dp_fabric_layout", "dtf_ps_ctl_bytes",
                                         "dtf_gemm_f32_sk_workspace_bytes")
                      else c_void_p if name == "dtf_gemm_f32_sk_workspace_alloc"
                      else None if name == "dtf_gemm_f32_sk_workspace_free" else c_int)
    lib.dtf_build_flavour.restype = c_char_p
    lib.dtf_build_flavour.argtypes = []
    flavour = lib.dtf_build_flavour().decode()
    if os.environ.get("DTF_KERNELS_LIB") or flavour != "release":
        import sys
        print(f"[dtf] kernel library: {p} (flavour: {flavour})"
              + ("  -- DIAGNOSTIC build: timings / results of this process are not production numbers" if flavour != "release" else
                 "  -- loaded through DTF_KERNELS_LIB instead of the in-tree build"), file=sys.stderr, flush=True)
    _lib = lib
    return lib


def build_flavour() -> str:
    return load().dtf_build_flavour().decode()


def ptr(t) -> Optional[int]:
    """Raw device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: code {rc}"
                           + (" (unsupported shape/config)" if rc == -1 else
                              " (bad argument)" if rc in (-2, -3) else " (HIP launch error)"))
