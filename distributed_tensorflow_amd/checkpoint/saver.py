"""tf.train.Saver-compatible checkpointing (reference .idea/MNISTDist.py:154 ``saver =
tf.train.Saver()`` handed to the Supervisor, :159-165).

* ``Saver.save(tensors, logdir/model.ckpt, global_step)`` writes
  ``model.ckpt-<step>.index`` / ``.data-00000-of-00001`` under the reference's
  variable names (``Variable`` ... ``Variable_8``; optimizer slots use TF's
  ``Variable/Adam`` naming) and maintains the text-proto ``checkpoint`` state file
  with ``max_to_keep`` (default 5) rotation, exactly like TF.
* ``latest_checkpoint(logdir)`` / ``Saver.restore(prefix)`` are the other direction; they also take
  the single-file Saver **V1** checkpoints that the TF 0.x the reference targets wrote (so a logdir
  left by the reference resumes here), and ``Saver(write_version=1)`` writes that format.
Native layouts equal TF's (HWIO / [in,out] / HWC flatten), so import/export is a
plain copy of each tensor.
"""
from __future__ import annotations

import os
import re
from typing import Dict, List, Optional

import numpy as np

from ..models.spec import TF_VARIABLE_NAMES
from . import bundle

_STATE_FILE = "checkpoint"


def to_tf_names(tensors: Dict[str, np.ndarray], global_step: Optional[int] = None,
                adam: Optional[Dict[str, Dict[str, np.ndarray]]] = None,
                adam_t: int = 0, betas=(0.9, 0.999)) -> Dict[str, np.ndarray]:
    """logical names (wc1, ...) -> TF checkpoint keys (Variable, ...)."""
    out = {}
    for name, arr in tensors.items():
        out[TF_VARIABLE_NAMES[name]] = np.asarray(arr, dtype=np.float32)
    if global_step is not None:
        out[TF_VARIABLE_NAMES["global_step"]] = np.asarray(global_step, dtype=np.int32)  # int32 scalar (M:147)
    if adam:
        for name, arr in adam.get("m", {}).items():
            out[TF_VARIABLE_NAMES[name] + "/Adam"] = np.asarray(arr, dtype=np.float32)
        for name, arr in adam.get("v", {}).items():
            out[TF_VARIABLE_NAMES[name] + "/Adam_1"] = np.asarray(arr, dtype=np.float32)
        out["beta1_power"] = np.asarray(betas[0] ** (adam_t + 1), dtype=np.float32)
        out["beta2_power"] = np.asarray(betas[1] ** (adam_t + 1), dtype=np.float32)
    return out


def from_tf_names(ckpt: Dict[str, np.ndarray]):
    """TF keys -> (params by logical name, global_step or None, adam slots or None)."""
    inv = {v: k for k, v in TF_VARIABLE_NAMES.items()}
    params, m, v = {}, {}, {}
    step = None
    for key, arr in ckpt.items():
        if key in inv:
            if inv[key] == "global_step":
                step = int(arr)
            else:
                params[inv[key]] = arr
        elif key.endswith("/Adam") and key[:-5] in inv:
            m[inv[key[:-5]]] = arr
        elif key.endswith("/Adam_1") and key[:-7] in inv:
            v[inv[key[:-7]]] = arr
    adam = {"m": m, "v": v} if m and v else None
    if adam is not None and "beta1_power" in ckpt:
        adam["beta1_power"] = float(ckpt["beta1_power"])
        adam["beta2_power"] = float(ckpt["beta2_power"])
    return params, step, adam


def read_checkpoint_state(logdir: str) -> List[str]:
    """All checkpoint prefixes recorded in ``logdir/checkpoint`` (oldest first)."""
    path = os.path.join(logdir, _STATE_FILE)
    if not os.path.exists(path):
        return []
    latest, allp = None, []
    for line in open(path):
        mm = re.match(r'\s*(model_checkpoint_path|all_model_checkpoint_paths)\s*:\s*"(.*)"\s*$', line)
        if not mm:
            continue
        p = mm.group(2)
        if not os.path.isabs(p):
            p = os.path.join(logdir, p)
        if mm.group(1) == "model_checkpoint_path":
            latest = p
        else:
            allp.append(p)
    if latest and latest not in allp:
        allp.append(latest)
    return allp


def latest_checkpoint(logdir: str) -> Optional[str]:
    """tf.train.latest_checkpoint: newest prefix whose files exist (.index, or the V1 file itself), else None."""
    for p in reversed(read_checkpoint_state(logdir)):
        if _exists(p):
            return p
    return None


def _exists(prefix: str) -> bool:
    return os.path.exists(prefix + ".index") or os.path.isfile(prefix)


def _write_state(logdir: str, prefixes: List[str]):
    rel = [os.path.basename(p) if os.path.dirname(p) == logdir.rstrip("/") else p for p in prefixes]
    tmp = os.path.join(logdir, _STATE_FILE + ".tmp")
    with open(tmp, "w") as f:
        f.write(f'model_checkpoint_path: "{rel[-1]}"\n')
        for r in rel:
            f.write(f'all_model_checkpoint_paths: "{r}"\n')
    os.replace(tmp, os.path.join(logdir, _STATE_FILE))


class Saver:
    def __init__(self, max_to_keep: int = 5, write_version: int = 2):
        self.max_to_keep = max_to_keep
        self.write_version = write_version
        self._kept: List[str] = []

    def save(self, tf_tensors: Dict[str, np.ndarray], save_path: str, global_step: Optional[int] = None) -> str:
        """Write a checkpoint; returns its prefix (``save_path-<global_step>``)."""
        prefix = save_path if global_step is None else f"{save_path}-{int(global_step)}"
        logdir = os.path.dirname(prefix) or "."
        os.makedirs(logdir, exist_ok=True)
        bundle.write_bundle(prefix, tf_tensors, version=self.write_version)
        if not self._kept:
            self._kept = [p for p in read_checkpoint_state(logdir) if _exists(p)]
        if prefix in self._kept:
            self._kept.remove(prefix)
        self._kept.append(prefix)
        while self.max_to_keep and len(self._kept) > self.max_to_keep:
            old = self._kept.pop(0)
            for suffix in (".index", ".data-00000-of-00001", "", ".meta"):
                try:
                    os.remove(old + suffix)
                except FileNotFoundError:
                    pass
        _write_state(logdir, self._kept)
        return prefix

    @staticmethod
    def restore(prefix: str) -> Dict[str, np.ndarray]:
        return bundle.read_bundle(prefix)
