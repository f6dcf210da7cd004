"""Supervisor: chief election, init-or-restore, timed checkpoints, stop coordination.

Mirrors what ``tf.train.Supervisor(is_chief, logdir, init_op, saver, global_step,
save_model_secs=600)`` + ``sv.managed_session`` / ``should_stop`` / ``stop`` give the reference
(reference .idea/MNISTDist.py:159-173, :192):
  * the chief (worker task 0) restores the newest checkpoint in ``logdir`` if there is one,
    otherwise runs the initialiser; every other task waits for the chief's variables (here: a
    broadcast from the chief),
  * the chief saves ``logdir/model.ckpt-<global_step>`` every ``save_model_secs`` seconds
    (TF-bundle files + ``checkpoint`` state file, newest 5 kept) and once more at stop,
  * ``should_stop()`` turns true after ``request_stop()`` or when the managed block raises.  In synchronous mode
    the request is job-wide: it travels in the control block of the gradient all-reduce
    (parallel/strategies.py::SyncAllReduce) and every worker leaves the loop after the same step -- a worker
    that just walked away would strand its peers in the next collective.
Saves snapshot the state between steps (never concurrently with an update), so a checkpoint is always consistent;
the file write itself runs on a background thread (checkpoint/async_writer.py).
"""
from __future__ import annotations

import contextlib
import os
import time
from typing import Callable, Optional

from .backends import Backend
from .checkpoint import Saver, from_tf_names, latest_checkpoint, to_tf_names
from .checkpoint.async_writer import AsyncCheckpointWriter


class Supervisor:
    def __init__(self, backend: Backend, is_chief: bool, logdir: Optional[str] = "/tmp/train_logs",
                 save_model_secs: float = 600.0, saver: Optional[Saver] = None, init_seed: int = 0,
                 checkpoint_basename: str = "model.ckpt", log: Callable[[str], None] = print,
                 adam_betas=(0.9, 0.999), save_optimizer_state: bool = True):
        self.b = backend
        self.is_chief = is_chief
        self.logdir = logdir
        self.save_model_secs = save_model_secs
        self.saver = saver or Saver()
        self.init_seed = init_seed
        self.basename = checkpoint_basename
        self.log = log
        self.betas = adam_betas
        self.save_optimizer_state = save_optimizer_state   # False in async mode: the slots live on the ps tasks
        self.pre_save_check: Optional[Callable[[], None]] = None   # raises if the state must not be checkpointed
        self._writer: Optional[AsyncCheckpointWriter] = None
        self.save_wait_seconds = 0.0      # time the training thread spent inside save() (snapshot + queueing only)
        self._stop = False
        self._strategy = None
        self._last_save = time.monotonic()
        self.restored_from: Optional[str] = None
        self.saved: list = []

    # -- session bring-up ----------------------------------------------------
    def init_or_restore(self) -> bool:
        """Chief only.  Returns True if a checkpoint was restored."""
        assert self.is_chief
        prefix = latest_checkpoint(self.logdir) if self.logdir else None
        if prefix:
            params, step, adam = from_tf_names(self.saver.restore(prefix))
            self.b.load_params(params)
            if step is not None:
                self.b.set_global_step(step)
            if adam:
                # Adam's step count comes from the saved beta1_power (= beta1^(t+1), TF's slot variable), not from
                # global_step: the two differ whenever global_step does not advance by one per Adam update
                t = step or 0
                b1p = adam.get("beta1_power")
                if b1p is not None and 0.0 < b1p < 1.0 and 0.0 < self.betas[0] < 1.0:
                    import math
                    t = max(0, int(round(math.log(b1p) / math.log(self.betas[0]))) - 1)
                self.b.load_optimizer_state({"m": adam["m"], "v": adam["v"], "t": t})
            self.restored_from = prefix
            self.log(f"Restored from checkpoint {prefix} (global_step={self.b.get_global_step()})")
            return True
        self.b.init_params(self.init_seed)
        self.b.set_global_step(0)
        return False

    @contextlib.contextmanager
    def managed_session(self, strategy, final_save: bool = True):
        """Init/restore on the chief, hand the state to everyone, run the block, stop cleanly."""
        self._strategy = strategy
        try:
            if self.is_chief:
                self.init_or_restore()
            strategy.broadcast_initial()
            self._last_save = time.monotonic()
            yield self
        except BaseException:
            self._stop = True
            raise
        else:
            if final_save and self.is_chief and self.logdir:
                self.save()
        finally:
            self._stop = True
            if self._writer is not None:
                import sys
                in_flight = sys.exc_info()[1]
                try:
                    self._writer.close()      # every queued checkpoint is on disk before the session ends
                except Exception as e:
                    # a failed write must be heard, but never INSTEAD of the error that is already propagating
                    if in_flight is None:
                        raise
                    self.log(f"background checkpoint writer also failed while closing: {type(e).__name__}: {e}")
                finally:
                    self._writer = None

    # -- stop coordination ---------------------------------------------------
    def should_stop(self) -> bool:
        st = self._strategy
        if not self._stop and st is not None and getattr(st, "coordinated_stop", False) and st.poll_stop():
            self._stop = True
        if st is not None and getattr(st, "gather_done", False):
            # sharded dense update: every worker has just gathered the master weights after the same step, at the
            # chief's request (maybe_save) -- the one moment a checkpoint of them is consistent
            st.gather_done = False
            if self.is_chief and self.logdir and not self._stop:     # (a stopping job saves once, at the end)
                self.save()
        return self._stop

    def request_stop(self):
        """Ask the job to stop.  Synchronous workers stop together (after the next step at which the flag is
        read); everything else stops this task at once."""
        st = self._strategy
        if st is not None and getattr(st, "coordinated_stop", False):
            st.request_stop()
        else:
            self._stop = True

    def stop(self):
        self._stop = True

    # -- checkpoints ---------------------------------------------------------
    def save(self, block: bool = False) -> Optional[str]:
        """Snapshot the state now (between two steps) and write it in the background; returns the checkpoint prefix
        the files will have.  ``block`` waits for the files (the final save does, through the writer's close)."""
        if not (self.is_chief and self.logdir):
            return None
        t0 = time.monotonic()
        if self.pre_save_check is not None:
            self.pre_save_check()             # e.g. a timed-out peer-to-peer sum: raise instead of checkpointing it
        st = self._strategy
        if st is not None and hasattr(st, "prepare_save"):
            st.prepare_save()                 # asynchronous push mode: the fp32 master lives on the ps tasks
        if st is not None and getattr(st, "_dp", None) is not None:
            if st._dp.master_stale:
                raise RuntimeError("the dense update is sharded over the workers and this task's copy of the other "
                                   "shards is stale: save through maybe_save() (job-wide gather) or after finish()")
            st._dp.flush()                    # a deferred dense update is ordered before the snapshot (GPU-side wait)
        if self._writer is None:
            self._writer = AsyncCheckpointWriter(self.log)
        self._writer.wait()                   # one snapshot buffer: the previous save must have been written
        finish = self.b.snapshot(with_optimizer=self.save_optimizer_state)
        base = os.path.join(self.logdir, self.basename)

        def job():
            params, step, opt = finish()
            tensors = to_tf_names(params, global_step=step, adam=opt, adam_t=(opt or {}).get("t", 0), betas=self.betas)
            self.saved.append(self.saver.save(tensors, base, global_step=step))

        self._writer.submit(job)
        self._last_save = time.monotonic()
        self.save_wait_seconds += self._last_save - t0
        if block:
            self._writer.wait()
        return base

    def maybe_save(self) -> Optional[str]:
        """Timed save (chief): call between steps."""
        if self.is_chief and self.logdir and self.save_model_secs > 0 and \
                time.monotonic() - self._last_save >= self.save_model_secs:
            st = self._strategy
            if st is not None and getattr(st, "sharded", False):
                st.request_gather()           # the save happens in should_stop(), when every worker has gathered
                return None
            return self.save()
        return None
