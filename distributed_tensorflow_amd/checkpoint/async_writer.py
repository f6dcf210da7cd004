"""Background checkpoint writer (reference: the Supervisor's checkpoint thread, ``save_model_secs=600``,
reference .idea/MNISTDist.py:163-165 -- TF saves from a service thread, never from the training loop).

``Supervisor.save()`` only takes a SNAPSHOT of the state between two steps (GPU: a device-to-device copy on a
side stream followed by an asynchronous copy into pinned host memory; CPU: a clone of the arena) and queues a job;
this thread waits for the snapshot to land, renames the tensors to their TF checkpoint keys, writes the bundle
files and rotates old checkpoints.  The training loop never waits for the disk -- only for a previous snapshot
that is still being written, when saves are requested faster than they complete.
"""
from __future__ import annotations

import queue
import threading
from typing import Callable, Optional


class AsyncCheckpointWriter:
    def __init__(self, log: Callable[[str], None] = print):
        self._q: "queue.Queue" = queue.Queue()
        self._log = log
        self._err: Optional[BaseException] = None
        self._idle = threading.Event()
        self._idle.set()
        self._pending = 0
        self._lock = threading.Lock()
        self._thread = threading.Thread(target=self._run, name="dtf-ckpt-writer", daemon=True)
        self._thread.start()

    def submit(self, job: Callable[[], None]):
        """``job`` runs on the writer thread (it waits for its snapshot itself)."""
        self.raise_if_failed()
        with self._lock:
            self._pending += 1
            self._idle.clear()
        self._q.put(job)

    def _run(self):
        while True:
            job = self._q.get()
            if job is None:
                return
            try:
                job()
            except BaseException as e:      # surfaced on the training thread at the next save / at close
                self._err = e
                self._log(f"checkpoint writer: {type(e).__name__}: {e}")
            finally:
                with self._lock:
                    self._pending -= 1
                    if self._pending == 0:
                        self._idle.set()

    def busy(self) -> bool:
        return not self._idle.is_set()

    def wait(self):
        """Block until every queued checkpoint is on disk."""
        self._idle.wait()
        self.raise_if_failed()

    def raise_if_failed(self):
        if self._err is not None:
            e, self._err = self._err, None
            raise RuntimeError(f"background checkpoint write failed: {type(e).__name__}: {e}") from e

    def close(self):
        self.wait()
        self._q.put(None)
        self._thread.join(timeout=5.0)
