"""Liveness / failure detection and job-wide control keys over the rendezvous store.

The reference gets this from the TF master session: ``sv.should_stop()`` turns true when a supervised thread
fails or the session dies, and ``managed_session`` re-raises (reference .idea/MNISTDist.py:159-173).  Here the
control plane is the job's TCPStore (the one ``torch.distributed`` rendezvoused on), reached through a SECOND
client connection owned by a monitor thread, so it keeps working while the main thread sits inside a collective:

  * every task publishes a heartbeat counter ``hb/<rank>`` once a second and watches the counters of the tasks it
    depends on (``watch``).  A counter that stops moving for ``timeout_s`` -- or a ``dead/<rank>`` note left by a
    task that is going down with an exception -- marks that peer as lost;
  * a lost peer first sets ``failed`` (the training loop raises a clear error at its next step boundary); if the
    main thread does not acknowledge within ``grace_s`` -- it is blocked in a collective that can never complete
    -- the monitor prints the reason and ends the process with exit status 3 (nothing is exec'ed, the GPU context
    just goes away with the process), instead of leaving it parked until the collective timeout;
  * ``done/<rank>`` marks an orderly exit, so finished tasks are not mistaken for dead ones;
  * ``set_flag`` / ``flag_set`` / ``wait_flag`` are plain store keys for job-wide events that must not be a timed
    collective (e.g. the ps tasks of a synchronous job wait for "workers_done" the way ``server.join()`` waits).

Asynchronous parameter-server jobs tolerate lost workers (the reference's contract): workers there watch only the
ps tasks, and the ps tasks watch nobody.
"""
from __future__ import annotations

import datetime
import os
import sys
import threading
import time
from typing import Callable, Iterable, Optional

import torch.distributed as dist

EXIT_PEER_LOST = 3


class PeerLost(RuntimeError):
    pass


class Liveness:
    def __init__(self, host: str, port: int, rank: int, watch: Iterable[int], name: str = "", interval_s: float = 1.0,
                 timeout_s: float = 15.0, grace_s: float = 10.0, log: Callable[[str], None] = print,
                 hard_exit: bool = True):
        self.rank, self.watch, self.name = rank, sorted(set(watch) - {rank}), name or f"rank {rank}"
        self.interval_s, self.timeout_s, self.grace_s = interval_s, timeout_s, grace_s
        self.log, self.hard_exit = log, hard_exit
        # a client of its own: the process group's store object is used by the main thread
        self.store = dist.TCPStore(host, port, is_master=False, timeout=datetime.timedelta(seconds=30),
                                   wait_for_workers=False)
        self.failed: Optional[str] = None
        self._acked = False
        self._stop = threading.Event()
        self._count = 0
        self._seen = {r: (None, time.monotonic()) for r in self.watch}
        self._flags_seen = set()
        self._peers = {}          # rank -> (last heartbeat value, since when): tasks checked on demand (peer_lost)
        self._lock = threading.Lock()
        self._beat()
        self._thread = threading.Thread(target=self._run, name="dtf-liveness", daemon=True)
        self._thread.start()

    # ------------------------------------------------------------------ keys
    def _beat(self):
        self._count += 1
        self.store.set(f"dtf/hb/{self.rank}", str(self._count))

    def _get(self, key: str) -> Optional[bytes]:
        try:
            if self.store.check([key]):
                return self.store.get(key)
        except Exception:      # store gone (its host task died): handled by the caller as "no progress"
            raise
        return None

    def set_flag(self, name: str, value: str = "1"):
        with self._lock:
            self.store.set(f"dtf/flag/{name}", value)

    def flag_set(self, name: str) -> bool:
        if name in self._flags_seen:
            return True
        with self._lock:
            ok = bool(self.store.check([f"dtf/flag/{name}"]))
        if ok:
            self._flags_seen.add(name)
        return ok

    def wait_flag(self, name: str, poll_s: float = 0.2):
        """Block (no timeout: like ``server.join()``) until the flag is set or a watched peer is lost."""
        while not self.flag_set(name):
            self.raise_if_failed()
            time.sleep(poll_s)

    def peer_lost(self, rank: int) -> bool:
        """On-demand check of a task that is NOT in ``watch`` (the ps of an asynchronous job tolerates lost workers but
        must stop waiting for them): True once the task has left a death note or its heartbeat has stood still for
        ``timeout_s``.  A task that finished in an orderly way is not lost."""
        now = time.monotonic()
        with self._lock:
            try:
                if self._get(f"dtf/done/{rank}") is not None:
                    return False
                if self._get(f"dtf/dead/{rank}") is not None:
                    return True
                v = self._get(f"dtf/hb/{rank}")
            except Exception:
                return False
        last, since = self._peers.get(rank, (None, now))
        if v != last:
            self._peers[rank] = (v, now)
            return False
        self._peers[rank] = (last, since)
        return now - since > self.timeout_s

    # --------------------------------------------------------------- monitor
    def _run(self):
        while not self._stop.wait(self.interval_s):
            try:
                with self._lock:
                    self._beat()
                    now = time.monotonic()
                    for r in self.watch:
                        if self._get(f"dtf/done/{r}") is not None:
                            self._seen[r] = ("done", now)
                            continue
                        dead = self._get(f"dtf/dead/{r}")
                        if dead is not None:
                            self._fail(f"task with rank {r} went down: {dead.decode(errors='replace')}")
                            break
                        v = self._get(f"dtf/hb/{r}")
                        last, since = self._seen[r]
                        if last == "done":
                            continue
                        if v != last:
                            self._seen[r] = (v, now)
                        elif now - since > self.timeout_s:
                            self._fail(f"no heartbeat from the task with rank {r} for {now - since:.0f} s (lost or hung)")
                            break
            except Exception as e:      # the store itself is unreachable: its host task is gone
                self._fail(f"rendezvous store unreachable ({type(e).__name__}): the task hosting it is gone")
            if self.failed is not None:
                break
        if self.failed is None or self._stop.is_set():
            return
        # give the main thread a chance to raise PeerLost itself; if it is stuck in a collective, end the process
        t0 = time.monotonic()
        while time.monotonic() - t0 < self.grace_s:
            if self._acked or self._stop.is_set():
                return
            time.sleep(0.1)
        if self.hard_exit and not self._acked and not self._stop.is_set():
            self.log(f"{self.name}: {self.failed}; this task is blocked in a step that cannot complete -- aborting "
                     f"(exit {EXIT_PEER_LOST}).  Restart the job: it resumes from the last checkpoint.")
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(EXIT_PEER_LOST)

    def _fail(self, why: str):
        if self.failed is None:
            self.failed = why

    def raise_if_failed(self):
        if self.failed is not None:
            self._acked = True
            raise PeerLost(self.failed)

    # ---------------------------------------------------------------- ending
    def announce_death(self, why: str):
        """Best effort, from an exception handler: tell the peers now instead of after the heartbeat timeout."""
        try:
            with self._lock:
                self.store.set(f"dtf/dead/{self.rank}", why[:200])
        except Exception:
            pass

    def close(self, orderly: bool = True):
        self._acked = True
        self._stop.set()
        if orderly:
            try:
                with self._lock:
                    self.store.set(f"dtf/done/{self.rank}", "1")
            except Exception:
                pass
        self._thread.join(timeout=2.0)
