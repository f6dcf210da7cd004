"""Exchange fabric of the data-parallel step: peer-mapped buffers + device-resident protocol state.

The reference moves every variable and every gradient through a parameter server per step (reference
.idea/MNISTDist.py:188).  Here the two messages of a synchronous step travel as peer-to-peer
stores over the xGMI mesh (csrc/kernels/dp.hip):

  * dense (wd1, 12.8 MB): reduce-scatter -> rank-ordered sum of this rank's 1/world shard -> optimizer on the shard ->
    all-gather of the tensor the fc1 kernels read (bf16 copy; fp32 weights in the fp32 configuration);
  * tail (0.25 MB): one-shot exchange + update in one launch, which is also the step's single COMMIT point.

Everything that changes from step to step -- sequence numbers, parities, verdicts -- lives in device memory, so the
whole step is captured once into a HIP graph (parallel/sync_dp.py) and replays with no host work.

Failure semantics (all-or-nothing): every in-kernel wait is bounded; a timed-out wait or a poisoned peer makes the
launch-wide verdict bad, sets the sticky POISON word and leaves parameters, moments and step state untouched from then
on; the poison travels with the flags, so every rank stops committing within one step.  ``health()`` reads the word;
the trainer turns it into exit status 3 (lost peer), as for a missed heartbeat.

Bring-up is collective over ``group`` and unanimous: either every rank has mapped every peer or ``RuntimeError`` is
raised on all of them (the caller then stays on the RCCL path).
"""
from __future__ import annotations

import ctypes
from ctypes import c_int, c_long, c_longlong, c_void_p
from typing import List, Optional

import torch
import torch.distributed as dist

from ..ops import native


class P2PHealthError(RuntimeError):
    """A peer-to-peer wait timed out or a peer reported itself poisoned: the synchronous job cannot continue."""


class FabricDesc(ctypes.Structure):
    """Mirror of ``struct DtfDpFabric`` (csrc/kernels/api.h)."""

    _fields_ = ([("base", c_void_p * 8), ("world", c_int), ("rank", c_int), ("dense_blocks", c_int), ("commit_blocks", c_int),
                 ("n_tail", c_long), ("n_tail_max", c_long), ("n_dense", c_long), ("csz", c_long), ("ag_bytes", c_int),
                 ("reserved", c_int)]
                + [(n, c_long) for n in ("off_tail_stage", "off_tail_flags", "off_tail_done", "off_dense_a", "off_dense_b",
                                         "off_flag_a", "off_flag_b", "off_done_a", "off_done_b")]
                + [("seq_tail", c_void_p), ("seq_dense", c_void_p), ("ctl", c_void_p), ("gsum", c_void_p),
                   ("timeout_ticks", c_longlong)])


CTL_POISON, CTL_WHY, CTL_DENSE_OK, CTL_MUTE_SEQ = 0, 1, 2, 3
TAIL_BLOCKS, MAX_BLOCKS = 64, 256


class DpFabric:
    MAX_WORLD = 8

    def __init__(self, n_dense: int, n_tail: int, device, dtype: str = "bf16", group=None, timeout_s: float = 60.0,
                 dense_blocks: Optional[int] = None, commit_blocks: Optional[int] = None):
        self.lib = native.load()
        self.group, self.device = group, torch.device(device)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if not 1 <= self.world <= self.MAX_WORLD:
            raise RuntimeError(f"the exchange fabric supports up to {self.MAX_WORLD} ranks, got {self.world}")
        if n_dense % (8 * self.world) or n_tail % 4:
            raise RuntimeError(f"the exchange fabric needs a dense range divisible by 8 x world (got {n_dense}, world "
                               f"{self.world}) and a tail that is a multiple of 4")
        self.timeout_s = float(timeout_s)
        d = FabricDesc()
        d.world, d.rank = self.world, self.rank
        import os
        if dense_blocks is None:       # DTF_DP_DENSE_BLOCKS: tuning override (listed in bench.py's env_overrides)
            dense_blocks = int(os.environ.get("DTF_DP_DENSE_BLOCKS", "64"))
        if commit_blocks is None and os.environ.get("DTF_DP_COMMIT_BLOCKS"):
            commit_blocks = int(os.environ["DTF_DP_COMMIT_BLOCKS"])
        if commit_blocks is None:
            # the commit workgroups fetch 6 float4 per thread before the step's verdict arrives (dp.hip): enough of
            # them that the whole shard is in registers by then (one rank: 256, eight ranks: 66)
            commit_blocks = max(16, min(MAX_BLOCKS, -(-(n_dense // self.world // 4) // (256 * 6))))
        d.dense_blocks, d.commit_blocks = int(dense_blocks), int(commit_blocks)
        d.n_tail, d.n_tail_max, d.n_dense = n_tail, n_tail, n_dense
        d.ag_bytes = 4 if dtype == "fp32" else 2
        d.timeout_ticks = int(self.timeout_s * 1e8)
        nbytes = int(self.lib.dtf_dp_fabric_layout(ctypes.byref(d)))
        if nbytes <= 0:
            raise RuntimeError("dtf_dp_fabric_layout rejected the configuration")
        # DTF_DP_FAKE_WORLD=k (one rank only; TIMING DIAGNOSTIC, the training result is wrong): the dense kernels work
        # on a shard of 1/k of wd1, as each rank of a k-rank job would -- what their local cost is at that world size
        self.fake_world = int(os.environ.get("DTF_DP_FAKE_WORLD", "0")) if self.world == 1 else 0
        if self.fake_world > 1:
            d.csz = n_dense // self.fake_world // 8 * 8
            if os.environ.get("DTF_DP_COMMIT_BLOCKS") is None:
                d.commit_blocks = max(16, min(MAX_BLOCKS, -(-(int(d.csz) // 4) // (256 * 6))))
        # protocol state in ordinary device memory of this rank
        self.seq_tail = torch.ones(TAIL_BLOCKS + MAX_BLOCKS, dtype=torch.int32, device=self.device)
        self.seq_dense = torch.ones(MAX_BLOCKS, dtype=torch.int32, device=self.device)
        self.ctl = torch.zeros(4, dtype=torch.int32, device=self.device)
        self.gsum = torch.zeros(int(d.csz), dtype=torch.float32, device=self.device)
        d.seq_tail, d.seq_dense = self.seq_tail.data_ptr(), self.seq_dense.data_ptr()
        d.ctl, d.gsum = self.ctl.data_ptr(), self.gsum.data_ptr()
        self.desc = d
        self._mapped: List[int] = []
        self._own = c_void_p()
        err: Optional[str] = None
        handle = ctypes.create_string_buffer(64)
        with torch.cuda.device(self.device):
            try:
                native.check(self.lib.dtf_p2p_alloc(nbytes, ctypes.byref(self._own)), "fabric alloc")
                native.check(self.lib.dtf_p2p_export(self._own, handle), "hipIpcGetMemHandle")
            except Exception as e:     # keep making the collective calls below: the peers are waiting in them
                err = repr(e)
            handles: List[Optional[bytes]] = [None] * self.world
            dist.all_gather_object(handles, None if err else handle.raw, group=group)
            # does every rank have a GPU to itself?  (ranks that share a device are time-sliced: the step then keeps
            # the kernel order that does not let one rank's spinning exchange starve the other's kernels, engine.hip)
            ids: List[Optional[str]] = [None] * self.world
            dist.all_gather_object(ids, self._device_identity(), group=group)
            self.exclusive_devices = len(set(ids)) == self.world
            if err is None and any(h is None for h in handles):
                err = "a peer could not allocate/export its buffer"
            if err is None:
                try:
                    for r, h in enumerate(handles):
                        if r == self.rank:
                            d.base[r] = self._own.value
                        else:
                            p = c_void_p()
                            native.check(self.lib.dtf_p2p_import(h, ctypes.byref(p)), f"hipIpcOpenMemHandle(rank {r})")
                            d.base[r] = p.value
                            self._mapped.append(p.value)
                except Exception as e:
                    err = repr(e)
        oks: List[Optional[str]] = [None] * self.world
        dist.all_gather_object(oks, err, group=group)     # unanimous: every rank has every mapping or nobody goes on
        if any(o is not None for o in oks):
            self.close()
            raise RuntimeError("exchange fabric bring-up failed: " + "; ".join(f"rank {r}: {o}" for r, o in enumerate(oks) if o))
        torch.cuda.synchronize(self.device)
        dist.barrier(group=group)     # nobody stores into a peer before everybody has mapped (and zeroed) everything

    def _device_identity(self) -> str:
        import socket
        p = torch.cuda.get_device_properties(self.device)
        uuid = getattr(p, "uuid", None)
        return f"{socket.gethostname()}/{uuid if uuid is not None else torch.cuda.current_device()}/{getattr(p, 'pci_bus_id', '')}"

    def byref(self):
        return ctypes.byref(self.desc)

    # ------------------------------------------------------------------ health
    def poisoned(self) -> int:
        """0 = healthy; else the reason bits (1 a wait timed out, 2 a peer was poisoned, 4 launch broken).  Synchronises."""
        c = self.ctl.tolist()
        return (c[CTL_WHY] or 1) if c[CTL_POISON] else 0

    def health(self):
        why = self.poisoned()
        if why:
            what = [s for bit, s in ((1, "a wait for a peer timed out"), (2, "a peer reported a failed exchange"),
                                     (4, "an exchange launch did not complete")) if why & bit]
            raise P2PHealthError("peer-to-peer exchange: " + ", ".join(what) + " (lost or stalled worker); no update has "
                                 "been applied since")

    def next_seq(self) -> int:
        """Sequence number the next step's exchanges will use (synchronises; tests and the fault hook)."""
        return int(self.seq_tail[0].item())

    def mute_step(self, seq: int):
        """Test hook: the step with this sequence number raises none of its flags on this rank (a stalled sender)."""
        self.ctl[CTL_MUTE_SEQ] = int(seq)

    def close(self):
        for p in self._mapped:
            self.lib.dtf_p2p_unmap(p)
        self._mapped = []
        if self._own.value:
            self.lib.dtf_p2p_free(self._own)
            self._own = c_void_p()
