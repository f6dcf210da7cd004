"""Rank body of test_distributed_gpu.py::test_fabric_stalled_peer_is_all_or_nothing (not a test module).

WORLD ranks share cuda:0, both messages of the step go through the exchange fabric (csrc/kernels/dp.hip).  After a few
good steps one rank's next step raises none of its flags (a stalled sender).  Required: every rank ends POISONED, and
on every rank parameters, moments and step state are bit-for-bit those of a COMPLETE step -- the snapshot taken before
the fault, or (for the rank that stalled without knowing it) nothing else either, because the peers' poison reaches it
in the same step -- and stay so through the steps that follow."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine  # noqa: E402
from distributed_tensorflow_amd.models.spec import ModelSpec  # noqa: E402
from distributed_tensorflow_amd.parallel import comm  # noqa: E402
from distributed_tensorflow_amd.parallel.dp_fabric import DpFabric, P2PHealthError  # noqa: E402
from distributed_tensorflow_amd.parallel.sync_dp import SyncDataParallel  # noqa: E402

rank, port, optimizer, mode, WORLD, DTYPE = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
B, GOOD, AFTER, STALLED = 64, 4, 3, WORLD - 1
torch.cuda.set_device(0)
comm.init_process_group("gloo", rank, WORLD, "127.0.0.1", port, timeout_s=120)
ds = DeviceDataset.synthetic(4096, seed=1234, dtype=DTYPE)
eng = HipEngine(ModelSpec(), B, ds, optimizer=optimizer, learning_rate=0.01 if optimizer == "sgd" else 1e-3, seed=1000 + rank,
                multi_stream=False, fuse_dense_update=False, dtype=DTYPE)
eng.init_params(seed=7)
n_dense = eng.layout.segment("wd1", "wd1")[1]
# the bound of the in-kernel waits: long enough for ranks that time-slice one GPU, short enough for a test
fabric = DpFabric(n_dense, eng.layout.total - n_dense, eng.device, dtype=DTYPE, timeout_s=6.0 if WORLD == 2 else 15.0)
dp = SyncDataParallel(eng, tail_group=dist.new_group(), fabric=fabric)
dp.broadcast_parameters()
if mode == "graph":
    dp.capture()
for _ in range(GOOD):
    dp.train_step()
torch.cuda.synchronize()
assert fabric.poisoned() == 0, "poisoned before the fault"
arenas = [eng.params, eng.state] + ([eng.adam_m, eng.adam_v] if optimizer == "adam" else [])
snap = [a.clone() for a in arenas]
assert eng.global_step == GOOD
dist.barrier()
if rank == STALLED:
    fabric.mute_step(fabric.next_seq())
for k in range(1 + AFTER):
    dp.train_step()
    torch.cuda.synchronize()
    for name, a, s0 in zip(("params", "state", "adam_m", "adam_v"), arenas, snap):
        assert torch.equal(a, s0), f"rank {rank}: {name} changed in step {k} after the fault (a half-applied step)"
why = fabric.poisoned()
assert why != 0, f"rank {rank} is not poisoned"
try:
    fabric.health()
    raise AssertionError("health() did not raise")
except P2PHealthError:
    pass
every = [None] * WORLD
dist.all_gather_object(every, why)
# every rank holds complete steps only, and the same ones: compare the arenas across ranks (own shards differ by design
# in the mixed-precision configuration -- masters of foreign shards are stale -- so compare what every rank owns: the tail)
tails = [torch.empty(eng.layout.total - n_dense) for _ in range(WORLD)]
dist.all_gather(tails, eng.params[n_dense:].cpu())
assert all(torch.equal(tails[0], t) for t in tails[1:]), "ranks hold different tail parameters"
dp.release_graph()
dist.barrier()
fabric.close()
print(f"DPFAULT rank {rank} ok: poisoned (reason bits per rank {every}), arenas equal the pre-fault snapshot", flush=True)
comm.shutdown()
