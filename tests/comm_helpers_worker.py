"""Rank body of test_comm_helpers.py (not a test module): the slice collectives of parallel/comm.py over gloo."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.parallel import comm  # noqa: E402

rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
comm.init_process_group("gloo", rank, world, "127.0.0.1", port, timeout_s=60)
n = 4 * world
base = torch.arange(n, dtype=torch.float32)
t = base * (rank + 1)
comm.reduce_scatter_sum_inplace(t, rank, world)
want = base * sum(range(1, world + 1))
assert torch.equal(t[4 * rank:4 * rank + 4], want[4 * rank:4 * rank + 4]), t
for dtype in (torch.bfloat16, torch.float32):
    b = torch.zeros(n, dtype=dtype)
    b[4 * rank:4 * rank + 4] = rank + 1
    comm.all_gather_inplace(b, rank, world)
    assert b.tolist() == [float(q + 1) for q in range(world) for _ in range(4)], b
print(f"COMM rank {rank} ok", flush=True)
dist.barrier()
comm.shutdown()
