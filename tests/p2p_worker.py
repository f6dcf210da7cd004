"""Rank body of test_distributed_gpu.py::test_p2p_oneshot_allreduce_* (not a test module): W ranks share cuda:0,
map each other's staging buffers through hipIpc and run the one-shot all-reduce of parallel/p2p.py."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.parallel import comm  # noqa: E402
from distributed_tensorflow_amd.parallel.p2p import P2PAllReduce  # noqa: E402

rank, world, port, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
comm.init_process_group("gloo", rank, world, "127.0.0.1", port, timeout_s=90)
two = mode == "twoshot"
N = 3136 * 1024 if two else 63372          # the dense and the tail message of the deep-MNIST step
# (generous wait bound: the ranks share ONE GPU, and a spinning exchange kernel can keep a peer's kernel off it for
#  seconds -- see tests/dp_world2_worker.py)
p2p = P2PAllReduce(N, dev, two_shot=two, timeout_s=60.0)
assert p2p.self_check(N, iters=6), "self check against the process group's all-reduce failed"

# back-to-back launches without host synchronisation: exact integer sums, both parities, slot reuse
x = torch.empty(N, device=dev)
for k in range(1, 41):
    x.fill_(float(k * (rank + 1)))
    p2p.allreduce(x)
    if k % 10 == 0:
        torch.cuda.synchronize()
        want = float(k * world * (world + 1) // 2)
        assert p2p.status() == 0, "a wait timed out"
        assert bool((x == want).all()), (k, x[:4].tolist(), want)

# a shorter vector, and bit-identical results on every rank for non-trivial data
g = torch.Generator(device=dev)
g.manual_seed(77 + rank)
M = 1024 * world if two else 1000
y = torch.randn(M, device=dev, generator=g)
p2p.allreduce(y)
torch.cuda.synchronize()
gathered = [torch.empty(M) for _ in range(world)]
dist.all_gather(gathered, y.cpu())
assert all(torch.equal(gathered[0], t) for t in gathered), "ranks disagree bitwise"
assert p2p.status() == 0
dist.barrier()
p2p.close()
print(f"P2P rank {rank}/{world} ok", flush=True)
comm.shutdown()
