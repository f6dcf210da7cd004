"""Rank body of test_distributed_gpu.py::test_overlapped_dp_schedule_world2_over_gloo (not a test module).

Two ranks share cuda:0 and talk over gloo; each also replays BOTH ranks with plain engines (compute
gradients, sum, scale by 1/2, apply) and compares the data-parallel result against that."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine  # noqa: E402
from distributed_tensorflow_amd.models.spec import ModelSpec  # noqa: E402
from distributed_tensorflow_amd.parallel import comm  # noqa: E402
from distributed_tensorflow_amd.parallel.sync_dp import SyncDataParallel  # noqa: E402

rank, port, optimizer, tail = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
B, STEPS, LR = 64, 5, (0.01 if optimizer == "sgd" else 1e-3)
torch.cuda.set_device(0)
comm.init_process_group("gloo", rank, 2, "127.0.0.1", port, timeout_s=90)
ds = DeviceDataset.synthetic(4096, seed=1234)


def engine(seed):
    return HipEngine(ModelSpec(), B, ds, optimizer=optimizer, learning_rate=LR, seed=seed, multi_stream=False,
                     fuse_dense_update=False)


eng = engine(1000 + rank)
eng.init_params(seed=7 + rank)            # rank 1 starts from other weights: the broadcast has to fix that
tail_p2p = dense_p2p = None
if tail.startswith("p2p"):     # both messages through the peer-to-peer kernels instead of the process group
    from distributed_tensorflow_amd.parallel.p2p import P2PAllReduce
    n_dense = eng.layout.segment("wd1", "wd1")[1]
    tail_p2p = P2PAllReduce(eng.layout.total - n_dense, eng.device)
    dense_p2p = P2PAllReduce(n_dense, eng.device, two_shot=True)
dp = SyncDataParallel(eng, tail_group=dist.new_group(), tail_p2p=tail_p2p, dense_p2p=dense_p2p,
                      p2p_tail_update=(tail != "p2p-plain"))
dp.broadcast_parameters()

refs = [engine(1000 + q) for q in (0, 1)]
for e in refs:
    e.init_params(seed=7)
    e.set_grad_scale(0.5)
assert torch.equal(eng.params, refs[0].params), "broadcast_parameters did not install rank 0's weights"

losses = []
for _ in range(STEPS):
    dp.train_step()
    for e in refs:
        e.launch_sample()
        e.launch_forward(ds.images, ds.labels, e.idx, backward=True)
        e.launch_backward_dense()
        e.launch_backward_conv(ds.images, e.idx)
    g = refs[0].grads + refs[1].grads
    for e in refs:
        e.grads.copy_(g)
        e.launch_update()
    losses.append((eng.last_metrics()[0], refs[rank].last_metrics()[0]))
dp.flush()
torch.cuda.synchronize()

for mine, ref in losses:                  # this rank saw the batches its replayed twin saw
    assert abs(mine - ref) <= 1e-5 * max(1.0, abs(ref)), losses
assert losses[-1][0] < losses[0][0], losses
err = (eng.params - refs[rank].params).abs().max().item()
assert err <= 2e-6, f"data-parallel parameters differ from the replayed two-rank step by {err}"
assert eng.global_step == refs[rank].global_step == STEPS
both = [torch.empty_like(eng.params) for _ in range(2)]
dist.all_gather(both, eng.params)
assert torch.equal(both[0], both[1]), "ranks diverged"
if tail_p2p is not None:
    assert tail_p2p.status() == 0 and dense_p2p.status() == 0
    dist.barrier()
    tail_p2p.close()
    dense_p2p.close()
print(f"DPW2 rank {rank} ok: max |dp - replay| = {err:.3g}, losses {losses[0][0]:.4f} -> {losses[-1][0]:.4f}", flush=True)
comm.shutdown()
