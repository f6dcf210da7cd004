"""Diagnostic: host enqueue cost vs GPU time of the world-1 data-parallel step (VERDICT r1 Weak #2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch
from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
from distributed_tensorflow_amd.models.spec import ModelSpec
from distributed_tensorflow_amd.parallel.sync_dp import SyncDataParallel
from distributed_tensorflow_amd.parallel import comm

torch.cuda.set_device(0)
comm.init_process_group(backend="nccl", rank=0, world_size=1, device=torch.device("cuda", 0))
spec = ModelSpec.deep_mnist()
ds = DeviceDataset.synthetic(60000, seed=1234)

def timeit(step, n=300):
    for _ in range(30): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    return 1e3 * t_host / n, 1e3 * t_all / n

eng = HipEngine(spec, 1024, ds, optimizer="sgd", learning_rate=1e-3, seed=1000, multi_stream=False, fuse_dense_update=True)
eng.init_params(seed=7)
print("fused single-GPU step: host %.4f ms, total %.4f ms" % timeit(eng.train_step))
eng2 = HipEngine(spec, 1024, ds, optimizer="sgd", learning_rate=1e-3, seed=1000, multi_stream=False, fuse_dense_update=False)
eng2.init_params(seed=7)
print("unfused single-GPU step (wgrad store + fused tail over the arena): host %.4f ms, total %.4f ms" % timeit(eng2.train_step))
tg = torch.distributed.new_group()
dp = SyncDataParallel(eng2, tail_group=tg)
dp.broadcast_parameters()
print("dp step, rccl both: host %.4f ms, total %.4f ms" % timeit(dp.train_step))
real = dp._allreduce
dp._allreduce = lambda lo, hi, g: None
print("dp step, no collectives enqueued: host %.4f ms, total %.4f ms" % timeit(dp.train_step))
dp._allreduce = real
dp2 = SyncDataParallel(eng2, tail_group=tg, defer_dense_update=False)
print("dp step, rccl, dense update not deferred: host %.4f ms, total %.4f ms" % timeit(dp2.train_step))
try:
    from distributed_tensorflow_amd.parallel.p2p import P2PAllReduce
    n_tail = eng2.layout.total - eng2.layout.segment("wd1", "wd1")[1]
    p = P2PAllReduce(n_tail, torch.device("cuda", 0), timeout_s=2.0, two_shot=False)
    dp3 = SyncDataParallel(eng2, tail_group=tg, tail_p2p=p)
    print("dp step, p2p tail+update: host %.4f ms, total %.4f ms" % timeit(dp3.train_step))
    dp3._allreduce = lambda lo, hi, g: None
    print("dp step, p2p tail+update, dense collective not enqueued: host %.4f ms, total %.4f ms" % timeit(dp3.train_step))
except Exception as e:
    print("p2p:", repr(e))
# E2 experiment: never wait for the deferred dense update (breaks the step; timing only)
dp4 = SyncDataParallel(eng2, tail_group=tg, tail_p2p=p)
real_step = dp4.train_step
def no_wait_step():
    dp4._have_pending_dense = False
    real_step()
print("dp step, p2p tail+update, NO wait for the dense update (E2 off): host %.4f ms, total %.4f ms" % timeit(no_wait_step))
