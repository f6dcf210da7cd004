"""Where the host time of the RCCL data-parallel step goes (world = 1 on one GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
from distributed_tensorflow_amd.models.spec import ModelSpec
from distributed_tensorflow_amd.parallel.sync_dp import SyncDataParallel
from distributed_tensorflow_amd.parallel import comm
from distributed_tensorflow_amd.ops.native import check
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
comm.init_process_group("nccl", 0, 1, device=torch.device("cuda", 0))
ds = DeviceDataset.synthetic(60000)
eng = HipEngine(ModelSpec(), 1024, ds); eng.init_params(7)
dp = SyncDataParallel(eng, tail_group=dist.new_group())
for _ in range(20): dp.train_step()
torch.cuda.synchronize()
def t(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    h = time.perf_counter() - t0
    torch.cuda.synchronize(); return h / n * 1e6, (time.perf_counter() - t0) / n * 1e6
lib, d = eng.lib, eng._d()
cur = torch.cuda.current_stream().cuda_stream
comm_s = dp.comm_stream.cuda_stream
tail_s = cur if dp.tail_stream is None else dp.tail_stream.cuda_stream
def p1(): check(lib.dtf_engine_dp_phase1(d, 0, cur, comm_s), "p1")
def p2(): check(lib.dtf_engine_dp_phase2(d, dp.n_dense, 3, 1, None, cur, comm_s, tail_s), "p2")
def p3(): check(lib.dtf_engine_dp_phase3(d, dp.n_dense, 4, 1, 0, cur, tail_s), "p3")
def ar_dense():
    with torch.cuda.stream(dp.comm_stream): dp._allreduce(0, dp.n_dense, dp.group)
def ar_tail():
    if dp.tail_stream is None: dp._allreduce(*dp.tail_bucket, dp.tail_group)
    else:
        with torch.cuda.stream(dp.tail_stream): dp._allreduce(*dp.tail_bucket, dp.tail_group)
def ctx():
    with torch.cuda.stream(dp.comm_stream): pass
for name, fn in (("phase1", p1), ("phase2", p2), ("phase3", p3), ("allreduce dense (13 MB)", ar_dense),
                 ("allreduce tail (0.25 MB)", ar_tail), ("stream ctx only", ctx), ("full step", dp.train_step)):
    h, w = t(fn)
    print(f"{name:28s} host {h:7.1f} us   wall {w:7.1f} us", flush=True)
