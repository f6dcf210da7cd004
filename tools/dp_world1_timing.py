"""Host-overhead check of the multi-GPU (RCCL) step path on one GPU (world = 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
from distributed_tensorflow_amd.models.spec import ModelSpec
from distributed_tensorflow_amd.parallel.sync_dp import SyncDataParallel
from distributed_tensorflow_amd.parallel import comm
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
comm.init_process_group("nccl", 0, 1, device=torch.device("cuda", 0))
ds = DeviceDataset.synthetic(60000)
for B in (1024,):
    eng = HipEngine(ModelSpec(), B, ds, multi_stream=os.environ.get("DTF_MS", "1") == "1"); eng.init_params(7)
    dp = SyncDataParallel(eng, tail_group=dist.new_group())
    for _ in range(20): dp.train_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): dp.train_step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"DP world1 B={B}: {t/200*1e3:.4f} ms/step (host enqueue {t_host/200*1e3:.4f} ms/step)")
