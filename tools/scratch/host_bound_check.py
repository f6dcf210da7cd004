"""Is the eager single-GPU step host-bound?  Time the enqueue of N steps (no sync) and the wall time to completion.
  python tools/scratch/host_bound_check.py <batch> [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
from distributed_tensorflow_amd.models.spec import ModelSpec

B = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
spec = ModelSpec.from_name("deep_mnist")
ds = DeviceDataset.synthetic(60000, seed=1)
eng = HipEngine(spec, B, ds, seed=1)
eng.init_params(seed=1)
for _ in range(300): eng.train_step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(N): eng.train_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"batch {B}: enqueue {1e6 * (t1 - t0) / N:6.1f} us/step, completion {1e6 * (t2 - t0) / N:6.1f} us/step, drain after enqueue {1e3 * (t2 - t1):7.2f} ms")
