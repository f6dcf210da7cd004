"""Loss-curve distance of the fp32 engine from the fp32 GPU oracle next to the distance between two oracles (GPU, CPU),
per step, for both tilings of the fc1 backward GEMMs (DTF_F32_T112).  Run via gpurun."""
import os, sys
import torch
from distributed_tensorflow_amd.data import batch_indices
from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
from distributed_tensorflow_amd.models.oracle import OracleModel
from distributed_tensorflow_amd.models.spec import ModelSpec
from distributed_tensorflow_amd.ops import native

lib = native.load()
ds = DeviceDataset.synthetic(4096, seed=1234, dtype="fp32")
for lr in (1e-3, 5e-3):
    for mode in (0, 1):
        lib.dtf_gemm_f32_force_t112(mode)
        spec, B, steps = ModelSpec(), 64, 24
        eng = HipEngine(spec, B, ds, optimizer="sgd", learning_rate=lr, seed=21, dtype="fp32")
        eng.init_params(seed=3)
        o = OracleModel(spec, device="cuda", optimizer="sgd", learning_rate=lr); o.load_params(eng.export_params())
        oc = OracleModel(spec, device="cpu", optimizer="sgd", learning_rate=lr); oc.load_params(eng.export_params())
        out = []
        for s in range(steps):
            idx = torch.as_tensor(batch_indices(s * B, B, ds.n, 21), device="cuda")
            x, y = ds.images[idx], ds.labels[idx]
            lr_, _ = o.train_step(x, y); lc, _ = oc.train_step(x.cpu(), y.cpu())
            eng.train_step(); l, _ = eng.last_metrics()
            out.append("%d:%.1e/%.1e" % (s, abs(l - lr_) / abs(lr_), abs(lr_ - lc) / abs(lr_)))
        print("lr", lr, "t112", mode, " ".join(out), flush=True)
