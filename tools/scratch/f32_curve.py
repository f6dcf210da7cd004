import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
from distributed_tensorflow_amd.models.oracle import OracleModel
from distributed_tensorflow_amd.models.spec import ModelSpec
from distributed_tensorflow_amd.data import batch_indices
ds = DeviceDataset.synthetic(4096, seed=1234, dtype="fp32")
for opt, lr in (("sgd", 0.02), ("sgd", 0.002), ("adam", 1e-3)):
    spec, B = ModelSpec(), 64
    eng = HipEngine(spec, B, ds, optimizer=opt, learning_rate=lr, seed=21, dtype="fp32"); eng.init_params(seed=3)
    o = OracleModel(spec, device="cuda", optimizer=opt, learning_rate=lr); o.load_params(eng.export_params())
    o64 = OracleModel(spec, device="cpu", optimizer=opt, learning_rate=lr); o64.load_params(eng.export_params())
    out = []
    for s in range(40):
        idx = torch.as_tensor(batch_indices(s * B, B, ds.n, 21), device="cuda")
        lr_, _ = o.train_step(ds.images[idx], ds.labels[idx])
        lc, _ = o64.train_step(ds.images[idx].cpu(), ds.labels[idx].cpu())
        eng.train_step(); l, _ = eng.last_metrics()
        out.append((round(l, 5), round(lr_, 5), round(lc, 5)))
    print(opt, lr, out)
