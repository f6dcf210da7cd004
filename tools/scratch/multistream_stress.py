"""Race screen of the single-GPU step with the backward fork/join on side streams (and under graph replay): every variant
must end bit-identical to the single-stream eager step, over many repetitions."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
from distributed_tensorflow_amd.models.spec import ModelSpec

def run(B, opt, dtype, multi, graph, fuse, steps=12):
    ds = DeviceDataset.synthetic(8192, seed=3, dtype=dtype)
    e = HipEngine(ModelSpec(), B, ds, optimizer=opt, learning_rate=0.01 if opt == "sgd" else 1e-3, seed=9, keep_prob=0.75,
                  multi_stream=multi, fuse_dense_update=fuse, dtype=dtype)
    e.init_params(seed=4)
    if graph:
        e.capture()
    for _ in range(steps):
        e.train_step()
    torch.cuda.synchronize()
    return e.params.clone(), e.state.clone()

bad = 0
for B, opt, dtype in ((2048, "sgd", "bf16"), (2048, "adam", "bf16"), (512, "sgd", "bf16"), (1024, "sgd", "fp32")):
    ref_p, ref_s = run(B, opt, dtype, False, False, False)
    for rep in range(6):
        for multi, graph, fuse in ((True, False, False), (True, True, False), (False, True, False), (True, True, opt == "sgd")):
            p, s = run(B, opt, dtype, multi, graph, fuse)
            ok = torch.equal(s, ref_s) and (torch.equal(p, ref_p) if not fuse else torch.allclose(p, ref_p, rtol=0, atol=1e-6))
            if not ok:
                bad += 1
                print(f"MISMATCH B={B} {opt} {dtype} multi={multi} graph={graph} fuse={fuse} rep={rep}: "
                      f"max diff {(p - ref_p).abs().max().item():.3g} state {s.tolist()} vs {ref_s.tolist()}", flush=True)
    print(f"B={B} {opt} {dtype}: done", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
