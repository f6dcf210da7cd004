"""Per-kernel fixed cost inside a HIP graph: N dependent launches of a near-empty kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import native
lib = native.load()
buf = torch.zeros(1 << 22, device="cuda")
def timed(n_elems, reps=40, branches=1):
    side = torch.cuda.Stream()
    others = [torch.cuda.Stream() for _ in range(branches - 1)]
    with torch.cuda.stream(side):
        lib.dtf_fill_f32(buf.data_ptr(), n_elems, 1.0, side.cuda_stream); side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            if branches == 1:
                for _ in range(reps): lib.dtf_fill_f32(buf.data_ptr(), n_elems, 1.0, side.cuda_stream)
            else:
                ev = torch.cuda.Event(); ev.record(side)
                for o in others: o.wait_event(ev)
                for _ in range(reps):
                    lib.dtf_fill_f32(buf.data_ptr(), n_elems, 1.0, side.cuda_stream)
                    for k, o in enumerate(others):
                        lib.dtf_fill_f32(buf.data_ptr() + (k + 1) * (1 << 22), n_elems, 1.0, o.cuda_stream)
                for o in others:
                    e2 = torch.cuda.Event(); e2.record(o); side.wait_event(e2)
        gr.replay(); side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(5): gr.replay()
        e1.record(side); side.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)
for n in (64, 4096, 1 << 16, 1 << 18, 1 << 20):
    print(f"fill {n:8d} floats: {timed(n):6.2f} us per dependent launch; 2 branches: {timed(n, branches=2):6.2f} us per pair", flush=True)
