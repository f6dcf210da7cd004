"""conv2 filter-gradient slabs of the in-tree library against another build of it (DTF_OTHER_LIB), bit for bit:
the tap-chain MFMA waves of conv5x5_wgrad_pc accumulate every tap over the same k-steps in the same order as the
tap-group version, so nothing may differ.  Run via gpurun."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd.ops import native
from distributed_tensorflow_amd.ops.native import ptr
new = native.load()
old = ctypes.CDLL(os.environ["DTF_OTHER_LIB"])
for lib in (old,):
    lib.dtf_conv2_wgrad_nt.argtypes = new.dtf_conv2_wgrad_nt.argtypes
    lib.dtf_conv2_wgrad_nt.restype = ctypes.c_int
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(3)
ok = True
for (B, G, C1, C2) in [(1, 1, 32, 64), (7, 2, 32, 64), (37, 2, 32, 64), (1024, 64, 32, 64), (100, 50, 32, 64), (33, 4, 128, 256)]:
    x = torch.randn(B, 14, 14, C1, generator=g).bfloat16().cuda()
    gp = torch.randn(B, 7, 7, C2, generator=g).bfloat16().cuda()
    arg = torch.randint(0, 5, (B, 7, 7, C2), generator=g, dtype=torch.uint8).cuda()
    outs = []
    for lib in (new, old):
        part = torch.full((G, 25 * C1 * C2 + C2), float("nan"), device="cuda")
        rc = lib.dtf_conv2_wgrad_nt(ptr(x), ptr(gp), ptr(arg), ptr(part), B, C1, C2, G, 0, s)
        torch.cuda.synchronize()
        assert rc == 0, rc
        outs.append(part[:min(G, B)].clone())
    same = torch.equal(outs[0], outs[1])
    fin = bool(torch.isfinite(outs[0]).all())
    print(f"B {B} G {G} C {C1}/{C2}: bitwise equal {same}, finite {fin}, max abs diff {float((outs[0] - outs[1]).abs().max()):.3e}", flush=True)
    ok = ok and same and fin
sys.exit(0 if ok else 1)
