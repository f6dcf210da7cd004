"""Stand-alone timing of the fc1 wgrad (TN, plain and fused-SGD) and forward (split-K NN) GEMMs with whatever kernel
library is loaded (DTF_KERNELS_LIB for A/B runs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd.ops import native
from distributed_tensorflow_amd.ops.native import ptr
lib = native.load()
s = torch.cuda.current_stream().cuda_stream
def bench(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
torch.manual_seed(0)
for (B, F, H) in [(1024, 3136, 1024), (4096, 3136, 1024), (1024, 12544, 4096)]:
    p2 = torch.randn(B, F, device="cuda").bfloat16(); da = torch.randn(B, H, device="cuda").bfloat16()
    W = torch.randn(F, H, device="cuda"); sh = W.bfloat16(); G = torch.empty(F, H, device="cuda")
    hyper = torch.tensor([0.0, 0.9, 0.999, 1e-8, 1.0, 0, 0, 0], device="cuda")
    part = torch.empty(8, B, H, device="cuda")
    t_tn = bench(lambda: lib.dtf_gemm_tn(ptr(p2), ptr(da), ptr(G), F, H, B, F, H, H, 128, s))
    ref = p2.float().t() @ da.float()
    e_tn = float((G - ref).norm() / ref.norm())
    t_sgd = bench(lambda: lib.dtf_gemm_tn_sgd(ptr(p2), ptr(da), ptr(W), ptr(sh), ptr(hyper), F, H, B, F, H, H, s))
    S = lib.dtf_gemm_nn_partial(ptr(p2), ptr(sh), ptr(part), B, H, F, F, H, 0, s)
    torch.cuda.synchronize()
    refn = p2.float() @ sh.float()
    e_nn = float((part[:S].sum(0) - refn).norm() / refn.norm())
    t_nn = bench(lambda: lib.dtf_gemm_nn_partial(ptr(p2), ptr(sh), ptr(part), B, H, F, F, H, 0, s))
    print(f"B{B} F{F} H{H}: tn {t_tn:6.1f} us (err {e_tn:.1e}) | tn+sgd {t_sgd:6.1f} us | nn split {S} {t_nn:6.1f} us (err {e_nn:.1e})", flush=True)
