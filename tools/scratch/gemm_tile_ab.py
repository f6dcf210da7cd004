"""A/B of the 128x128 and 256x128 workgroup tiles of the dense GEMMs at the fc1 shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd.ops import native
from distributed_tensorflow_amd.ops.native import ptr
lib = native.load()
s = torch.cuda.current_stream().cuda_stream
def bench(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
torch.manual_seed(0)
for (M, N, K) in [(1024, 3136, 1024), (4096, 3136, 1024), (1024, 12544, 4096)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    ref = (A.float() @ B.float().t())
    for tile in (128, 129, 256):
        C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        rc = lib.dtf_gemm_nt(ptr(A), ptr(B), ptr(C), None, M, N, K, K, K, N, 0, 1.0, 0, None, tile, s)
        torch.cuda.synchronize()
        err = float((C.float() - ref).norm() / ref.norm())
        us = bench(lambda: lib.dtf_gemm_nt(ptr(A), ptr(B), ptr(C), None, M, N, K, K, K, N, 0, 1.0, 0, None, tile, s))
        print(f"nt M{M} N{N} K{K} tile {tile}: rc {rc} rel err {err:.2e}  {us:7.1f} us {2*M*N*K/us/1e6:6.0f} TF", flush=True)
    ut = bench(lambda: torch.matmul(A, B.t()))
    print(f"   hipBLASLt {ut:7.1f} us {2*M*N*K/ut/1e6:6.0f} TF", flush=True)
