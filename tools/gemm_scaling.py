"""K- and size-scaling of the NT GEMM (separates fixed cost from the per-k-step slope)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import native
from distributed_tensorflow_amd.ops.native import ptr
lib = native.load()
s = torch.cuda.current_stream().cuda_stream
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
shapes = [(1024, 3136, k) for k in (64, 256, 512, 1024, 2048, 4096)] + \
         [(128, 128, 1024), (1024, 1024, 1024), (2048, 2048, 1024), (4096, 4096, 1024), (4096, 4096, 4096), (8192, 8192, 4096)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    us = bench(lambda: lib.dtf_gemm_nt(ptr(A), ptr(B), ptr(C), None, M, N, K, K, K, N, 0, 1.0, 0, None, 128, s))
    ut = bench(lambda: torch.matmul(A, B.t()))
    print(f"nt M{M} N{N} K{K}: ours {us:7.1f} us {2*M*N*K/us/1e6:6.0f} TF | hipBLASLt {ut:7.1f} us {2*M*N*K/ut/1e6:6.0f} TF", flush=True)
