"""Timing diagnostic for the GEMM kernels (store epilogue on/off, tile sizes)."""
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
for (M, N, K) in [(1024, 1024, 3136), (1024, 3136, 1024)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for tile in (64, 128):
        for epi in (0, 2):
            us = bench(lambda: lib.dtf_gemm_nt(ptr(A), ptr(B), ptr(C), None, M, N, K, K, K, N, epi, 1.0, 0, None, tile, s))
            print(f"nt M{M} N{N} K{K} tile{tile} epi{epi}: {us:.1f} us  {2*M*N*K/us/1e6:.0f} TF")
    us = bench(lambda: torch.matmul(A, B.t()))
    print(f"   torch.matmul (hipBLASLt reference): {us:.1f} us {2*M*N*K/us/1e6:.0f} TF")
M, N, K = 3136, 1024, 1024
A = torch.randn(K, M, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16()
C = torch.empty(M, N, device="cuda")
for tile in (64, 128):
    us = bench(lambda: lib.dtf_gemm_tn(ptr(A), ptr(B), ptr(C), M, N, K, M, N, N, tile, s))
    print(f"tn M{M} N{N} K{K} tile{tile}: {us:.1f} us {2*M*N*K/us/1e6:.0f} TF")
us = bench(lambda: torch.matmul(A.t(), B))
print(f"   torch.matmul: {us:.1f} us {2*M*N*K/us/1e6:.0f} TF")
