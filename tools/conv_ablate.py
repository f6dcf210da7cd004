"""Ablation timing of the conv2 wgrad kernel (edit VARIANTS/call for fwd/dgrad): builds diagnostic copies of conv2.hip with parts
of the kernel removed (-DDTF_ABLATE bits, see conv2.hip) and times each on the GPU.
Build step runs on the CPU box: python tools/conv_ablate.py build ; timing: ... run"""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "diag_build")
VARIANTS = {0: "full", 32: "no x staging", 128: "no dY staging", 160: "no staging at all", 64: "no MFMA",
            256: "no k-loop", 416: "syncs+slab write only", 512: "no slab write", 928: "empty"}
if sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    for v in VARIANTS:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               f"-DDTF_ABLATE={v}", "-I", ROOT + "/csrc/kernels", ROOT + "/csrc/kernels/conv2.hip",
                               "-o", f"{OUT}/conv2_ab{v}.so"])
    print("built", len(VARIANTS))
else:
    import torch
    B = 1024
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    x = torch.rand(B, 14, 14, 32, device="cuda").bfloat16()
    gp = torch.randn(B, 7, 7, 64, device="cuda").bfloat16()
    pooled = torch.rand(B, 7, 7, 64, device="cuda").bfloat16()
    arg = torch.randint(0, 4, (B, 7, 7, 64), device="cuda", dtype=torch.uint8)
    part = torch.empty(G, 25 * 32 * 64 + 64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for v, name in VARIANTS.items():
        lib = ctypes.CDLL(f"{OUT}/conv2_ab{v}.so")
        f = lib.dtf_conv2_wgrad
        f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        call = lambda: f(x.data_ptr(), gp.data_ptr(), arg.data_ptr(), pooled.data_ptr(), part.data_ptr(), B, 32, 64, G, s)
        for _ in range(5): call()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): call()
        torch.cuda.synchronize()
        print(f"ablate={v:2d} {name:22s} {(time.perf_counter()-t0)/50*1e6:7.1f} us")
