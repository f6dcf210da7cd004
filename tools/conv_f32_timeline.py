"""Phase timeline of the fp32 conv2 implicit-GEMM kernels (forward / dgrad) from in-kernel clock stamps.
  python tools/conv_f32_timeline.py build          (CPU box: diag_build/conv2_f32_timing.so, -DDTF_F32_TIMING=1)
  python tools/conv_f32_timeline.py run [B]        (GPU)
Stamps per workgroup (100 MHz wall clock), see F32_STAMP in csrc/kernels/conv2_f32.hip.
DTF_F32_WG_PER_CU=1 runs one workgroup (one wave per SIMD) per CU instead of two.

What it showed (1024 images, profiles/r3_f32_igemm_timeline.txt): two workgroups per CU spend 80 us in the k-loops of
two images and 9 us everywhere else.  A lone wave per SIMD needs 44.8 us per image; with the B-fragment loads taken out
of the k-loop 42.4, with one LDS read per step instead of thirteen 40.4, with both 38.2 us (throw-away ablation builds)
-- that last figure is nothing but 2450 MFMAs per SIMD issued back to back, i.e. the matrix pipe at the clock this
kernel gets (78.4 k cycles in 38.2 us = 2.05 GHz).  Two waves per SIMD take 40 us per image: 95 % of that floor.
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "diag_build")
if sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DDTF_F32_TIMING=1", "-I", ROOT + "/csrc/kernels", ROOT + "/csrc/kernels/conv2_f32.hip",
                           "-o", f"{OUT}/conv2_f32_timing.so"])
    print("built")
else:
    import torch
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    C1, C2 = 32, 64
    lib = ctypes.CDLL(f"{OUT}/conv2_f32_timing.so")
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.dtf_conv2_f32_pack_weights.argtypes = [P, P, I, I, P]
    lib.dtf_conv2_f32_fwd.argtypes = [P, P, P, P, P, I, I, I, P]
    lib.dtf_conv2_f32_dgrad.argtypes = [P, P, P, P, I, I, I, P]
    lib.dtf_conv2_f32_set_timing_buf.argtypes = [P]
    x = torch.rand(B, 14, 14, C1, device="cuda")
    w = torch.randn(5, 5, C1, C2, device="cuda") * 0.05
    bias = torch.randn(C2, device="cuda") * 0.1
    packed = torch.empty(2, 25 * C1 * C2, device="cuda")
    pooled = torch.empty(B, 7, 7, C2, device="cuda")
    arg = torch.empty(B, 7, 7, C2, dtype=torch.uint8, device="cuda")
    gp = torch.randn(B, 7, 7, C2, device="cuda")
    dx = torch.empty(B, 14, 14, C1, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    lib.dtf_conv2_f32_pack_weights(w.data_ptr(), packed.data_ptr(), C1, C2, s)
    calls = {
        "forward": lambda: lib.dtf_conv2_f32_fwd(x.data_ptr(), packed[0].data_ptr(), bias.data_ptr(), pooled.data_ptr(), arg.data_ptr(), B, C1, C2, s),
        "dgrad": lambda: lib.dtf_conv2_f32_dgrad(gp.data_ptr(), arg.data_ptr(), packed[1].data_ptr(), dx.data_ptr(), B, C1, C2, s),
    }
    for name, call in calls.items():
        lib.dtf_conv2_f32_set_timing_buf(None)
        for _ in range(5):
            assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record(); torch.cuda.synchronize()
        buf = torch.zeros(2048, 16, dtype=torch.int64, device="cuda")
        assert lib.dtf_conv2_f32_set_timing_buf(buf.data_ptr()) == 0
        call(); torch.cuda.synchronize()
        t = buf.cpu().double()
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        us = (t - t0) / 100.0
        us[t == 0] = float("nan")
        print(f"{name}: B {B}, {t.shape[0]} workgroups, {e0.elapsed_time(e1) * 50:.1f} us per launch (20 launches, untimed build)")
        med = lambda c: float(us[:, c].nanmedian())
        mx = lambda c: float(us[:, c][~us[:, c].isnan()].max())
        print(f"  start        median {med(0):7.2f} max {mx(0):7.2f} us")
        n_img = int(((t[:, 1:] > 0).sum(1).max() + 3) // 4)
        for n in range(n_img):
            c = 1 + 4 * n
            if c + 3 > 15 or not bool((t[:, c + 3] > 0).any()):
                break
            print(f"  image {n}: at barrier {med(c):7.2f} | staged {med(c + 1):7.2f} (+{med(c + 1) - med(c):5.2f}) | k-loop end {med(c + 2):7.2f} "
                  f"(+{med(c + 2) - med(c + 1):6.2f}, max end {mx(c + 2):7.2f}) | stored {med(c + 3):7.2f} (+{med(c + 3) - med(c + 2):5.2f}, max {mx(c + 3):7.2f})")
