"""Phase timeline of the conv2 forward implicit-GEMM kernel from in-kernel clock stamps.
  python tools/conv_timeline.py build      (CPU box: diag_build/conv2_timing.so, -DDTF_TIMING=1)
  python tools/conv_timeline.py run        (GPU)
Stamps per workgroup (100 MHz wall clock): 0 start, 1 tile staged (+sync), 2 wave 0 left its k-loop,
3 all waves left the k-loop, 4 epilogue in LDS (+sync), 5 stores issued; 6/7 = shader clock at start/end.
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "diag_build")
if sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DDTF_TIMING=1", "-DDTF_DIAG=1", "-I", ROOT + "/csrc/kernels", ROOT + "/csrc/kernels/conv2.hip",
                           "-o", f"{OUT}/conv2_timing.so"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DDTF_TIMING=1", "-DDTF_DIAG=1", "-I", ROOT + "/csrc/kernels", ROOT + "/csrc/kernels/head.hip",
                           "-o", f"{OUT}/head_timing.so"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DDTF_TIMING=1", "-DDTF_DIAG=1", "-I", ROOT + "/csrc/kernels", ROOT + "/csrc/kernels/gemm.hip",
                           ROOT + "/csrc/kernels/head.hip", "-o", f"{OUT}/gemm_timing.so"])
    print("built")
elif len(sys.argv) > 3 and sys.argv[3] == "gemm":
    import torch
    B = int(sys.argv[2]); F_, H = 3136, 1024
    lib = ctypes.CDLL(f"{OUT}/gemm_timing.so")
    P, I = ctypes.c_void_p, ctypes.c_int
    a2 = (torch.randn(B, F_, device="cuda") * 0.3).bfloat16()       # pooled conv2 output (flat)
    wd1 = (torch.randn(F_, H, device="cuda") * 0.03).bfloat16()
    da = (torch.randn(B, H, device="cuda") * 0.1).bfloat16()
    dflat = torch.empty(B, F_, device="cuda", dtype=torch.bfloat16)
    part = torch.empty(8, B, H, device="cuda")
    W = torch.randn(F_, H, device="cuda"); shadow = torch.empty(F_, H, device="cuda", dtype=torch.bfloat16)
    hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 1.0, 0, 0, 0], device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    lib.dtf_gemm_nn_partial.argtypes = [P, P, P, I, I, I, I, I, I, P]
    lib.dtf_gemm_nt.argtypes = [P, P, P, P, I, I, I, I, I, I, I, ctypes.c_float, ctypes.c_uint64, P, I, P]
    lib.dtf_gemm_tn_sgd.argtypes = [P, P, P, P, P, I, I, I, I, I, I, P]
    lib.dtf_gemm_set_timing_buf.argtypes = [P]
    calls = {
        "nn  (fc1 fwd, split-K)": (lambda: lib.dtf_gemm_nn_partial(a2.data_ptr(), wd1.data_ptr(), part.data_ptr(), B, H, F_, F_, H, 0, s), None),
        "nt  (fc1 dgrad)": (lambda: lib.dtf_gemm_nt(da.data_ptr(), wd1.data_ptr(), dflat.data_ptr(), None, B, F_, H, H, H, F_, 0, 1.0, 0, None, 0, s), None),
        "tn  (fc1 wgrad + SGD)": (lambda: lib.dtf_gemm_tn_sgd(a2.data_ptr(), da.data_ptr(), W.data_ptr(), shadow.data_ptr(), hyper.data_ptr(), F_, H, B, F_, H, H, s), None),
    }
    # the training step's launch: fc1 dgrad tiles followed by the head wgrad workgroups
    KS = 32
    h1 = (torch.rand(B, H, device="cuda")).bfloat16(); dlog = torch.randn(B, 10, device="cuda") / B
    hpart = torch.empty(KS, H * 10 + 12 + H, device="cuda")
    lib.dtf_gemm_nt_head_wgrad.argtypes = [P, P, P, I, I, I, I, I, I, P, P, P, P, I, I, I, P]
    lib.dtf_gemm_set_timing_buf(None)
    fused = lambda: lib.dtf_gemm_nt_head_wgrad(da.data_ptr(), wd1.data_ptr(), dflat.data_ptr(), B, F_, H, H, H, F_, h1.data_ptr(),
                                               dlog.data_ptr(), da.data_ptr(), hpart.data_ptr(), B, H, KS, s)
    for _ in range(5): fused()
    torch.cuda.synchronize()
    buf = torch.zeros(4096, 4, dtype=torch.int64, device="cuda")
    lib.dtf_gemm_set_timing_buf(buf.data_ptr())
    fused(); torch.cuda.synchronize()
    t = buf.cpu().double()
    ntiles = ((B + 127) // 128) * ((F_ + 127) // 128)
    tt, th = t[:ntiles], t[ntiles:ntiles + 16 * KS]
    t0 = min(tt[:, 0].min(), th[:, 0].min())
    print(f"nt + head wgrad in one launch: {ntiles} tiles end median {(tt[:, 3].median() - t0) / 100:5.2f} max {(tt[:, 3].max() - t0) / 100:5.2f} us | "
          f"{th.shape[0]} head workgroups: start median {(th[:, 0].median() - t0) / 100:5.2f} max {(th[:, 0].max() - t0) / 100:5.2f}, "
          f"duration median {((th[:, 3] - th[:, 0]).median()) / 100:5.2f} max {((th[:, 3] - th[:, 0]).max()) / 100:5.2f}, end max {(th[:, 3].max() - t0) / 100:5.2f} us")
    for name, (call, _) in calls.items():
        lib.dtf_gemm_set_timing_buf(None)
        for _ in range(5): rc = call()
        torch.cuda.synchronize()
        buf = torch.zeros(4096, 4, dtype=torch.int64, device="cuda")
        assert lib.dtf_gemm_set_timing_buf(buf.data_ptr()) == 0
        call(); torch.cuda.synchronize()
        t = buf.cpu().double()
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        ns = (t - t0) * 10.0
        d = ns[:, 1:] - ns[:, :-1]
        print(f"{name}: {t.shape[0]} workgroups; start median {ns[:, 0].median() / 1e3:5.2f} max {ns[:, 0].max() / 1e3:5.2f} | "
              f"prologue {d[:, 0].median() / 1e3:5.2f} | main loop {d[:, 1].median() / 1e3:5.2f} (max {d[:, 1].max() / 1e3:5.2f}) | "
              f"epilogue {d[:, 2].median() / 1e3:5.2f} (max {d[:, 2].max() / 1e3:5.2f}) | end median {ns[:, 3].median() / 1e3:5.2f} max {ns[:, 3].max() / 1e3:5.2f} us")
elif len(sys.argv) > 3 and sys.argv[3] == "head":
    import torch
    B = int(sys.argv[2]); H, S = 1024, (int(sys.argv[4]) if len(sys.argv) > 4 else 4)
    lib = ctypes.CDLL(f"{OUT}/head_timing.so")
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    part = torch.randn(S, B, H, device="cuda") * 0.3
    bias1 = torch.randn(H, device="cuda") * 0.1
    h1 = torch.empty(B, H, device="cuda", dtype=torch.bfloat16)
    wT = torch.randn(10, H, device="cuda") * 0.05
    bias = torch.zeros(10, device="cuda")
    labels = torch.randint(0, 10, (B,), device="cuda", dtype=torch.int32)
    logits = torch.empty(B, 10, device="cuda"); lrows = torch.empty(B, device="cuda"); crows = torch.empty(B, device="cuda")
    dlog = torch.empty(B, 10, device="cuda"); da = torch.empty(B, H, device="cuda", dtype=torch.bfloat16)
    state = torch.zeros(8, dtype=torch.int64, device="cuda")
    f = lib.dtf_head_fused_fwd_bwd
    f.argtypes = [P, I, P, F, ctypes.c_uint64, P, P, P, P, P, P, P, P, P, P, P, I, I, F, F, I, P]
    s = torch.cuda.current_stream().cuda_stream
    call = lambda: f(part.data_ptr(), S, bias1.data_ptr(), 1.0, 0, state.data_ptr(), h1.data_ptr(), wT.data_ptr(), bias.data_ptr(),
                     labels.data_ptr(), None, logits.data_ptr(), lrows.data_ptr(), crows.data_ptr(), dlog.data_ptr(),
                     da.data_ptr(), B, H, 1.0 / B, 1.0, 1, s)
    lib.dtf_head_set_timing_buf.argtypes = [P]
    for _ in range(5): assert call() == 0
    torch.cuda.synchronize()
    nblk = min(B, 4096)
    buf = torch.zeros(nblk, 8, dtype=torch.int64, device="cuda")
    assert lib.dtf_head_set_timing_buf(buf.data_ptr()) == 0
    call(); torch.cuda.synchronize()
    t = buf.cpu().double()
    t0 = t[:, 0].min()
    ns = (t[:, :5] - t0) * 10.0
    print(f"head_fwd_bwd (fused fc1 epilogue) B={B}: {nblk} workgroups, first row of each")
    for i, nm in enumerate(["start", "h1 row + partial logits", "logits reduced", "softmax/loss", "da row stored"]):
        c = ns[:, i]
        print(f"  {nm:26s} min {c.min() / 1e3:6.2f}  median {c.median() / 1e3:6.2f}  max {c.max() / 1e3:6.2f} us")
    d = ns[:, 1:] - ns[:, :-1]
    for i, nm in enumerate(["partials -> h1 -> dot", "reductions", "softmax", "backward (da)"]):
        print(f"  phase {nm:24s} median {d[:, i].median() / 1e3:6.2f}  max {d[:, i].max() / 1e3:6.2f} us")
else:
    import torch
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    lib = ctypes.CDLL(os.environ.get("DTF_TIMING_LIB", f"{OUT}/conv2_timing.so"))
    P = ctypes.c_void_p
    x = torch.rand(B, 14, 14, 32, device="cuda").bfloat16()
    w = (torch.randn(2, 25 * 64 * 32, device="cuda") * 0.05).bfloat16()
    bias = torch.zeros(64, device="cuda")
    pooled = torch.empty(B, 7, 7, 64, device="cuda", dtype=torch.bfloat16)
    arg = torch.empty(B, 7, 7, 64, device="cuda", dtype=torch.uint8)
    nblk = (B + 1) // 2
    buf = torch.zeros(nblk, 16, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    f = lib.dtf_conv2_fwd; f.argtypes = [P] * 5 + [ctypes.c_int] * 3 + [P]
    lib.dtf_conv2_set_timing_buf.argtypes = [P]
    for _ in range(5):
        f(x.data_ptr(), w.data_ptr(), bias.data_ptr(), pooled.data_ptr(), arg.data_ptr(), B, 32, 64, s)
    torch.cuda.synchronize()
    assert lib.dtf_conv2_set_timing_buf(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f(x.data_ptr(), w.data_ptr(), bias.data_ptr(), pooled.data_ptr(), arg.data_ptr(), B, 32, 64, s)
    e1.record(); torch.cuda.synchronize()
    if len(sys.argv) > 3 and sys.argv[3] == "wgrad":
        G = 64
        gp = torch.randn(B, 7, 7, 64, device="cuda").bfloat16()
        arg = torch.randint(0, 5, (B, 7, 7, 64), device="cuda", dtype=torch.uint8)
        part = torch.empty(G, 25 * 32 * 64 + 64, device="cuda")
        pc = os.environ.get("DTF_C2W_PC", "1") != "0"
        buf2 = torch.zeros(G * 4, 32 if pc else 16, dtype=torch.int64, device="cuda")   # the pc kernel stamps 32 slots
        fw = lib.dtf_conv2_wgrad; fw.argtypes = [P] * 4 + [ctypes.c_int] * 4 + [P]
        call = lambda: fw(x.data_ptr(), gp.data_ptr(), arg.data_ptr(), part.data_ptr(), B, 32, 64, G, s)
        lib.dtf_conv2_set_timing_buf(None)
        for _ in range(5): call()
        torch.cuda.synchronize()
        assert lib.dtf_conv2_set_timing_buf(buf2.data_ptr()) == 0
        call(); torch.cuda.synchronize()
        full = buf2.cpu().double()
        t = full[:, :16].reshape(G * 4, 2, 8)
        t0 = t[:, 0, 0].min()
        if pc:
            left, wait = (full[:, 16:24] - t0) * 10.0 / 1e3, full[:, 24:32] * 10.0 / 1e3
            print("MFMA waves 0..7, loop left (median us): " + " ".join(f"{left[:, w].median():6.2f}" for w in range(8)))
            print("                 summed wait (median) : " + " ".join(f"{wait[:, w].median():6.2f}" for w in range(8)))
            print(f"                 last wave of a workgroup leaves at median {left.max(dim=1).values.median():.2f} us, first at {left.min(dim=1).values.median():.2f}")
        for wv, name in ((0, "wave 0 (MFMA wave)"), (1, "wave 8 (producer)")) if pc else ((0, "wave 0 (stages x + dY)"), (1, "wave 7 (stages x only)")):
            a = t[:, wv, :]
            print(f"wgrad B={B} {name}: {G * 4} workgroups, {B // G} images each")
            for i, n in enumerate(["start", "loop entered", "loop left", "reduced"]):
                c = (a[:, i] - t0) * 10.0
                print(f"  {n:14s} min {c.min() / 1e3:6.2f}  median {c.median() / 1e3:6.2f}  max {c.max() / 1e3:6.2f} us")
            if pc and wv == 0:     # slots 4 / 6 of the MFMA wave: shader-clock cycles at loop entry / exit
                mhz = (a[:, 6] - a[:, 4]) / ((a[:, 2] - a[:, 1]) * 10.0) * 1e3
                print(f"  shader clock over the loop: median {mhz.median():.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f})")
                for i, n in ((5, "sum k-loop"), (7, "sum barrier wait")):
                    c = a[:, i] * 10.0
                    print(f"  {n:26s} median {c.median() / 1e3:6.2f}  max {c.max() / 1e3:6.2f} us")
                continue
            for i, n in ((5, "sum k-loop"), (4, "sum wait for prefetch"), (6, "sum stage+prefetch issue"), (7, "sum barrier wait")):
                c = a[:, i] * 10.0
                print(f"  {n:26s} median {c.median() / 1e3:6.2f}  max {c.max() / 1e3:6.2f} us")
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "fused_dgrad":
        n = 60000
        images = torch.rand(n, 784, device="cuda").bfloat16()
        idx = torch.randint(0, n, (B,), device="cuda", dtype=torch.int32)
        gp = torch.randn(B, 7, 7, 64, device="cuda").bfloat16()
        arg2 = torch.randint(0, 5, (B, 7, 7, 64), device="cuda", dtype=torch.uint8)
        arg1 = torch.randint(0, 5, (B, 14, 14, 32), device="cuda", dtype=torch.uint8)
        part1 = torch.empty((B + 1) // 2, 26 * 32, device="cuda")
        fd = lib.dtf_conv2_dgrad_conv1_wgrad
        fd.argtypes = [P] * 7 + [ctypes.c_int, P]
        call = lambda: fd(gp.data_ptr(), arg2.data_ptr(), w[1].data_ptr(), images.data_ptr(), idx.data_ptr(),
                          arg1.data_ptr(), part1.data_ptr(), B, s)
        lib.dtf_conv2_set_timing_buf(None)
        for _ in range(5): assert call() == 0
        torch.cuda.synchronize()
        buf.zero_()
        assert lib.dtf_conv2_set_timing_buf(buf.data_ptr()) == 0
        call(); torch.cuda.synchronize()
        t = buf.cpu().double()
        t0 = t[:, 0].min()
        order = [0, 1, 2, 3, 4, 5]
        names = ["start", "gradient tile staged", "wave0 k-loop end", "all k-loops end", "dx in LDS + c1w loads", "conv1 wgrad slab written"]
        ns = (t[:, order] - t0) * 10.0
        print(f"fused conv2 dgrad + conv1 wgrad B={B}: {nblk} workgroups")
        for i, nm in enumerate(names):
            c = ns[:, i]
            print(f"  {nm:26s} min {c.min() / 1e3:6.2f}  median {c.median() / 1e3:6.2f}  max {c.max() / 1e3:6.2f} us")
        d = ns[:, 1:] - ns[:, :-1]
        for i, nm in enumerate(["staging (pool/ReLU grad)", "k-loop (wave 0)", "k-loop skew", "dx -> LDS, zero fill", "conv1 wgrad (2 images)"]):
            print(f"  phase {nm:26s} median {d[:, i].median() / 1e3:6.2f}  max {d[:, i].max() / 1e3:6.2f}   first half {d[: nblk // 2, i].median() / 1e3:5.2f}  second half {d[nblk // 2:, i].median() / 1e3:5.2f} us")
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "fused":
        n = 60000
        images = torch.rand(n, 784, device="cuda").bfloat16()
        state = torch.zeros(8, dtype=torch.int64, device="cuda")
        idx_out = torch.zeros(B, dtype=torch.int32, device="cuda")
        w1 = torch.randn(25, 32, device="cuda") * 0.1
        b1 = torch.full((32,), 0.1, device="cuda")
        p1 = torch.empty(B, 14, 14, 32, device="cuda", dtype=torch.bfloat16)
        arg1 = torch.empty(B, 14, 14, 32, device="cuda", dtype=torch.uint8)
        ff = lib.dtf_conv12_fwd
        U, I = ctypes.c_uint, ctypes.c_int
        ff.argtypes = [P, P, P, P, U, U, U, I, P, P, P, P, P, P, P, P, I, P]
        call = lambda: ff(images.data_ptr(), None, idx_out.data_ptr(), state.data_ptr(), n, 8, 1234, 1, w1.data_ptr(), b1.data_ptr(),
                          p1.data_ptr(), arg1.data_ptr(), w.data_ptr(), bias.data_ptr(), pooled.data_ptr(), arg.data_ptr(), B, s)
        lib.dtf_conv2_set_timing_buf(None)
        for _ in range(5): assert call() == 0
        torch.cuda.synchronize()
        buf.zero_()
        assert lib.dtf_conv2_set_timing_buf(buf.data_ptr()) == 0
        call(); torch.cuda.synchronize()
        t = buf.cpu().double()
        t0 = t[:, 0].min()
        order = [0, 6, 1, 2, 3, 4, 5]
        names = ["start", "images in LDS", "conv1 done, tile staged", "wave0 k-loop end", "all k-loops end", "epilogue in LDS", "stores issued"]
        ns = (t[:, order] - t0) * 10.0
        print(f"fused conv1+conv2 forward B={B}: {nblk} workgroups")
        for i, nm in enumerate(names):
            c = ns[:, i]
            print(f"  {nm:24s} min {c.min() / 1e3:6.2f}  median {c.median() / 1e3:6.2f}  max {c.max() / 1e3:6.2f} us")
        d = ns[:, 1:] - ns[:, :-1]
        for i, nm in enumerate(["zero fill + image load", "conv1 rows", "k-loop (wave 0)", "k-loop skew", "epilogue math", "stores"]):
            print(f"  phase {nm:24s} median {d[:, i].median() / 1e3:6.2f}  max {d[:, i].max() / 1e3:6.2f} us")
        rows = t[:, 8:10] * 10.0 / 1e3 / 4.0
        print(f"  wave 0, per pooled row  : LDS reads + MFMAs + pooling epilogue {rows[:, 0].median():5.2f} us, copy-out {rows[:, 1].median():5.2f} us"
              f"   (first half {rows[: nblk // 2, 0].median():5.2f} / {rows[: nblk // 2, 1].median():5.2f}, second half {rows[nblk // 2:, 0].median():5.2f} / {rows[nblk // 2:, 1].median():5.2f})")
        # who is slow?  by XCD (blockIdx % 8), by dispatch half, and the deciles of the conv1 phase
        c1 = d[:, 1] / 1e3
        bid = torch.arange(nblk)
        print("  conv1 phase by XCD      :", " ".join(f"{c1[bid % 8 == x].median():5.2f}" for x in range(8)))
        print("  conv1 phase by half     :", f"{c1[: nblk // 2].median():5.2f} {c1[nblk // 2:].median():5.2f}")
        qs = torch.quantile(c1, torch.linspace(0, 1, 11, dtype=torch.double))
        print("  conv1 phase deciles     :", " ".join(f"{q:5.2f}" for q in qs))
        end = ns[:, 6] / 1e3
        qs = torch.quantile(end, torch.linspace(0, 1, 11, dtype=torch.double))
        print("  workgroup end deciles   :", " ".join(f"{q:5.2f}" for q in qs))
        print("  end by XCD (median/max) :", " ".join(f"{end[bid % 8 == x].median():5.1f}/{end[bid % 8 == x].max():4.1f}" for x in range(8)))
        sys.exit(0)
    t = buf.cpu().double()
    t0 = t[:, 0].min()
    ns = (t[:, :6] - t0) * 10.0     # 100 MHz -> ns
    names = ["start", "staged", "wave0 k-loop end", "all k-loops end", "epilogue in LDS", "stores issued"]
    print(f"B={B}: {nblk} workgroups, event time {e0.elapsed_time(e1) * 1e3:.1f} us")
    for i, n in enumerate(names):
        c = ns[:, i]
        print(f"  {n:18s} min {c.min() / 1e3:6.2f}  median {c.median() / 1e3:6.2f}  max {c.max() / 1e3:6.2f} us")
    d = ns[:, 1:] - ns[:, :-1]
    for i, n in enumerate(["staging", "k-loop (wave 0)", "k-loop skew", "epilogue math", "stores"]):
        print(f"  phase {n:16s} median {d[:, i].median() / 1e3:6.2f}  max {d[:, i].max() / 1e3:6.2f} us")
    mhz = ((t[:, 7] - t[:, 6]) / ((t[:, 5] - t[:, 0]) * 10.0) * 1e3)
    print(f"  shader clock over the workgroup lifetime: median {mhz.median():.0f} MHz")
