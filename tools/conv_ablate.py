"""Ablation timing of the conv2 kernels: builds diagnostic copies of conv2.hip with parts of the
kernel removed (-DDTF_ABLATE bits, see conv2.hip) and times each on the GPU.
  python tools/conv_ablate.py build            (CPU box; writes diag_build/)
  python tools/conv_ablate.py run [wgrad|fwd|dgrad] [G]
"""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "diag_build")
WG = {0: "full", 32: "no x staging", 128: "no dY staging", 160: "no staging at all", 64: "no MFMA",
      256: "no k-loop", 416: "syncs+slab write only", 512: "no slab write", 928: "empty"}
WG_PC = {0: "full", 64: "no MFMA", 1024: "no fragment reads in the k-loop", 1088: "no MFMA, no fragment reads", 160: "no staging at all",
         256: "no k-loop", 1184: "k-loop MFMAs only (no staging, no reads)"}
IG = {0: "full", 1: "no B global loads", 2: "no A LDS reads", 4: "no MFMA", 8: "no staging loads", 16: "no epilogue stores",
      7: "no k-loop work", 23: "staging only", 31: "shell only"}
CFGS = {"fwd": ["32,32,64,1,2,MODE_FWD,4", "32,32,64,2,1,MODE_FWD,2", "32,32,64,1,1,MODE_FWD,2", "32,32,64,4,1,MODE_FWD,2",
                "32,32,64,2,2,MODE_FWD,3", "32,32,64,2,2,MODE_FWD,4"],
        "dgrad": ["64,32,32,1,2,MODE_DGRAD,4", "64,32,32,2,1,MODE_DGRAD,2", "64,64,32,1,1,MODE_DGRAD,2", "64,64,32,2,1,MODE_DGRAD,2",
                  "64,32,32,4,1,MODE_DGRAD,4", "64,32,32,2,1,MODE_DGRAD,4", "64,64,32,1,1,MODE_DGRAD,4"]}
STAGGER = [0, 3, 6, 9]
if sys.argv[1] == "buildcfg":
    os.makedirs(OUT, exist_ok=True)
    for kind, lst in CFGS.items():
        for i, c in enumerate(lst):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                                   f"-DDTF_{kind.upper()}_CFG={c}", "-DDTF_DIAG=1", "-I", ROOT + "/csrc/kernels", ROOT + "/csrc/kernels/conv2.hip",
                                   "-o", f"{OUT}/conv2_{kind}{i}.so"])
    print("built cfgs")
elif sys.argv[1] == "buildstagger":
    os.makedirs(OUT, exist_ok=True)
    mode = os.environ.get("STAGGER_MODE", "0")
    for d in STAGGER:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               f"-DDTF_STAGGER={d}", f"-DDTF_STAGGER_MODE={mode}", "-DDTF_DIAG=1", "-I", ROOT + "/csrc/kernels", ROOT + "/csrc/kernels/conv2.hip",
                               "-o", f"{OUT}/conv2_st{d}.so"])
    print("built stagger")
elif sys.argv[1] == "buildpc":
    os.makedirs(OUT, exist_ok=True)
    for v in sorted(WG_PC):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               f"-DDTF_ABLATE={v}", "-DDTF_DIAG=1", "-I", ROOT + "/csrc/kernels", ROOT + "/csrc/kernels/conv2.hip",
                               "-o", f"{OUT}/conv2_ab{v}.so"])
    print("built", len(WG_PC))
elif sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    for v in sorted(set(WG) | set(IG)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               f"-DDTF_ABLATE={v}", "-DDTF_DIAG=1", "-I", ROOT + "/csrc/kernels", ROOT + "/csrc/kernels/conv2.hip",
                               "-o", f"{OUT}/conv2_ab{v}.so"])
    print("built", len(set(WG) | set(IG)))
else:
    import torch
    which = sys.argv[2] if len(sys.argv) > 2 else "wgrad"
    B = 1024
    G = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    x = torch.rand(B, 14, 14, 32, device="cuda").bfloat16()
    gp = torch.randn(B, 7, 7, 64, device="cuda").bfloat16()
    pooled = torch.rand(B, 7, 7, 64, device="cuda").bfloat16()
    arg = torch.randint(0, 5, (B, 7, 7, 64), device="cuda", dtype=torch.uint8)
    part = torch.empty(G, 25 * 32 * 64 + 64, device="cuda")
    w = (torch.randn(25, 64, 32, device="cuda") * 0.05).bfloat16()
    bias = torch.zeros(64, device="cuda")
    dx = torch.empty(B, 14, 14, 32, device="cuda", dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    P = ctypes.c_void_p
    cfg = which.startswith("cfg")
    stg = which.startswith("st_")
    if stg:
        which = which[3:]
        items = {d: f"second half +{d} us" for d in STAGGER}
    elif cfg:
        which = which[3:]
        items = {i: c for i, c in enumerate(CFGS[which])}
    else:
        items = WG_PC if which == "wgrad_pc" else WG if which == "wgrad" else IG
        which = "wgrad" if which == "wgrad_pc" else which
    for v, name in items.items():
        lib = ctypes.CDLL(f"{OUT}/conv2_st{v}.so" if stg else f"{OUT}/conv2_{which}{v}.so" if cfg else f"{OUT}/conv2_ab{v}.so")
        if which == "wgrad":
            f = lib.dtf_conv2_wgrad; f.argtypes = [P] * 4 + [ctypes.c_int] * 4 + [P]
            call = lambda: f(x.data_ptr(), gp.data_ptr(), arg.data_ptr(), part.data_ptr(), B, 32, 64, G, s)
        elif which == "fwd":
            f = lib.dtf_conv2_fwd; f.argtypes = [P] * 5 + [ctypes.c_int] * 3 + [P]
            call = lambda: f(x.data_ptr(), w.data_ptr(), bias.data_ptr(), pooled.data_ptr(), arg.data_ptr(), B, 32, 64, s)
        else:
            f = lib.dtf_conv2_dgrad; f.argtypes = [P] * 4 + [ctypes.c_int] * 3 + [P]
            call = lambda: f(gp.data_ptr(), arg.data_ptr(), w.data_ptr(), dx.data_ptr(), B, 32, 64, s)
        # time inside a HIP graph (20 launches per replay) so host launch cost does not floor the result
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            s = side.cuda_stream
            for _ in range(5): call()
            side.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                for _ in range(20): call()
            gr.replay(); side.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(5): gr.replay()
            e1.record(side); side.synchronize()
        print(f"{which} ablate={v:3d} {name:22s} {e0.elapsed_time(e1) * 1e3 / 100:7.1f} us", flush=True)
