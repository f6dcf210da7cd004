"""Numerics of the exact-fp32 kernel set (csrc/kernels/*_f32.hip, --dtype fp32) against plain PyTorch
references of the same ops.  The f32-input MFMA is bit-for-bit an fmaf chain, so these kernels are held to
fp32 rounding: <= 1e-5 relative (norm-wise) against an fp64 evaluation of the same fp32 inputs -- the same bound
the PyTorch fp32 op itself meets.  The reference precision is float32 (reference .idea/MNISTDist.py:113-115).

Run on the GPU box: python -m pytest tests -m gpu
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from distributed_tensorflow_amd.ops import reference as R
from distributed_tensorflow_amd.ops.native import check, ptr

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-5


def stream():
    return torch.cuda.current_stream().cuda_stream


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def conv_block_ref64(x_nhwc, w_hwio, b):
    """fp64 conv + bias + relu, then 2x2 pool with argmax (first max wins) of the fp32 inputs."""
    y = R.conv2d_same(x_nhwc.double(), w_hwio.double(), b.double())
    return R.pool_argmax(y)


def dead_code(arg, pooled):
    return torch.where(pooled > 0, arg, torch.full_like(arg, 4)).contiguous()


# ----------------------------------------------------------------------- GEMMs
@pytest.mark.parametrize("M,N,K,S", [(1024, 1024, 3136, 0), (100, 1024, 3136, 0), (257, 384, 192, 3), (64, 128, 64, 1),
                                     (1, 1024, 3136, 0), (100, 1024, 3136, 32), (100, 1024, 3136, 99)])
def test_gemm_f32_nn_partial(gpu_lib, M, N, K, S):
    g = torch.Generator(device="cpu").manual_seed(15)
    # asymmetric operands so a transposed / garbled fragment map cannot pass
    A = torch.randn(M, K, generator=g).to(DEV)
    Bm = (torch.randn(K, N, generator=g) * 0.05 + torch.arange(N).float()[None, :] * 1e-3).to(DEV)
    part = torch.full((32, M, N), float("nan"), device=DEV)         # the wrapper never uses more than 32 splits
    S_used = gpu_lib.dtf_gemm_f32_nn_partial(ptr(A), ptr(Bm), ptr(part), M, N, K, K, N, S, stream())
    assert 0 < S_used <= 32
    got = part[:S_used].sum(0)
    assert bool(torch.isfinite(got).all())
    assert rel_err(got, A.double() @ Bm.double()) < TOL


@pytest.fixture
def tiling(gpu_lib, request):
    """Both tilings of the NT / TN products on every shape: 128 x 128 (0) and the 112-wide tiles (1); the library picks
    by shape otherwise (-1, restored afterwards)."""
    gpu_lib.dtf_gemm_f32_force_t112(request.param)
    yield request.param
    gpu_lib.dtf_gemm_f32_force_t112(-1)


@pytest.mark.parametrize("tiling", [0, 1, -1], indirect=True)
@pytest.mark.parametrize("M,N,K", [(1024, 3136, 1024), (100, 3136, 1024), (257, 192, 64), (64, 64, 128), (5, 130, 32),
                                   (300, 230, 96)])
def test_gemm_f32_nt(gpu_lib, M, N, K, tiling):
    g = torch.Generator(device="cpu").manual_seed(5)
    A = torch.randn(M, K, generator=g).to(DEV)
    Bm = (torch.randn(N, K, generator=g) * 0.05 + torch.arange(N).float()[:, None] * 1e-3).to(DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    check(gpu_lib.dtf_gemm_f32_nt(ptr(A), ptr(Bm), ptr(C), M, N, K, K, K, N, stream()), "gemm_f32_nt")
    assert bool(torch.isfinite(C).all())
    assert rel_err(C, A.double() @ Bm.double().t()) < TOL


@pytest.mark.parametrize("M,N,K,S", [(100, 3136, 1024, 8), (100, 3136, 1024, 3), (37, 260, 96, 8), (128, 128, 64, 2),
                                     (100, 3136, 1024, 1)])
def test_gemm_f32_nt_splitk(gpu_lib, M, N, K, S):
    """Small-batch fc1 dgrad: K ranges into partial slabs + slab reduction == the plain product."""
    g = torch.Generator(device="cpu").manual_seed(6)
    A = torch.randn(M, K, generator=g).to(DEV)
    Bm = (torch.randn(N, K, generator=g) * 0.05 + torch.arange(N).float()[:, None] * 1e-3).to(DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    part = torch.full((S, M, N), float("nan"), device=DEV)
    check(gpu_lib.dtf_gemm_f32_nt_splitk(ptr(A), ptr(Bm), ptr(part), ptr(C), M, N, K, K, K, N, S, stream()),
          "gemm_f32_nt_splitk")
    assert bool(torch.isfinite(C).all())
    assert rel_err(C, A.double() @ Bm.double().t()) < TOL
    # a C that is not dense cannot be the target of the slab reduction
    if S > 1:
        assert gpu_lib.dtf_gemm_f32_nt_splitk(ptr(A), ptr(Bm), ptr(part), ptr(C), M, N, K, K, K, N + 4, S, stream()) == -3


@pytest.mark.parametrize("tiling", [0, 1, -1], indirect=True)
@pytest.mark.parametrize("M,N,K", [(3136, 1024, 100), (3136, 1024, 256), (200, 136, 1000), (128, 128, 64), (36, 260, 7),
                                   (236, 300, 33)])
def test_gemm_f32_tn_and_fused_sgd(gpu_lib, M, N, K, tiling):
    g = torch.Generator(device="cpu").manual_seed(7)
    A = (torch.randn(K, M, generator=g) + torch.arange(M).float()[None, :] * 1e-3).to(DEV)
    Bm = torch.randn(K, N, generator=g).to(DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    check(gpu_lib.dtf_gemm_f32_tn(ptr(A), ptr(Bm), ptr(C), M, N, K, M, N, N, None, stream()), "gemm_f32_tn")
    ref = A.double().t() @ Bm.double()
    assert rel_err(C, ref) < TOL
    # fused SGD epilogue: W -= lr * grad_scale * (A^T B), in place, bitwise fmaf(-lr, g, w) of the stored gradient
    hyper = torch.tensor([0.05, 0.9, 0.999, 1e-8, 0.5, 0.0, 0.0, 0.0], device=DEV)
    W0 = torch.randn(M, N, generator=g).to(DEV)
    W = W0.clone()
    check(gpu_lib.dtf_gemm_f32_tn(ptr(A), ptr(Bm), ptr(W), M, N, K, M, N, N, ptr(hyper), stream()), "gemm_f32_tn sgd")
    lr = float(hyper[0] * hyper[4])
    want = torch.addcmul(W0.double(), C.double(), torch.tensor(-lr, dtype=torch.float64, device=DEV)).float()
    assert float((W - want).abs().max()) <= 2e-7 * float(W0.abs().max() + lr * C.abs().max())


@pytest.mark.parametrize("K", [1024, 2048, 100])
def test_gemm_f32_tn_stream_k(gpu_lib, K):
    """fc1 wgrad as stream-K (gemm_f32_sk): 200 tiles x K/32 iterations cut into 256 equal ranges, partial tiles handed
    over through the uncached workspace.  Against fp64; two runs bitwise equal (fixed summation order); the flags are
    back to zero (graph replay needs no per-launch argument); the fused SGD epilogue; K = 100 falls back to the plain
    launch (the iterations do not divide)."""
    M, N = 3136, 1024
    g = torch.Generator(device="cpu").manual_seed(17)
    A = (torch.randn(K, M, generator=g) + torch.arange(M).float()[None, :] * 1e-3).to(DEV)
    Bm = torch.randn(K, N, generator=g).to(DEV)
    ws = gpu_lib.dtf_gemm_f32_sk_workspace_alloc()
    assert ws
    try:
        ref = A.double().t() @ Bm.double()
        outs = []
        for _ in range(2):
            C = torch.full((M, N), float("nan"), device=DEV)
            check(gpu_lib.dtf_gemm_f32_tn_sk(ptr(A), ptr(Bm), ptr(C), M, N, K, M, N, N, None, ws, None, stream()), "tn_sk")
            torch.cuda.synchronize()
            assert bool(torch.isfinite(C).all())
            assert rel_err(C, ref) < TOL
            outs.append(C)
        assert torch.equal(outs[0], outs[1])
        plain = torch.empty(M, N, device=DEV)
        check(gpu_lib.dtf_gemm_f32_tn(ptr(A), ptr(Bm), ptr(plain), M, N, K, M, N, N, None, stream()), "tn")
        if K % 1024:
            assert torch.equal(plain, outs[0])          # the fallback IS the plain kernel
        hyper = torch.tensor([0.05, 0.9, 0.999, 1e-8, 0.5, 0.0, 0.0, 0.0], device=DEV)
        W0 = torch.randn(M, N, generator=g).to(DEV)
        W = W0.clone()
        check(gpu_lib.dtf_gemm_f32_tn_sk(ptr(A), ptr(Bm), ptr(W), M, N, K, M, N, N, ptr(hyper), ws, None, stream()), "tn_sk sgd")
        lr = float(hyper[0] * hyper[4])
        want = torch.addcmul(W0.double(), outs[0].double(), torch.tensor(-lr, dtype=torch.float64, device=DEV)).float()
        assert float((W - want).abs().max()) <= 2e-7 * float(W0.abs().max() + lr * outs[0].abs().max())
        import ctypes
        flags = (ctypes.c_uint * 256)()
        torch.cuda.synchronize()
        from distributed_tensorflow_amd.ops import native as _n
        rt = ctypes.CDLL("libamdhip64.so")
        rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        assert rt.hipMemcpy(flags, ctypes.c_void_p(ws + 256 * 128 * 128 * 4), 1024, 2) == 0
        assert not any(flags)
    finally:
        gpu_lib.dtf_gemm_f32_sk_workspace_free(ws)


# ----------------------------------------------------------------------- conv1
@pytest.mark.parametrize("B,C", [(1, 32), (37, 32), (8, 128)])
def test_conv1_f32_fwd(gpu_lib, B, C):
    g = torch.Generator(device="cpu").manual_seed(1)
    N = 64
    images = torch.rand(N, 784, generator=g).to(DEV)
    idx = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(DEV)
    w = (torch.randn(5, 5, 1, C, generator=g) * 0.1).to(DEV)
    b = (torch.randn(C, generator=g) * 0.1).to(DEV)
    pooled = torch.full((B, 14, 14, C), float("nan"), device=DEV)
    arg = torch.full((B, 14, 14, C), 9, dtype=torch.uint8, device=DEV)
    check(gpu_lib.dtf_conv1_f32_fwd(ptr(images), ptr(idx), ptr(w), ptr(b), ptr(pooled), ptr(arg), B, C, 0, stream()),
          "conv1_f32_fwd")
    x = images[idx.long()].reshape(B, 28, 28, 1)
    ref, ref_arg = conv_block_ref64(x, w, b)
    assert rel_err(pooled, ref) < TOL
    live = ref > 1e-3
    assert float((arg == ref_arg)[live].float().mean()) > 0.9995      # ties / near-ties only
    assert bool(((arg == 4) == (pooled <= 0)).all())


def test_conv1_f32_fwd_fused_sampler(gpu_lib):
    from distributed_tensorflow_amd.data import half_bits
    g = torch.Generator(device="cpu").manual_seed(11)
    N, B, C, seed = 500, 96, 32, 5
    images = torch.rand(N, 784, generator=g).to(DEV)
    w = (torch.randn(5, 5, 1, C, generator=g) * 0.1).to(DEV)
    b = (torch.randn(C, generator=g) * 0.1).to(DEV)
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    for cursor in (0, 450, 7 * N + 123):
        state[0] = cursor
        idx = torch.zeros(B, dtype=torch.int32, device=DEV)
        check(gpu_lib.dtf_sample_indices(ptr(state), ptr(idx), B, N, half_bits(N), seed, 1, stream()), "sample")
        p0 = torch.empty(B, 14, 14, C, device=DEV)
        a0 = torch.empty(B, 14, 14, C, dtype=torch.uint8, device=DEV)
        check(gpu_lib.dtf_conv1_f32_fwd(ptr(images), ptr(idx), ptr(w), ptr(b), ptr(p0), ptr(a0), B, C, 0, stream()), "c1")
        idx1 = torch.full((B,), -1, dtype=torch.int32, device=DEV)
        p1, a1 = torch.empty_like(p0), torch.empty_like(a0)
        check(gpu_lib.dtf_conv1_f32_fwd_sampled(ptr(images), ptr(idx1), ptr(state), N, half_bits(N), seed, 1, ptr(w), ptr(b),
                                                ptr(p1), ptr(a1), B, C, 0, stream()), "c1 sampled")
        assert torch.equal(idx, idx1)
        assert torch.equal(p0, p1) and torch.equal(a0, a1)


@pytest.mark.parametrize("B,C,G", [(33, 32, 7), (6, 128, 6), (1, 32, 1), (40, 32, 64)])
def test_conv1_f32_wgrad(gpu_lib, B, C, G):
    g = torch.Generator(device="cpu").manual_seed(2)
    images = torch.rand(B, 784, generator=g).to(DEV)
    w = (torch.randn(5, 5, 1, C, generator=g) * 0.1).double().to(DEV).requires_grad_(True)
    b = (torch.randn(C, generator=g) * 0.1).double().to(DEV).requires_grad_(True)
    x = images.double().reshape(B, 28, 28, 1)
    y = R.maxpool2d(R.conv2d_same(x, w, b))
    gp = torch.randn(y.shape, generator=g).to(DEV)
    y.backward(gp.double())
    pooled, arg = conv_block_ref64(x, w.detach(), b.detach())
    code = dead_code(arg, pooled)
    part = torch.full((G, 26 * C), float("nan"), device=DEV)
    check(gpu_lib.dtf_conv1_f32_wgrad(ptr(images), None, ptr(gp), ptr(code), ptr(part), B, C, G, stream()), "conv1_f32_wgrad")
    got = part.sum(0)
    assert bool(torch.isfinite(got).all())
    assert rel_err(got[:25 * C].reshape(5, 5, 1, C), w.grad) < TOL
    assert rel_err(got[25 * C:], b.grad) < TOL


# ----------------------------------------------------------------------- conv2
pack_ref_f32 = R.conv2_pack_reference_f32


def _conv2_inputs(B, C1, C2, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand(B, 14, 14, C1, generator=g).to(DEV)
    w = (torch.randn(5, 5, C1, C2, generator=g) * (1.0 / (25 * C1) ** 0.5)).to(DEV)
    b = (torch.randn(C2, generator=g) * 0.1).to(DEV)
    return g, x, w, b


# B * (C2 / 64) workgroups against 256 CUs picks the M ranges per image: 3 (<= 85), 2 (<= 128), 1 (see conv2_f32.hip)
@pytest.mark.parametrize("B,C1,C2", [(1, 32, 64), (5, 32, 64), (64, 32, 64), (3, 128, 256), (100, 32, 64), (260, 32, 64),
                                     (30, 128, 256), (70, 128, 256)])
def test_conv2_f32_fwd(gpu_lib, B, C1, C2):
    g, x, w, b = _conv2_inputs(B, C1, C2, 3)
    packed = torch.empty(2, 25 * C1 * C2, device=DEV)
    check(gpu_lib.dtf_conv2_f32_pack_weights(ptr(w), ptr(packed), C1, C2, stream()), "pack")
    f_ref, d_ref = pack_ref_f32(w.reshape(25, C1, C2))
    assert torch.equal(packed[0], f_ref) and torch.equal(packed[1], d_ref)
    pooled = torch.full((B, 7, 7, C2), float("nan"), device=DEV)
    arg = torch.full((B, 7, 7, C2), 9, dtype=torch.uint8, device=DEV)
    check(gpu_lib.dtf_conv2_f32_fwd(ptr(x), ptr(packed[0]), ptr(b), ptr(pooled), ptr(arg), B, C1, C2, stream()), "conv2_f32_fwd")
    ref, ref_arg = conv_block_ref64(x, w, b)
    assert bool(torch.isfinite(pooled).all())
    assert rel_err(pooled, ref) < TOL
    live = ref > 1e-2
    assert float((arg == ref_arg)[live].float().mean()) > 0.9995
    assert bool(((arg == 4) == (pooled <= 0)).all())


# likewise for dgrad (M ranges) and wgrad (G * tiles <= 128: two tap ranges)
@pytest.mark.parametrize("B,C1,C2,G", [(1, 32, 64, 1), (5, 32, 64, 3), (32, 32, 64, 8), (3, 128, 256, 2), (9, 32, 64, 16),
                                       (100, 32, 64, 100), (260, 32, 64, 200), (30, 128, 256, 16), (70, 128, 256, 4)])
def test_conv2_f32_backward(gpu_lib, B, C1, C2, G):
    g, x, w, b = _conv2_inputs(B, C1, C2, 4)
    xf = x.double().requires_grad_(True)
    wf = w.double().requires_grad_(True)
    bfp = b.double().requires_grad_(True)
    y = R.maxpool2d(R.conv2d_same(xf, wf, bfp))
    gp = torch.randn(y.shape, generator=g).to(DEV)
    y.backward(gp.double())
    pooled, arg = conv_block_ref64(x, w, b)
    arg = dead_code(arg, pooled)
    packed = torch.empty(2, 25 * C1 * C2, device=DEV)
    check(gpu_lib.dtf_conv2_f32_pack_weights(ptr(w), ptr(packed), C1, C2, stream()), "pack")
    dx = torch.full((B, 14, 14, C1), float("nan"), device=DEV)
    check(gpu_lib.dtf_conv2_f32_dgrad(ptr(gp), ptr(arg), ptr(packed[1]), ptr(dx), B, C1, C2, stream()), "conv2_f32_dgrad")
    assert bool(torch.isfinite(dx).all())
    assert rel_err(dx, xf.grad) < TOL
    stride = 25 * C1 * C2 + C2
    part = torch.full((G, stride), float("nan"), device=DEV)
    check(gpu_lib.dtf_conv2_f32_wgrad(ptr(x), ptr(gp), ptr(arg), ptr(part), B, C1, C2, G, stream()), "conv2_f32_wgrad")
    got = part.sum(0)
    assert bool(torch.isfinite(got).all())
    assert rel_err(got[:25 * C1 * C2].reshape(5, 5, C1, C2), wf.grad) < TOL
    assert rel_err(got[25 * C1 * C2:], bfp.grad) < TOL


# ------------------------------------------------------------------------ head
@pytest.mark.parametrize("B,H", [(1, 1024), (100, 1024), (37, 4096)])
def test_head_f32(gpu_lib, B, H):
    g = torch.Generator(device="cpu").manual_seed(8)
    h1 = F.relu(torch.randn(B, H, generator=g)).to(DEV)
    w = (torch.randn(H, 10, generator=g) * 0.1).double().to(DEV).requires_grad_(True)
    b = (torch.randn(10, generator=g) * 0.1).double().to(DEV).requires_grad_(True)
    labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32).to(DEV)
    keep = 0.5
    hf = h1.double().requires_grad_(True)
    z = hf @ w + b
    loss_rows = -F.log_softmax(z, -1).gather(1, labels.long().unsqueeze(1)).squeeze(1)
    loss_rows.mean().backward()

    logits, lrows, crows = torch.empty(B, 10, device=DEV), torch.empty(B, device=DEV), torch.empty(B, device=DEV)
    dlog, da = torch.empty(B, 10, device=DEV), torch.empty(B, H, device=DEV)
    w32, b32 = w.detach().float(), b.detach().float()
    wT = torch.empty(10, H, device=DEV)
    check(gpu_lib.dtf_transpose_head_w(ptr(w32), ptr(wT), H, stream()), "transpose_head_w")
    check(gpu_lib.dtf_head_f32_fwd_bwd(ptr(h1), ptr(wT), ptr(b32), ptr(labels), None, ptr(logits), ptr(lrows), ptr(crows),
                                       ptr(dlog), ptr(da), B, H, 1.0 / B, 1.0 / keep, 1, stream()), "head_f32")
    assert rel_err(logits, z) < TOL
    assert rel_err(lrows, loss_rows) < TOL
    assert float((crows - (z.argmax(-1) == labels.long()).float()).abs().sum()) <= 1     # argmax near-ties
    dz = (torch.softmax(z.detach(), -1) - F.one_hot(labels.long(), 10).double()) / B
    assert rel_err(dlog, dz) < TOL
    want_da = torch.where(h1 > 0, hf.grad / keep, torch.zeros_like(hf.grad))
    assert rel_err(da, want_da) < TOL
    KS = 3
    stride = H * 10 + 12 + H
    part = torch.empty(KS, stride, device=DEV)
    check(gpu_lib.dtf_head_f32_wgrad(ptr(h1), ptr(dlog), ptr(da), ptr(part), B, H, KS, stream()), "head_f32_wgrad")
    got = part.sum(0)
    assert rel_err(got[:H * 10].reshape(H, 10), w.grad) < TOL
    assert rel_err(got[H * 10:H * 10 + 10], b.grad) < TOL
    assert rel_err(got[H * 10 + 12:], da.double().sum(0)) < TOL


@pytest.mark.parametrize("B,KS", [(1024, 16), (1000, 7), (640, 10), (100, 8)])
def test_fc1_dgrad_f32_with_head_wgrad_workgroups_matches_the_two_kernels(gpu_lib, B, KS):
    """One launch = the fp32 fc1 dgrad GEMM tiles + one extra grid layer of output-layer wgrad workgroups (gemm_f32
    HEADW).  The GEMM result is bitwise the plain kernel's, and so is the reduction: the stand-alone fp32 kernel runs
    the same 8 waves per workgroup, so the two routes sum in one order (the two-rank replay test relies on it).  B = 100 is the small-batch split-K product
    (25 tiles x 8 K ranges + two guest layers, partial products summed by the slab reduction)."""
    H, F_ = 1024, 3136
    g = torch.Generator(device="cpu").manual_seed(61)
    da = (torch.randn(B, H, generator=g) * (torch.rand(B, H, generator=g) > 0.5)).to(DEV)
    w = (torch.randn(F_, H, generator=g) * 0.03).to(DEV)
    h1 = torch.randn(B, H, generator=g).clamp_min(0).to(DEV)
    dlog = (torch.randn(B, 10, generator=g) / B).to(DEV)
    stride = H * 10 + 12 + H
    C0 = torch.empty(B, F_, device=DEV)
    C1 = torch.full((B, F_), 7.0, device=DEV)
    part0 = torch.empty(KS, stride, device=DEV)
    part1 = torch.full((KS, stride), 3.0, device=DEV)
    check(gpu_lib.dtf_gemm_f32_nt(ptr(da), ptr(w), ptr(C0), B, F_, H, H, H, F_, stream()), "nt")
    check(gpu_lib.dtf_head_f32_wgrad(ptr(h1), ptr(dlog), ptr(da), ptr(part0), B, H, KS, stream()), "head_wgrad")
    S = 8 if B <= 128 else 1               # the engine's split-K of a small batch (partials summed by the slab reduction)
    skp = torch.empty(S, B, F_, device=DEV) if S > 1 else None
    check(gpu_lib.dtf_gemm_f32_nt_head_wgrad(ptr(da), ptr(w), ptr(skp) if S > 1 else None, ptr(C1), B, F_, H, H, H, F_, S,
                                             ptr(h1), ptr(dlog), ptr(da), ptr(part1), B, H, KS, stream()), "nt+head_wgrad")
    torch.cuda.synchronize()
    if S == 1:
        assert torch.equal(C0, C1)
    assert rel_err(C1, (da.double() @ w.double().t())) < 1e-5
    assert torch.equal(part1, part0)
    got = part1.double().sum(0)
    assert rel_err(got[:H * 10].reshape(H, 10), h1.double().t() @ dlog.double()) < 1e-5
    assert rel_err(got[H * 10:H * 10 + 10], dlog.double().sum(0)) < 1e-5
    assert rel_err(got[H * 10 + 12:], da.double().sum(0)) < 1e-5


@pytest.mark.parametrize("keep", [1.0, 0.6])
def test_head_f32_fused_fc1_epilogue(gpu_lib, keep):
    """split-K partials -> (bias, ReLU, dropout, h1, logits, loss, dlogits, da) in one launch, fp32 activations:
    h1 against an fp64 evaluation, dropout keeps the right fraction and scales by 1/keep, and da is zero exactly
    where h1 is."""
    M, N, K = 200, 1024, 3136
    g = torch.Generator(device="cpu").manual_seed(21)
    A = torch.randn(M, K, generator=g).to(DEV)
    Bm = (torch.randn(K, N, generator=g) * 0.03).to(DEV)
    bias1 = torch.randn(N, generator=g).to(DEV)
    wT = (torch.randn(10, N, generator=g) * 0.1).to(DEV)
    b = (torch.randn(10, generator=g) * 0.1).to(DEV)
    labels = torch.randint(0, 10, (M,), generator=g, dtype=torch.int32).to(DEV)
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    state[2] = 5
    part = torch.empty(8, M, N, device=DEV)
    h1 = torch.zeros(M, N, device=DEV)
    logits, lrows, crows = torch.empty(M, 10, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    dlog, da = torch.empty(M, 10, device=DEV), torch.empty(M, N, device=DEV)
    S = gpu_lib.dtf_gemm_f32_nn_partial(ptr(A), ptr(Bm), ptr(part), M, N, K, K, N, 0, stream())
    assert S > 0
    check(gpu_lib.dtf_head_f32_fused_fwd_bwd(ptr(part), S, ptr(bias1), keep, 9, ptr(state), ptr(h1), ptr(wT), ptr(b),
                                             ptr(labels), None, ptr(logits), ptr(lrows), ptr(crows), ptr(dlog), ptr(da),
                                             M, N, 1.0 / M, 1.0 / keep, 1, stream()), "fused")
    ref = F.relu(A.double() @ Bm.double() + bias1.double())
    kept = h1 > 0
    if keep == 1.0:
        assert rel_err(h1, ref) < TOL
    else:
        assert abs(float(kept.float().mean()) / float((ref > 0).float().mean()) - keep) < 0.02
        assert rel_err(h1[kept], (ref / keep)[kept]) < TOL
    assert rel_err(logits, h1.double() @ wT.double().t() + b.double()) < TOL
    assert bool((da[~kept] == 0).all())


# ------------------------------------------------------------------- optimizer
def test_fused_reduce_sgd_fp32_configuration(gpu_lib):
    """shadow == NULL selects the fp32 configuration of the fused tail: no bf16 copies, and the conv2 packs are the
    fp32 fragment packs of conv2_f32.hip.  Bitwise against reduce_segments + sgd_update + pack_conv2_weights_f32."""
    c1, c2 = 32, 64
    H = 16
    Gs = (5, 33, 7)
    n_head, n_dense, n_c2, n_c1 = H * 10 + 12 + H, 1000, 25 * c1 * c2 + c2, 40
    offs = [0, n_head, n_head + n_dense, n_head + n_dense + n_c2]
    total = offs[3] + n_c1
    g = torch.Generator(device="cpu").manual_seed(3)
    w0 = torch.randn(total, generator=g).to(DEV)
    grads = torch.randn(total, generator=g).to(DEV)
    parts = [torch.randn(G, n, generator=g).to(DEV) for G, n in zip(Gs, (n_head, n_c2, n_c1))]
    hyper = torch.tensor([0.1, 0.9, 0.999, 1e-8, 0.5, 0.0, 0.0, 0.0], device=DEV)
    w_ref, g_ref = w0.clone(), grads.clone()
    for o, p in zip((offs[0], offs[2], offs[3]), parts):
        pa = (ctypes.c_void_p * 1)(p.data_ptr()); oa = (ctypes.c_void_p * 1)(g_ref.data_ptr() + 4 * o)
        check(gpu_lib.dtf_reduce_segments(1, pa, oa, (ctypes.c_int * 1)(p.shape[0]), (ctypes.c_int * 1)(p.shape[1]),
                                          (ctypes.c_long * 1)(p.shape[1]), stream()), "reduce")
    check(gpu_lib.dtf_sgd_update(ptr(w_ref), ptr(g_ref), None, ptr(hyper), total, None, 0, 0, 0, stream()), "sgd")
    pk_ref = torch.zeros(2, 25 * c1 * c2, device=DEV)
    check(gpu_lib.dtf_conv2_f32_pack_weights(w_ref.data_ptr() + 4 * offs[2], ptr(pk_ref), c1, c2, stream()), "pack")
    w1, g1 = w0.clone(), grads.clone()
    pk = torch.zeros(2, 25 * c1 * c2, device=DEV)
    owT = torch.zeros(10, H, device=DEV)
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    pa = (ctypes.c_void_p * 3)(*[p.data_ptr() for p in parts])
    check(gpu_lib.dtf_fused_reduce_sgd(3, pa, (ctypes.c_int * 3)(*Gs), (ctypes.c_int * 3)(n_head, n_c2, n_c1),
                                       (ctypes.c_long * 3)(n_head, n_c2, n_c1), (ctypes.c_long * 3)(offs[0], offs[2], offs[3]),
                                       ptr(w1), ptr(g1), None, ptr(hyper), offs[1], offs[2], ptr(pk), offs[2], c1, c2,
                                       ptr(owT), offs[0], H, ptr(state), 100, 1, 1, stream()), "fused")
    torch.cuda.synchronize()
    assert torch.equal(g1, g_ref) and torch.equal(w1, w_ref)
    assert torch.equal(pk, pk_ref)
    assert torch.equal(owT, w_ref[:H * 10].reshape(H, 10).t().contiguous())
    assert state.tolist()[:3] == [100, 1, 1]
