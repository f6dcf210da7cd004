"""Numerics of every hand-written HIP kernel against a plain PyTorch fp32 reference
of the same op (inputs pre-rounded to bf16 where the kernel consumes bf16).

Run on the GPU box: python -m pytest tests -m gpu
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from distributed_tensorflow_amd.data import batch_indices, synthetic_mnist
from distributed_tensorflow_amd.ops import reference as R
from distributed_tensorflow_amd.ops.native import check, ptr

pytestmark = pytest.mark.gpu
DEV = "cuda"


def stream():
    return torch.cuda.current_stream().cuda_stream


def bf(x):
    return x.to(torch.bfloat16)


def rel_err(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-12))


def conv_block_ref(x_nhwc, w_hwio, b):
    """fp32 conv+bias+relu, then 2x2 pool with argmax (first max wins)."""
    y = R.conv2d_same(x_nhwc.float(), w_hwio.float(), b.float())
    return R.pool_argmax(y)


# ------------------------------------------------------------------ data path
def dead_code(arg, pooled):
    """argmax code as the forward kernels store it: 4 (kPoolDead) where the pooled activation is not positive."""
    return torch.where(bf(pooled) > 0, arg, torch.full_like(arg, 4)).contiguous()


def test_sample_indices_match_host(gpu_lib):
    n, B, seed = 55000, 300, 77
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    idx = torch.zeros(B, dtype=torch.int32, device=DEV)
    from distributed_tensorflow_amd.data import half_bits
    for cursor in (0, 54900, 3 * 55000 + 17):
        state[0] = cursor
        check(gpu_lib.dtf_sample_indices(ptr(state), ptr(idx), B, n, half_bits(n), seed, 1, stream()), "sample")
        want = batch_indices(cursor, B, n, seed)
        assert np.array_equal(idx.cpu().numpy().astype(np.int64), want)
    # one epoch is a permutation
    state[0] = 0
    big = torch.zeros(n, dtype=torch.int32, device=DEV)
    check(gpu_lib.dtf_sample_indices(ptr(state), ptr(big), n, n, half_bits(n), seed, 1, stream()), "sample")
    assert len(np.unique(big.cpu().numpy())) == n


def test_synthetic_generator_matches_host(gpu_lib):
    from distributed_tensorflow_amd.engine import DeviceDataset
    ds = DeviceDataset.synthetic(500, seed=1234)
    x, y = synthetic_mnist(500, seed=1234)
    assert np.array_equal(ds.labels.cpu().numpy(), y)
    assert np.array_equal(ds.images.float().cpu().numpy(), x)


def test_init_kernels(gpu_lib):
    n = 1 << 20
    w = torch.empty(n, device=DEV)
    check(gpu_lib.dtf_init_truncated_normal(ptr(w), n, 0.1, 42, 3, stream()), "init")
    assert float(w.abs().max()) <= 0.2 + 1e-6
    assert abs(float(w.mean())) < 1e-3
    # std of a 2-sigma truncated normal is 0.8796 sigma
    assert abs(float(w.std()) / 0.1 - 0.8796) < 0.01
    check(gpu_lib.dtf_fill_f32(ptr(w), n, 0.1, stream()), "fill")
    assert float(w.min()) == float(w.max()) == pytest.approx(0.1)


# ---------------------------------------------------------------------- conv1
@pytest.mark.parametrize("B,C", [(1, 32), (37, 32), (8, 128)])
def test_conv1_fwd(gpu_lib, B, C):
    g = torch.Generator(device="cpu").manual_seed(1)
    N = 64
    images = bf(torch.rand(N, 784, generator=g)).to(DEV)
    idx = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(DEV)
    w = (torch.randn(5, 5, 1, C, generator=g) * 0.1).to(DEV)
    b = (torch.randn(C, generator=g) * 0.1).to(DEV)
    pooled = torch.empty(B, 14, 14, C, dtype=torch.bfloat16, device=DEV)
    arg = torch.empty(B, 14, 14, C, dtype=torch.uint8, device=DEV)
    check(gpu_lib.dtf_conv1_fwd(ptr(images), ptr(idx), ptr(w), ptr(b), ptr(pooled), ptr(arg), B, C, 0, stream()),
          "conv1_fwd")
    x = images[idx.long()].float().reshape(B, 28, 28, 1)
    ref, ref_arg = conv_block_ref(x, w, b)
    assert rel_err(pooled, ref) < 5e-3
    live = ref > 1e-3
    agree = (arg == ref_arg)[live].float().mean()
    assert agree > 0.995, agree
    assert bool(((arg == 4) == (pooled.float() <= 0)).all())   # ReLU-dead windows carry the dead code


def test_conv1_fwd_fused_sampler(gpu_lib):
    """conv1 forward drawing its own batch from the device cursor == sample_indices + gathered conv1 forward."""
    from distributed_tensorflow_amd.data import half_bits
    g = torch.Generator(device="cpu").manual_seed(11)
    N, B, C, seed = 500, 96, 32, 5
    images = bf(torch.rand(N, 784, generator=g)).to(DEV)
    w = (torch.randn(5, 5, 1, C, generator=g) * 0.1).to(DEV)
    b = (torch.randn(C, generator=g) * 0.1).to(DEV)
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    for cursor in (0, 450, 7 * N + 123):   # the middle one spans an epoch boundary
        state[0] = cursor
        idx = torch.zeros(B, dtype=torch.int32, device=DEV)
        check(gpu_lib.dtf_sample_indices(ptr(state), ptr(idx), B, N, half_bits(N), seed, 1, stream()), "sample")
        p0 = torch.empty(B, 14, 14, C, dtype=torch.bfloat16, device=DEV)
        a0 = torch.empty(B, 14, 14, C, dtype=torch.uint8, device=DEV)
        check(gpu_lib.dtf_conv1_fwd(ptr(images), ptr(idx), ptr(w), ptr(b), ptr(p0), ptr(a0), B, C, 0, stream()), "c1")
        idx1 = torch.full((B,), -1, dtype=torch.int32, device=DEV)
        p1, a1 = torch.empty_like(p0), torch.empty_like(a0)
        check(gpu_lib.dtf_conv1_fwd_sampled(ptr(images), ptr(idx1), ptr(state), N, half_bits(N), seed, 1, ptr(w), ptr(b),
                                            ptr(p1), ptr(a1), B, C, 0, stream()), "c1 sampled")
        assert torch.equal(idx, idx1)
        assert torch.equal(p0, p1) and torch.equal(a0, a1)


@pytest.mark.parametrize("B,C", [(33, 32), (6, 128)])
def test_conv1_wgrad(gpu_lib, B, C):
    g = torch.Generator(device="cpu").manual_seed(2)
    images = bf(torch.rand(B, 784, generator=g)).to(DEV)
    w = (torch.randn(5, 5, 1, C, generator=g) * 0.1).to(DEV).requires_grad_(True)
    b = (torch.randn(C, generator=g) * 0.1).to(DEV).requires_grad_(True)
    x = images.float().reshape(B, 28, 28, 1)
    y = R.maxpool2d(R.conv2d_same(x, w, b))
    gp = bf(torch.randn(y.shape, generator=g)).to(DEV)
    y.backward(gp.float())
    pooled, arg = conv_block_ref(x, w.detach(), b.detach())
    G = 7
    part = torch.empty(G, 26 * C, device=DEV)
    code = dead_code(arg, pooled)
    check(gpu_lib.dtf_conv1_wgrad(ptr(images), None, ptr(gp), ptr(code), ptr(part), B, C, G, stream()), "conv1_wgrad")
    got = part.sum(0)
    assert rel_err(got[:25 * C].reshape(5, 5, 1, C), w.grad) < 2e-3
    assert rel_err(got[25 * C:], b.grad) < 2e-3


# ---------------------------------------------------------------------- conv2
def _conv2_inputs(B, C1, C2, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = bf(torch.rand(B, 14, 14, C1, generator=g)).to(DEV)
    w = bf(torch.randn(5, 5, C1, C2, generator=g) * (1.0 / (25 * C1) ** 0.5)).to(DEV)
    b = (torch.randn(C2, generator=g) * 0.1).to(DEV)
    return g, x, w, b


@pytest.mark.parametrize("B,C1,C2", [(1, 32, 64), (5, 32, 64), (64, 32, 64), (3, 128, 256)])
def test_conv2_fwd(gpu_lib, B, C1, C2):
    g, x, w, b = _conv2_inputs(B, C1, C2, 3)
    packed = torch.empty(2, 25 * C1 * C2, dtype=torch.bfloat16, device=DEV)
    check(gpu_lib.dtf_conv2_pack_weights(ptr(w), ptr(packed), C1, C2, stream()), "pack")
    f_ref, d_ref = R.conv2_pack_reference(w.reshape(25, C1, C2))
    assert torch.equal(packed[0], f_ref) and torch.equal(packed[1], d_ref)
    wT = packed[0]
    pooled = torch.empty(B, 7, 7, C2, dtype=torch.bfloat16, device=DEV)
    arg = torch.empty(B, 7, 7, C2, dtype=torch.uint8, device=DEV)
    check(gpu_lib.dtf_conv2_fwd(ptr(x), ptr(wT), ptr(b), ptr(pooled), ptr(arg), B, C1, C2, stream()), "conv2_fwd")
    ref, ref_arg = conv_block_ref(x, w, b)
    assert rel_err(pooled, ref) < 6e-3
    live = ref > 1e-2
    agree = (arg == ref_arg)[live].float().mean()
    assert agree > 0.99, agree
    assert bool(((arg == 4) == (pooled.float() <= 0)).all())


@pytest.mark.parametrize("B", [1, 2, 37, 256, 640, 641, 801])
def test_fused_conv1_conv2_forward_is_bitwise_the_two_kernels(gpu_lib, B):
    """conv1 computed inside conv2's staging phase (one launch) == conv1_fwd then conv2_fwd, with an
    explicit gather and with the on-device batch sampler.  Up to dtf_conv2_small_batch() (640) images the fused
    kernel takes one image per 8-wave workgroup, above that two."""
    assert gpu_lib.dtf_conv2_small_batch() == 640
    from distributed_tensorflow_amd.data import half_bits
    g = torch.Generator(device="cpu").manual_seed(31)
    N, seed = 300, 11
    images = bf(torch.rand(N, 784, generator=g)).to(DEV)
    w1 = (torch.randn(5, 5, 1, 32, generator=g) * 0.1).to(DEV)
    b1 = (torch.randn(32, generator=g) * 0.1).to(DEV)
    _, _, w2, b2 = _conv2_inputs(1, 32, 64, 32)
    packed = torch.empty(2, 25 * 32 * 64, dtype=torch.bfloat16, device=DEV)
    check(gpu_lib.dtf_conv2_pack_weights(ptr(w2), ptr(packed), 32, 64, stream()), "pack")
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    state[0] = 250   # spans an epoch boundary for B > 50

    def buffers():
        return (torch.zeros(B, 14, 14, 32, dtype=torch.bfloat16, device=DEV), torch.zeros(B, 14, 14, 32, dtype=torch.uint8, device=DEV),
                torch.zeros(B, 7, 7, 64, dtype=torch.bfloat16, device=DEV), torch.zeros(B, 7, 7, 64, dtype=torch.uint8, device=DEV))

    for sampled in (False, True):
        idx = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(DEV)
        p1, a1, p2, a2 = buffers()
        if sampled:
            check(gpu_lib.dtf_sample_indices(ptr(state), ptr(idx), B, N, half_bits(N), seed, 1, stream()), "sample")
        check(gpu_lib.dtf_conv1_fwd(ptr(images), ptr(idx), ptr(w1), ptr(b1), ptr(p1), ptr(a1), B, 32, 0, stream()), "c1")
        check(gpu_lib.dtf_conv2_fwd(ptr(p1), ptr(packed[0]), ptr(b2), ptr(p2), ptr(a2), B, 32, 64, stream()), "c2")
        q1, c1, q2, c2 = buffers()
        idx_out = torch.full((B,), -1, dtype=torch.int32, device=DEV)
        check(gpu_lib.dtf_conv12_fwd(ptr(images), None if sampled else ptr(idx), ptr(idx_out), ptr(state) if sampled else None,
                                     N, half_bits(N), seed, 1, ptr(w1), ptr(b1), ptr(q1), ptr(c1), ptr(packed[0]), ptr(b2),
                                     ptr(q2), ptr(c2), B, stream()), "fused")
        torch.cuda.synchronize()
        assert torch.equal(p1, q1) and torch.equal(a1, c1)
        assert torch.equal(p2, q2) and torch.equal(a2, c2)
        if sampled:
            assert torch.equal(idx_out, idx)
    assert float(p2.float().abs().sum()) > 0


@pytest.mark.parametrize("B", [1, 2, 33, 128])
@pytest.mark.parametrize("imgs", [1, 2])
def test_fused_conv2_dgrad_conv1_wgrad_matches_the_two_kernels(gpu_lib, B, imgs):
    """dx kept in LDS and consumed by the conv1 filter gradient in the same launch == conv2_dgrad then
    conv1_wgrad (same arithmetic per image; only the image -> slab assignment differs), with two images per
    workgroup and with one (the small-batch launch: B slabs)."""
    g = torch.Generator(device="cpu").manual_seed(41)
    N = 200
    images = bf(torch.rand(N, 784, generator=g)).to(DEV)
    idx = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(DEV)
    _, _, w2, _ = _conv2_inputs(1, 32, 64, 42)
    packed = torch.empty(2, 25 * 32 * 64, dtype=torch.bfloat16, device=DEV)
    check(gpu_lib.dtf_conv2_pack_weights(ptr(w2), ptr(packed), 32, 64, stream()), "pack")
    gp = bf(torch.randn(B, 7, 7, 64, generator=g)).to(DEV)
    arg2 = torch.randint(0, 5, (B, 7, 7, 64), generator=g, dtype=torch.uint8).to(DEV)
    arg1 = torch.randint(0, 5, (B, 14, 14, 32), generator=g, dtype=torch.uint8).to(DEV)
    dx = torch.empty(B, 14, 14, 32, dtype=torch.bfloat16, device=DEV)
    check(gpu_lib.dtf_conv2_dgrad(ptr(gp), ptr(arg2), ptr(packed[1]), ptr(dx), B, 32, 64, stream()), "dgrad")
    G = (B + imgs - 1) // imgs
    ref = torch.zeros(G, 26 * 32, device=DEV)
    check(gpu_lib.dtf_conv1_wgrad(ptr(images), ptr(idx), ptr(dx), ptr(arg1), ptr(ref), B, 32, G, stream()), "conv1_wgrad")
    got = torch.full((G, 26 * 32), float("nan"), device=DEV)
    if imgs == 2:
        check(gpu_lib.dtf_conv2_dgrad_conv1_wgrad(ptr(gp), ptr(arg2), ptr(packed[1]), ptr(images), ptr(idx), ptr(arg1),
                                                  ptr(got), B, stream()), "fused")
    elif B == 1:
        pytest.skip("one image: the two launches coincide")
    else:
        check(gpu_lib.dtf_conv2_dgrad_conv1_wgrad_g(ptr(gp), ptr(arg2), ptr(packed[1]), ptr(images), ptr(idx), ptr(arg1),
                                                    ptr(got), B, G, stream()), "fused, one image per workgroup")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got).all())
    assert rel_err(got.sum(0), ref.sum(0)) < 1e-5
    assert float(ref.sum(0).abs().sum()) > 0


@pytest.mark.parametrize("B,C1,C2", [(1, 32, 64), (5, 32, 64), (32, 32, 64), (3, 128, 256), (9, 128, 256)])
def test_conv2_backward(gpu_lib, B, C1, C2):
    g, x, w, b = _conv2_inputs(B, C1, C2, 4)
    xf = x.float().requires_grad_(True)
    wf = w.float().requires_grad_(True)
    bfp = b.clone().requires_grad_(True)
    y = R.maxpool2d(R.conv2d_same(xf, wf, bfp))
    gp = bf(torch.randn(y.shape, generator=g)).to(DEV)
    y.backward(gp.float())
    pooled, arg = conv_block_ref(x, w, b)
    arg = dead_code(arg, pooled)
    # dgrad
    dx = torch.empty(B, 14, 14, C1, dtype=torch.bfloat16, device=DEV)
    packed = torch.empty(2, 25 * C1 * C2, dtype=torch.bfloat16, device=DEV)
    check(gpu_lib.dtf_conv2_pack_weights(ptr(w), ptr(packed), C1, C2, stream()), "pack")
    check(gpu_lib.dtf_conv2_dgrad(ptr(gp), ptr(arg), ptr(packed[1]), ptr(dx), B, C1, C2, stream()),
          "conv2_dgrad")
    assert rel_err(dx, xf.grad) < 8e-3
    # wgrad: the default launch, then both wave tiles (16 ci x 32 co and 16 ci x 64 co) explicitly
    G = 3
    stride = 25 * C1 * C2 + C2
    results = []
    for nt in (None, 2, 4):
        part = torch.full((G, stride), float("nan"), device=DEV)
        if nt is None:
            check(gpu_lib.dtf_conv2_wgrad(ptr(x), ptr(gp), ptr(arg), ptr(part), B, C1, C2, G, stream()), "conv2_wgrad")
        else:
            check(gpu_lib.dtf_conv2_wgrad_nt(ptr(x), ptr(gp), ptr(arg), ptr(part), B, C1, C2, G, nt, stream()),
                  f"conv2_wgrad_nt{nt}")
        got = part[:min(G, B)].sum(0)            # image chunks past the batch write nothing
        assert bool(torch.isfinite(got).all()), nt
        assert rel_err(got[:25 * C1 * C2].reshape(5, 5, C1, C2), wf.grad) < 3e-3, nt
        assert rel_err(got[25 * C1 * C2:], bfp.grad) < 3e-3, nt
        results.append(got)
    # same products, same summation order over the images: the two wave tiles agree bit for bit
    assert torch.equal(results[1], results[2])


@pytest.mark.parametrize("B,G", [(7, 1), (37, 2), (64, 5), (33, 33), (96, 3)])
def test_conv2_wgrad_producer_consumer_kernel_matches_the_self_staging_one(gpu_lib, B, G):
    """conv5x5_wgrad_pc (the default launch of the 32/64 model: producer waves, two-image rounds handed over through
    LDS progress words, three buffer pairs) against conv5x5_wgrad<2>: odd image counts per workgroup (the kh = 1 waves
    sit out the last round), more rounds than buffer pairs (reuse behind the `done` words), one image per workgroup.
    The two kernels sum the same bf16 products in a different order, fp32 accumulators."""
    g, x, w, b = _conv2_inputs(B, 32, 64, 11)
    gp = bf(torch.randn(B, 7, 7, 64, generator=g)).to(DEV)
    arg = torch.randint(0, 5, (B, 7, 7, 64), generator=g, dtype=torch.uint8).to(DEV)
    stride = 25 * 32 * 64 + 64
    out = []
    for nt in (0, 2):
        part = torch.full((G, stride), float("nan"), device=DEV)
        for _ in range(3):     # repeated launches: a stale progress word or buffer would show
            check(gpu_lib.dtf_conv2_wgrad_nt(ptr(x), ptr(gp), ptr(arg), ptr(part), B, 32, 64, G, nt, stream()), f"wgrad nt={nt}")
        got = part[:min(G, B)].sum(0)
        assert bool(torch.isfinite(got).all())
        out.append(got)
    assert rel_err(out[0][:-64], out[1][:-64]) < 2e-6
    assert rel_err(out[0][-64:], out[1][-64:]) < 2e-6      # bias: ones x dY on the matrix core vs the VALU sum
    assert float(out[1].abs().sum()) > 0


# ----------------------------------------------------------------------- GEMMs
@pytest.mark.parametrize("M,N,K,tile", [(100, 1024, 3136, 64), (128, 1024, 3136, 128), (257, 192, 64, 64),
                                        (1000, 3136, 1024, 128), (64, 64, 128, 0)])
def test_gemm_nt(gpu_lib, M, N, K, tile):
    g = torch.Generator(device="cpu").manual_seed(5)
    # asymmetric operands so a transposed/garbled fragment map cannot pass
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    Bm = bf(torch.randn(N, K, generator=g) * 0.05 + torch.arange(N).float()[:, None] * 1e-3).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    check(gpu_lib.dtf_gemm_nt(ptr(A), ptr(Bm), ptr(C), None, M, N, K, K, K, N, 0, 1.0, 0, None, tile, stream()),
          "gemm_nt")
    ref = A.float() @ Bm.float().t()
    assert rel_err(C, ref) < 4e-3
    check(gpu_lib.dtf_gemm_nt(ptr(A), ptr(Bm), ptr(C), ptr(bias), M, N, K, K, K, N, 1, 1.0, 0, None, tile, stream()),
          "gemm_nt epi")
    assert rel_err(C, F.relu(ref + bias)) < 4e-3


@pytest.mark.parametrize("M,N,K", [(4096, 3136, 1024), (4000, 3000, 256), (300, 200, 128)])
def test_gemm_nt_256x128_tile(gpu_lib, M, N, K):
    """The 256 x 128 workgroup tile (4 x 2 waves of 64 x 64; selected automatically once it fills the machine 1.5
    times over): full and ragged tiles, forced with tile = 256 and through the automatic choice."""
    g = torch.Generator(device="cpu").manual_seed(25)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    Bm = bf(torch.randn(N, K, generator=g) * 0.05 + torch.arange(N).float()[:, None] * 1e-3).to(DEV)
    ref = A.float() @ Bm.float().t()
    for tile in (256, 0):
        C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        check(gpu_lib.dtf_gemm_nt(ptr(A), ptr(Bm), ptr(C), None, M, N, K, K, K, N, 0, 1.0, 0, None, tile, stream()),
              "gemm_nt 256")
        assert rel_err(C, ref) < 4e-3, tile



@pytest.mark.parametrize("M,N,K,S", [(1024, 1024, 3136, 0), (100, 1024, 3136, 0), (257, 384, 192, 3), (64, 128, 64, 1),
                                     (130, 256, 200, 2), (130, 136, 72, 1)])     # K tails: rows past K must stage as zeros
def test_gemm_nn_splitk_forward(gpu_lib, M, N, K, S):
    g = torch.Generator(device="cpu").manual_seed(15)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    Bm = bf(torch.randn(K, N, generator=g) * 0.05 + torch.arange(N).float()[None, :] * 1e-3).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    part = torch.empty(8, M, N, device=DEV)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    check(gpu_lib.dtf_gemm_nn_fwd(ptr(A), ptr(Bm), ptr(part), ptr(out), ptr(bias), M, N, K, K, N, S, 1.0, 0, None,
                                  stream()), "gemm_nn_fwd")
    ref = F.relu(A.float() @ Bm.float() + bias)
    assert rel_err(out, ref) < 4e-3
    # dropout through the split-K epilogue: same mask semantics as the fused GEMM epilogue
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    o1 = torch.empty_like(out)
    check(gpu_lib.dtf_gemm_nn_fwd(ptr(A), ptr(Bm), ptr(part), ptr(o1), ptr(bias), M, N, K, K, N, S, 0.5, 3, ptr(state),
                                  stream()), "gemm_nn_fwd drop")
    kept = o1.float() > 0
    assert abs(float(kept.float().mean()) / float((ref > 0).float().mean()) - 0.5) < 0.05
    assert rel_err(o1.float()[kept], (2 * ref)[kept]) < 4e-3


def test_gemm_nt_dropout(gpu_lib):
    M, N, K, keep = 256, 1024, 256, 0.75
    g = torch.Generator(device="cpu").manual_seed(6)
    A = bf(torch.rand(M, K, generator=g)).to(DEV)
    Bm = bf(torch.rand(N, K, generator=g)).to(DEV)
    bias = torch.zeros(N, device=DEV)
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    C0 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    C1 = torch.empty_like(C0)
    C2 = torch.empty_like(C0)
    args = (ptr(A), ptr(Bm))
    check(gpu_lib.dtf_gemm_nt(*args, ptr(C0), ptr(bias), M, N, K, K, K, N, 1, keep, 9, ptr(state), 0, stream()), "d0")
    check(gpu_lib.dtf_gemm_nt(*args, ptr(C1), ptr(bias), M, N, K, K, K, N, 1, keep, 9, ptr(state), 0, stream()), "d1")
    state[2] = 1
    check(gpu_lib.dtf_gemm_nt(*args, ptr(C2), ptr(bias), M, N, K, K, K, N, 1, keep, 9, ptr(state), 0, stream()), "d2")
    ref = (A.float() @ Bm.float().t())
    kept = C0.float() > 0
    frac = float(kept.float().mean())
    assert abs(frac - keep) < 0.01, frac
    assert torch.equal(C0, C1)              # same offset -> same mask
    assert not torch.equal(C0, C2)          # new offset -> new mask
    assert rel_err(C0.float()[kept], (ref / keep)[kept]) < 4e-3


@pytest.mark.parametrize("M,N,K,tile", [(3136, 1024, 100, 128), (3136, 1024, 256, 64), (200, 136, 1000, 64),
                                        (128, 128, 64, 128)])
def test_gemm_tn(gpu_lib, M, N, K, tile):
    g = torch.Generator(device="cpu").manual_seed(7)
    A = bf(torch.randn(K, M, generator=g) + torch.arange(M).float()[None, :] * 1e-3).to(DEV)
    Bm = bf(torch.randn(K, N, generator=g)).to(DEV)
    C = torch.empty(M, N, device=DEV)
    check(gpu_lib.dtf_gemm_tn(ptr(A), ptr(Bm), ptr(C), M, N, K, M, N, N, tile, stream()), "gemm_tn")
    ref = A.float().t() @ Bm.float()
    assert rel_err(C, ref) < 1e-4


@pytest.mark.parametrize("M,N,K", [(3136, 1024, 128), (200, 136, 1000)])
def test_gemm_tn_bf16_gradient_store(gpu_lib, M, N, K):
    """The 16-bit gradient epilogue (grad_comm="bf16"): exactly the fp32 product of the plain kernel, rounded once."""
    g = torch.Generator(device="cpu").manual_seed(9)
    A = bf(torch.randn(K, M, generator=g) + torch.arange(M).float()[None, :] * 1e-3).to(DEV)
    Bm = bf(torch.randn(K, N, generator=g)).to(DEV)
    C = torch.empty(M, N, device=DEV)
    check(gpu_lib.dtf_gemm_tn(ptr(A), ptr(Bm), ptr(C), M, N, K, M, N, N, 128, stream()), "gemm_tn")
    C16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    check(gpu_lib.dtf_gemm_tn_g16_ev(ptr(A), ptr(Bm), ptr(C16), M, N, K, M, N, N, None, stream()), "gemm_tn_g16")
    assert torch.equal(C16, C.to(torch.bfloat16))


def test_sgd_and_adam_with_bf16_gradient(gpu_lib):
    """The optimizer kernels reading a bf16 gradient == the same kernels on that gradient widened to fp32, bit for bit."""
    n = 4 * 1000 + 4
    w0 = torch.randn(n, device=DEV)
    g16 = torch.randn(n, device=DEV).to(torch.bfloat16)
    g32 = g16.float()
    hyper = torch.tensor([0.01, 0.9, 0.999, 1e-8, 0.5, 0, 0, 0], device=DEV)
    outs = []
    for wide in (True, False):
        state = torch.zeros(8, dtype=torch.int64, device=DEV)
        w, sh = w0.clone(), torch.empty(n, dtype=torch.bfloat16, device=DEV)
        if wide:
            check(gpu_lib.dtf_sgd_update(ptr(w), ptr(g32), ptr(sh), ptr(hyper), n, ptr(state), 128, 1, 1, stream()), "sgd")
        else:
            check(gpu_lib.dtf_sgd_update_g16(ptr(w), ptr(g16), ptr(sh), ptr(hyper), n, ptr(state), 128, 1, 1, stream()), "sgd16")
        assert state[:4].tolist() == [128, 1, 1, 0]
        m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        for _ in range(3):
            f = gpu_lib.dtf_adam_update if wide else gpu_lib.dtf_adam_update_g16
            check(f(ptr(w), ptr(g32 if wide else g16), ptr(m), ptr(v), ptr(sh), ptr(hyper), n, ptr(state), 3, ptr(state),
                    0, 1, 1, stream()), "adam")
        assert int(state[3]) == 3
        outs.append((w, sh, m, v))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)


def test_step_state_counters_advance_independently_on_two_streams(gpu_lib):
    """The data-parallel Adam step bumps the dense range's step counter on the comm stream while the tail kernel advances
    cursor / global step / dropout offset on the compute stream.  A counter advanced by 0 must not be touched at all:
    "+= 0" wrote a stale value back over the other stream's increment (lost batches in the two-rank Adam runs)."""
    n, iters = 256, 1500
    w, g = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    hyper = torch.tensor([0.01, 0.9, 0.999, 1e-8, 1.0, 0, 0, 0], device=DEV)
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    # both streams first spin for ~50 ms, so that all launches below are queued before any of them runs: the two
    # streams then drain back to back and their one-thread kernels really overlap (launched eagerly they alternate)
    for st in (sa, sb):
        with torch.cuda.stream(st):
            torch.cuda._sleep(100_000_000)
    for _ in range(iters):
        # stream A: Adam on a small range, bumps state[3] only (state == NULL: no cursor / step advance)
        check(gpu_lib.dtf_adam_update(ptr(w), ptr(g), ptr(m), ptr(v), None, ptr(hyper), n, ptr(state), 3, None, 0, 0, 0,
                                      sa.cuda_stream), "adam")
        # stream B: the step advance
        check(gpu_lib.dtf_advance_state(ptr(state), 64, 1, 2, sb.cuda_stream), "advance")
    torch.cuda.synchronize()
    assert state[:4].tolist() == [64 * iters, iters, 2 * iters, iters], state.tolist()


@pytest.mark.parametrize("M,N,K", [(1024, 3136, 1024), (1000, 3136, 1024), (257, 384, 192), (128, 128, 64)])
def test_gemm_lds_dma_staging_is_bitwise_the_register_staged_kernel(gpu_lib, M, N, K):
    """The LDS-DMA pipeline (buffer_load ... lds, four LDS buffers, fragments read a phase ahead, counted vmcnt + raw
    barriers) multiplies the same fragments in the same order as the register-staged one: bit-identical outputs, for
    the NT (fc1 dgrad) and TN (fc1 wgrad) kernels, over many launches while another stream keeps the memory system
    busy (a mis-ordered read of a staged buffer shows up as rare wrong tiles, not as a steady error)."""
    g = torch.Generator(device="cpu").manual_seed(12)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    Bm = bf(torch.randn(N, K, generator=g) * 0.05 + torch.arange(N).float()[:, None] * 1e-3).to(DEV)
    At, Bt = A.t().contiguous(), Bm.t().contiguous()          # TN operands: [K][M], [K][N]
    Mp, Np = (M + 7) // 8 * 8, (N + 7) // 8 * 8               # the TN kernel wants M, N multiples of 8
    At = torch.nn.functional.pad(At, (0, Mp - M)).contiguous()
    Bt = torch.nn.functional.pad(Bt, (0, Np - N)).contiguous()
    ref_nt = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ref_tn = torch.empty(Mp, Np, device=DEV)
    check(gpu_lib.dtf_gemm_nt(ptr(A), ptr(Bm), ptr(ref_nt), None, M, N, K, K, K, N, 0, 1.0, 0, None, 129, stream()), "nt reg")
    check(gpu_lib.dtf_gemm_tn(ptr(At), ptr(Bt), ptr(ref_tn), Mp, Np, K, Mp, Np, Np, 129, stream()), "tn reg")
    torch.cuda.synchronize()
    assert rel_err(ref_nt, A.float() @ Bm.float().t()) < 4e-3
    noise = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    out_nt = torch.empty_like(ref_nt)
    out_tn = torch.empty_like(ref_tn)
    for it in range(60):
        with torch.cuda.stream(noise):
            junk.copy_(junk.flip(0)) if it % 3 == 0 else junk.add_(1)
        out_nt.fill_(float("nan"))
        out_tn.fill_(float("nan"))
        check(gpu_lib.dtf_gemm_nt(ptr(A), ptr(Bm), ptr(out_nt), None, M, N, K, K, K, N, 0, 1.0, 0, None, 130, stream()), "nt dma")
        check(gpu_lib.dtf_gemm_tn(ptr(At), ptr(Bt), ptr(out_tn), Mp, Np, K, Mp, Np, Np, 130, stream()), "tn dma")
        assert torch.equal(out_nt, ref_nt), f"NT, launch {it}"
        assert torch.equal(out_tn, ref_tn), f"TN, launch {it}"
    torch.cuda.synchronize()


def test_transpose(gpu_lib):
    x = bf(torch.randn(3, 100, 72)).to(DEV)
    out = torch.empty(3, 72, 100, dtype=torch.bfloat16, device=DEV)
    check(gpu_lib.dtf_transpose_bf16(ptr(x), ptr(out), 3, 100, 72, stream()), "transpose")
    assert torch.equal(out, x.transpose(1, 2).contiguous())


# ------------------------------------------------------------------------ head
@pytest.mark.parametrize("B,H", [(1, 1024), (100, 1024), (37, 4096)])
def test_head(gpu_lib, B, H):
    g = torch.Generator(device="cpu").manual_seed(8)
    h1 = bf(F.relu(torch.randn(B, H, generator=g))).to(DEV)
    w = (torch.randn(H, 10, generator=g) * 0.1).to(DEV).requires_grad_(True)
    b = (torch.randn(10, generator=g) * 0.1).to(DEV).requires_grad_(True)
    labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32).to(DEV)
    keep = 0.5
    hf = h1.float().requires_grad_(True)
    z = hf @ w + b
    loss_rows = R.softmax_cross_entropy_with_logits(z, labels)
    loss_rows.mean().backward()

    logits = torch.empty(B, 10, device=DEV)
    lrows = torch.empty(B, device=DEV)
    crows = torch.empty(B, device=DEV)
    dlog = torch.empty(B, 10, device=DEV)
    da = torch.empty(B, H, dtype=torch.bfloat16, device=DEV)
    wT = torch.empty(10, H, device=DEV)
    check(gpu_lib.dtf_transpose_head_w(ptr(w), ptr(wT), H, stream()), "transpose_head_w")
    assert torch.equal(wT, w.detach().t().contiguous())
    check(gpu_lib.dtf_head_fwd_bwd(ptr(h1), ptr(wT), ptr(b), ptr(labels), None, ptr(logits), ptr(lrows), ptr(crows),
                                   ptr(dlog), ptr(da), B, H, 1.0 / B, 1.0 / keep, 1, stream()), "head")
    assert rel_err(logits, z) < 1e-5
    assert rel_err(lrows, loss_rows) < 1e-5
    assert torch.equal(crows, (z.argmax(-1) == labels.long()).float())
    dz = (torch.softmax(z.detach(), -1) - F.one_hot(labels.long(), 10).float()) / B
    assert rel_err(dlog, dz) < 1e-5
    want_da = torch.where(h1.float() > 0, hf.grad / keep, torch.zeros_like(hf.grad))
    assert rel_err(da, want_da) < 4e-3

    KS = 3
    stride = H * 10 + 12 + H
    part = torch.empty(KS, stride, device=DEV)
    check(gpu_lib.dtf_head_wgrad(ptr(h1), ptr(dlog), ptr(da), ptr(part), B, H, KS, stream()), "head_wgrad")
    got = part.sum(0)
    assert rel_err(got[:H * 10].reshape(H, 10), w.grad) < 1e-4
    assert rel_err(got[H * 10:H * 10 + 10], b.grad) < 1e-4
    assert float(got[H * 10 + 10:H * 10 + 12].abs().max()) == 0.0
    assert rel_err(got[H * 10 + 12:], da.float().sum(0)) < 1e-4


@pytest.mark.parametrize("B,KS", [(1024, 32), (1000, 7), (64, 8), (100, 8), (256, 8), (300, 8)])
def test_fc1_dgrad_with_head_wgrad_workgroups_matches_the_two_kernels(gpu_lib, B, KS):
    """One launch = the fc1 dgrad GEMM tiles + the output-layer wgrad workgroups (gemm.hip HEADW).
    The GEMM result is bitwise the plain kernel's; the reduction (8 row phases instead of 4, so a
    different fp32 summation order) matches the stand-alone kernel and the fp32 reference.
    Below a round of 128 x 128 tiles (B < 961 here) the 64 x 64 tile kernel carries the reduction workgroups."""
    H, F_ = 1024, 3136
    g = torch.Generator(device="cpu").manual_seed(33)
    da = bf(torch.randn(B, H, generator=g) * (torch.rand(B, H, generator=g) > 0.5)).to(DEV)
    w = bf(torch.randn(F_, H, generator=g) * 0.03).to(DEV)            # wd1 [flat][hidden]: dflat = da . wd1^T
    h1 = bf(torch.randn(B, H, generator=g).clamp_min(0)).to(DEV)
    dlog = (torch.randn(B, 10, generator=g) / B).to(DEV)
    stride = H * 10 + 12 + H
    C0 = torch.empty(B, F_, dtype=torch.bfloat16, device=DEV)
    C1 = torch.full((B, F_), 7.0, dtype=torch.bfloat16, device=DEV)
    part0 = torch.empty(KS, stride, device=DEV)
    part1 = torch.full((KS, stride), 3.0, device=DEV)
    check(gpu_lib.dtf_gemm_nt(ptr(da), ptr(w), ptr(C0), None, B, F_, H, H, H, F_, 0, 1.0, 0, None, 0, stream()), "nt")
    check(gpu_lib.dtf_head_wgrad(ptr(h1), ptr(dlog), ptr(da), ptr(part0), B, H, KS, stream()), "head_wgrad")
    check(gpu_lib.dtf_gemm_nt_head_wgrad(ptr(da), ptr(w), ptr(C1), B, F_, H, H, H, F_, ptr(h1), ptr(dlog), ptr(da),
                                         ptr(part1), B, H, KS, stream()), "nt+head_wgrad")
    torch.cuda.synchronize()
    assert torch.equal(C0, C1)
    assert rel_err(C1, da.float() @ w.float().t()) < 4e-3
    # padding floats 10, 11 of the db_out block are written as zeros by both
    assert rel_err(part1, part0) < 1e-5
    got = part1.sum(0)
    assert rel_err(got[:H * 10].reshape(H, 10), h1.float().t() @ dlog) < 1e-4
    assert rel_err(got[H * 10:H * 10 + 10], dlog.sum(0)) < 1e-4
    assert rel_err(got[H * 10 + 12:], da.float().sum(0)) < 1e-4


@pytest.mark.parametrize("keep", [1.0, 0.6])
def test_head_fused_fc1_epilogue_matches_two_kernel_path(gpu_lib, keep):
    """split-K partials -> (bias, ReLU, dropout, h1, logits, loss, dlogits, da) in one launch ==
    splitk_bias_relu_dropout followed by head_fwd_bwd, bitwise."""
    M, N, K = 200, 1024, 3136
    g = torch.Generator(device="cpu").manual_seed(21)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    Bm = bf(torch.randn(K, N, generator=g) * 0.03).to(DEV)
    bias1 = torch.randn(N, generator=g).to(DEV)
    wT = (torch.randn(10, N, generator=g) * 0.1).to(DEV)
    b = (torch.randn(10, generator=g) * 0.1).to(DEV)
    labels = torch.randint(0, 10, (M,), generator=g, dtype=torch.int32).to(DEV)
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    state[2] = 5
    part = torch.empty(8, M, N, device=DEV)
    outs = []
    for fused in (False, True):
        h1 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        logits, lrows, crows = torch.empty(M, 10, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        dlog, da = torch.empty(M, 10, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        tail = (ptr(wT), ptr(b), ptr(labels), None, ptr(logits), ptr(lrows), ptr(crows), ptr(dlog), ptr(da), M, N,
                1.0 / M, 1.0 / keep, 1, stream())
        if fused:
            S = gpu_lib.dtf_gemm_nn_partial(ptr(A), ptr(Bm), ptr(part), M, N, K, K, N, 0, stream())
            assert S > 0
            check(gpu_lib.dtf_head_fused_fwd_bwd(ptr(part), S, ptr(bias1), keep, 9, ptr(state), ptr(h1), *tail), "fused")
        else:
            check(gpu_lib.dtf_gemm_nn_fwd(ptr(A), ptr(Bm), ptr(part), ptr(h1), ptr(bias1), M, N, K, K, N, 0, keep, 9,
                                          ptr(state), stream()), "nn_fwd")
            check(gpu_lib.dtf_head_fwd_bwd(ptr(h1), *tail), "head")
        torch.cuda.synchronize()
        outs.append((h1, logits, lrows, crows, dlog, da))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert float(outs[0][0].float().abs().sum()) > 0


@pytest.mark.parametrize("S", [1, 2, 3, 4, 6, 8, 9])
@pytest.mark.parametrize("gather", [False, True])
def test_head_one_pass_kernel_matches_the_general_one_and_sums_slabs_in_order(gpu_lib, S, gather):
    """H == 1024 runs a variant with two waves per row, eight hidden units per thread, the row's activations and
    output-weight slices kept in registers for the backward half (head.hip ONEPASS).  h1 is bit for bit the general
    kernel's and exactly relu(((p0 + p1) + p2) + ... + bias) rounded to bf16 -- the slab order of the stand-alone
    epilogue; the logits are summed in a different (fixed) order, so everything behind them agrees to rounding."""
    M, N = 333, 1024
    g = torch.Generator(device="cpu").manual_seed(100 + S)
    part = torch.randn(S, M, N, generator=g).to(DEV)
    bias1 = torch.randn(N, generator=g).to(DEV)
    wT = (torch.randn(10, N, generator=g) * 0.1).to(DEV)
    b = (torch.randn(10, generator=g) * 0.1).to(DEV)
    labels = torch.randint(0, 10, (4 * M,), generator=g, dtype=torch.int32).to(DEV)
    idx = torch.randperm(4 * M, generator=g)[:M].to(torch.int32).to(DEV) if gather else None
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    outs = []
    try:
        for onepass in (0, 1):
            gpu_lib.dtf_head_set_onepass(onepass)
            h1 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            logits, lrows, crows = torch.empty(M, 10, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
            dlog, da = torch.empty(M, 10, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            check(gpu_lib.dtf_head_fused_fwd_bwd(ptr(part), S, ptr(bias1), 1.0, 9, ptr(state), ptr(h1), ptr(wT), ptr(b),
                                                 ptr(labels), ptr(idx) if gather else None, ptr(logits), ptr(lrows),
                                                 ptr(crows), ptr(dlog), ptr(da), M, N, 1.0 / M, 1.0, 1, stream()), "fused")
            torch.cuda.synchronize()
            outs.append((h1, logits, lrows, crows, dlog, da))
    finally:
        gpu_lib.dtf_head_set_onepass(-1)     # back to the default variant
    assert torch.equal(outs[0][0], outs[1][0])
    for k, tol in ((1, 1e-5), (2, 1e-5), (4, 1e-5), (5, 4e-3)):
        assert rel_err(outs[1][k], outs[0][k]) < tol
    assert float((outs[0][3] != outs[1][3]).float().mean()) < 0.01     # a tie in the logits may flip a row
    acc = part[0].clone()
    for s_ in range(1, S):
        acc += part[s_]
    assert torch.equal(outs[1][0], F.relu(acc + bias1).to(torch.bfloat16))
    lab = labels[idx.long()] if gather else labels[:M]
    z = outs[1][0].float() @ wT.t() + b
    assert rel_err(outs[1][1], z) < 1e-5
    assert torch.equal(outs[1][3], (outs[1][1].argmax(-1) == lab.long()).float())
    assert rel_err(outs[1][2], R.softmax_cross_entropy_with_logits(z, lab)) < 1e-5


# ------------------------------------------------------------------- optimizer
def test_reduce_segments(gpu_lib):
    a = torch.randn(5, 1000, device=DEV)
    b = torch.randn(3, 76, device=DEV)
    out = torch.zeros(1000 + 80, device=DEV)
    parts = (ctypes.c_void_p * 2)(a.data_ptr(), b.data_ptr())
    outs = (ctypes.c_void_p * 2)(out.data_ptr(), out.data_ptr() + 4 * 1000)
    G = (ctypes.c_int * 2)(5, 3)
    n = (ctypes.c_int * 2)(1000, 76)
    st = (ctypes.c_long * 2)(1000, 76)
    check(gpu_lib.dtf_reduce_segments(2, parts, outs, G, n, st, stream()), "reduce")
    assert torch.allclose(out[:1000], a.sum(0), atol=1e-5)
    assert torch.allclose(out[1000:1076], b.sum(0), atol=1e-5)
    assert float(out[1076:].abs().max()) == 0.0


@pytest.mark.parametrize("Gs", [(5, 33, 7), (200, 129, 513)])
def test_fused_reduce_sgd(gpu_lib, Gs):
    """One launch == reduce_segments + sgd_update + wc2 transpose, bitwise (the second case runs the
    eight-loads-in-flight loops of both reductions, which must keep the same summation order)."""
    c1, c2 = 32, 64
    H = 16
    n_head, n_dense, n_c2, n_c1 = H * 10 + 12 + H, 1000, 25 * c1 * c2 + c2, 40
    offs = [0, n_head, n_head + n_dense, n_head + n_dense + n_c2]
    total = offs[3] + n_c1
    g = torch.Generator(device="cpu").manual_seed(3)
    w0 = torch.randn(total, generator=g).to(DEV)
    grads = torch.randn(total, generator=g).to(DEV)
    parts = [torch.randn(G, n, generator=g).to(DEV) for G, n in zip(Gs, (n_head, n_c2, n_c1))]
    hyper = torch.tensor([0.1, 0.9, 0.999, 1e-8, 0.5, 0.0, 0.0, 0.0], device=DEV)
    # reference path
    w_ref, g_ref = w0.clone(), grads.clone()
    for k, (o, p) in enumerate(zip((offs[0], offs[2], offs[3]), parts)):
        pa = (ctypes.c_void_p * 1)(p.data_ptr()); oa = (ctypes.c_void_p * 1)(g_ref.data_ptr() + 4 * o)
        check(gpu_lib.dtf_reduce_segments(1, pa, oa, (ctypes.c_int * 1)(p.shape[0]), (ctypes.c_int * 1)(p.shape[1]),
                                          (ctypes.c_long * 1)(p.shape[1]), stream()), "reduce")
    sh_ref = torch.zeros(total, dtype=torch.bfloat16, device=DEV)
    check(gpu_lib.dtf_sgd_update(ptr(w_ref), ptr(g_ref), ptr(sh_ref), ptr(hyper), total, None, 0, 0, 0, stream()), "sgd")
    # fused path
    w1, g1 = w0.clone(), grads.clone()
    sh1 = torch.zeros(total, dtype=torch.bfloat16, device=DEV)
    wT = torch.zeros(2, 25 * c1 * c2, dtype=torch.bfloat16, device=DEV)
    owT = torch.zeros(10, H, device=DEV)
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    pa = (ctypes.c_void_p * 3)(*[p.data_ptr() for p in parts])
    check(gpu_lib.dtf_fused_reduce_sgd(3, pa, (ctypes.c_int * 3)(*Gs), (ctypes.c_int * 3)(n_head, n_c2, n_c1),
                                       (ctypes.c_long * 3)(n_head, n_c2, n_c1), (ctypes.c_long * 3)(offs[0], offs[2], offs[3]),
                                       ptr(w1), ptr(g1), ptr(sh1), ptr(hyper), offs[1], offs[2], ptr(wT), offs[2], c1, c2,
                                       ptr(owT), offs[0], H, ptr(state), 100, 1, 1, stream()), "fused")
    for o, p_ in zip((offs[0], offs[2], offs[3]), parts):     # and both agree with an fp64 sum of the slabs
        want = p_.double().sum(0)
        assert float((g1[o:o + p_.shape[1]].double() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
    assert torch.equal(owT, w_ref[:H * 10].reshape(H, 10).t().contiguous())
    assert torch.equal(g1, g_ref) and torch.equal(w1, w_ref) and torch.equal(sh1, sh_ref)
    f_ref, d_ref = R.conv2_pack_reference(sh_ref[offs[2]:offs[2] + 25 * c1 * c2].reshape(25, c1, c2))
    assert torch.equal(wT[0], f_ref) and torch.equal(wT[1], d_ref)
    assert state.tolist()[:3] == [100, 1, 1]


def test_fused_reduce_adam(gpu_lib):
    """One launch == reduce_segments + adam_update (whole arena) + wc2 packs + out_w transpose + state advance,
    bitwise, over three consecutive steps (the step counter is advanced by the last workgroup to finish)."""
    c1, c2 = 32, 64
    H = 16
    n_dense, n_head, n_c2, n_c1 = 40000, H * 10 + 12 + H, 25 * c1 * c2 + c2, 40
    offs = [0, n_dense, n_dense + n_head, n_dense + n_head + n_c2]     # arena = [dense | head | conv2 | conv1]
    total = offs[3] + n_c1
    g = torch.Generator(device="cpu").manual_seed(4)
    w0 = torch.randn(total, generator=g).to(DEV)
    hyper = torch.tensor([0.01, 0.9, 0.999, 1e-8, 0.5, 0.0, 0.0, 0.0], device=DEV)
    w_ref, w1 = w0.clone(), w0.clone()
    m_ref, v_ref, m1, v1 = (torch.zeros(total, device=DEV) for _ in range(4))
    sh_ref = torch.zeros(total, dtype=torch.bfloat16, device=DEV)
    sh1 = torch.zeros(total, dtype=torch.bfloat16, device=DEV)
    st_ref = torch.zeros(8, dtype=torch.int64, device=DEV)
    st1 = torch.zeros(8, dtype=torch.int64, device=DEV)
    wT = torch.zeros(2, 25 * c1 * c2, dtype=torch.bfloat16, device=DEV)
    owT = torch.zeros(10, H, device=DEV)
    for step in range(3):
        grads = torch.randn(total, generator=g).to(DEV)
        parts = [torch.randn(G, n, generator=g).to(DEV) for G, n in ((5, n_head), (33, n_c2), (7, n_c1))]
        g_ref, g1 = grads.clone(), grads.clone()
        for o, p in zip((offs[1], offs[2], offs[3]), parts):
            pa = (ctypes.c_void_p * 1)(p.data_ptr()); oa = (ctypes.c_void_p * 1)(g_ref.data_ptr() + 4 * o)
            check(gpu_lib.dtf_reduce_segments(1, pa, oa, (ctypes.c_int * 1)(p.shape[0]), (ctypes.c_int * 1)(p.shape[1]),
                                              (ctypes.c_long * 1)(p.shape[1]), stream()), "reduce")
        check(gpu_lib.dtf_adam_update(ptr(w_ref), ptr(g_ref), ptr(m_ref), ptr(v_ref), ptr(sh_ref), ptr(hyper), total,
                                      ptr(st_ref), 3, ptr(st_ref), 100, 1, 1, stream()), "adam")
        pa = (ctypes.c_void_p * 3)(*[p.data_ptr() for p in parts])
        check(gpu_lib.dtf_fused_reduce_adam(3, pa, (ctypes.c_int * 3)(5, 33, 7), (ctypes.c_int * 3)(n_head, n_c2, n_c1),
                                            (ctypes.c_long * 3)(n_head, n_c2, n_c1), (ctypes.c_long * 3)(offs[1], offs[2], offs[3]),
                                            ptr(w1), ptr(g1), ptr(m1), ptr(v1), ptr(sh1), ptr(hyper), offs[0], offs[1],
                                            ptr(wT), offs[2], c1, c2, ptr(owT), offs[1], H, ptr(st1), 3, 100, 1, 1, stream()),
              "fused adam")
        torch.cuda.synchronize()
        assert torch.equal(g1, g_ref)
        assert torch.equal(m1, m_ref) and torch.equal(v1, v_ref)
        assert torch.equal(w1, w_ref) and torch.equal(sh1, sh_ref)
        assert st1.tolist() == st_ref.tolist() == [100 * (step + 1), step + 1, step + 1, step + 1, 0, 0, 0, 0]
    assert torch.equal(owT, w_ref[offs[1]:offs[1] + H * 10].reshape(H, 10).t().contiguous())
    f_ref, d_ref = R.conv2_pack_reference(sh_ref[offs[2]:offs[2] + 25 * c1 * c2].reshape(25, c1, c2))
    assert torch.equal(wT[0], f_ref) and torch.equal(wT[1], d_ref)


def test_sgd_and_adam(gpu_lib):
    n = 4 * 1000 + 4
    w0 = torch.randn(n, device=DEV)
    g = torch.randn(n, device=DEV)
    hyper = torch.tensor([0.01, 0.9, 0.999, 1e-8, 0.5, 0, 0, 0], device=DEV)
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    w = w0.clone()
    sh = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    check(gpu_lib.dtf_sgd_update(ptr(w), ptr(g), ptr(sh), ptr(hyper), n, ptr(state), 128, 1, 1, stream()), "sgd")
    assert torch.allclose(w, w0 - 0.01 * 0.5 * g, atol=1e-6)
    assert torch.equal(sh, w.to(torch.bfloat16))
    assert state[:4].tolist() == [128, 1, 1, 0]
    # adam, 3 steps against torch.optim.Adam (grad pre-scaled by hyper[4])
    state.zero_()
    w = w0.clone()
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    # reference: TF AdamOptimizer update rule (epsilon outside the bias-corrected sqrt), fp64
    wr, mr, vr = w0.double().clone(), torch.zeros(n, dtype=torch.float64, device=DEV), torch.zeros(n, dtype=torch.float64, device=DEV)
    b1, b2 = float(hyper[1]), float(hyper[2])
    for t in range(1, 4):
        check(gpu_lib.dtf_adam_update(ptr(w), ptr(g), ptr(m), ptr(v), ptr(sh), ptr(hyper), n, ptr(state), 3,
                                      ptr(state), 0, 1, 1, stream()), "adam")
        ge = g.double() * 0.5
        mr = b1 * mr + (1 - b1) * ge
        vr = b2 * vr + (1 - b2) * ge * ge
        lr_t = 0.01 * (1 - b2 ** t) ** 0.5 / (1 - b1 ** t)
        wr = wr - lr_t * mr / (vr.sqrt() + 1e-8)
    assert float((w.double() - wr).abs().max()) < 2e-5
    assert torch.equal(sh, w.to(torch.bfloat16))
    assert int(state[3]) == 3 and int(state[1]) == 3
