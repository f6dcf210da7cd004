"""End-to-end HipEngine checks on a real MI355X: one step vs the fp32 oracle,
convergence on synthetic data, HIP-graph replay == eager launches."""
import numpy as np
import pytest
import torch

from distributed_tensorflow_amd.data import batch_indices
from distributed_tensorflow_amd.models.oracle import OracleModel
from distributed_tensorflow_amd.models.spec import ModelSpec

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.fixture(scope="module")
def dataset(gpu_lib):
    from distributed_tensorflow_amd.engine import DeviceDataset
    return DeviceDataset.synthetic(4096, seed=1234)


@pytest.mark.parametrize("B", [100, 256])
def test_step_matches_oracle(gpu_lib, dataset, B):
    from distributed_tensorflow_amd.engine import HipEngine
    spec = ModelSpec()
    eng = HipEngine(spec, B, dataset, optimizer="sgd", learning_rate=0.01, seed=5)
    eng.init_params()
    # the oracle rounds to bf16 where the kernels store bf16 (tests/test_engine_numerics_gpu.py derives the bounds)
    from distributed_tensorflow_amd.models.oracle import Bf16EmulatedOracle
    oracle = Bf16EmulatedOracle(spec, device="cuda", optimizer="sgd", learning_rate=0.01)
    p0 = eng.export_params()
    oracle.load_params(p0)
    idx = batch_indices(0, B, dataset.n, 5)
    x = dataset.images[torch.as_tensor(idx, device="cuda")].float()
    y = dataset.labels[torch.as_tensor(idx, device="cuda")]
    loss_ref, acc_ref = oracle.compute_gradients(x, y)

    eng.launch_train_step()
    torch.cuda.synchronize()
    assert np.array_equal(eng.idx.cpu().numpy().astype(np.int64), idx)
    loss, acc = eng.last_metrics()
    assert abs(loss - loss_ref) / abs(loss_ref) < 5e-5, (loss, loss_ref)
    assert abs(acc - acc_ref) <= 1.0 / B
    g_ref, g = oracle.grads(), eng.export_grads()
    for name in eng.layout.names():
        e = rel_err(g[name], g_ref[name])
        assert e < 8e-4, (name, e)
    assert eng.global_step == 1
    assert int(eng.state[0]) == B
    # update applied: the parameter deltas agree as well as the gradients do
    oracle.apply_gradients()
    p_ref, p = oracle.export_params(), eng.export_params()
    for name in eng.layout.names():
        assert rel_err(p[name] - p0[name], p_ref[name] - p0[name]) < 8e-4, name
    # bf16 shadows/transposes follow the master weights
    assert torch.equal(eng.shadow, eng.params.to(torch.bfloat16))
    wc2 = eng.pview("wc2").reshape(25, spec.c1, spec.c2)
    from distributed_tensorflow_amd.ops.reference import conv2_pack_reference
    f_ref, d_ref = conv2_pack_reference(wc2.to(torch.bfloat16))
    assert torch.equal(eng.wc2T[0], f_ref) and torch.equal(eng.wc2T[1], d_ref)
    assert torch.equal(eng.out_wT, eng.pview("out_w").reshape(spec.hidden, 10).t().contiguous())


def test_training_converges_and_graph_replay(gpu_lib, dataset):
    from distributed_tensorflow_amd.engine import HipEngine
    spec = ModelSpec()

    def run(graph):
        eng = HipEngine(spec, 128, dataset, optimizer="sgd", learning_rate=0.01, seed=3)
        eng.init_params(seed=11)
        if graph:
            eng.capture()
            assert eng.global_step == 0
        losses = []
        for _ in range(60):
            eng.train_step()
            losses.append(eng.last_metrics()[0])
        return eng, losses

    e0, l0 = run(False)
    assert l0[-1] < 0.5 * l0[0], (l0[0], l0[-1])
    assert e0.global_step == 60
    # evaluation agrees with the (bf16-storage-emulating) oracle holding the same weights on the same images
    from distributed_tensorflow_amd.models.oracle import Bf16EmulatedOracle
    emu = Bf16EmulatedOracle(spec, device="cuda")
    emu.load_params(e0.export_params())
    l_ref, a_ref = emu.loss_and_accuracy(dataset.images[:1000].float(), dataset.labels[:1000])
    loss, acc = e0.evaluate(dataset.images[:1000], dataset.labels[:1000])
    assert abs(acc - float(a_ref)) <= 2.0 / 1000 and abs(loss - float(l_ref)) <= 2e-3 * max(1.0, float(l_ref)), \
        (loss, acc, float(l_ref), float(a_ref))
    e1, l1 = run(True)
    assert e1.global_step == 60
    np.testing.assert_allclose(l1, l0, rtol=1e-5, atol=1e-6)
    assert torch.equal(e0.params, e1.params)  # split-K slabs, no float atomics -> bitwise reproducible


def test_fused_dense_update_is_bitwise_the_same_training(gpu_lib, dataset):
    """fc1 SGD update applied inside the wgrad GEMM epilogue == gradient arena + optimizer kernel."""
    from distributed_tensorflow_amd.engine import HipEngine
    engines = []
    for fuse in (False, True):
        eng = HipEngine(ModelSpec(), 96, dataset, optimizer="sgd", learning_rate=0.01, seed=3, fuse_dense_update=fuse)
        eng.init_params(seed=13)
        if fuse:
            eng.capture()
        for _ in range(12):
            eng.train_step()
        engines.append(eng)
    a, b = engines
    assert a.global_step == b.global_step == 12
    assert torch.equal(a.params, b.params) and torch.equal(a.shadow, b.shadow) and torch.equal(a.wc2T, b.wc2T)
    assert torch.equal(a.out_wT, b.out_wT)
    assert a.last_metrics() == b.last_metrics()


def test_adam_and_dropout_train(gpu_lib, dataset):
    from distributed_tensorflow_amd.engine import HipEngine
    eng = HipEngine(ModelSpec(), 128, dataset, optimizer="adam", learning_rate=1e-3, keep_prob=0.5, seed=9)
    eng.init_params()
    first = None
    for i in range(80):
        eng.train_step()
        if i == 0:
            first = eng.last_metrics()[0]
    last = eng.last_metrics()[0]
    assert last < 0.3 * first, (first, last)
    from distributed_tensorflow_amd.models.oracle import Bf16EmulatedOracle
    emu = Bf16EmulatedOracle(ModelSpec(), device="cuda")
    emu.load_params(eng.export_params())
    l_ref, a_ref = emu.loss_and_accuracy(dataset.images[:1024].float(), dataset.labels[:1024])
    loss, acc = eng.evaluate(dataset.images[:1024], dataset.labels[:1024])
    assert abs(acc - float(a_ref)) <= 2.0 / 1024 and abs(loss - float(l_ref)) <= 2e-3 * max(1.0, float(l_ref)), \
        (loss, acc, float(l_ref), float(a_ref))


def test_sync_dp_world1_rccl_matches_plain_engine(gpu_lib, dataset):
    """The bucketed/overlapped all-reduce path over RCCL (world = 1 on the single test GPU: RCCL
    cannot host two ranks on one device) must be a no-op numerically."""
    import os
    import torch.distributed as dist
    from distributed_tensorflow_amd.engine import HipEngine
    from distributed_tensorflow_amd.parallel import comm
    from distributed_tensorflow_amd.parallel.sync_dp import SyncDataParallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29611"
    comm.init_process_group("nccl", 0, 1, device=torch.device("cuda", 0))
    try:
        a = HipEngine(ModelSpec(), 128, dataset, learning_rate=0.01, seed=4)
        b = HipEngine(ModelSpec(), 128, dataset, learning_rate=0.01, seed=4)
        a.init_params(seed=2)
        b.init_params(seed=2)
        tail_group = dist.new_group([0])
        dp = SyncDataParallel(b, bucket_mb=1.0, tail_group=tail_group)
        assert len(dp.dense_buckets) >= 12 and dp.tail_bucket[1] == b.layout.total
        dp.broadcast_parameters()
        for _ in range(5):
            a.launch_train_step()
            dp.train_step()
        dp.flush()
        torch.cuda.synchronize()
        assert torch.equal(a.params, b.params)
        assert torch.equal(a.shadow, b.shadow) and torch.equal(a.wc2T, b.wc2T)
        assert a.global_step == b.global_step == 5
        # one communicator only (bench.py's fallback): the tail collective rides the comm stream behind the dense one
        c = HipEngine(ModelSpec(), 128, dataset, learning_rate=0.01, seed=4)
        c.init_params(seed=2)
        dp1 = SyncDataParallel(c)
        assert dp1.tail_stream is dp1.comm_stream
        dp1.broadcast_parameters()
        for _ in range(5):
            dp1.train_step()
        dp1.flush()
        torch.cuda.synchronize()
        assert torch.equal(a.params, c.params) and torch.equal(a.wc2T, c.wc2T)
        # Adam: per-range step counters keep the deferred dense update and the tail update in step
        a2 = HipEngine(ModelSpec(), 64, dataset, optimizer="adam", learning_rate=1e-3, seed=4)
        b2 = HipEngine(ModelSpec(), 64, dataset, optimizer="adam", learning_rate=1e-3, seed=4)
        a2.init_params(seed=2)
        b2.init_params(seed=2)
        dp2 = SyncDataParallel(b2, tail_group=tail_group)
        for _ in range(4):
            a2.launch_train_step()
            dp2.train_step()
        dp2.flush()
        torch.cuda.synchronize()
        assert torch.allclose(a2.params, b2.params, rtol=0, atol=1e-6)
        # grad_comm="bf16" (opt-in): the fc1 weight gradient is produced, communicated and consumed in bf16.  First step:
        # exactly the fp32 path's gradient rounded once; everything outside wd1 untouched.  Then 150 steps whose loss
        # curve stays with the fp32-gradient run.
        e32 = HipEngine(ModelSpec(), 128, dataset, learning_rate=0.01, seed=4, keep_prob=0.75)
        e16 = HipEngine(ModelSpec(), 128, dataset, learning_rate=0.01, seed=4, keep_prob=0.75)
        e32.init_params(seed=2)
        e16.init_params(seed=2)
        dp32 = SyncDataParallel(e32, tail_group=tail_group)
        dp16 = SyncDataParallel(e16, tail_group=tail_group, grad_comm="bf16")
        w0 = e16.params.clone()
        nd = dp16.n_dense
        dp32.train_step(); dp32.flush()
        dp16.train_step(); dp16.flush()
        torch.cuda.synchronize()
        g = e32.grads[0:nd]
        assert float(g.abs().max()) > 0
        assert torch.equal(e16.dense_g16, g.to(torch.bfloat16))
        assert torch.allclose(e16.params[0:nd], w0[0:nd] - 0.01 * g.to(torch.bfloat16).float(), rtol=0, atol=5e-8)
        assert torch.equal(e16.params[nd:], e32.params[nd:])
        assert torch.equal(e16.shadow[0:nd], e16.params[0:nd].to(torch.bfloat16))
        l0 = e32.last_metrics()[0]
        curve = []
        for i in range(150):
            dp32.train_step()
            dp16.train_step()
            if i % 10 == 9:
                curve.append((e32.last_metrics()[0], e16.last_metrics()[0]))
        dp32.flush(); dp16.flush()
        assert curve[-1][0] < 0.5 * l0, (l0, curve)
        for l32, l16 in curve:
            assert abs(l16 - l32) <= 0.1 * max(l32, 0.05 * l0), curve
        with pytest.raises(ValueError):
            SyncDataParallel(HipEngine(ModelSpec(), 32, None, dtype="fp32"), tail_group=tail_group, grad_comm="bf16")
    finally:
        dist.destroy_process_group()


def test_cli_trains_on_gpu(gpu_lib, tmp_path):
    """mnistdist.py end to end on the MI355X backend: console format, checkpoint, test accuracy."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    logdir = str(tmp_path / "logs")
    r = subprocess.run([sys.executable, os.path.join(root, "mnistdist.py"), "--synthetic", "--backend=hip",
                        "--steps=300", "--batch_size=128", "--learning_rate=0.01", "--display_step=100",
                        f"--logdir={logdir}", f"--metrics_file={tmp_path}/m.jsonl"],
                       capture_output=True, text=True, cwd=root, timeout=280)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = re.findall(r"^job: worker/0 step:  (\d+) mini_batch loss:  ([\d.e-]+) training accuracy:  ([\d.]+)$",
                       r.stdout, re.M)
    assert [int(l[0]) for l in lines] == [0, 100, 200]
    acc = float(re.search(r"test accuracy: ([\d.]+)", r.stdout).group(1))
    assert acc > 0.95, r.stdout
    assert "Optimization Finished!" in r.stdout
    from distributed_tensorflow_amd.checkpoint import read_bundle
    ck = read_bundle(os.path.join(logdir, "model.ckpt-300"))
    assert int(ck["Variable_8"]) == 300 and ck["Variable_1"].shape == (5, 5, 32, 64)


def test_wide_model_step_matches_oracle(gpu_lib, dataset):
    """BASELINE.json config #5: widened CNN (conv 128/256, fc4096) through the same kernels."""
    from distributed_tensorflow_amd.engine import HipEngine
    spec = ModelSpec.wide()
    B = 24
    eng = HipEngine(spec, B, dataset, optimizer="sgd", learning_rate=0.001, seed=8)
    eng.init_params()
    from distributed_tensorflow_amd.models.oracle import Bf16EmulatedOracle
    oracle = Bf16EmulatedOracle(spec, device="cuda", optimizer="sgd", learning_rate=0.001)
    oracle.load_params(eng.export_params())
    idx = torch.as_tensor(batch_indices(0, B, dataset.n, 8), device="cuda")
    loss_ref, _ = oracle.compute_gradients(dataset.images[idx].float(), dataset.labels[idx])
    eng.launch_train_step()
    torch.cuda.synchronize()
    loss, _ = eng.last_metrics()
    assert abs(loss - loss_ref) / abs(loss_ref) < 1e-4, (loss, loss_ref)
    g_ref, g = oracle.grads(), eng.export_grads()
    for name in eng.layout.names():
        e = rel_err(g[name], g_ref[name])
        assert e < 2e-3, (name, e)


def test_background_checkpoint_does_not_stall_training(gpu_lib, tmp_path):
    """Supervisor.save() on the GPU backend: a side-stream snapshot + background write (reference: the Supervisor's
    checkpoint thread, M:163-165).  The training thread spends far less than a disk write inside save(), training
    continues while the file is written, and the checkpoint holds exactly the state at the step it was taken."""
    import os
    import time
    from distributed_tensorflow_amd.backends import make_backend
    from distributed_tensorflow_amd.checkpoint import read_bundle
    from distributed_tensorflow_amd.data.mnist import read_data_sets
    from distributed_tensorflow_amd.parallel.strategies import LocalStrategy
    from distributed_tensorflow_amd.supervisor import Supervisor
    data = read_data_sets("/nonexistent", one_hot=False, synthetic=True, seed=3)
    for optimizer in ("sgd", "adam"):
        logdir = str(tmp_path / f"logs_{optimizer}")
        b = make_backend("hip", ModelSpec(), 256, data, optimizer=optimizer, learning_rate=1e-3, seed=1, use_graph=False)
        sv = Supervisor(b, is_chief=True, logdir=logdir, save_model_secs=0, log=lambda *_: None)
        strategy = LocalStrategy(b)
        with sv.managed_session(strategy, final_save=False):
            for _ in range(10):
                strategy.step()
            t0 = time.perf_counter()
            sv.save()                                # the first save also allocates the pinned snapshot buffers
            t_first = time.perf_counter() - t0
            sv._writer.wait()                        # (one snapshot buffer: a save waits for the previous write)
            for _ in range(20):
                strategy.step()
            b.synchronize()
            want = {k: v.copy() for k, v in b.export_params().items()}
            want_opt = b.export_optimizer_state()
            t0 = time.perf_counter()
            sv.save()
            t_save = time.perf_counter() - t0
            for _ in range(50):                      # keeps training while the writer thread works
                strategy.step()
            b.synchronize()
        # snapshot + enqueue only (13-39 MB of D2H + file I/O happen elsewhere)
        assert t_save < 0.1, t_save
        assert t_first < 2.0, t_first
        ck = read_bundle(os.path.join(logdir, "model.ckpt-30"))
        assert int(ck["Variable_8"]) == 30
        from distributed_tensorflow_amd.models.spec import TF_VARIABLE_NAMES
        for name, arr in want.items():
            assert np.array_equal(ck[TF_VARIABLE_NAMES[name]], arr), name      # the state of step 30, not of step 80
        if optimizer == "adam":
            assert np.array_equal(ck[TF_VARIABLE_NAMES["wd1"] + "/Adam"], want_opt["m"]["wd1"])
            assert np.array_equal(ck[TF_VARIABLE_NAMES["wc2"] + "/Adam_1"], want_opt["v"]["wc2"])
        assert b.get_global_step() == 80


def test_evaluate_matches_oracle_with_one_sync(gpu_lib, dataset):
    """K9: evaluate() enqueues every chunk's forward + metrics reduction and synchronises once; its loss / accuracy equal
    the fp32 oracle's on the same (bf16-rounded) weights within bf16 tolerance, for a row count that is not a multiple
    of the engine batch."""
    from distributed_tensorflow_amd.engine import HipEngine
    eng = HipEngine(ModelSpec(), 128, dataset, optimizer="sgd", learning_rate=0.01, seed=3)
    eng.init_params(seed=11)
    for _ in range(40):
        eng.train_step()
    oracle = OracleModel(ModelSpec(), device="cuda")
    oracle.load_params({k: v.to(torch.bfloat16).float() for k, v in eng.export_params().items()})
    n = 1000
    l_ref, a_ref = oracle.loss_and_accuracy(dataset.images[:n].float(), dataset.labels[:n])
    loss, acc = eng.evaluate(dataset.images[:n], dataset.labels[:n])
    assert abs(loss - float(l_ref)) < 2e-2 * max(1.0, abs(float(l_ref)))
    assert abs(acc - float(a_ref)) <= 0.01
    # the reduction itself: exact against the row buffers of the last chunk
    m_loss, m_acc = eng.last_metrics()
    assert abs(m_loss - float(eng.loss_rows.double().mean())) < 1e-6
    assert abs(m_acc - float(eng.correct_rows.double().mean())) < 1e-9


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_ragged_evaluation_chunk_keeps_fc1_partials_in_bounds(gpu_lib, dtype):
    """Evaluation walks a dataset in chunks of the engine batch; the last chunk is ragged.  With 4096 rows per chunk
    the fc1 forward GEMM runs unsplit, but a 3000-row chunk has few enough tiles that the split-K heuristic alone
    would pick two splits of 3000 rows -- more than the 4096-row partial buffer holds (found by a memory fault in
    tools/lr_scaling_sweep.py).  The executor caps the split count; the result must match the oracle."""
    from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
    from distributed_tensorflow_amd.models.oracle import Bf16EmulatedOracle
    ds = DeviceDataset.synthetic(4096 + 3000, seed=99, dtype=dtype)
    eng = HipEngine(ModelSpec(), 4096, ds, optimizer="sgd", learning_rate=0.01, seed=2, dtype=dtype)
    eng.init_params(seed=5)
    for _ in range(3):
        eng.train_step()
    guard = eng.logits.clone()          # allocated right behind fc1_part: an overrun would land in a neighbour
    loss, acc = eng.evaluate(ds.images, ds.labels)
    oracle = (OracleModel if dtype == "fp32" else Bf16EmulatedOracle)(ModelSpec(), device="cuda")
    oracle.load_params(eng.export_params())
    tl = ta = 0.0
    for lo in range(0, ds.n, 1024):
        l, a = oracle.loss_and_accuracy(ds.images[lo:lo + 1024].float(), ds.labels[lo:lo + 1024])
        k = min(1024, ds.n - lo)
        tl, ta = tl + float(l) * k, ta + float(a) * k
    assert abs(acc - ta / ds.n) <= 3.0 / ds.n, (acc, ta / ds.n)
    assert abs(loss - tl / ds.n) <= 2e-3 * max(1.0, tl / ds.n), (loss, tl / ds.n)
    assert guard.shape == eng.logits.shape
    # and a chunk larger than the engine batch is refused, not run
    with pytest.raises(RuntimeError):
        eng.launch_forward(ds.images, ds.labels, None, backward=False, n=4097, train_mode=False)
