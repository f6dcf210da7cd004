"""CPU-only unit tests: flag surface, cluster/role resolution, model spec/arena, data pipeline,
sampler bijection, console format, oracle numerics, sync-DP bucket plan."""
import gzip
import os

import numpy as np
import pytest
import torch

from distributed_tensorflow_amd import flags as F
from distributed_tensorflow_amd.cluster import ClusterSpec, TaskInfo, replica_device_setter
from distributed_tensorflow_amd.data import batch_indices, feistel_permute, half_bits, synthetic_mnist
from distributed_tensorflow_amd.data import mnist as M
from distributed_tensorflow_amd.models.oracle import OracleModel
from distributed_tensorflow_amd.models.spec import ArenaLayout, ModelSpec, TF_VARIABLE_NAMES
from distributed_tensorflow_amd.ops import reference as R
from distributed_tensorflow_amd.utils.metrics import format_display_line


# ------------------------------------------------------------------------ flags
def _fresh_flags():
    fv = F.FlagValues()
    fv._define("data_dir", "/tmp/mnist-data", "", "string")
    fv._define("task_index", 0, "", "integer")
    fv._define("learning_rate", 0.001, "", "float")
    fv._define("training_iter", 10000, "", "integer")
    fv._define("synthetic", False, "", "boolean")
    fv._alias("steps", "training_iter")
    return fv


def test_flag_syntaxes_and_defaults():
    fv = _fresh_flags()
    rest = fv._parse(["--task_index=3", "--learning_rate", "0.5", "--steps", "77", "--synthetic", "pos", "--unknown=1"])
    assert fv.task_index == 3 and fv.learning_rate == 0.5 and fv.training_iter == 77 and fv.synthetic is True
    assert fv.steps == 77 and fv.data_dir == "/tmp/mnist-data"
    assert rest == ["pos", "--unknown=1"]
    assert fv.is_set("task_index") and not fv.is_set("data_dir")
    fv._reset()
    fv._parse(["--nosynthetic", "--synthetic=false"])
    assert fv.synthetic is False and fv.training_iter == 10000
    with pytest.raises(ValueError):
        fv._parse(["--task_index=abc"])
    with pytest.raises(ValueError):
        fv._parse(["--task_index"])


def test_cli_keeps_reference_flag_names_and_defaults():
    """reference .idea/MNISTDist.py:14-31"""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    saved = dict(F.FLAGS._flags), dict(F.FLAGS._aliases)
    F.FLAGS._flags.clear()
    F.FLAGS._aliases.clear()
    try:
        spec = importlib.util.spec_from_file_location("mnistdist_cli", os.path.join(root, "mnistdist.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        F.FLAGS._parse([])
        want = {"data_dir": "/tmp/mnist-data", "ps_hosts": "", "worker_hosts": "", "job_name": "", "task_index": 0,
                "hidden_units": 100, "batch_size": 128, "training_iter": 10000, "learning_rate": 0.001,
                "display_step": 100}
        got = F.FLAGS.flag_values_dict()
        for k, v in want.items():
            assert got[k] == v, k
        assert F.FLAGS.steps == 10000 and got["logdir"] == "/tmp/train_logs" and got["save_model_secs"] == 600.0
    finally:
        F.FLAGS._flags.clear()
        F.FLAGS._aliases.clear()
        F.FLAGS._flags.update(saved[0])
        F.FLAGS._aliases.update(saved[1])


# ---------------------------------------------------------------------- cluster
def test_cluster_roles_and_ranks():
    c = ClusterSpec.from_flags("h0:2222,h1:2222", "h2:2223,h3:2223,h4:2223")
    assert c.world_size == 5 and c.master() == ("h0", 2222)
    assert [c.rank_of("ps", i) for i in range(2)] == [0, 1]
    assert [c.rank_of("worker", i) for i in range(3)] == [2, 3, 4]
    assert c.worker_ranks() == [2, 3, 4] and c.ps_ranks() == [0, 1]
    t = TaskInfo(c, "worker", 0)
    assert t.is_chief and t.is_worker and not t.is_ps and t.num_workers == 3
    assert not TaskInfo(c, "worker", 1).is_chief and TaskInfo(c, "ps", 1).is_ps
    with pytest.raises(ValueError):
        c.rank_of("worker", 3)
    with pytest.raises(ValueError):
        c.rank_of("evaluator", 0)
    assert ClusterSpec.from_flags("", "a:1,b:2").master() == ("a", 1)
    # TF's default round-robin placement in creation order
    place = replica_device_setter(c, list(TF_VARIABLE_NAMES))
    assert place == {"wc1": 0, "wc2": 1, "wd1": 0, "out_w": 1, "bc1": 0, "bc2": 1, "bd1": 0, "out_b": 1,
                     "global_step": 0}


# ------------------------------------------------------------------ model spec
def test_spec_matches_reference_contract():
    s = ModelSpec()
    assert s.num_params() == 3274634
    fwd, fb = s.flops_per_image()
    assert round(fwd / 1e6, 2) == 27.77
    lay = ArenaLayout(s)
    assert lay.order[0] == "wd1" and lay.order[1] == "out_w"      # dominant gradient first, small tail after
    assert all(o % 4 == 0 for o in lay.offsets.values())
    arena = torch.arange(lay.total, dtype=torch.float32)
    assert lay.view(arena, "wd1").shape == (3136, 1024) and lay.view(arena, "wc1").shape == (5, 5, 1, 32)
    lo, hi = lay.segment("out_w", "bd1")
    assert (lo, hi) == (3136 * 1024, 3136 * 1024 + 1024 * 10 + 12 + 1024)
    w = ModelSpec.wide()
    assert (w.c1, w.c2, w.hidden, w.flat) == (128, 256, 4096, 12544)
    assert list(TF_VARIABLE_NAMES.values()) == ["Variable"] + [f"Variable_{i}" for i in range(1, 9)]


# ------------------------------------------------------------------------- data
def test_feistel_sampler_is_an_epoch_permutation():
    for n in (10, 1000, 55000):
        p = feistel_permute(np.arange(n), n, key=0xABCDEF)
        assert sorted(p.tolist()) == list(range(n))
        assert (1 << (2 * half_bits(n))) >= n
    a = batch_indices(0, 64, 1000, seed=3)
    b = batch_indices(64, 64, 1000, seed=3)
    assert len(set(a) & set(b)) == 0
    # crossing an epoch boundary reshuffles
    e0 = batch_indices(0, 1000, 1000, seed=3)
    e1 = batch_indices(1000, 1000, 1000, seed=3)
    assert sorted(e0) == sorted(e1) and not np.array_equal(e0, e1)
    c = batch_indices(990, 20, 1000, seed=3)
    assert np.array_equal(c[:10], e0[990:]) and np.array_equal(c[10:], e1[:10])
    assert np.array_equal(batch_indices(5, 4, 10, seed=0, shuffle=False), [5, 6, 7, 8])


def test_mnist_idx_reader_and_next_batch_semantics(tmp_path):
    rng = np.random.RandomState(0)
    d = str(tmp_path)
    tr_x = rng.randint(0, 256, (300, 28, 28)).astype(np.uint8)
    tr_y = rng.randint(0, 10, 300).astype(np.uint8)
    te_x = rng.randint(0, 256, (50, 28, 28)).astype(np.uint8)
    te_y = rng.randint(0, 10, 50).astype(np.uint8)
    M.write_idx_images(os.path.join(d, M.TRAIN_IMAGES), tr_x)
    M.write_idx_labels(os.path.join(d, M.TRAIN_LABELS), tr_y)
    M.write_idx_images(os.path.join(d, M.TEST_IMAGES), te_x, gz=False)
    M.write_idx_labels(os.path.join(d, M.TEST_LABELS), te_y, gz=False)
    ds = M.read_data_sets(d, one_hot=True, validation_size=100, seed=1)
    assert not ds.synthetic
    assert ds.train.num_examples == 200 and ds.validation.num_examples == 100 and ds.test.num_examples == 50
    assert ds.train.images.shape == (200, 784) and ds.train.images.max() <= 1.0 and ds.train.labels.shape == (200, 10)
    np.testing.assert_allclose(ds.test.images[3], te_x[3].reshape(-1) / 255.0)
    assert np.array_equal(ds.test.labels.argmax(1), te_y)
    seen = []
    for _ in range(3):                      # 3 * 80 = 240 > 200: crosses an epoch boundary
        x, y = ds.train.next_batch(80)
        assert x.shape == (80, 784) and y.shape == (80, 10)
        seen.append(x)
    assert ds.train.epochs_completed == 1
    first_epoch = np.concatenate(seen)[:200]
    assert len({r.tobytes() for r in first_epoch}) == 200     # every example exactly once per epoch
    with pytest.raises(ValueError):
        with gzip.open(os.path.join(d, "bad.gz"), "wb") as f:
            f.write(b"\0" * 32)
        M.read_idx_images(os.path.join(d, "bad"))
    with pytest.raises(FileNotFoundError):
        M.read_data_sets(str(tmp_path / "nope"), allow_synthetic=False)
    assert M.read_data_sets(str(tmp_path / "nope"), synthetic_sizes=(600, 100)).synthetic


def test_synthetic_data_is_bf16_exact_and_learnable_shape():
    x, y = synthetic_mnist(256, seed=5)
    assert x.shape == (256, 784) and y.shape == (256,) and set(np.unique(y)) <= set(range(10))
    assert np.array_equal(torch.tensor(x).to(torch.bfloat16).float().numpy(), x)
    x2, y2 = synthetic_mnist(256, seed=5)
    assert np.array_equal(x, x2) and np.array_equal(y, y2)


# ---------------------------------------------------------------------- console
def test_display_line_is_the_reference_print():
    # reference M:183-186 under Python 3: print("job: %s/%s" % ..., "step: ", step, "mini_batch loss: ", loss, ...)
    assert format_display_line("worker", 0, 100, 2.5, 0.125) == \
        "job: worker/0 step:  100 mini_batch loss:  2.5 training accuracy:  0.125"


# ----------------------------------------------------------------------- oracle
def test_oracle_matches_nchw_torch_model():
    torch.manual_seed(0)
    m = OracleModel(seed=1)
    x = torch.rand(4, 784)
    z = m.logits(x)
    p = m.params
    # same network expressed the usual PyTorch way (NCHW, OIHW), flatten permuted to HWC
    import torch.nn.functional as Fn
    h = Fn.max_pool2d(Fn.relu(Fn.conv2d(x.view(4, 1, 28, 28), p["wc1"].permute(3, 2, 0, 1), p["bc1"], padding=2)), 2)
    h = Fn.max_pool2d(Fn.relu(Fn.conv2d(h, p["wc2"].permute(3, 2, 0, 1), p["bc2"], padding=2)), 2)
    h = h.permute(0, 2, 3, 1).reshape(4, -1)
    z2 = Fn.relu(h @ p["wd1"] + p["bd1"]) @ p["out_w"] + p["out_b"]
    assert torch.allclose(z, z2, atol=1e-4)
    # TF init statistics: truncated normal sigma 0.1 (|w| <= 0.2), biases 0.1
    assert float(p["wd1"].abs().max()) <= 0.2 and abs(float(p["wd1"].std()) - 0.0880) < 2e-3
    assert torch.all(p["bd1"] == 0.1)
    onehot = torch.eye(10)[torch.tensor([1, 2, 3, 4])]
    l1 = R.softmax_cross_entropy_with_logits(z, onehot)
    l2 = R.softmax_cross_entropy_with_logits(z, torch.tensor([1, 2, 3, 4]))
    assert torch.allclose(l1, l2)
    pooled, arg = R.pool_argmax(torch.tensor([[[[1.0], [1.0]], [[0.5], [1.0]]]]))
    assert float(pooled) == 1.0 and int(arg) == 0             # first max wins


def test_oracle_sgd_and_adam_learn():
    x, y = synthetic_mnist(128)
    x, y = torch.tensor(x), torch.tensor(y)
    for opt, lr in (("sgd", 0.01), ("adam", 1e-3)):
        m = OracleModel(optimizer=opt, learning_rate=lr, seed=2)
        first = m.train_step(x, y)[0]
        for _ in range(15):
            last = m.train_step(x, y)[0]
        assert last < 0.5 * first and m.global_step == 16


def test_bucket_plan_covers_arena_in_backward_order():
    from distributed_tensorflow_amd.parallel.sync_dp import make_buckets
    lay = ArenaLayout(ModelSpec())
    n_dense = lay.segment("wd1", "wd1")[1]
    b = make_buckets(0, n_dense, 1 << 20)
    assert b[0][0] == 0 and b[-1][1] == n_dense and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
    assert len(b) == 4 and max(hi - lo for lo, hi in b) <= 1 << 20


def test_conv2_pack_reference_matches_the_documented_index_formula():
    """The fragment-ordered conv2 weight layout (csrc/kernels/conv2.hip pack_conv2_weights) as an explicit
    index formula vs the reshape/permute reference used by the GPU tests."""
    from distributed_tensorflow_amd.ops.reference import conv2_pack_reference
    cin, cout = 32, 64
    w = torch.arange(25 * cin * cout, dtype=torch.float32).reshape(25, cin, cout)
    f, d = conv2_pack_reference(w)
    g = torch.Generator().manual_seed(0)
    for _ in range(200):
        tap, ci, co = (int(torch.randint(0, n, (1,), generator=g)) for n in (25, cin, cout))
        fi = ((tap * (cin // 32) + ci // 32) * (cout // 16) + co // 16) * 512 + (((ci % 32) // 8) * 16 + co % 16) * 8 + ci % 8
        di = ((tap * (cout // 32) + co // 32) * (cin // 16) + ci // 16) * 512 + (((co % 32) // 8) * 16 + ci % 16) * 8 + co % 8
        assert f[fi] == w[tap, ci, co] and d[di] == w[tap, ci, co]
    assert sorted(f.tolist()) == sorted(w.reshape(-1).tolist())   # a permutation
