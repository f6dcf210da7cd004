"""Multi-process plumbing on CPU (gloo): the reference's launch style -- one OS process per
task, discovered only through --ps_hosts/--worker_hosts (reference .idea/MNISTDist.py:94-107).
Covers BASELINE.json config #1 (1 ps + 2 workers, batch 100, sync), both async parameter-server
implementations, worker-loss tolerance and chief restart/resume."""
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "mnistdist.py")


def free_ports(n):
    socks, ports = [], []
    for _ in range(n):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        socks.append(s)
        ports.append(s.getsockname()[1])
    for s in socks:
        s.close()
    return ports


def launch(n_ps, n_workers, extra, logdir, per_task_extra=None, timeout=240, backend="cpu"):
    ports = free_ports(n_ps + n_workers)
    ps_hosts = ",".join(f"127.0.0.1:{p}" for p in ports[:n_ps])
    wk_hosts = ",".join(f"127.0.0.1:{p}" for p in ports[n_ps:])
    env = dict(os.environ, OMP_NUM_THREADS="2", MKL_NUM_THREADS="2")
    procs = []
    for job, n in (("ps", n_ps), ("worker", n_workers)):
        for i in range(n):
            cmd = [sys.executable, "-u", CLI, f"--job_name={job}", f"--task_index={i}", f"--ps_hosts={ps_hosts}",
                   f"--worker_hosts={wk_hosts}", f"--backend={backend}", "--synthetic", f"--logdir={logdir}",
                   "--comm_timeout=120"] + list(extra) + list((per_task_extra or {}).get((job, i), []))
            procs.append(((job, i), subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                                                     env=env, cwd=ROOT)))
    out = {}
    deadline = time.time() + timeout
    for key, p in procs:
        try:
            o, _ = p.communicate(timeout=max(1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            for _, q in procs:
                q.kill()
            o = p.communicate()[0]
            raise AssertionError(f"{key} timed out:\n{o}")
        out[key] = (p.returncode, o)
    return out


DISPLAY = re.compile(r"^job: worker/(\d+) step:  (\d+) mini_batch loss:  ([-\d.einf]+) training accuracy:  ([\d.]+)$", re.M)


def test_sync_1ps_2workers_batch100(tmp_path):
    logdir = str(tmp_path / "logs")
    out = launch(1, 2, ["--mode=sync", "--batch_size=100", "--steps=12", "--display_step=4",
                        "--learning_rate=0.01", "--noeval"], logdir)
    for key, (rc, o) in out.items():
        assert rc == 0, (key, o)
    assert "ps/0: done" in out[("ps", 0)][1]
    w0, w1 = out[("worker", 0)][1], out[("worker", 1)][1]
    for o in (w0, w1):
        assert "Optimization Finished!" in o
        steps = [int(m.group(2)) for m in DISPLAY.finditer(o)]
        assert steps == [0, 4, 8], o          # reference console format, every display_step
    l0 = [float(m.group(3)) for m in DISPLAY.finditer(w0)]
    assert l0[-1] < l0[0]
    # chief-only checkpoint with the final global step
    files = os.listdir(logdir)
    assert "checkpoint" in files and "model.ckpt-12.index" in files and "model.ckpt-12.data-00000-of-00001" in files
    from distributed_tensorflow_amd.checkpoint import read_bundle
    ck = read_bundle(os.path.join(logdir, "model.ckpt-12"))
    assert int(ck["Variable_8"]) == 12 and ck["Variable_2"].shape == (3136, 1024)


def test_async_p2p_two_ps_shards_and_global_step_semantics(tmp_path):
    logdir = str(tmp_path / "logs")
    out = launch(2, 2, ["--mode=async", "--async_impl=p2p", "--batch_size=32", "--steps=20", "--display_step=1",
                        "--learning_rate=0.01", "--noeval"], logdir)
    for key, (rc, o) in out.items():
        assert rc == 0, (key, o)
    # training_iter bounds the GLOBAL step: the two workers share the 20 updates
    steps = []
    for w in (0, 1):
        steps.append([int(m.group(2)) for m in DISPLAY.finditer(out[("worker", w)][1])])
    assert 1 <= len(steps[0]) < 20 and 1 <= len(steps[1]) < 20
    assert len(steps[0]) + len(steps[1]) in range(20, 23)
    allsteps = sorted(steps[0] + steps[1])
    assert allsteps[0] == 0 and allsteps[-1] <= 20


def test_async_broadcast_rounds(tmp_path):
    logdir = str(tmp_path / "logs")
    out = launch(1, 2, ["--mode=async", "--async_impl=broadcast", "--staleness=1", "--batch_size=32", "--steps=16",
                        "--display_step=2", "--learning_rate=0.01", "--noeval"], logdir)
    for key, (rc, o) in out.items():
        assert rc == 0, (key, o)
    w0 = out[("worker", 0)][1]
    steps = [int(m.group(2)) for m in DISPLAY.finditer(w0)]
    assert steps == [0, 2, 4, 6, 8, 10, 12, 14]       # global_step advances by #workers per round
    losses = [float(m.group(3)) for m in DISPLAY.finditer(w0)]
    assert losses[-1] < losses[0]
    assert "model.ckpt-16.index" in os.listdir(logdir)


def test_async_survives_worker_loss(tmp_path):
    logdir = str(tmp_path / "logs")
    out = launch(1, 2, ["--mode=async", "--async_impl=p2p", "--batch_size=16", "--steps=24", "--display_step=4",
                        "--learning_rate=0.01", "--noeval"], logdir,
                 per_task_extra={("worker", 1): ["--fault_inject_step=3"]})
    assert out[("worker", 1)][0] == 17                       # injected crash
    assert out[("worker", 0)][0] == 0, out[("worker", 0)][1]
    assert out[("ps", 0)][0] == 0, out[("ps", 0)][1]
    assert "Optimization Finished!" in out[("worker", 0)][1]
    assert "lost" in out[("ps", 0)][1]


def test_chief_restart_resumes_from_checkpoint(tmp_path):
    logdir = str(tmp_path / "logs")
    args = ["--synthetic", "--backend=cpu", "--batch_size=32", "--learning_rate=0.01", "--display_step=5",
            f"--logdir={logdir}", "--noeval"]
    r1 = subprocess.run([sys.executable, CLI, "--steps=10"] + args, capture_output=True, text=True, cwd=ROOT)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    assert "model.ckpt-10.index" in os.listdir(logdir)
    r2 = subprocess.run([sys.executable, CLI, "--steps=20"] + args, capture_output=True, text=True, cwd=ROOT)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    assert "Restored from checkpoint" in r2.stdout and "model.ckpt-10" in r2.stdout
    steps = [int(m.group(2)) for m in DISPLAY.finditer(r2.stdout)]
    assert steps == [10, 15]                                  # continues at the restored global_step
    assert "model.ckpt-20.index" in os.listdir(logdir)


def test_resumes_from_a_logdir_left_by_the_tf_0x_reference(tmp_path):
    """The reference's Supervisor wrote Saver-V1 files (`model.ckpt-N`, one sstable, no .index) under the names
    Variable..Variable_8.  Such a logdir must restore here, continue at its global_step, and the run may go on
    writing either format next to it (reference .idea/MNISTDist.py:154-165)."""
    import numpy as np
    from distributed_tensorflow_amd.checkpoint import Saver, bundle, to_tf_names
    from distributed_tensorflow_amd.models.spec import ModelSpec
    logdir = str(tmp_path / "train_logs")
    rng = np.random.RandomState(0)
    shapes = ModelSpec().shapes()
    params = {n: (rng.randn(*s) * 0.05).astype(np.float32) for n, s in shapes.items()}
    Saver(write_version=1).save(to_tf_names(params, global_step=40), os.path.join(logdir, "model.ckpt"), global_step=40)
    assert sorted(os.listdir(logdir)) == ["checkpoint", "model.ckpt-40"]
    assert bundle.checkpoint_format(os.path.join(logdir, "model.ckpt-40")) == 1
    args = ["--synthetic", "--backend=cpu", "--batch_size=16", "--learning_rate=0.01", "--display_step=5",
            f"--logdir={logdir}", "--noeval"]
    r = subprocess.run([sys.executable, CLI, "--steps=50", "--saver_write_version=1"] + args, capture_output=True,
                       text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Restored from checkpoint" in r.stdout and "model.ckpt-40" in r.stdout
    assert [int(m.group(2)) for m in DISPLAY.finditer(r.stdout)] == [40, 45]
    assert "model.ckpt-50" in os.listdir(logdir) and "model.ckpt-50.index" not in os.listdir(logdir)
    got = bundle.read_bundle(os.path.join(logdir, "model.ckpt-50"))
    assert int(got["Variable_8"]) == 50 and got["Variable_2"].shape == (3136, 1024)
    assert not np.array_equal(got["Variable_1"], params["wc2"])          # it trained on from the restored weights
    assert np.abs(got["Variable_1"] - params["wc2"]).max() < 0.05


def test_sync_aborts_cleanly_on_worker_loss_and_restart_resumes(tmp_path):
    """Synchronous mode cannot outlive a worker: when worker 1 dies hard mid-run, the surviving worker
    and the ps task notice within the heartbeat timeout, say so in one line and exit with status 3 -- they do not
    sit in a collective until its timeout.  A restarted job resumes from the last timed checkpoint."""
    logdir = str(tmp_path / "logs")
    common = ["--mode=sync", "--batch_size=16", "--display_step=5", "--learning_rate=0.01", "--noeval",
              "--save_model_secs=0.05", "--heartbeat_timeout=6"]
    t0 = time.time()
    out = launch(1, 2, common + ["--steps=400"], logdir, per_task_extra={("worker", 1): ["--fault_inject_step=12"]},
                 timeout=150)
    took = time.time() - t0
    assert out[("worker", 1)][0] == 17                                   # injected crash (os._exit, no goodbye)
    rc0, o0 = out[("worker", 0)]
    assert rc0 == 3, o0
    assert "lost contact with a peer" in o0 or "aborting" in o0, o0
    assert "resumes from the last checkpoint" in o0
    assert out[("ps", 0)][0] == 3, out[("ps", 0)][1]
    assert took < 100, took                                              # nowhere near --comm_timeout
    saved = sorted(int(f.split("-")[1].split(".")[0]) for f in os.listdir(logdir) if f.endswith(".index"))
    assert saved and saved[-1] <= 13, saved                              # timed saves happened before the loss
    # restart: resumes at the last checkpoint and finishes
    out2 = launch(1, 2, common + [f"--steps={saved[-1] + 10}"], logdir)
    for key, (rc, o) in out2.items():
        assert rc == 0, (key, o)
    assert f"model.ckpt-{saved[-1]}" in out2[("worker", 0)][1] and "Restored from checkpoint" in out2[("worker", 0)][1]
    assert f"model.ckpt-{saved[-1] + 10}.index" in os.listdir(logdir)


def test_sync_stop_request_is_job_wide(tmp_path):
    """sv.request_stop() on ONE synchronous worker (here a test hook; in production SIGTERM) stops EVERY worker
    after the same step, the chief checkpoints that step, and every task exits 0 (reference: sv.should_stop(),
    M:173 -- a worker that just left the loop would strand the other one in the next all-reduce)."""
    logdir = str(tmp_path / "logs")
    out = launch(1, 2, ["--mode=sync", "--batch_size=16", "--steps=500", "--display_step=1", "--learning_rate=0.01",
                        "--noeval"], logdir, per_task_extra={("worker", 1): ["--request_stop_step=7"]})
    for key, (rc, o) in out.items():
        assert rc == 0, (key, o)
    steps = [[int(m.group(2)) for m in DISPLAY.finditer(out[("worker", w)][1])] for w in (0, 1)]
    assert steps[0] == steps[1]
    assert 7 <= len(steps[0]) <= 9, steps                                # stopped right after the request, together
    assert f"model.ckpt-{len(steps[0])}.index" in os.listdir(logdir)
    assert "Optimization Finished!" in out[("worker", 0)][1]


def test_sync_adam_restore_keeps_replicas_identical(tmp_path):
    """A restored Adam run: the chief loads m / v / t from the checkpoint and must hand them to the other workers with
    the parameters, or the replicas apply different updates to the same gradient.  Each worker dumps its
    parameters at the end; they must be bitwise equal after training on from the restored state."""
    logdir = str(tmp_path / "logs")
    args = ["--mode=sync", "--optimizer=adam", "--batch_size=16", "--display_step=5", "--learning_rate=0.001", "--noeval"]
    out = launch(0, 2, args + ["--steps=6"], logdir)
    for key, (rc, o) in out.items():
        assert rc == 0, (key, o)
    dumps = {w: str(tmp_path / f"params_w{w}.npy") for w in (0, 1)}
    out = launch(0, 2, args + ["--steps=14"], logdir,
                 per_task_extra={("worker", w): [f"--dump_params={dumps[w]}"] for w in (0, 1)})
    for key, (rc, o) in out.items():
        assert rc == 0, (key, o)
    assert "Restored from checkpoint" in out[("worker", 0)][1]
    a, b = np.load(dumps[0]), np.load(dumps[1])
    assert a.shape == b.shape and np.array_equal(a, b)


def test_async_rounds_rejects_adam(tmp_path):
    out = launch(1, 1, ["--mode=async", "--async_impl=broadcast", "--optimizer=adam", "--steps=4", "--noeval"],
                 str(tmp_path / "logs"))
    assert any(rc != 0 and "SGD only" in o for rc, o in out.values()), out
