"""Rank body of test_distributed_gpu.py::test_async_push_single_worker_equals_local_training (not a test module).

rank 0 = ps, rank 1 = the only worker, both on cuda:0 (mailboxes mapped through hipIpc on the same device).  With ONE
worker and staleness 0 the asynchronous parameter server is plain sequential training, so the worker replays every
step on a second, local engine and the two must agree bit for bit: compute copy after every step, fp32 master at the end."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.backends import HipBackend  # noqa: E402
from distributed_tensorflow_amd.cluster import ClusterSpec, TaskInfo  # noqa: E402
from distributed_tensorflow_amd.data.mnist import read_data_sets  # noqa: E402
from distributed_tensorflow_amd.engine import HipEngine  # noqa: E402
from distributed_tensorflow_amd.models.spec import ModelSpec  # noqa: E402
from distributed_tensorflow_amd.parallel import comm  # noqa: E402
from distributed_tensorflow_amd.parallel.async_push import AsyncPushServer, AsyncPushWorker  # noqa: E402

rank, port, optimizer, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
STEPS, B, LR = 12, 64, (0.01 if optimizer == "sgd" else 1e-3)
torch.cuda.set_device(0)
comm.init_process_group("gloo", rank, 2, "127.0.0.1", port, timeout_s=90)
cluster = ClusterSpec({"ps": [f"127.0.0.1:{port}"], "worker": [f"127.0.0.1:{port + 1}"]})
task = TaskInfo(cluster, "ps" if rank == 0 else "worker", 0)
data = read_data_sets("", synthetic=True, synthetic_sizes=(4096, 256))
backend = HipBackend(ModelSpec(), B, data, optimizer=optimizer, learning_rate=LR, seed=11, dtype=dtype)
backend.init_params(5 + rank)          # the worker starts from other weights: the initial broadcast has to fix that

if rank == 0:
    import time
    t_start = time.monotonic()          # no liveness plane in this script: a worker that died is given up on after 60 s
    srv = AsyncPushServer(backend, task, chief_rank=1, training_iter=STEPS, staleness=0, log=print,
                          peer_lost=lambda r: time.monotonic() - t_start > 60.0)
    srv.broadcast_initial()
    step = srv.serve()
    assert step == STEPS and srv.applied == STEPS, (step, srv.applied)
    assert srv.staleness_hist == {0: STEPS}, srv.staleness_hist      # one worker, pull-after-push: never stale
    srv.fab.close()
    print("APUSH rank 0 ok", flush=True)
else:
    eng = backend.engine
    wk = AsyncPushWorker(backend, task, chief_rank=1, staleness=0, timeout_s=20.0)
    backend.init_params(5)             # chief == this worker: its weights are the job's
    wk.broadcast_initial()
    ref = HipEngine(ModelSpec(), B, backend.train_ds, optimizer=optimizer, learning_rate=LR, seed=11, dtype=dtype)
    ref.init_params(5)
    assert torch.equal(ref.params, eng.params)
    for s in range(STEPS):
        got = wk.step()
        ref.launch_sample()
        ref.launch_forward(backend.train_ds.images, backend.train_ds.labels, ref.idx, backward=True)
        ref.launch_backward_dense()
        ref.launch_backward_conv(backend.train_ds.images, ref.idx)
        ref.launch_update()
        torch.cuda.synchronize()
        assert got == s + 1 == ref.global_step, (got, s, ref.global_step)
        assert eng.last_metrics() == ref.last_metrics(), (s, eng.last_metrics(), ref.last_metrics())
        if dtype == "bf16":
            assert torch.equal(eng.shadow, ref.shadow), f"compute copy differs after step {s}"
            assert torch.equal(eng.wc2T, ref.wc2T) and torch.equal(eng.out_wT, ref.out_wT)
            lo = wk.fab.n_dense
            assert torch.equal(eng.params[lo:], ref.params[lo:])
            if s == STEPS - 1:
                # the worker's fp32 copy of wd1 is NOT maintained step by step (only its bf16 compute copy is) ...
                assert not torch.equal(eng.params[:lo], ref.params[:lo])
        else:
            assert torch.equal(eng.params, ref.params), f"parameters differ after step {s}"
    wk.finish()                        # ... until the master is fetched (chief: before a checkpoint / at the end)
    torch.cuda.synchronize()
    assert torch.equal(eng.params[:eng.layout.ctrl], ref.params[:eng.layout.ctrl]), "fp32 master differs from local training"
    assert eng.global_step == STEPS
    wk.fab.close()
    print("APUSH rank 1 ok", flush=True)
comm.shutdown()
