"""Stand-alone times of the two exchange kernels of the data-parallel step (csrc/kernels/dp.hip) with a world of one
rank: what they cost locally, without links.  Usage: python tools/dp_kernel_bench.py [bf16|fp32] [sgd|adam]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
import distributed_tensorflow_amd  # noqa: F401
import torch
from distributed_tensorflow_amd.engine import DeviceDataset, HipEngine
from distributed_tensorflow_amd.models.spec import ModelSpec
from distributed_tensorflow_amd.parallel import comm
from distributed_tensorflow_amd.parallel.dp_fabric import DpFabric
from distributed_tensorflow_amd.ops.native import check

DTYPE = sys.argv[1] if len(sys.argv) > 1 else "bf16"
OPT = sys.argv[2] if len(sys.argv) > 2 else "sgd"
torch.cuda.set_device(0)
comm.init_process_group("gloo", 0, 1, timeout_s=60)
ds = DeviceDataset.synthetic(8192, seed=1, dtype=DTYPE)
eng = HipEngine(ModelSpec(), 1024, ds, optimizer=OPT, learning_rate=1e-3, seed=0, multi_stream=False, dtype=DTYPE)
eng.init_params(seed=7)
lib, e = eng.lib, eng._desc
lib.dtf_dp_dense_exchange.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 8
lib.dtf_dp_tail_commit.argtypes = ([ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 7 + [ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_long] * 5 + [ctypes.c_void_p])
n_dense = eng.layout.segment("wd1", "wd1")[1]
adam = int(OPT == "adam")
s = torch.cuda.current_stream().cuda_stream
for db, cb in ((64, None), (128, None), (256, None), (128, 64), (128, 128)):
    fab = DpFabric(n_dense, eng.layout.total - n_dense, eng.device, dtype=DTYPE, timeout_s=2.0, dense_blocks=db, commit_blocks=cb)

    def dense():
        check(lib.dtf_dp_dense_exchange(fab.byref(), adam, eng.grads.data_ptr(), eng.params.data_ptr(),
                                        eng.adam_m.data_ptr() if adam else None, eng.adam_v.data_ptr() if adam else None,
                                        eng.shadow.data_ptr() if DTYPE == "bf16" else None, eng.hyper.data_ptr(),
                                        eng.state.data_ptr() + 24 if adam else None, s), "dense")

    def tail(with_dense):
        check(lib.dtf_dp_tail_commit(fab.byref(), adam, with_dense, eng.params.data_ptr(), eng.grads.data_ptr(),
                                     eng.adam_m.data_ptr() if adam else None, eng.adam_v.data_ptr() if adam else None,
                                     eng.shadow.data_ptr() if DTYPE == "bf16" else None, eng.hyper.data_ptr(), eng.wc2T.data_ptr(),
                                     e.off_wc2, e.c1, e.c2, eng.out_wT.data_ptr(), e.off_out_w, e.hidden, eng.state.data_ptr(), 3, 4,
                                     1024, 1, 1, e.off_out_w, 0, s), "tail")

    def timed(fn, n=200):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(n): fn()
        ev1.record(); torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) * 1e3 / n

    def both():
        dense(); tail(1)
    t_pair = timed(both)
    # the dense kernel alone desynchronises the two sequence counters: do it on a fresh fabric at the end
    t_tail0 = None
    fab2 = DpFabric(n_dense, eng.layout.total - n_dense, eng.device, dtype=DTYPE, timeout_s=2.0, dense_blocks=db, commit_blocks=cb)
    fab_keep, fab = fab, fab2
    t_tail0 = timed(lambda: tail(0))
    fab3 = DpFabric(n_dense, eng.layout.total - n_dense, eng.device, dtype=DTYPE, timeout_s=2.0, dense_blocks=db, commit_blocks=cb)
    fab = fab3
    t_dense = timed(dense)
    print(f"{DTYPE} {OPT} dense_blocks {db:3d} commit_blocks {fab.desc.commit_blocks:3d}: dense {t_dense:6.2f} us | tail (no dense commit) "
          f"{t_tail0:6.2f} us | dense + tail-with-commit back to back {t_pair:6.2f} us   poisoned {fab_keep.poisoned()}")
    for x in (fab_keep, fab2, fab3): x.close()
