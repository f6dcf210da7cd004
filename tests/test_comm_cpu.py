"""parallel/comm.py on CPU (gloo, world_size 2): bring-up and the all-reduce latency probe that
bench.py reports next to its multi-GPU numbers."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import sys
sys.path.insert(0, {root!r})
from distributed_tensorflow_amd.parallel import comm
rank, port = int(sys.argv[1]), int(sys.argv[2])
comm.init_process_group("gloo", rank, 2, "127.0.0.1", port, timeout_s=60)
assert comm.world_size() == 2 and comm.rank() == rank
us = comm.allreduce_latency_us(4096, iters=3, warmup=1)
print("LAT", us)
comm.shutdown()
"""


def test_allreduce_latency_probe_agrees_across_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = WORKER.format(root=ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, env=dict(os.environ, OMP_NUM_THREADS="1"))
             for r in range(2)]
    vals = []
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out
        vals.append(float([l for l in out.splitlines() if l.startswith("LAT")][0].split()[1]))
    assert vals[0] == vals[1] > 0.0   # the probe returns the max over ranks on every rank
