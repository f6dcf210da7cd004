"""parallel/comm.py slice collectives (reduce-scatter / all-gather in place) as the sharded dense update uses them,
over gloo on the CPU: world 2 and 3, fp32 and bf16 payloads."""
import os
import subprocess
import sys

import pytest

from test_distributed_cpu import free_ports


@pytest.mark.parametrize("world", [2, 3])
def test_slice_collectives_over_gloo(world):
    port = free_ports(1)[0]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "comm_helpers_worker.py")
    procs = [subprocess.Popen([sys.executable, "-u", worker, str(r), str(world), str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    for r, p in enumerate(procs):
        try:
            o, _ = p.communicate(timeout=90)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("timed out")
        assert p.returncode == 0 and f"COMM rank {r} ok" in o, o[-2000:]
