"""Large-batch learning-rate rule (utils/lr_schedule.py; BASELINE.json config 5) and its CLI flags."""
import math
import os
import subprocess
import sys

import pytest

from distributed_tensorflow_amd.utils.lr_schedule import LrSchedule, scaled_rate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rules():
    assert scaled_rate(0.001, 128, 128, "linear") == 0.001
    assert scaled_rate(0.001, 8192, 128, "none") == 0.001
    assert scaled_rate(0.001, 8192, 128, "linear") == pytest.approx(0.064)
    assert scaled_rate(0.001, 8192, 128, "sqrt") == pytest.approx(0.008)
    with pytest.raises(ValueError):
        scaled_rate(0.001, 256, 128, "cubic")


def test_warmup_ramps_linearly_to_the_scaled_rate():
    s = LrSchedule(0.01, 1024, 128, "linear", warmup_steps=4)
    assert s.peak_lr == pytest.approx(0.08)
    got = [s(t) for t in range(6)]
    assert got == pytest.approx([0.01 + 0.07 * k / 4 for k in (1, 2, 3, 4)] + [0.08, 0.08])
    assert LrSchedule(0.01, 128, 128, "linear", warmup_steps=4)(0) == 0.01      # nothing to warm up to
    assert LrSchedule(0.01, 512, 128, "sqrt").constant and LrSchedule(0.01, 512, 128, "sqrt")(0) == pytest.approx(0.02)


def test_cli_applies_the_rule(tmp_path):
    """Local CPU run: the scaled rate is announced and used (loss after 6 steps differs from the unscaled run's)."""
    def run(extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "mnistdist.py"), "--backend=cpu", "--synthetic", "--steps=6",
                            "--batch_size=256", "--learning_rate=0.002", "--display_step=5", "--noeval", "--logdir="] + extra,
                           capture_output=True, text=True, cwd=ROOT, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout + r.stderr

    plain = run([])
    scaled = run(["--lr_scaling=linear", "--lr_warmup_steps=3"])
    assert "learning rate 0.002 x (256/128, linear) = 0.004, warm-up over 3 steps" in scaled, scaled
    assert "learning rate" not in plain

    def loss_at_5(o):
        line = [l for l in o.splitlines() if "step:  5 " in l][0]
        return float(line.split("mini_batch loss:")[1].split()[0])
    assert loss_at_5(plain) != loss_at_5(scaled)
    assert math.isfinite(loss_at_5(scaled))
