"""Damaged checkpoint files must be refused cleanly (restore reads files it did not write).

The codec (csrc/checkpoint/bundle.cc) is compiled together with csrc/checkpoint/fuzz_main.cc under
AddressSanitizer + UBSan (host code only) and driven over thousands of damaged copies of a valid V2 bundle and a
valid Saver V1 file: bit flips, stamped bytes, truncations, spliced maximal varints, shifted tails, with the block
checksum check compiled out so that the damage reaches the table / protobuf parsers.  A sanitizer report fails the
test (exit 99 / abort); reader errors are the expected outcome.  A few hand-made cases pin the Python surface.
"""
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from distributed_tensorflow_amd.checkpoint.bundle import CheckpointError, list_bundle, read_bundle, write_bundle

ROOT = Path(__file__).resolve().parent.parent
SRC = [ROOT / "csrc" / "checkpoint" / "bundle.cc", ROOT / "csrc" / "checkpoint" / "fuzz_main.cc"]


def _tensors():
    rng = np.random.default_rng(0)
    t = {"Variable": rng.standard_normal((5, 5, 1, 4)).astype(np.float32),
         "Variable_1": rng.standard_normal((40, 3)).astype(np.float32),
         "Variable_8": np.array(17, dtype=np.int64),
         "beta1_power": np.array(0.9, dtype=np.float32),
         "ids": np.arange(-3, 9, dtype=np.int32),
         "empty": np.zeros((0, 3), dtype=np.float32)}
    for i in range(40):      # enough keys for prefix compression and several restart points
        t[f"extra/var_{i}"] = rng.standard_normal((7, 3)).astype(np.float32)
    return t


@pytest.fixture(scope="module")
def fuzz_binary(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    out = tmp_path_factory.mktemp("ckpt_fuzz") / "ckpt_fuzz"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-DDTF_CKPT_FUZZ",
           "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-static-libasan"] + [str(s) for s in SRC] + ["-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return out


@pytest.mark.parametrize("version", [2, 1])
@pytest.mark.parametrize("seed", [1, 2])
def test_damaged_files_never_crash_the_codec(fuzz_binary, tmp_path, version, seed):
    prefix = tmp_path / "model.ckpt-17"
    write_bundle(str(prefix), _tensors(), version=version)
    env = dict(os.environ, ASAN_OPTIONS="exitcode=99:detect_leaks=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([str(fuzz_binary), str(prefix), str(tmp_path / "work"), "4000", str(seed)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, f"exit {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-6000:]}"
    fields = dict(kv.split("=") for kv in r.stdout.split())
    assert fields["format"] == f"v{version}"
    assert int(fields["refused"]) > 1000          # damage is noticed ...
    assert int(fields["loaded"]) + int(fields["refused"]) == 4000   # ... and every case came back


@pytest.mark.parametrize("version", [2, 1])
def test_every_truncation_is_a_checkpoint_error(tmp_path, version):
    """Python surface, release library (checksums on): a file cut anywhere raises CheckpointError, nothing else."""
    prefix = tmp_path / "m"
    write_bundle(str(prefix), _tensors(), version=version)
    table = Path(str(prefix) + ".index") if version == 2 else prefix
    whole = table.read_bytes()
    cuts = sorted(set(list(range(0, len(whole), max(1, len(whole) // 60))) + list(range(len(whole) - 64, len(whole)))))
    for n in cuts:
        table.write_bytes(whole[:n])
        with pytest.raises(CheckpointError):
            read_bundle(str(prefix))
    table.write_bytes(whole)
    assert set(read_bundle(str(prefix))) == set(_tensors())
    if version == 2:       # the data shard cut short: refused when the bundle is opened, before any allocation
        shard = Path(str(prefix) + ".data-00000-of-00001")
        data = shard.read_bytes()
        shard.write_bytes(data[:len(data) // 2])
        with pytest.raises(CheckpointError, match="past the end"):
            read_bundle(str(prefix))
        shard.write_bytes(data[:-1] + bytes([data[-1] ^ 1]))
        with pytest.raises(CheckpointError, match="checksum"):
            read_bundle(str(prefix))


def test_rank_above_eight_is_refused_not_misreported(tmp_path):
    prefix = tmp_path / "deep"
    write_bundle(str(prefix), {"t": np.zeros((1,) * 9, dtype=np.float32)})
    with pytest.raises(CheckpointError, match="8 dimensions"):
        list_bundle(str(prefix))
    with pytest.raises(CheckpointError, match="8 dimensions"):
        read_bundle(str(prefix))
