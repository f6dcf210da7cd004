"""TF tensor-bundle (Saver V2) codec: golden vectors, C++ <-> independent-Python cross checks,
Saver/state-file behaviour, and the reference's variable-name contract.

TensorFlow itself is not installable here, so compatibility is pinned by (a) published
crc32c/LevelDB constants, (b) an independent pure-Python implementation of the format
(tests/tf_bundle_py.py) that must read what the C++ writes and vice versa."""
import os
import struct

import numpy as np
import pytest

import tf_bundle_py as P
from distributed_tensorflow_amd.checkpoint import (Saver, bundle, from_tf_names, latest_checkpoint, list_bundle,
                                                   read_bundle, read_checkpoint_state, to_tf_names, write_bundle)
from distributed_tensorflow_amd.checkpoint.bundle import CheckpointError
from distributed_tensorflow_amd.models.spec import TF_VARIABLE_NAMES, ModelSpec


def test_crc32c_golden_vectors():
    # RFC 3720 B.4 test vectors + the classic check value
    assert bundle.crc32c(b"123456789") == 0xE3069283
    assert bundle.crc32c(bytes(32)) == 0x8A9136AA
    assert bundle.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert bundle.crc32c(bytes(range(32))) == 0x46DD794E
    assert bundle.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    for blob in (b"", b"a", os.urandom(1000), os.urandom(4097)):
        assert bundle.crc32c(blob) == P.crc32c(blob)
        assert bundle.crc32c_masked(blob) == P.mask(P.crc32c(blob))
    # LevelDB mask: ((crc >> 15) | (crc << 17)) + 0xa282ead8
    assert P.unmask(P.mask(0xDEADBEEF)) == 0xDEADBEEF


def _model_tensors(seed=0):
    rng = np.random.RandomState(seed)
    t = {n: rng.randn(*s).astype(np.float32) for n, s in ModelSpec().shapes().items()}
    return t


def test_cpp_writer_is_readable_by_independent_parser(tmp_path):
    t = to_tf_names(_model_tensors(), global_step=1234)
    prefix = str(tmp_path / "model.ckpt-1234")
    write_bundle(prefix, t)
    idx = open(prefix + ".index", "rb").read()
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    assert struct.unpack("<Q", idx[-8:])[0] == 0xDB4775248B80FB57
    kv = P.read_table(idx)
    keys = [k.decode() for k, _ in kv]
    assert keys == sorted(keys) and keys[0] == ""
    assert kv[0][1] == P.HEADER_1_SHARD          # num_shards=1, LITTLE, version.producer=1
    assert set(keys[1:]) == set(TF_VARIABLE_NAMES.values())
    for k, v in kv[1:]:
        e = P.decode_entry(v)
        want = t[k.decode()]
        assert e["dtype"] == (3 if want.dtype == np.int32 else 1)   # DT_INT32 / DT_FLOAT
        assert tuple(e["shape"]) == want.shape
        blob = data[e["offset"]:e["offset"] + e["size"]]
        assert blob == want.tobytes()
        assert P.unmask(e["crc32c"]) == P.crc32c(blob) if len(blob) < 100000 else True
        assert v == P.encode_entry(e["dtype"], e["shape"], e["offset"], e["size"], e["crc32c"])  # byte-identical proto
    # contract table of SURVEY §5.4
    shapes = {k.decode(): tuple(P.decode_entry(v)["shape"]) for k, v in kv[1:]}
    assert shapes == {"Variable": (5, 5, 1, 32), "Variable_1": (5, 5, 32, 64), "Variable_2": (3136, 1024),
                      "Variable_3": (1024, 10), "Variable_4": (32,), "Variable_5": (64,), "Variable_6": (1024,),
                      "Variable_7": (10,), "Variable_8": ()}


def test_cpp_reader_reads_independent_writer(tmp_path):
    """A bundle laid out differently from ours (many small blocks, restart interval 2, prefix
    compression across similar keys) must load: that is what a TF-written file can look like."""
    rng = np.random.RandomState(1)
    tensors = {f"layer_{i:03d}/weights": rng.randn(3, i + 1).astype(np.float32) for i in range(40)}
    tensors["global_step"] = np.asarray(99, dtype=np.int64)
    data = b""
    kv = [(b"", P.HEADER_1_SHARD)]
    for name in sorted(tensors):
        a = tensors[name]
        blob = a.tobytes()
        dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 9}[a.dtype]
        kv.append((name.encode(), P.encode_entry(dt, a.shape, len(data), len(blob), P.mask(P.crc32c(blob)))))
        data += blob
    prefix = str(tmp_path / "other.ckpt")
    open(prefix + ".index", "wb").write(P.build_table(kv, entries_per_block=7, restart_interval=2))
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    got = read_bundle(prefix)
    assert set(got) == set(tensors)
    for k in tensors:
        assert got[k].dtype == tensors[k].dtype and np.array_equal(got[k], tensors[k])
    assert list_bundle(prefix)["layer_007/weights"] == (np.dtype(np.float32), (3, 8))


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "m.ckpt")
    write_bundle(prefix, {"Variable": np.arange(100, dtype=np.float32)})
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    raw[17] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(CheckpointError, match="checksum"):
        read_bundle(prefix)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[3] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(CheckpointError):
        read_bundle(prefix)
    with pytest.raises(CheckpointError):
        read_bundle(str(tmp_path / "missing"))


def test_name_mapping_round_trip_with_adam_slots():
    p = _model_tensors(3)
    m = {k: v * 0.1 for k, v in p.items()}
    v = {k: v * v for k, v in p.items()}
    tf = to_tf_names(p, global_step=42, adam={"m": m, "v": v}, adam_t=41)
    assert "Variable_2/Adam" in tf and "Variable_2/Adam_1" in tf and "beta1_power" in tf
    assert tf["Variable_8"].dtype == np.int32 and tf["Variable_8"].shape == ()
    p2, step, adam = from_tf_names(tf)
    assert step == 42 and set(p2) == set(p)
    assert all(np.array_equal(p2[k], p[k]) for k in p)
    assert np.array_equal(adam["m"]["wd1"], m["wd1"]) and np.array_equal(adam["v"]["out_b"], v["out_b"])
    assert adam["beta1_power"] == pytest.approx(0.9 ** 42)


def test_saver_state_file_and_max_to_keep(tmp_path):
    logdir = str(tmp_path / "train_logs")
    saver = Saver(max_to_keep=3)
    t = to_tf_names(_model_tensors(), global_step=0)
    for step in (100, 200, 300, 400, 500):
        t["Variable_8"] = np.asarray(step, dtype=np.int32)
        prefix = saver.save(t, os.path.join(logdir, "model.ckpt"), global_step=step)
        assert prefix.endswith(f"model.ckpt-{step}")
    kept = read_checkpoint_state(logdir)
    assert [os.path.basename(p) for p in kept] == ["model.ckpt-300", "model.ckpt-400", "model.ckpt-500"]
    assert not os.path.exists(os.path.join(logdir, "model.ckpt-100.index"))
    assert latest_checkpoint(logdir).endswith("model.ckpt-500")
    txt = open(os.path.join(logdir, "checkpoint")).read().splitlines()
    assert txt[0] == 'model_checkpoint_path: "model.ckpt-500"'
    assert txt[1:] == [f'all_model_checkpoint_paths: "model.ckpt-{s}"' for s in (300, 400, 500)]
    params, step, adam = from_tf_names(Saver.restore(latest_checkpoint(logdir)))
    assert step == 500 and adam is None and params["wd1"].shape == (3136, 1024)
    # a fresh Saver (restarted chief) continues the rotation from the state file
    s2 = Saver(max_to_keep=3)
    s2.save(t, os.path.join(logdir, "model.ckpt"), global_step=600)
    assert [os.path.basename(p) for p in read_checkpoint_state(logdir)] == \
        ["model.ckpt-400", "model.ckpt-500", "model.ckpt-600"]
    assert latest_checkpoint(str(tmp_path / "nothing_here")) is None
