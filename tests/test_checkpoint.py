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


# ------------------------------------------------------------------ Saver V1 (TF < 0.12, the reference's era)
def _ld(field, payload):  # length-delimited protobuf field
    return P.varint((field << 3) | 2) + P.varint(len(payload)) + payload


def _shape_proto(shape):
    return b"".join(_ld(2, (b"\x08" + P.varint(d)) if d else b"") for d in shape)


def _slice_proto(extents):
    """extents: per dimension None (everything) or (start, length)."""
    out = b""
    for e in extents:
        body = b"" if e is None else ((b"\x08" + P.varint(e[0]) if e[0] else b"") + b"\x10" + P.varint(e[1]))
        out += _ld(1, body)
    return out


def test_v1_cpp_round_trip_and_key_encoding(tmp_path):
    t = to_tf_names(_model_tensors(5), global_step=777)
    t["neg"] = np.asarray([-3, 0, 7, -(2 ** 31)], dtype=np.int32)
    t["wide"] = np.asarray([[-5, 2 ** 40]], dtype=np.int64)
    prefix = str(tmp_path / "model.ckpt-777")
    bundle.write_v1(prefix, t)
    assert os.path.isfile(prefix) and not os.path.exists(prefix + ".index")
    assert bundle.checkpoint_format(prefix) == 1
    got = read_bundle(prefix)
    assert set(got) == set(t)
    for k in t:
        assert got[k].dtype == t[k].dtype and got[k].shape == t[k].shape and np.array_equal(got[k], t[k]), k
    assert list_bundle(prefix)["Variable_2"] == (np.dtype(np.float32), (3136, 1024))
    # the table keys are what TensorFlow's checkpoint::EncodeTensorNameSlice produces:
    # OrderedCode num(0) | string(name) | num(ndims) | signed(start=0), signed(length=-1) per dimension
    kv = dict(P.read_table(open(prefix, "rb").read()))
    assert b"" in kv
    assert b"\x00Variable_2\x00\x01\x01\x02\x80\x7f\x80\x7f" in kv
    assert b"\x00Variable_8\x00\x01\x00" in kv                      # scalar: zero dimensions, num(0) is the byte 0
    assert b"\x00Variable\x00\x01\x01\x04" + b"\x80\x7f" * 4 in kv
    # meta entry: SavedTensorSlices.meta.tensor[*] lists name / shape / dtype of everything in the file
    (f, _, meta), = [x for x in P.pb_fields(kv[b""]) if x[0] == 1]
    names = set()
    for f, _, v in P.pb_fields(meta):
        if f == 1:
            d = {ff: vv for ff, _, vv in P.pb_fields(v)}
            names.add(d[1].decode())
            if d[1] == b"Variable_8":
                assert d[3] == 3 and d[2] == b""        # DT_INT32 scalar
    assert names == set(t)
    # the float payload is TensorProto.float_val, packed (field 5, wire type 2)
    (_, _, ss), = [x for x in P.pb_fields(kv[b"\x00Variable_4\x00\x01\x01\x01\x80\x7f"]) if x[0] == 2]
    d = {ff: vv for ff, _, vv in P.pb_fields(ss)}
    assert d[1] == b"Variable_4" and P.pb_fields(d[3]) == [(5, 2, t["Variable_4"].tobytes())]


def test_v1_reader_takes_files_laid_out_differently(tmp_path):
    """What an old TensorFlow could have written: a partitioned variable saved as two slices, unpacked
    repeated floats, ints as varints (negative = ten bytes), tensor_content, small table blocks."""
    rng = np.random.RandomState(2)
    w = rng.randn(6, 5).astype(np.float32)
    b = rng.randn(7).astype(np.float32)
    step = np.asarray(-12345, dtype=np.int32)
    c = rng.randn(2, 3, 4).astype(np.float64)

    def meta_tensor(name, shape, dt, slices):
        return _ld(1, _ld(1, name) + _ld(2, _shape_proto(shape)) + b"\x18" + P.varint(dt) +
                   b"".join(_ld(4, _slice_proto(s)) for s in slices))

    def data(name, extents, tensor_proto):
        return _ld(2, _ld(1, name) + _ld(2, _slice_proto(extents)) + _ld(3, tensor_proto))

    meta = (meta_tensor(b"w", w.shape, 1, [[(0, 4), None], [(4, 2), None]]) + meta_tensor(b"b", b.shape, 1, [[None]]) +
            meta_tensor(b"global_step", (), 3, [[]]) + meta_tensor(b"c", c.shape, 2, [[None, (1, 2), None], [None, (0, 1), None]]))
    kv = [
        (b"", _ld(1, meta)),
        (b"\x00b", data(b"b", [None], b"".join(b"\x2d" + x.tobytes() for x in b))),             # unpacked float_val
        (b"\x00c0", data(b"c", [None, (1, 2), None], _ld(6, np.ascontiguousarray(c[:, 1:3]).tobytes()))),
        (b"\x00c1", data(b"c", [None, (0, 1), None], _ld(4, np.ascontiguousarray(c[:, 0:1]).tobytes()))),  # tensor_content
        (b"\x00g", data(b"global_step", [], b"\x38" + P.varint((-12345) & (2 ** 64 - 1)))),        # unpacked int_val
        (b"\x00w0", data(b"w", [(0, 4), None], _ld(5, w[0:4].tobytes()))),
        (b"\x00w1", data(b"w", [(4, 2), None], _ld(5, w[4:6].tobytes()))),
    ]
    prefix = str(tmp_path / "old.ckpt-5")
    open(prefix, "wb").write(P.build_table(kv, entries_per_block=2, restart_interval=2))
    got = read_bundle(prefix)
    assert np.array_equal(got["w"], w) and np.array_equal(got["b"], b) and np.array_equal(got["c"], c)
    assert got["global_step"].dtype == np.int32 and int(got["global_step"]) == -12345
    # a slice that does not cover its tensor is an error, not zeros
    open(prefix, "wb").write(P.build_table([k for k in kv if k[0] != b"\x00w1"], entries_per_block=3))
    with pytest.raises(CheckpointError, match="cover"):
        read_bundle(prefix)


def test_saver_writes_and_resumes_v1_logdirs(tmp_path):
    logdir = str(tmp_path / "train_logs")
    saver = Saver(max_to_keep=2, write_version=1)
    t = to_tf_names(_model_tensors(), global_step=0)
    for step in (10, 20, 30):
        t["Variable_8"] = np.asarray(step, dtype=np.int32)
        saver.save(t, os.path.join(logdir, "model.ckpt"), global_step=step)
    assert sorted(os.listdir(logdir)) == ["checkpoint", "model.ckpt-20", "model.ckpt-30"]
    latest = latest_checkpoint(logdir)
    assert latest.endswith("model.ckpt-30")
    params, step, _ = from_tf_names(Saver.restore(latest))
    assert step == 30 and np.array_equal(params["wc2"], t["Variable_1"])
    # a V2 saver takes the same logdir over (TF >= 0.12 picking up an old run)
    Saver(max_to_keep=2).save(t, os.path.join(logdir, "model.ckpt"), global_step=40)
    assert [os.path.basename(p) for p in read_checkpoint_state(logdir)] == ["model.ckpt-30", "model.ckpt-40"]
    assert not os.path.exists(os.path.join(logdir, "model.ckpt-20"))
    assert bundle.checkpoint_format(latest_checkpoint(logdir)) == 2
