"""Property tests of the checkpoint codec (SURVEY §4 "unit - checkpoint"): any dictionary of named tensors written by the
C++ writer comes back unchanged through the C++ reader, in both Saver formats, and -- for the V2 bundle -- through
the independent pure-Python parser of the format (tests/tf_bundle_py.py).  Names share prefixes (LevelDB prefix
compression, several restart points), tensors may be empty, scalar or rank 4."""
import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

import tf_bundle_py as P
from distributed_tensorflow_amd.checkpoint import list_bundle, read_bundle, write_bundle

DTYPES = [np.float32, np.float64, np.int32, np.int64, np.uint8]
DT_CODE = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.uint8): 4,
           np.dtype(np.int64): 9}
# TF variable-name alphabet plus the slot-name separators; a few fixed stems make shared prefixes likely
_name = st.builds(lambda stem, tail: stem + tail,
                  st.sampled_from(["Variable", "Variable_1", "layer/kernel", "layer/kernel/Adam", "beta1_power", "a", "z"]),
                  st.text(alphabet="abcXYZ019_/.-", min_size=0, max_size=12))


@st.composite
def tensors(draw, dtypes=tuple(DTYPES)):
    out = {}
    for name in draw(st.lists(_name, min_size=1, max_size=40, unique=True)):
        dtype = np.dtype(draw(st.sampled_from(dtypes)))
        shape = tuple(draw(st.lists(st.integers(0, 5), min_size=0, max_size=4)))
        n = int(np.prod(shape)) if shape else 1
        raw = draw(st.binary(min_size=n * dtype.itemsize, max_size=n * dtype.itemsize))
        out[name] = np.frombuffer(raw, dtype=dtype).reshape(shape).copy()
    return out


def _same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()    # bytes: NaN payloads too


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(t=tensors(dtypes=(np.float32, np.float64, np.int32, np.int64)), version=st.sampled_from([1, 2]))
def test_write_read_round_trip_any_tensor_dictionary(tmp_path_factory, t, version):
    # (the V1 writer takes the four types SavedTensorSlices has typed fields for here; uint8 is covered by the V2 test)
    prefix = str(tmp_path_factory.mktemp("ck") / "model.ckpt-7")
    write_bundle(prefix, t, version=version)
    got = read_bundle(prefix)
    assert set(got) == set(t)
    for k in t:
        assert _same(got[k], t[k]), (k, version)
    listed = list_bundle(prefix)
    assert {k: (v[0], tuple(v[1])) for k, v in listed.items()} == {k: (a.dtype, a.shape) for k, a in t.items()}


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(t=tensors())
def test_v2_bundle_parses_with_the_independent_implementation(tmp_path_factory, t):
    prefix = str(tmp_path_factory.mktemp("ck") / "model.ckpt-7")
    write_bundle(prefix, t)
    idx = open(prefix + ".index", "rb").read()
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    kv = P.read_table(idx)
    keys = [k for k, _ in kv]
    assert keys == sorted(keys) and keys[0] == b"" and kv[0][1] == P.HEADER_1_SHARD
    assert {k.decode() for k in keys[1:]} == set(t)
    for k, v in kv[1:]:
        e, want = P.decode_entry(v), t[k.decode()]
        assert e["dtype"] == DT_CODE[want.dtype] and tuple(e["shape"]) == want.shape
        blob = data[e["offset"]:e["offset"] + e["size"]]
        assert blob == want.tobytes()
        assert P.unmask(e["crc32c"]) == P.crc32c(blob)
