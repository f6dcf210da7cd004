"""Independent pure-Python implementation of the TF tensor-bundle index format, used only to
cross-check the C++ codec (csrc/checkpoint/bundle.cc): what one writes the other must read.
Written from the format description (LevelDB table_format.md + tensor_bundle.proto)."""
import struct

MAGIC = 0xDB4775248B80FB57


def _table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_T = _table()


def crc32c(b: bytes) -> int:
    c = 0xFFFFFFFF
    for x in b:
        c = _T[(c ^ x) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def unmask(m):
    r = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


def varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def read_varint(b, p):
    r, s = 0, 0
    while True:
        x = b[p]
        p += 1
        r |= (x & 0x7F) << s
        if not x & 0x80:
            return r, p
        s += 7


# ---- protobuf ----
def pb_fields(b):
    p, out = 0, []
    while p < len(b):
        tag, p = read_varint(b, p)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, p = read_varint(b, p)
        elif wt == 2:
            l, p = read_varint(b, p)
            v = b[p:p + l]
            p += l
        elif wt == 5:
            v = struct.unpack("<I", b[p:p + 4])[0]
            p += 4
        elif wt == 1:
            v = struct.unpack("<Q", b[p:p + 8])[0]
            p += 8
        else:
            raise ValueError("wire type")
        out.append((f, wt, v))
    return out


def decode_entry(b):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=0)
    for f, wt, v in pb_fields(b):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            for f2, _, dim in pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, x in pb_fields(dim):
                        if f3 == 1:
                            size = x
                    e["shape"].append(size)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
    return e


def encode_entry(dtype, shape, offset, size, crc, shard_id=0):
    sh = b""
    for d in shape:
        dim = (b"\x08" + varint(d)) if d else b""
        sh += b"\x12" + varint(len(dim)) + dim
    out = b""
    if dtype:
        out += b"\x08" + varint(dtype)
    out += b"\x12" + varint(len(sh)) + sh
    if shard_id:
        out += b"\x18" + varint(shard_id)
    if offset:
        out += b"\x20" + varint(offset)
    if size:
        out += b"\x28" + varint(size)
    if crc:
        out += b"\x35" + struct.pack("<I", crc)
    return out


HEADER_1_SHARD = bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01])


# ---- table ----
def parse_block(b):
    n = struct.unpack("<I", b[-4:])[0]
    end = len(b) - 4 - 4 * n
    p, key, out = 0, b"", []
    while p < end:
        sh, p = read_varint(b, p)
        ns, p = read_varint(b, p)
        vl, p = read_varint(b, p)
        key = key[:sh] + b[p:p + ns]
        p += ns
        out.append((key, b[p:p + vl]))
        p += vl
    return out


def read_block(f, off, size):
    body, typ, crc = f[off:off + size], f[off + size], struct.unpack("<I", f[off + size + 1:off + size + 5])[0]
    assert typ == 0
    assert unmask(crc) == crc32c(body + bytes([typ])), "block crc"
    return body


def read_table(f):
    foot = f[-48:]
    assert struct.unpack("<Q", foot[40:])[0] == MAGIC
    p = 0
    moff, p = read_varint(foot, p)
    msz, p = read_varint(foot, p)
    ioff, p = read_varint(foot, p)
    isz, p = read_varint(foot, p)
    kv = []
    for _, h in parse_block(read_block(f, ioff, isz)):
        off, q = read_varint(h, 0)
        sz, q = read_varint(h, q)
        kv += parse_block(read_block(f, off, sz))
    return kv


def build_block(kv, restart_interval=16):
    out, restarts, last, cnt = bytearray(), [0], b"", 0
    for k, v in kv:
        sh = 0
        if cnt < restart_interval:
            while sh < min(len(last), len(k)) and last[sh] == k[sh]:
                sh += 1
        else:
            restarts.append(len(out))
            cnt = 0
        out += varint(sh) + varint(len(k) - sh) + varint(len(v)) + k[sh:] + v
        last = k
        cnt += 1
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def build_table(kv, entries_per_block=1000, restart_interval=16):
    f = bytearray()
    index = []

    def emit(block):
        off = len(f)
        f.extend(block)
        f.append(0)
        f.extend(struct.pack("<I", mask(crc32c(block + b"\x00"))))
        return off, len(block)

    for i in range(0, len(kv), entries_per_block):
        chunk = kv[i:i + entries_per_block]
        off, sz = emit(build_block(chunk, restart_interval))
        index.append((chunk[-1][0], varint(off) + varint(sz)))  # last key itself is a valid separator
    moff, msz = emit(build_block([]))
    ioff, isz = emit(build_block(index, 1))
    foot = varint(moff) + varint(msz) + varint(ioff) + varint(isz)
    foot += b"\x00" * (40 - len(foot)) + struct.pack("<Q", MAGIC)
    f.extend(foot)
    return bytes(f)
