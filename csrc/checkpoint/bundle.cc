// TensorFlow "tensor bundle" (tf.train.Saver V2) checkpoint codec, from the
// documented on-disk format -- no TensorFlow dependency.
//
// The reference saves its 9 unnamed variables with tf.train.Saver() through the
// Supervisor (reference .idea/MNISTDist.py:154, :159-165); TF >= 0.12 writes that
// as a bundle:
//   <prefix>.index                 LevelDB-style SSTable: key "" -> BundleHeaderProto,
//                                  key <tensor name> -> BundleEntryProto
//   <prefix>.data-00000-of-00001   raw little-endian tensor bytes, back to back
// This file implements both directions (so checkpoints written here load in
// TensorFlow and TF-written ones load here):
//   * crc32c (Castagnoli) + the LevelDB/TF mask
//   * varint32/64, fixed32/64
//   * SSTable blocks with prefix compression + restart array, block trailer
//     (type byte + masked crc), index block, empty metaindex block, 48-byte footer
//     with magic 0xdb4775248b80fb57
//   * minimal protobuf wire encode/decode for BundleHeaderProto, BundleEntryProto,
//     TensorShapeProto, VersionDef
//
// TF < 0.12 -- the version the reference script was written for (tf.initialize_all_variables,
// reference .idea/MNISTDist.py:156) -- wrote Saver "V1" files instead: ONE sstable <prefix> whose
// key "" holds a SavedTensorSlices{meta} proto (names, shapes, dtypes, slices) and whose other keys
// (OrderedCode(0, name, ndims, start/length per dim)) hold SavedTensorSlices{data} protos with the
// values in a TensorProto.  Both directions of that are here too (reader_open detects the format),
// so logdirs written by the reference resume here.
// Exposed through a C ABI (see bottom) and bound from Python with ctypes.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace dtfckpt {

// ------------------------------------------------------------------ crc32c
static uint32_t g_crc_table[8][256];
static bool g_crc_init = false;

static void crc_init() {
  if (g_crc_init) return;
  const uint32_t poly = 0x82f63b78u;  // reflected Castagnoli
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ poly : (c >> 1);
    g_crc_table[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = g_crc_table[0][i];
    for (int t = 1; t < 8; ++t) {
      c = g_crc_table[0][c & 0xff] ^ (c >> 8);
      g_crc_table[t][i] = c;
    }
  }
  g_crc_init = true;
}

static uint32_t crc32c_extend(uint32_t crc, const uint8_t* p, size_t n) {
  crc_init();
  uint32_t c = crc ^ 0xffffffffu;
  while (n >= 8) {  // slicing-by-8
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = g_crc_table[7][lo & 0xff] ^ g_crc_table[6][(lo >> 8) & 0xff] ^ g_crc_table[5][(lo >> 16) & 0xff] ^
        g_crc_table[4][lo >> 24] ^ g_crc_table[3][hi & 0xff] ^ g_crc_table[2][(hi >> 8) & 0xff] ^
        g_crc_table[1][(hi >> 16) & 0xff] ^ g_crc_table[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = g_crc_table[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xffffffffu;
}

static const uint32_t kMaskDelta = 0xa282ead8u;
static inline uint32_t crc_mask(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + kMaskDelta; }
static inline uint32_t crc_unmask(uint32_t m) {
  uint32_t rot = m - kMaskDelta;
  return (rot >> 17) | (rot << 15);
}

// ------------------------------------------------------------------ coding
static void put_fixed32(std::string* s, uint32_t v) {
  char b[4] = {(char)(v & 0xff), (char)((v >> 8) & 0xff), (char)((v >> 16) & 0xff), (char)((v >> 24) & 0xff)};
  s->append(b, 4);
}
static void put_fixed64(std::string* s, uint64_t v) {
  put_fixed32(s, (uint32_t)v);
  put_fixed32(s, (uint32_t)(v >> 32));
}
static uint32_t get_fixed32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t get_fixed64(const uint8_t* p) { return (uint64_t)get_fixed32(p) | ((uint64_t)get_fixed32(p + 4) << 32); }
static void put_varint64(std::string* s, uint64_t v) {
  while (v >= 0x80) {
    s->push_back((char)((v & 0x7f) | 0x80));
    v >>= 7;
  }
  s->push_back((char)v);
}
static void put_varint32(std::string* s, uint32_t v) { put_varint64(s, v); }
// returns bytes consumed, 0 on error
static size_t get_varint64(const uint8_t* p, const uint8_t* end, uint64_t* out) {
  uint64_t r = 0;
  int shift = 0;
  const uint8_t* q = p;
  while (q < end && shift <= 63) {
    uint8_t b = *q++;
    r |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) {
      *out = r;
      return (size_t)(q - p);
    }
    shift += 7;
  }
  return 0;
}

// ------------------------------------------------------------------ protobuf (minimal)
static void pb_varint_field(std::string* s, int field, uint64_t v) {
  put_varint32(s, (uint32_t)(field << 3) | 0);
  put_varint64(s, v);
}
static void pb_bytes_field(std::string* s, int field, const std::string& v) {
  put_varint32(s, (uint32_t)(field << 3) | 2);
  put_varint64(s, v.size());
  s->append(v);
}
static void pb_fixed32_field(std::string* s, int field, uint32_t v) {
  put_varint32(s, (uint32_t)(field << 3) | 5);
  put_fixed32(s, v);
}

struct Entry {
  std::string name;
  int dtype = 0;  // tensorflow.DataType
  std::vector<int64_t> dims;
  int shard_id = 0;
  int64_t offset = 0;
  int64_t size = 0;
  uint32_t crc = 0;  // masked crc32c of the tensor bytes
  bool has_slices = false;
};

static std::string encode_shape(const std::vector<int64_t>& dims) {
  std::string shape;
  for (int64_t d : dims) {
    std::string dim;
    pb_varint_field(&dim, 1, (uint64_t)d);  // Dim.size (proto3 omits 0, TF still reads it as 0)
    if (d == 0) dim.clear();
    pb_bytes_field(&shape, 2, dim);  // TensorShapeProto.dim
  }
  return shape;
}

static std::string encode_entry(const Entry& e) {
  std::string s;
  if (e.dtype) pb_varint_field(&s, 1, (uint64_t)e.dtype);
  pb_bytes_field(&s, 2, encode_shape(e.dims));  // TF always emits the (possibly empty) shape message
  if (e.shard_id) pb_varint_field(&s, 3, (uint64_t)e.shard_id);
  if (e.offset) pb_varint_field(&s, 4, (uint64_t)e.offset);
  if (e.size) pb_varint_field(&s, 5, (uint64_t)e.size);
  if (e.crc) pb_fixed32_field(&s, 6, e.crc);
  return s;
}

static std::string encode_header(int num_shards) {
  std::string s;
  pb_varint_field(&s, 1, (uint64_t)num_shards);  // num_shards
  // endianness LITTLE = 0 (default, omitted)
  std::string ver;
  pb_varint_field(&ver, 1, 1);  // VersionDef.producer = kTensorBundleVersion (1)
  pb_bytes_field(&s, 3, ver);
  return s;
}

struct PbReader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  bool next(int* field, int* wt) {
    if (p >= end) return false;
    uint64_t tag;
    size_t n = get_varint64(p, end, &tag);
    if (!n) { ok = false; return false; }
    p += n;
    *field = (int)(tag >> 3);
    *wt = (int)(tag & 7);
    return true;
  }
  uint64_t varint() {
    uint64_t v = 0;
    size_t n = get_varint64(p, end, &v);
    if (!n) { ok = false; p = end; return 0; }
    p += n;
    return v;
  }
  bool bytes(const uint8_t** b, size_t* len) {
    uint64_t l = varint();
    if (!ok || l > (uint64_t)(end - p)) { ok = false; p = end; return false; }
    *b = p;
    *len = (size_t)l;
    p += l;
    return true;
  }
  void skip(int wt) {
    if (wt == 0) varint();
    else if (wt == 1) { if (end - p < 8) { ok = false; p = end; } else p += 8; }
    else if (wt == 2) { const uint8_t* b; size_t l; bytes(&b, &l); }
    else if (wt == 5) { if (end - p < 4) { ok = false; p = end; } else p += 4; }
    else { ok = false; p = end; }
  }
};

static bool decode_shape(const uint8_t* b, size_t len, std::vector<int64_t>* dims) {
  PbReader r{b, b + len};
  int f, wt;
  while (r.next(&f, &wt)) {
    if (f == 2 && wt == 2) {
      const uint8_t* db; size_t dl;
      if (!r.bytes(&db, &dl)) return false;
      PbReader d{db, db + dl};
      int64_t size = 0;
      int f2, wt2;
      while (d.next(&f2, &wt2)) {
        if (f2 == 1 && wt2 == 0) size = (int64_t)d.varint();
        else d.skip(wt2);
      }
      if (!d.ok) return false;
      dims->push_back(size);
    } else {
      r.skip(wt);
    }
  }
  return r.ok;
}

static bool decode_entry(const uint8_t* b, size_t len, Entry* e) {
  PbReader r{b, b + len};
  int f, wt;
  while (r.next(&f, &wt)) {
    if (f == 1 && wt == 0) e->dtype = (int)r.varint();
    else if (f == 2 && wt == 2) {
      const uint8_t* sb; size_t sl;
      if (!r.bytes(&sb, &sl) || !decode_shape(sb, sl, &e->dims)) return false;
    } else if (f == 3 && wt == 0) e->shard_id = (int)r.varint();
    else if (f == 4 && wt == 0) e->offset = (int64_t)r.varint();
    else if (f == 5 && wt == 0) e->size = (int64_t)r.varint();
    else if (f == 6 && wt == 5) {
      if (r.end - r.p < 4) return false;
      e->crc = get_fixed32(r.p);
      r.p += 4;
    } else if (f == 7) { e->has_slices = true; r.skip(wt); }
    else r.skip(wt);
  }
  return r.ok;
}

static bool decode_header(const uint8_t* b, size_t len, int* num_shards, int* endianness) {
  PbReader r{b, b + len};
  int f, wt;
  *num_shards = 0;
  *endianness = 0;
  while (r.next(&f, &wt)) {
    if (f == 1 && wt == 0) *num_shards = (int)r.varint();
    else if (f == 2 && wt == 0) *endianness = (int)r.varint();
    else r.skip(wt);
  }
  return r.ok;
}

// ------------------------------------------------------------------ SSTable writer
static const uint64_t kTableMagic = 0xdb4775248b80fb57ull;
static const int kRestartInterval = 16;

class BlockBuilder {
 public:
  explicit BlockBuilder(int restart_interval) : interval_(restart_interval) { restarts_.push_back(0); }
  void add(const std::string& key, const std::string& value) {
    size_t shared = 0;
    if (counter_ < interval_) {
      size_t m = std::min(last_key_.size(), key.size());
      while (shared < m && last_key_[shared] == key[shared]) ++shared;
    } else {
      restarts_.push_back((uint32_t)buf_.size());
      counter_ = 0;
    }
    put_varint32(&buf_, (uint32_t)shared);
    put_varint32(&buf_, (uint32_t)(key.size() - shared));
    put_varint32(&buf_, (uint32_t)value.size());
    buf_.append(key, shared, std::string::npos);
    buf_.append(value);
    last_key_ = key;
    ++counter_;
  }
  std::string finish() {
    for (uint32_t r : restarts_) put_fixed32(&buf_, r);
    put_fixed32(&buf_, (uint32_t)restarts_.size());
    return buf_;
  }
  bool empty() const { return buf_.empty(); }
  size_t size_estimate() const { return buf_.size() + restarts_.size() * 4 + 4; }
  const std::string& last_key() const { return last_key_; }

 private:
  int interval_;
  int counter_ = 0;
  std::string buf_, last_key_;
  std::vector<uint32_t> restarts_;
};

static std::string encode_handle(uint64_t off, uint64_t size) {
  std::string s;
  put_varint64(&s, off);
  put_varint64(&s, size);
  return s;
}

// shortest string >= key (LevelDB BytewiseComparator::FindShortSuccessor)
static std::string short_successor(const std::string& key) {
  for (size_t i = 0; i < key.size(); ++i) {
    uint8_t b = (uint8_t)key[i];
    if (b != 0xff) {
      std::string r = key.substr(0, i + 1);
      r[i] = (char)(b + 1);
      return r;
    }
  }
  return key;
}
// LevelDB FindShortestSeparator(start, limit)
static std::string shortest_separator(const std::string& start, const std::string& limit) {
  size_t m = std::min(start.size(), limit.size()), d = 0;
  while (d < m && start[d] == limit[d]) ++d;
  if (d < m) {
    uint8_t b = (uint8_t)start[d];
    if (b < 0xff && b + 1 < (uint8_t)limit[d]) {
      std::string r = start.substr(0, d + 1);
      r[d] = (char)(b + 1);
      return r;
    }
  }
  return start;
}

static void append_block(std::string* file, const std::string& contents, uint64_t* off, uint64_t* size) {
  *off = file->size();
  *size = contents.size();
  file->append(contents);
  char type = 0;  // kNoCompression
  uint32_t crc = crc32c_extend(0, (const uint8_t*)contents.data(), contents.size());
  crc = crc32c_extend(crc, (const uint8_t*)&type, 1);
  file->push_back(type);
  put_fixed32(file, crc_mask(crc));
}

// keys must be added in sorted order
static std::string build_table(const std::vector<std::pair<std::string, std::string>>& kv, size_t block_size) {
  std::string file;
  BlockBuilder index(1);
  std::unique_ptr<BlockBuilder> data(new BlockBuilder(kRestartInterval));
  std::string pending_last_key;
  uint64_t pending_off = 0, pending_size = 0;
  bool pending = false;
  for (size_t i = 0; i < kv.size(); ++i) {
    if (pending) {
      index.add(shortest_separator(pending_last_key, kv[i].first), encode_handle(pending_off, pending_size));
      pending = false;
    }
    data->add(kv[i].first, kv[i].second);
    if (data->size_estimate() >= block_size) {
      pending_last_key = data->last_key();
      append_block(&file, data->finish(), &pending_off, &pending_size);
      data.reset(new BlockBuilder(kRestartInterval));
      pending = true;
    }
  }
  if (!data->empty()) {
    if (pending) {  // cannot happen (pending is flushed before each add) but keep the invariant explicit
      index.add(short_successor(pending_last_key), encode_handle(pending_off, pending_size));
      pending = false;
    }
    pending_last_key = data->last_key();
    append_block(&file, data->finish(), &pending_off, &pending_size);
    pending = true;
  }
  if (pending) index.add(short_successor(pending_last_key), encode_handle(pending_off, pending_size));
  uint64_t meta_off, meta_size, idx_off, idx_size;
  BlockBuilder meta(kRestartInterval);
  append_block(&file, meta.finish(), &meta_off, &meta_size);
  append_block(&file, index.finish(), &idx_off, &idx_size);
  std::string footer = encode_handle(meta_off, meta_size) + encode_handle(idx_off, idx_size);
  footer.resize(40, '\0');
  put_fixed64(&footer, kTableMagic);
  file.append(footer);
  return file;
}

// ------------------------------------------------------------------ SSTable reader
static bool read_file(const std::string& path, std::string* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n < 0) { fclose(f); return false; }   // a directory, a pipe
  out->resize((size_t)n);
  size_t got = n ? fread(&(*out)[0], 1, (size_t)n, f) : 0;
  fclose(f);
  return got == (size_t)n;
}

static bool read_block(const std::string& file, uint64_t off, uint64_t size, std::string* contents,
                       std::string* err) {
  // written so that a crafted handle cannot wrap the uint64 sum past the check
  if (file.size() < 5 || size > file.size() - 5 || off > file.size() - 5 - size) { *err = "block handle out of range"; return false; }
  const uint8_t* p = (const uint8_t*)file.data() + off;
  uint8_t type = p[size];
  uint32_t want = crc_unmask(get_fixed32(p + size + 1));
  uint32_t got = crc32c_extend(0, p, size + 1);
#ifndef DTF_CKPT_FUZZ   // the damaged-file harness (fuzz_main.cc) lets damage through to the parsers behind the checksum
  if (want != got) { *err = "block checksum mismatch"; return false; }
#else
  (void)want; (void)got;
#endif
  if (type != 0) { *err = "compressed table blocks are not supported (TF writes bundles uncompressed)"; return false; }
  contents->assign((const char*)p, size);
  return true;
}

static bool parse_block(const std::string& b, std::vector<std::pair<std::string, std::string>>* out,
                        std::string* err) {
  if (b.size() < 4) { *err = "block too small"; return false; }
  uint32_t nrestart = get_fixed32((const uint8_t*)b.data() + b.size() - 4);
  if ((uint64_t)nrestart * 4 + 4 > b.size()) { *err = "bad restart count"; return false; }
  const uint8_t* p = (const uint8_t*)b.data();
  const uint8_t* end = p + b.size() - 4 - (size_t)nrestart * 4;
  std::string key;
  while (p < end) {
    uint64_t shared, non_shared, vlen;
    size_t n;
    if (!(n = get_varint64(p, end, &shared))) { *err = "bad entry"; return false; }
    p += n;
    if (!(n = get_varint64(p, end, &non_shared))) { *err = "bad entry"; return false; }
    p += n;
    if (!(n = get_varint64(p, end, &vlen))) { *err = "bad entry"; return false; }
    p += n;
    const uint64_t rem = (uint64_t)(end - p);   // compared piecewise: the sum of two crafted lengths can wrap
    if (shared > key.size() || non_shared > rem || vlen > rem - non_shared) { *err = "bad entry lengths"; return false; }
    key.resize((size_t)shared);
    key.append((const char*)p, (size_t)non_shared);
    p += non_shared;
    out->emplace_back(key, std::string((const char*)p, (size_t)vlen));
    p += vlen;
  }
  return true;
}

static bool read_table(const std::string& file, std::vector<std::pair<std::string, std::string>>* kv,
                       std::string* err) {
  if (file.size() < 48) { *err = "index file too small for an sstable footer"; return false; }
  const uint8_t* foot = (const uint8_t*)file.data() + file.size() - 48;
  if (get_fixed64(foot + 40) != kTableMagic) { *err = "bad sstable magic (not a V2 checkpoint index?)"; return false; }
  uint64_t meta_off, meta_size, idx_off, idx_size;
  const uint8_t* p = foot;
  const uint8_t* end = foot + 40;
  size_t n;
  if (!(n = get_varint64(p, end, &meta_off))) { *err = "bad footer"; return false; }
  p += n;
  if (!(n = get_varint64(p, end, &meta_size))) { *err = "bad footer"; return false; }
  p += n;
  if (!(n = get_varint64(p, end, &idx_off))) { *err = "bad footer"; return false; }
  p += n;
  if (!(n = get_varint64(p, end, &idx_size))) { *err = "bad footer"; return false; }
  std::string idx;
  if (!read_block(file, idx_off, idx_size, &idx, err)) return false;
  std::vector<std::pair<std::string, std::string>> handles;
  if (!parse_block(idx, &handles, err)) return false;
  for (auto& h : handles) {
    uint64_t off, size;
    const uint8_t* hp = (const uint8_t*)h.second.data();
    const uint8_t* he = hp + h.second.size();
    if (!(n = get_varint64(hp, he, &off))) { *err = "bad block handle"; return false; }
    hp += n;
    if (!(n = get_varint64(hp, he, &size))) { *err = "bad block handle"; return false; }
    std::string blk;
    if (!read_block(file, off, size, &blk, err)) return false;
    if (!parse_block(blk, kv, err)) return false;
  }
  return true;
}

// ------------------------------------------------------------------ bundle writer / reader
static std::string shard_name(const std::string& prefix, int shard, int num_shards) {
  char buf[64];
  snprintf(buf, sizeof buf, ".data-%05d-of-%05d", shard, num_shards);
  return prefix + buf;
}

static bool write_file_atomic(const std::string& path, const std::string& data, std::string* err) {
  std::string tmp = path + ".tempstate";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) { *err = "cannot open " + tmp; return false; }
  bool ok = data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size();
  ok = (fclose(f) == 0) && ok;
  if (!ok) { *err = "short write to " + tmp; return false; }
  if (rename(tmp.c_str(), path.c_str()) != 0) { *err = "rename failed for " + path; return false; }
  return true;
}

struct Writer {
  std::string prefix;
  std::map<std::string, Entry> entries;  // sorted by name
  std::map<std::string, std::string> payload;
  std::string err;
};

struct Reader {
  std::string prefix;
  int num_shards = 1;
  std::vector<Entry> entries;  // sorted
  std::string err;
  bool v1 = false;
  std::vector<std::string> mem;  // V1: assembled tensor bytes, parallel to entries
};

// ------------------------------------------------------------------ Saver V1 (tensor-slice files)
static size_t dtype_size(int dt) {
  switch (dt) {
    case 1: case 3: return 4;   // DT_FLOAT, DT_INT32
    case 2: case 9: return 8;   // DT_DOUBLE, DT_INT64
    case 4: case 10: return 1;  // DT_UINT8, DT_BOOL
    default: return 0;
  }
}

// bytes of a tensor of `dims` x `es`, refused when a dimension is negative or the product passes `limit`
// (sizes in a file are untrusted: a crafted shape must not overflow int64 or ask for an absurd allocation)
static bool checked_bytes(const std::vector<int64_t>& dims, size_t es, uint64_t limit, int64_t* out) {
  uint64_t n = es;
  bool zero = false;
  for (int64_t d : dims) {
    if (d < 0) return false;
    if (d == 0) zero = true;
  }
  if (zero) { *out = 0; return true; }
  for (int64_t d : dims) {
    if (n > limit / (uint64_t)d) return false;
    n *= (uint64_t)d;
  }
  *out = (int64_t)n;
  return true;
}

struct Extent {
  int64_t start = 0;
  int64_t length = -1;  // -1: the whole dimension
};

static bool decode_slice(const uint8_t* b, size_t len, std::vector<Extent>* out) {
  PbReader r{b, b + len};
  int f, wt;
  while (r.next(&f, &wt)) {
    if (f == 1 && wt == 2) {
      const uint8_t* eb; size_t el;
      if (!r.bytes(&eb, &el)) return false;
      PbReader e{eb, eb + el};
      Extent x;
      int f2, wt2;
      while (e.next(&f2, &wt2)) {
        if (f2 == 1 && wt2 == 0) x.start = (int64_t)e.varint();
        else if (f2 == 2 && wt2 == 0) x.length = (int64_t)e.varint();
        else e.skip(wt2);
      }
      if (!e.ok) return false;
      out->push_back(x);
    } else {
      r.skip(wt);
    }
  }
  return r.ok;
}

// TensorProto -> raw little-endian element bytes of `dtype` (the type comes from the meta entry: the
// V1 writer fills only the typed *_val field).  Accepts packed and unpacked repeated encodings and
// tensor_content.
static bool decode_tensor_proto(const uint8_t* b, size_t len, int dtype, std::string* out) {
  PbReader r{b, b + len};
  int f, wt;
  while (r.next(&f, &wt)) {
    const bool fixed4 = (f == 5 && dtype == 1), fixed8 = (f == 6 && dtype == 2);
    const bool vint = (f == 7 && (dtype == 3 || dtype == 4)) || (f == 10 && dtype == 9) || (f == 11 && dtype == 10);
    if (f == 4 && wt == 2) {  // tensor_content
      const uint8_t* c; size_t cl;
      if (!r.bytes(&c, &cl)) return false;
      out->append((const char*)c, cl);
    } else if ((fixed4 || fixed8) && wt == 2) {  // packed floats / doubles are already little-endian bytes
      const uint8_t* c; size_t cl;
      if (!r.bytes(&c, &cl) || cl % (fixed4 ? 4 : 8)) return false;
      out->append((const char*)c, cl);
    } else if (fixed4 && wt == 5) {
      if (r.end - r.p < 4) return false;
      out->append((const char*)r.p, 4);
      r.p += 4;
    } else if (fixed8 && wt == 1) {
      if (r.end - r.p < 8) return false;
      out->append((const char*)r.p, 8);
      r.p += 8;
    } else if (vint && (wt == 2 || wt == 0)) {
      const size_t es = dtype_size(dtype);
      PbReader v{r.p, r.end};
      if (wt == 2) {
        const uint8_t* c; size_t cl;
        if (!r.bytes(&c, &cl)) return false;
        v = PbReader{c, c + cl};
        if (!cl) continue;
      }
      do {
        uint64_t x = v.varint();
        if (!v.ok) return false;
        out->append((const char*)&x, es);  // little-endian host: the low `es` bytes are the value
      } while (wt == 2 && v.p < v.end);
      if (wt == 0) r.p = v.p;
    } else {
      r.skip(wt);
    }
  }
  return r.ok;
}

// copy a slice (extents `ex`, row-major `src`) into the full row-major tensor `dst` of shape `dims`
static bool copy_slice(std::string* dst, const std::vector<int64_t>& dims, const std::vector<Extent>& ex,
                       const std::string& src, size_t es) {
  const size_t nd = dims.size();
  if (ex.size() != nd) return false;
  std::vector<int64_t> start(nd), len(nd);
  int64_t n = 1;
  for (size_t d = 0; d < nd; ++d) {
    start[d] = ex[d].length < 0 ? 0 : ex[d].start;
    len[d] = ex[d].length < 0 ? dims[d] : ex[d].length;
    if (start[d] < 0 || len[d] < 0 || start[d] > dims[d] || len[d] > dims[d] - start[d]) return false;
    n *= len[d];   // <= the product of dims, which the caller has bounded
  }
  if ((uint64_t)n * es != src.size()) return false;
  if (n == 0) return true;
  if (nd == 0) { memcpy(&(*dst)[0], src.data(), es); return true; }
  std::vector<int64_t> stride(nd, 1);
  for (size_t d = nd - 1; d-- > 0;) stride[d] = stride[d + 1] * dims[d + 1];
  const size_t run = (size_t)len[nd - 1] * es;
  std::vector<int64_t> idx(nd, 0);
  const char* sp = src.data();
  for (;;) {
    int64_t off = 0;
    for (size_t d = 0; d < nd; ++d) off += (start[d] + idx[d]) * stride[d];
    memcpy(&(*dst)[(size_t)off * es], sp, run);
    sp += run;
    size_t d = nd - 1;  // odometer over all dimensions but the last
    for (;;) {
      if (d == 0) return true;
      --d;
      if (++idx[d] < len[d]) break;
      idx[d] = 0;
    }
  }
}

static bool load_v1(const std::string& file, Reader* r) {
  std::vector<std::pair<std::string, std::string>> kv;
  if (!read_table(file, &kv, &r->err)) return false;
  std::map<std::string, size_t> by_name;
  bool have_meta = false;
  uint64_t declared = 0;
  // pass 1: the meta entry (key "") names every tensor with its full shape and type
  for (auto& e : kv) {
    if (!e.first.empty()) continue;
    PbReader top{(const uint8_t*)e.second.data(), (const uint8_t*)e.second.data() + e.second.size()};
    int f, wt;
    while (top.next(&f, &wt)) {
      if (!(f == 1 && wt == 2)) { top.skip(wt); continue; }
      const uint8_t* mb; size_t ml;
      if (!top.bytes(&mb, &ml)) break;
      PbReader m{mb, mb + ml};
      int f2, wt2;
      while (m.next(&f2, &wt2)) {
        if (!(f2 == 1 && wt2 == 2)) { m.skip(wt2); continue; }
        const uint8_t* tb; size_t tl;
        if (!m.bytes(&tb, &tl)) break;
        PbReader t{tb, tb + tl};
        Entry en;
        int f3, wt3;
        while (t.next(&f3, &wt3)) {
          const uint8_t* xb; size_t xl;
          if (f3 == 1 && wt3 == 2) { if (t.bytes(&xb, &xl)) en.name.assign((const char*)xb, xl); }
          else if (f3 == 2 && wt3 == 2) { if (!t.bytes(&xb, &xl) || !decode_shape(xb, xl, &en.dims)) t.ok = false; }
          else if (f3 == 3 && wt3 == 0) en.dtype = (int)t.varint();
          else t.skip(wt3);
        }
        if (!t.ok) { r->err = "bad SavedSliceMeta"; return false; }
        const size_t es = dtype_size(en.dtype);
        if (!es) { r->err = "unsupported dtype in V1 checkpoint for " + en.name; return false; }
        // a V1 file holds every tensor inline; packed varints shrink an element to no less than one byte
        if (!checked_bytes(en.dims, es, 8 * (uint64_t)file.size() + 64, &en.size) ||
            (declared += (uint64_t)en.size) > 8 * (uint64_t)file.size() + 64) {
          r->err = "tensor " + en.name + " declares more data than the file can hold";
          return false;
        }
        by_name[en.name] = r->entries.size();
        r->entries.push_back(en);
        r->mem.emplace_back((size_t)en.size, '\0');
      }
      if (!m.ok) { r->err = "bad SavedTensorSliceMeta"; return false; }
      have_meta = true;
    }
    if (!top.ok) { r->err = "bad SavedTensorSlices (meta)"; return false; }
  }
  if (!have_meta) { r->err = "no SavedTensorSliceMeta entry: not a Saver V1 file"; return false; }
  // pass 2: data entries; every slice is copied into place (a plain Saver writes one full slice per tensor)
  std::vector<int64_t> filled(r->entries.size(), 0);
  for (auto& e : kv) {
    if (e.first.empty()) continue;
    PbReader top{(const uint8_t*)e.second.data(), (const uint8_t*)e.second.data() + e.second.size()};
    int f, wt;
    while (top.next(&f, &wt)) {
      if (!(f == 2 && wt == 2)) { top.skip(wt); continue; }
      const uint8_t* sb; size_t sl;
      if (!top.bytes(&sb, &sl)) break;
      PbReader ss{sb, sb + sl};
      std::string name;
      std::vector<Extent> ex;
      const uint8_t* tp = nullptr; size_t tpl = 0;
      int f2, wt2;
      while (ss.next(&f2, &wt2)) {
        const uint8_t* xb; size_t xl;
        if (f2 == 1 && wt2 == 2) { if (ss.bytes(&xb, &xl)) name.assign((const char*)xb, xl); }
        else if (f2 == 2 && wt2 == 2) { if (!ss.bytes(&xb, &xl) || !decode_slice(xb, xl, &ex)) ss.ok = false; }
        else if (f2 == 3 && wt2 == 2) { ss.bytes(&tp, &tpl); }
        else ss.skip(wt2);
      }
      if (!ss.ok) { r->err = "bad SavedSlice"; return false; }
      auto it = by_name.find(name);
      if (it == by_name.end()) { r->err = "slice of a tensor the meta entry does not list: " + name; return false; }
      const Entry& en = r->entries[it->second];
      std::string bytes;
      if (tp && !decode_tensor_proto(tp, tpl, en.dtype, &bytes)) { r->err = "bad TensorProto for " + name; return false; }
      if (!copy_slice(&r->mem[it->second], en.dims, ex, bytes, dtype_size(en.dtype))) {
        r->err = "slice extents / data size do not fit the tensor shape for " + name;
        return false;
      }
      filled[it->second] += (int64_t)bytes.size();
    }
    if (!top.ok) { r->err = "bad SavedTensorSlices (data)"; return false; }
  }
  for (size_t i = 0; i < r->entries.size(); ++i)
    if (filled[i] != r->entries[i].size) { r->err = "slices do not cover tensor " + r->entries[i].name; return false; }
  r->v1 = true;
  return true;
}

// OrderedCode pieces of tensorflow/core/lib/strings/ordered_code (only what slice keys use)
static void ordered_num(std::string* s, uint64_t v) {
  uint8_t buf[9];
  int len = 0;
  while (v > 0) { ++len; buf[9 - len] = (uint8_t)(v & 0xff); v >>= 8; }
  buf[9 - len - 1] = (uint8_t)len;
  ++len;
  s->append((const char*)buf + 9 - len, (size_t)len);
}
static void ordered_string(std::string* s, const std::string& v) {
  for (char c : v) {
    if (c == '\0') s->append("\0\xff", 2);
    else if ((uint8_t)c == 0xff) s->append("\xff\0", 2);
    else s->push_back(c);
  }
  s->append("\0\x01", 2);
}
static void ordered_small_signed(std::string* s, int64_t v) { s->push_back((char)(0x80 ^ (uint8_t)v)); }  // |v| < 64

static std::string full_slice_proto(size_t ndim) {
  std::string s;
  for (size_t d = 0; d < ndim; ++d) pb_bytes_field(&s, 1, std::string());  // Extent{}: start 0, no length = everything
  return s;
}

static std::string v1_slice_key(const std::string& name, size_t ndim) {
  std::string k;
  ordered_num(&k, 0);
  ordered_string(&k, name);
  ordered_num(&k, ndim);
  for (size_t d = 0; d < ndim; ++d) {
    ordered_small_signed(&k, 0);   // start
    ordered_small_signed(&k, -1);  // length: kFullExtent
  }
  return k;
}

static bool encode_tensor_proto(const Entry& e, const std::string& bytes, std::string* out, std::string* err) {
  const size_t es = dtype_size(e.dtype);
  if (e.dtype == 1 || e.dtype == 2) {
    pb_bytes_field(out, e.dtype == 1 ? 5 : 6, bytes);  // packed float_val / double_val
  } else if (e.dtype == 3 || e.dtype == 9) {
    std::string packed;
    for (size_t i = 0; i + es <= bytes.size(); i += es) {
      int64_t v;
      if (es == 4) { int32_t x; memcpy(&x, &bytes[i], 4); v = x; } else memcpy(&v, &bytes[i], 8);
      put_varint64(&packed, (uint64_t)v);  // negative values sign-extend to ten bytes, as protobuf does
    }
    pb_bytes_field(out, e.dtype == 3 ? 7 : 10, packed);
  } else {
    *err = "Saver V1 writer supports float/double/int32/int64 only (" + e.name + ")";
    return false;
  }
  return true;
}

}  // namespace dtfckpt

using namespace dtfckpt;

extern "C" {

uint32_t dtf_crc32c(const void* data, size_t n) { return crc32c_extend(0, (const uint8_t*)data, n); }
uint32_t dtf_crc32c_mask(uint32_t crc) { return crc_mask(crc); }
uint32_t dtf_crc32c_unmask(uint32_t crc) { return crc_unmask(crc); }

void* dtf_ckpt_writer_new(const char* prefix) {
  Writer* w = new Writer();
  w->prefix = prefix;
  return w;
}

// dtype is a tensorflow.DataType enum value (DT_FLOAT = 1, DT_INT32 = 3, DT_INT64 = 9, ...)
int dtf_ckpt_writer_add(void* wp, const char* name, int dtype, int ndim, const int64_t* dims, const void* data,
                        int64_t nbytes) {
  Writer* w = (Writer*)wp;
  if (!*name) { w->err = "empty tensor name is reserved for the bundle header"; return -1; }
  if (w->entries.count(name)) { w->err = std::string("duplicate tensor ") + name; return -1; }
  Entry e;
  e.name = name;
  e.dtype = dtype;
  e.dims.assign(dims, dims + ndim);
  e.size = nbytes;
  w->entries[name] = e;
  w->payload[name].assign((const char*)data, (size_t)nbytes);
  return 0;
}

int dtf_ckpt_writer_finish(void* wp) {
  Writer* w = (Writer*)wp;
  std::string data;
  std::vector<std::pair<std::string, std::string>> kv;
  kv.emplace_back("", encode_header(1));
  for (auto& it : w->entries) {  // std::map iterates in bytewise key order, as the table requires
    Entry& e = it.second;
    const std::string& bytes = w->payload[it.first];
    e.offset = (int64_t)data.size();
    e.crc = crc_mask(crc32c_extend(0, (const uint8_t*)bytes.data(), bytes.size()));
    data.append(bytes);
    kv.emplace_back(e.name, encode_entry(e));
  }
  if (!write_file_atomic(shard_name(w->prefix, 0, 1), data, &w->err)) return -1;
  if (!write_file_atomic(w->prefix + ".index", build_table(kv, 262144), &w->err)) return -1;
  return 0;
}

// Saver V1: one sstable <prefix> (meta entry + one full slice per tensor); see the file header.
int dtf_ckpt_writer_finish_v1(void* wp) {
  Writer* w = (Writer*)wp;
  std::map<std::string, std::string> kv;  // the table wants bytewise key order
  std::string meta;
  for (auto& it : w->entries) {
    const Entry& e = it.second;
    std::string t;
    pb_bytes_field(&t, 1, e.name);
    pb_bytes_field(&t, 2, encode_shape(e.dims));
    pb_varint_field(&t, 3, (uint64_t)e.dtype);
    pb_bytes_field(&t, 4, full_slice_proto(e.dims.size()));
    pb_bytes_field(&meta, 1, t);  // SavedTensorSliceMeta.tensor
    std::string ss, tp;
    pb_bytes_field(&ss, 1, e.name);
    pb_bytes_field(&ss, 2, full_slice_proto(e.dims.size()));
    if (!encode_tensor_proto(e, w->payload[it.first], &tp, &w->err)) return -1;
    pb_bytes_field(&ss, 3, tp);
    std::string sts;
    pb_bytes_field(&sts, 2, ss);  // SavedTensorSlices.data
    kv[v1_slice_key(e.name, e.dims.size())] = sts;
  }
  std::string ver;
  pb_varint_field(&ver, 1, 1);  // VersionDef.producer = TF_CHECKPOINT_VERSION (1)
  pb_bytes_field(&meta, 2, ver);
  std::string top;
  pb_bytes_field(&top, 1, meta);  // SavedTensorSlices.meta
  kv[""] = top;
  std::vector<std::pair<std::string, std::string>> sorted(kv.begin(), kv.end());
  if (!write_file_atomic(w->prefix, build_table(sorted, 262144), &w->err)) return -1;
  return 0;
}

int dtf_ckpt_reader_is_v1(void* rp) { return ((Reader*)rp)->v1 ? 1 : 0; }

const char* dtf_ckpt_writer_error(void* wp) { return ((Writer*)wp)->err.c_str(); }
void dtf_ckpt_writer_free(void* wp) { delete (Writer*)wp; }

static Reader* reader_open(Reader* r);

// Never throws across the C boundary: an allocation a damaged file provokes becomes a reader error.
void* dtf_ckpt_reader_open(const char* prefix) {
  Reader* r = new Reader();
  r->prefix = prefix;
  try {
    reader_open(r);
  } catch (const std::exception& ex) {
    r->entries.clear();
    r->mem.clear();
    r->err = std::string("cannot load checkpoint: ") + ex.what();
  }
  return r;
}

static Reader* reader_open(Reader* r) {
  std::string file;
  if (!read_file(r->prefix + ".index", &file)) {
    // no bundle index: a Saver V1 checkpoint is the single file <prefix> itself
    if (read_file(r->prefix, &file) && file.size() >= 48 &&
        get_fixed64((const uint8_t*)file.data() + file.size() - 8) == kTableMagic) {
      load_v1(file, r);
      return r;
    }
    r->err = "cannot read " + r->prefix + ".index (V2) and " + r->prefix + " is not a Saver V1 file";
    return r;
  }
  std::vector<std::pair<std::string, std::string>> kv;
  if (!read_table(file, &kv, &r->err)) return r;
  bool have_header = false;
  for (auto& e : kv) {
    if (e.first.empty()) {
      int endian = 0;
      if (!decode_header((const uint8_t*)e.second.data(), e.second.size(), &r->num_shards, &endian)) {
        r->err = "bad BundleHeaderProto";
        return r;
      }
      if (endian != 0) { r->err = "big-endian bundles are not supported"; return r; }
      have_header = true;
      continue;
    }
    Entry en;
    en.name = e.first;
    if (!decode_entry((const uint8_t*)e.second.data(), e.second.size(), &en)) {
      r->err = "bad BundleEntryProto for " + e.first;
      return r;
    }
    int64_t want = 0;
    const size_t es = dtype_size(en.dtype);
    if (en.size < 0 || en.offset < 0 || en.offset > INT64_MAX - en.size || en.shard_id < 0 ||
        (have_header && en.shard_id >= r->num_shards) ||
        !checked_bytes(en.dims, es ? es : 1, (uint64_t)INT64_MAX, &want) ||
        (es && !en.has_slices && want != en.size)) {
      r->err = "inconsistent BundleEntryProto for " + e.first + " (shape, size, offset or shard)";
      return r;
    }
    r->entries.push_back(en);
  }
  if (!have_header) { r->err = "bundle header (empty key) missing"; return r; }
  // an entry must lie inside its shard (checked when the shard is there: listing variables needs only the index)
  std::map<int, int64_t> shard_bytes;
  for (const Entry& en : r->entries) {
    if (en.shard_id >= r->num_shards) { r->err = "entry " + en.name + " names a shard the header does not have"; return r; }
    auto it = shard_bytes.find(en.shard_id);
    if (it == shard_bytes.end()) {
      int64_t n = -1;
      if (FILE* f = fopen(shard_name(r->prefix, en.shard_id, r->num_shards).c_str(), "rb")) {
        if (fseek(f, 0, SEEK_END) == 0) n = (int64_t)ftell(f);
        fclose(f);
      }
      it = shard_bytes.emplace(en.shard_id, n).first;
    }
    if (it->second >= 0 && en.offset + en.size > it->second) {
      r->err = "entry " + en.name + " reaches past the end of its data shard (truncated checkpoint?)";
      return r;
    }
  }
  return r;
}

const char* dtf_ckpt_reader_error(void* rp) { return ((Reader*)rp)->err.c_str(); }
int dtf_ckpt_reader_num(void* rp) { return (int)((Reader*)rp)->entries.size(); }
int dtf_ckpt_reader_num_shards(void* rp) { return ((Reader*)rp)->num_shards; }

// Fills name (up to name_cap), dtype, ndim, dims (up to 8), nbytes.  Returns 0 on success.
int dtf_ckpt_reader_entry(void* rp, int i, char* name, int name_cap, int* dtype, int* ndim, int64_t* dims,
                          int64_t* nbytes, int* sliced) {
  Reader* r = (Reader*)rp;
  if (i < 0 || i >= (int)r->entries.size()) return -1;
  const Entry& e = r->entries[i];
  snprintf(name, (size_t)name_cap, "%s", e.name.c_str());
  if (e.dims.size() > 8) return -2;   // the caller's dims[] holds 8: refuse rather than report a rank it cannot see
  *dtype = e.dtype;
  *ndim = (int)e.dims.size();
  for (size_t k = 0; k < e.dims.size(); ++k) dims[k] = e.dims[k];
  *nbytes = e.size;
  *sliced = e.has_slices ? 1 : 0;
  return 0;
}

// Reads tensor i into dst (nbytes must equal the entry size); verifies the crc32c.
int dtf_ckpt_reader_read(void* rp, int i, void* dst, int64_t nbytes) {
  Reader* r = (Reader*)rp;
  if (i < 0 || i >= (int)r->entries.size()) return -1;
  const Entry& e = r->entries[i];
  if (nbytes != e.size) { r->err = "size mismatch for " + e.name; return -1; }
  if (r->v1) {
    if (nbytes) memcpy(dst, r->mem[(size_t)i].data(), (size_t)nbytes);
    return 0;
  }
  std::string path = shard_name(r->prefix, e.shard_id, r->num_shards);
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { r->err = "cannot open " + path; return -1; }
  bool ok = fseek(f, (long)e.offset, SEEK_SET) == 0 &&
            (nbytes == 0 || fread(dst, 1, (size_t)nbytes, f) == (size_t)nbytes);
  fclose(f);
  if (!ok) { r->err = "short read from " + path; return -1; }
  uint32_t got = crc32c_extend(0, (const uint8_t*)dst, (size_t)nbytes);
  if (e.crc != 0 && crc_unmask(e.crc) != got) { r->err = "tensor checksum mismatch for " + e.name; return -2; }
  return 0;
}

void dtf_ckpt_reader_free(void* rp) { delete (Reader*)rp; }

}  // extern "C"
