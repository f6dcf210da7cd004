// Damaged-file driver for the checkpoint codec (bundle.cc), built with AddressSanitizer + UBSan on the HOST
// (tests/test_checkpoint_fuzz.py).  It takes a valid checkpoint, damages copies of its table file in-process
// (bit flips, byte stamps, truncations, spliced varints) and loads every copy through the public C API the way
// bundle.py does.  Any load may fail with a reader error; none may crash, read out of bounds or overflow.
// The build defines DTF_CKPT_FUZZ, which makes read_block() accept a wrong block checksum so that damage reaches
// the block / protobuf parsers behind it.
//
//   ckpt_fuzz <valid_prefix> <work_prefix> <iterations> <seed>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
void* dtf_ckpt_reader_open(const char* prefix);
const char* dtf_ckpt_reader_error(void* rp);
int dtf_ckpt_reader_num(void* rp);
int dtf_ckpt_reader_entry(void* rp, int i, char* name, int name_cap, int* dtype, int* ndim, int64_t* dims,
                          int64_t* nbytes, int* sliced);
int dtf_ckpt_reader_read(void* rp, int i, void* dst, int64_t nbytes);
void dtf_ckpt_reader_free(void* rp);
}

static uint64_t g_rng;
static uint64_t rnd() {
  g_rng ^= g_rng << 13;
  g_rng ^= g_rng >> 7;
  g_rng ^= g_rng << 17;
  return g_rng;
}

static bool slurp(const std::string& path, std::string* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[65536];
  size_t n;
  out->clear();
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
  fclose(f);
  return true;
}
static void spit(const std::string& path, const std::string& data) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { perror(path.c_str()); exit(2); }
  if (!data.empty()) fwrite(data.data(), 1, data.size(), f);
  fclose(f);
}

static void damage(std::string* s) {
  const int edits = 1 + (int)(rnd() % 4);
  for (int e = 0; e < edits && !s->empty(); ++e) {
    const size_t at = (size_t)(rnd() % s->size());
    switch (rnd() % 6) {
      case 0: (*s)[at] ^= (char)(1u << (rnd() % 8)); break;
      case 1: (*s)[at] = (char)(rnd() & 0xff); break;
      case 2: { static const unsigned char stamps[] = {0x00, 0x7f, 0x80, 0xff}; (*s)[at] = (char)stamps[rnd() % 4]; break; }
      case 3: s->resize(at); break;                                   // truncation
      case 4: {                                                       // a maximal varint spliced over the bytes
        static const char big[] = "\xff\xff\xff\xff\xff\xff\xff\xff\xff\x01";
        for (size_t k = 0; k < 10 && at + k < s->size(); ++k) (*s)[at + k] = big[k];
        break;
      }
      default: s->insert(at, 1, (char)(rnd() & 0xff)); break;         // everything behind shifts by one
    }
  }
}

// what bundle.py does with a prefix; returns 1 when every tensor loaded
static int load(const std::string& prefix, long* tensors) {
  void* r = dtf_ckpt_reader_open(prefix.c_str());
  int ok = 1;
  if (*dtf_ckpt_reader_error(r)) ok = 0;
  const int n = ok ? dtf_ckpt_reader_num(r) : 0;
  for (int i = 0; i < n && ok; ++i) {
    char name[1024];
    int dtype = 0, ndim = 0, sliced = 0;
    int64_t dims[8], nbytes = 0;
    if (dtf_ckpt_reader_entry(r, i, name, (int)sizeof name, &dtype, &ndim, dims, &nbytes, &sliced) != 0) { ok = 0; break; }
    if (ndim < 0 || ndim > 8 || nbytes < 0 || nbytes > (int64_t)(256u << 20)) { ok = 0; break; }  // the caller's own sanity cap
    std::vector<char> buf((size_t)nbytes + 1);
    if (dtf_ckpt_reader_read(r, i, buf.data(), nbytes) != 0) ok = 0;
    else ++*tensors;
  }
  dtf_ckpt_reader_free(r);
  return ok;
}

int main(int argc, char** argv) {
  if (argc != 5) { fprintf(stderr, "usage: %s <valid_prefix> <work_prefix> <iterations> <seed>\n", argv[0]); return 2; }
  const std::string src = argv[1], work = argv[2];
  const long iters = atol(argv[3]);
  g_rng = 0x9e3779b97f4a7c15ull ^ (uint64_t)atoll(argv[4]);
  std::string table, shard;
  const std::string shard_suffix = ".data-00000-of-00001";
  const bool v2 = slurp(src + ".index", &table);
  if (v2) {
    if (!slurp(src + shard_suffix, &shard)) { fprintf(stderr, "no data shard next to %s.index\n", src.c_str()); return 2; }
    spit(work + shard_suffix, shard);
  } else if (!slurp(src, &table)) {
    fprintf(stderr, "cannot read %s(.index)\n", src.c_str());
    return 2;
  }
  const std::string table_path = v2 ? work + ".index" : work;
  spit(table_path, table);
  long tensors = 0;
  if (!load(work, &tensors) || tensors == 0) { fprintf(stderr, "the undamaged copy does not load\n"); return 2; }
  long loaded = 0, refused = 0;
  for (long it = 0; it < iters; ++it) {
    std::string t = table;
    damage(&t);
    spit(table_path, t);
    if (v2 && it % 8 == 7) {            // now and then the data shard is damaged too (short reads, checksum)
      std::string d = shard;
      damage(&d);
      spit(work + shard_suffix, d);
    } else if (v2 && it % 8 == 0) {
      spit(work + shard_suffix, shard);
    }
    long dummy = 0;
    if (load(work, &dummy)) ++loaded; else ++refused;
  }
  printf("format=%s iterations=%ld loaded=%ld refused=%ld\n", v2 ? "v2" : "v1", iters, loaded, refused);
  return 0;
}
