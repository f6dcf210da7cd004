// Asynchronous parameter server over peer-to-peer stores (the reference's only training mode: workers push
// gradients to the ps tasks, the ps applies them in arrival order with no locking and workers read the variables
// back; reference .idea/MNISTDist.py:98-111, :174-188 -- there through gRPC, 13.1 MB each way per worker step).
//
// MI355X-native transport: xGMI is a point-to-point mesh and every task maps every other task's mailbox (hipIpc), so a
// message is plain stores into the peer's uncached memory, a system-scope fence and a sequence flag -- no host round
// trip, no collective, nobody waits for anybody who is not the addressee:
//
//   worker w, message n (gradient of its n-th step):   ps_push
//       grads[ranges of ps p] -> mailbox(p).slot[w][n % NB], header {params version used, kind}, flags[b] = n
//       (write-through stores; the flag goes up when they are acknowledged: peer_store.hpp)
//   ps p:   ps_serve, the resident service kernel: its leader wave polls the mailbox flags and decides which complete
//           message is served next; its 64 service workgroups apply it to the fp32 master (+ Adam moments) of the
//           ps's ranges and, in the same pass, push what the worker's kernels read -- bf16 for wd1, fp32 for the rest
//           -- into box(w).slot[n % NB] with the header {global_step, stop} and flags[b] = n.  Arrival order = the
//           order the leader's poll finds them in; global_step and the staleness histogram are device counters.
//   worker w, before computing gradient n + 1:   ps_install of response n - k (k = staleness bound): waits for it,
//           copies it into the parameter arena / bf16 compute copy.  k = 0 is the reference's pull-after-push; k > 0
//           overlaps the round trip with k steps of compute.  NB = k + 1 slots make every buffer reuse safe without
//           further handshakes (slot n % NB is rewritten by message n + NB, which the worker only sends after
//           installing response n + NB - 1 - k = n).
//
// Work split: the ps's ranges are concatenated into one float4 index space cut into 256-float4 chunks; workgroup b
// of EVERY kernel owns chunks b, b + 64, ... and depends only on workgroup b of its peer (flag b), so there is no
// grid-wide barrier anywhere.  Every wait is bounded and reports through a status word.
#include "common.hpp"
#include "peer_store.hpp"

namespace dtf {

constexpr int kPsBlocks = 64;       // workgroups = flags per message
constexpr int kPsThreads = 256;     // one float4 per thread per chunk
constexpr int kPsMaxRanges = 8;
constexpr int kPsMaxWorkers = 16;

struct PsRanges {
  int count;
  long lo[kPsMaxRanges], hi[kPsMaxRanges];   // float offsets in the arena, multiples of 4
};

__device__ __forceinline__ long ps_total4(const PsRanges& r) {
  long t = 0;
  for (int k = 0; k < r.count; ++k) t += (r.hi[k] - r.lo[k]) >> 2;
  return t;
}
// concatenated float4 index -> arena float offset
__device__ __forceinline__ long ps_map(const PsRanges& r, long j4) {
  for (int k = 0; k < r.count; ++k) {
    const long n4 = (r.hi[k] - r.lo[k]) >> 2;
    if (j4 < n4) return r.lo[k] + 4 * j4;
    j4 -= n4;
  }
  return -1;
}

__device__ __forceinline__ void ps_raise(unsigned* flags, unsigned seq) {
  dp_puts_done();           // this thread's (write-through) stores into the peer's buffer are acknowledged ...
  __syncthreads();          // ... for every thread of the workgroup, before its flag goes up
  if (threadIdx.x == 0) dp_raise(flags + blockIdx.x, seq);
}

// bounded wait of workgroup b for flag b; false on timeout (status set)
__device__ __forceinline__ bool ps_wait(const unsigned* flags, unsigned seq, long long timeout_ticks, int* status) {
  __shared__ int ok;
  __syncthreads();          // (a second wait in the same kernel: nobody still reads the previous verdict)
  if (threadIdx.x == 0) {
    ok = 1;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flags + blockIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      if (wall_clock64() - t0 > timeout_ticks) {
        *status = 1;
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  return ok != 0;
}

// ------------------------------------------------------------------------------------------ worker -> ps
struct PsPushArgs {
  const float* grads;        // local gradient arena
  float* dst;                // mailbox slot on the ps, arena-indexed [n_total]
  unsigned* flags;           // [kPsBlocks] of that slot
  unsigned* hdr;             // [4] of that slot: {version lo, version hi, kind, 0}
  const int64_t* version;    // device word: global_step of the parameters this gradient was computed with
  PsRanges r;
  unsigned seq, kind;        // kind 1 = gradient, 2 = goodbye, 3 / 4 / 5 = request for the fp32 master / Adam m / Adam v (no data)
};

__global__ __launch_bounds__(kPsThreads) void ps_push(PsPushArgs a) {
  if (a.kind == 1) {
    const long total4 = ps_total4(a.r);
    for (long c0 = blockIdx.x; c0 * kPsThreads < total4; c0 += 4 * kPsBlocks) {   // four loads in flight per thread
      long off[4];
      f32x4 g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long j = (c0 + (long)u * kPsBlocks) * kPsThreads + threadIdx.x;
        off[u] = j < total4 ? ps_map(a.r, j) : -1;
        g[u] = *reinterpret_cast<const f32x4*>(a.grads + (off[u] < 0 ? 0 : off[u]));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (off[u] >= 0) dp_put16(a.dst + off[u], g[u]);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const uint64_t v = (uint64_t)*a.version;
    uc_store(a.hdr + 0, (unsigned)v);
    uc_store(a.hdr + 1, (unsigned)(v >> 32));
    uc_store(a.hdr + 2, a.kind);
  }
  ps_raise(a.flags, a.seq);
}

// ------------------------------------------------------------------------------------------ worker: install a response
struct PsInstallArgs {
  const unsigned* box_flags;
  const unsigned* box_hdr;
  const float* box_f32;
  const bf16* box_bf16;      // nullptr: everything arrives as fp32
  long n_dense;
  PsRanges r;
  unsigned seq;
  float* params;             // local fp32 arena (tail tensors; everything in the fp32 configuration)
  bf16* shadow;              // local bf16 compute copy (receives [0, n_dense) directly)
  const float* full;         // != nullptr: ALSO take the fp32 master of the ranges from here (checkpoint pull)
  const unsigned* full_flags;
  int64_t* state;            // device step state: state[1] = global_step   (header owner only)
  int64_t* version;          // device word: version of the installed parameters (header owner only)
  float* full_dst;           // where the `full` area goes: nullptr = params, else this arena (Adam m or v)
  unsigned* host_hdr;        // host-visible [8]: {global_step lo, hi, stop, seq, adam updates}   (header owner only)
  int owner;                 // 1: this ps owns global_step
  int* status;
  long long timeout_ticks;
};

__global__ __launch_bounds__(kPsThreads) void ps_install(PsInstallArgs a) {
  // (a timed-out wait is mirrored into the host-visible header: a worker that runs ahead of its installs -- staleness
  //  k > 0 -- never synchronises the stream, and must still hear that its parameters are no longer being refreshed)
  if (!ps_wait(a.box_flags, a.seq, a.timeout_ticks, a.status) ||
      (a.full && !ps_wait(a.full_flags, a.seq, a.timeout_ticks, a.status))) {
    if (threadIdx.x == 0) {
      a.host_hdr[5] = 1;
      __threadfence_system();
    }
    return;
  }
  const long total4 = ps_total4(a.r);
  for (long c0 = blockIdx.x; c0 * kPsThreads < total4; c0 += 4 * kPsBlocks) {     // four loads in flight per thread
    long off[4];
    f32x4 x[4];
    bf16x4 h[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long j = (c0 + (long)u * kPsBlocks) * kPsThreads + threadIdx.x;
      off[u] = j < total4 ? ps_map(a.r, j) : -1;
      const long o = off[u] < 0 ? 0 : off[u];
      if (a.box_bf16 && o < a.n_dense) h[u] = *reinterpret_cast<const bf16x4*>(a.box_bf16 + o);
      else x[u] = *reinterpret_cast<const f32x4*>(a.box_f32 + o);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (off[u] < 0) continue;
      if (a.box_bf16 && off[u] < a.n_dense) *reinterpret_cast<bf16x4*>(a.shadow + off[u]) = h[u];
      else *reinterpret_cast<f32x4*>(a.params + off[u]) = x[u];
      if (a.full)
        *reinterpret_cast<f32x4*>((a.full_dst ? a.full_dst : a.params) + off[u]) = *reinterpret_cast<const f32x4*>(a.full + off[u]);
    }
  }
  if (a.owner && blockIdx.x == 0 && threadIdx.x == 0) {
    const uint64_t step = (uint64_t)a.box_hdr[0] | ((uint64_t)a.box_hdr[1] << 32);
    a.state[1] = (int64_t)step;
    *a.version = (int64_t)step;
    // sequence number LAST (and zeroed first): the host reads it before and after the other words and takes them
    // only if it saw the same non-zero number twice -- the words are rewritten while the host may be looking
    a.host_hdr[3] = 0;
    __threadfence_system();
    a.host_hdr[0] = a.box_hdr[0];
    a.host_hdr[1] = a.box_hdr[1];
    a.host_hdr[2] = a.box_hdr[2];
    a.host_hdr[4] = a.box_hdr[3];
    __threadfence_system();
    a.host_hdr[3] = a.seq;
    __threadfence_system();
  }
}

// ------------------------------------------------------------------------------------------ ps: resident service loop
// The reference's ps is a passive, always-on server (M:105-106: server.join()).  ps_serve is its GPU form: ONE launch
// serves every message that arrives while it runs -- poll, apply in arrival order, answer, count -- with no host round
// trip per gradient (the first version synchronised the stream and went through Python once per message: 2.5 x the
// local step at staleness 0).  The host only relaunches it (it leaves after `budget` or when nothing has arrived for
// `idle`, so that lost workers can be noticed and the GPU never holds a kernel that cannot end) and reads the counters
// the leader leaves in pinned memory.
//
//   workgroup 64 (one wave) = LEADER: polls the mailbox flags round-robin, decides what is served next and publishes the
//       decision {round, worker, message, kind, apply?, global_step, stop, Adam count} in a ring in uncached memory;
//       global_step, the per-worker sequence numbers and the staleness histogram are its private state.
//   workgroups 0 .. 63 = SERVICE: each owns chunks b, b + 64, ... of the ps's ranges, takes the decisions in order and
//       applies / answers its chunks; its flag b on the worker's box goes up when they are out.  No workgroup ever
//       touches another one's chunks, so the only ordering needed is "decisions in order", which the ring gives.
// Every wait in here ends: the leader always publishes a final "leave" decision, and the service workgroups' wait
// for a decision is itself bounded.
constexpr int kPsRing = 32;
constexpr int kPsHistBins = 64;
enum {
  PSC_ROUND = 0, PSC_START = 1, PSC_STEP_LO = 2, PSC_STEP_HI = 3, PSC_APPLIED = 4, PSC_ALL_DONE = 5, PSC_DROPPED = 6,
  PSC_EXPECT = 8,                       // [16]
  PSC_HIST = 32,                        // [64]
  PSC_PROG = 96,                        // [64] decisions consumed by service workgroup b
  PSC_RING = 160,                       // [kPsRing][8]: {tag = round + 1, worker, message, kind (0 = leave), flags, step lo, step hi, Adam count}
  PSC_WORDS = 160 + kPsRing * 8
};
// host-visible summary (pinned): {round, step lo, step hi, applied, all_done, error, -, -, expect[16], hist[64]}
constexpr int kPsOutWords = 8 + 16 + kPsHistBins;

struct PsServeArgs {
  unsigned* ctl;
  char* own;                            // this ps's mailbox
  char* box[kPsMaxWorkers];             // every worker's box as mapped here
  int W, NB, p;
  long n_total, n_dense;
  long mb_flags, mb_hdr, mb_data, bx_flags, bx_hdr, bx_full_flags, bx_f32, bx_bf16, bx_full;   // async_push.py _Layout
  PsRanges r;
  float* w;
  float* m;
  float* v;
  const float* hyper;
  unsigned adam_t0;
  unsigned long long training_iter;
  unsigned round0;
  long long budget_ticks, idle_ticks;
  unsigned* out;
};

template <bool ADAM>
__global__ __launch_bounds__(kPsThreads) void ps_serve(PsServeArgs a) {
  const int tid = threadIdx.x, b = blockIdx.x;
  unsigned* const c = a.ctl;
  if (b == kPsBlocks) {
    // ------------------------------------------------------------------ leader (one wave)
    if (tid >= 64) return;
    __shared__ unsigned s_expect[kPsMaxWorkers], s_hist[kPsHistBins];
    const int lane = tid;
    if (lane < kPsMaxWorkers) s_expect[lane] = uc_load(c + PSC_EXPECT + lane);
    s_hist[lane] = 0;
    unsigned round = a.round0, start = uc_load(c + PSC_START), applied = uc_load(c + PSC_APPLIED);
    unsigned long long step = (unsigned long long)uc_load(c + PSC_STEP_LO) | ((unsigned long long)uc_load(c + PSC_STEP_HI) << 32);
    unsigned min_prog = a.round0 >= (unsigned)kPsRing ? a.round0 - kPsRing + 1 : 0;   // every workgroup consumed the previous launches' rounds
    int all_done = 0, error = 0;
    const long long t_begin = wall_clock64();
    long long t_idle = t_begin;
    for (;;) {
      const unsigned dropped = uc_load(c + PSC_DROPPED);
      int pick = -1;
      unsigned e_pick = 0;
      bool alive = false;
      for (int i = 0; i < a.W; ++i) {
        const int w = (start + i) % a.W;
        const unsigned e = s_expect[w];
        if (!e || ((dropped >> w) & 1u)) continue;
        alive = true;
        const unsigned* f = reinterpret_cast<const unsigned*>(a.own + a.mb_flags) + ((size_t)w * a.NB + e % a.NB) * kPsBlocks;
        if (__ballot(uc_load(f + lane) == e) == ~0ull) {
          pick = w;
          e_pick = e;
          break;
        }
      }
      const long long now = wall_clock64();
      unsigned kind = 0, vlo = 0, vhi = 0;
      if (pick >= 0) {
        const unsigned* h = reinterpret_cast<const unsigned*>(a.own + a.mb_hdr) + ((size_t)pick * a.NB + e_pick % a.NB) * 4;
        vlo = uc_load(h);
        vhi = uc_load(h + 1);
        kind = uc_load(h + 2);
        if (kind == 2) {                 // goodbye: nothing to answer
          if (lane == 0) s_expect[pick] = 0;
          t_idle = now;
          continue;
        }
      } else {
        if (!alive) all_done = 1;
        else if (now - t_begin <= a.budget_ticks && now - t_idle <= a.idle_ticks) {
          __builtin_amdgcn_s_sleep(8);
          continue;
        }
      }
      // ring back-pressure: the entry about to be overwritten (round - kPsRing) must have been consumed by everybody
      if (round >= (unsigned)kPsRing && min_prog < round - kPsRing + 1) {
        const long long t0 = wall_clock64();
        for (;;) {
          unsigned mn = 0xffffffffu;
          if (lane < kPsBlocks / 4) {
            const u32x4 pr = uc_load4(c + PSC_PROG + 4 * lane);
            mn = min(min(pr[0], pr[1]), min(pr[2], pr[3]));
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, o, 64));
          min_prog = mn;
          if (min_prog >= round - kPsRing + 1) break;
          if (wall_clock64() - t0 > a.budget_ticks + 100000000ll) {   // a service workgroup is stuck: give up loudly
            error = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(4);
        }
      }
      unsigned flags = 0;
      if (pick >= 0 && !error) {
        const unsigned long long version = (unsigned long long)vlo | ((unsigned long long)vhi << 32);
        const bool apply = kind == 1 && step < a.training_iter;
        if (apply) {
          const unsigned long long d = step > version ? step - version : 0;
          if (lane == 0) s_hist[d < (unsigned long long)kPsHistBins ? (int)d : kPsHistBins - 1] += 1;
          ++step;
          ++applied;
        }
        flags = (apply ? 1u : 0u) | (step >= a.training_iter ? 2u : 0u);
        if (lane == 0) s_expect[pick] = e_pick + 1;
        start = (start + 1) % a.W;
        t_idle = now;
      } else {
        kind = 0;                        // leave
      }
      if (lane == 0) {
        unsigned* e = c + PSC_RING + (round % kPsRing) * 8;
        uc_store(e + 1, (unsigned)pick);
        uc_store(e + 2, e_pick);
        uc_store(e + 3, kind);
        uc_store(e + 4, flags);
        uc_store(e + 5, (unsigned)step);
        uc_store(e + 6, (unsigned)(step >> 32));
        uc_store(e + 7, a.adam_t0 + applied);
        dp_puts_done();
        uc_store(e, round + 1);
      }
      ++round;
      if (kind == 0) break;
    }
    // persistent state + the host's summary
    if (lane < kPsMaxWorkers) {
      uc_store(c + PSC_EXPECT + lane, s_expect[lane]);
      a.out[8 + lane] = s_expect[lane];
    }
    const unsigned h = uc_load(c + PSC_HIST + lane) + s_hist[lane];
    uc_store(c + PSC_HIST + lane, h);
    a.out[8 + 16 + lane] = h;
    if (lane == 0) {
      uc_store(c + PSC_ROUND, round);
      uc_store(c + PSC_START, start);
      uc_store(c + PSC_STEP_LO, (unsigned)step);
      uc_store(c + PSC_STEP_HI, (unsigned)(step >> 32));
      uc_store(c + PSC_APPLIED, applied);
      uc_store(c + PSC_ALL_DONE, (unsigned)all_done);
      a.out[0] = round;
      a.out[1] = (unsigned)step;
      a.out[2] = (unsigned)(step >> 32);
      a.out[3] = applied;
      a.out[4] = (unsigned)all_done;
      a.out[5] = (unsigned)error;
    }
    __threadfence_system();
    return;
  }

  // -------------------------------------------------------------------- service workgroup b
  __shared__ unsigned s_dec[8];
  const float lr = a.hyper[0];
  const float b1 = ADAM ? a.hyper[1] : 0.f, b2 = ADAM ? a.hyper[2] : 0.f, eps = ADAM ? a.hyper[3] : 0.f;
  const long total4 = ps_total4(a.r);
  unsigned round = a.round0;
  for (;;) {
    if (tid == 0) {
      const unsigned* e = c + PSC_RING + (round % kPsRing) * 8;
      const long long t0 = wall_clock64();
      bool ok = true;
      while (uc_load(e) != round + 1) {
        if (wall_clock64() - t0 > 4 * a.budget_ticks + 200000000ll) {   // the leader always sends a "leave": never expected
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      const u32x4 lo = uc_load4(e), hi = uc_load4(e + 4);
      s_dec[0] = lo[0]; s_dec[1] = lo[1]; s_dec[2] = lo[2]; s_dec[3] = ok ? lo[3] : 0u;
      s_dec[4] = hi[0]; s_dec[5] = hi[1]; s_dec[6] = hi[2]; s_dec[7] = hi[3];
    }
    __syncthreads();
    const unsigned kind = s_dec[3];
    if (kind == 0) break;
    const int w = (int)s_dec[1];
    const unsigned n = s_dec[2], fl = s_dec[4];
    const bool apply = (fl & 1u) != 0;
    const int slot = (int)(n % (unsigned)a.NB);
    const float* grad = reinterpret_cast<const float*>(a.own + a.mb_data) + ((size_t)w * a.NB + slot) * a.n_total;
    char* const box = a.box[w];
    const size_t bi = (size_t)a.p * a.NB + slot;
    unsigned* box_flags = reinterpret_cast<unsigned*>(box + a.bx_flags) + bi * kPsBlocks;
    unsigned* box_hdr = reinterpret_cast<unsigned*>(box + a.bx_hdr) + bi * 4;
    float* box_f32 = reinterpret_cast<float*>(box + a.bx_f32) + (size_t)slot * a.n_total;
    bf16* box_bf16 = a.n_dense > 0 ? reinterpret_cast<bf16*>(box + a.bx_bf16) + (size_t)slot * a.n_dense : nullptr;
    float* full = kind >= 3 ? reinterpret_cast<float*>(box + a.bx_full) : nullptr;
    unsigned* full_flags = reinterpret_cast<unsigned*>(box + a.bx_full_flags) + (size_t)a.p * kPsBlocks;
    const float* full_src = kind == 4 ? a.m : (kind == 5 ? a.v : nullptr);
    float lr_t = 0.f;
    if (ADAM) {
      const float t = (float)s_dec[7];           // 1-based count of this update (ps.hip ps_apply_push)
      lr_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
    }
    for (long ch0 = b; ch0 * kPsThreads < total4; ch0 += 4 * kPsBlocks) {     // four chunks in flight per thread
      long off[4];
      f32x4 wv[4], g[4], mv[4], vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long j = (ch0 + (long)u * kPsBlocks) * kPsThreads + tid;
        off[u] = j < total4 ? ps_map(a.r, j) : -1;
        const long o = off[u] < 0 ? 0 : off[u];
        wv[u] = *reinterpret_cast<const f32x4*>(a.w + o);
        if (apply) {
          g[u] = *reinterpret_cast<const f32x4*>(grad + o);
          if (ADAM) {
            mv[u] = *reinterpret_cast<const f32x4*>(a.m + o);
            vv[u] = *reinterpret_cast<const f32x4*>(a.v + o);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (off[u] < 0) continue;
        const long o = off[u];
        if (apply) {
          if (ADAM) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // TF AdamOptimizer, the arithmetic of optim.hip adam4
              mv[u][e] = fmaf(b1, mv[u][e], (1.f - b1) * g[u][e]);
              vv[u][e] = fmaf(b2, vv[u][e], (1.f - b2) * g[u][e] * g[u][e]);
              wv[u][e] -= lr_t * mv[u][e] / (sqrtf(vv[u][e]) + eps);
            }
            *reinterpret_cast<f32x4*>(a.m + o) = mv[u];
            *reinterpret_cast<f32x4*>(a.v + o) = vv[u];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) wv[u][e] = fmaf(-lr, g[u][e], wv[u][e]);   // un-averaged, as the reference applies them
          }
          *reinterpret_cast<f32x4*>(a.w + o) = wv[u];
        }
        if (box_bf16 && o < a.n_dense) {
          const bf16x4 hv = {(bf16)wv[u][0], (bf16)wv[u][1], (bf16)wv[u][2], (bf16)wv[u][3]};
          dp_put8(box_bf16 + o, __builtin_bit_cast(u32x2, hv));
        } else {
          dp_put16(box_f32 + o, wv[u]);
        }
        if (full) dp_put16(full + o, full_src ? *reinterpret_cast<const f32x4*>(full_src + o) : wv[u]);
      }
    }
    if (b == 0 && tid == 0) {
      uc_store(box_hdr + 0, s_dec[5]);
      uc_store(box_hdr + 1, s_dec[6]);
      uc_store(box_hdr + 2, (fl >> 1) & 1u);
      uc_store(box_hdr + 3, s_dec[7]);
    }
    dp_puts_done();          // the answer's stores are write-through: acknowledged = visible to the worker
    __syncthreads();
    if (tid == 0) {
      if (full) dp_raise(full_flags + b, n);
      dp_raise(box_flags + b, n);
      uc_store(c + PSC_PROG + b, round + 1);
    }
    ++round;
    __syncthreads();         // s_dec is rewritten by the next round
  }
  if (tid == 0) uc_store(c + PSC_PROG + b, round + 1);   // the "leave" decision is consumed too (ring back-pressure counts rounds)
}

__global__ void ps_serve_init(unsigned* ctl, int W, unsigned long long step) {
  const int t = threadIdx.x;
  if (t < PSC_WORDS) ctl[t] = 0;
  __syncthreads();
  if (t < W) ctl[PSC_EXPECT + t] = 1;
  if (t == 0) {
    ctl[PSC_STEP_LO] = (unsigned)step;
    ctl[PSC_STEP_HI] = (unsigned)(step >> 32);
  }
}
__global__ void ps_ctl_set(unsigned* ctl, int word, unsigned value) { ctl[word] = value; }

static bool ranges_ok(int count, const long* lo, const long* hi, long n_total) {
  if (count < 1 || count > kPsMaxRanges) return false;
  for (int k = 0; k < count; ++k)
    if (lo[k] < 0 || hi[k] <= lo[k] || hi[k] > n_total || (lo[k] & 3) || (hi[k] & 3)) return false;
  return true;
}
static PsRanges make_ranges(int count, const long* lo, const long* hi) {
  PsRanges r;
  r.count = count;
  for (int k = 0; k < kPsMaxRanges; ++k) {
    r.lo[k] = k < count ? lo[k] : 0;
    r.hi[k] = k < count ? hi[k] : 0;
  }
  return r;
}

}  // namespace dtf

using namespace dtf;

extern "C" {

int dtf_ps_blocks() { return kPsBlocks; }
int dtf_ps_max_workers() { return kPsMaxWorkers; }

// Every pointer below is an address inside a mailbox / box as mapped into THIS process (parallel/async_push.py
// computes them from the layout); n_total bounds every range.
int dtf_ps_push(const void* grads, void* dst, void* flags, void* hdr, const void* version, int nr, const long* lo,
                const long* hi, long n_total, unsigned seq, unsigned kind, hipStream_t s) {
  if (!ranges_ok(nr, lo, hi, n_total) || seq == 0 || kind < 1 || kind > 5) return -2;
  if (((uintptr_t)grads | (uintptr_t)dst) & 15) return -3;
  PsPushArgs a{(const float*)grads, (float*)dst, (unsigned*)flags, (unsigned*)hdr, (const int64_t*)version,
               make_ranges(nr, lo, hi), seq, kind};
  hipLaunchKernelGGL(ps_push, dim3(kPsBlocks), dim3(kPsThreads), 0, s, a);
  return (int)hipGetLastError();
}

int dtf_ps_install(const void* box_flags, const void* box_hdr, const void* box_f32, const void* box_bf16, long n_dense,
                   int nr, const long* lo, const long* hi, long n_total, unsigned seq, void* params, void* shadow,
                   const void* full, const void* full_flags, void* full_dst, void* state, void* version, void* host_hdr,
                   int owner, int* status, double timeout_s, hipStream_t s) {
  if (!ranges_ok(nr, lo, hi, n_total) || seq == 0 || (n_dense & 3)) return -2;
  if (box_bf16 && !shadow) return -3;
  PsInstallArgs a;
  a.box_flags = (const unsigned*)box_flags;
  a.box_hdr = (const unsigned*)box_hdr;
  a.box_f32 = (const float*)box_f32;
  a.box_bf16 = (const bf16*)box_bf16;
  a.n_dense = n_dense;
  a.r = make_ranges(nr, lo, hi);
  a.seq = seq;
  a.params = (float*)params;
  a.shadow = (bf16*)shadow;
  a.full = (const float*)full;
  a.full_flags = (const unsigned*)full_flags;
  a.full_dst = (float*)full_dst;
  a.state = (int64_t*)state;
  a.version = (int64_t*)version;
  a.host_hdr = (unsigned*)host_hdr;
  a.owner = owner;
  a.status = status;
  a.timeout_ticks = (long long)(timeout_s * 1e8);
  hipLaunchKernelGGL(ps_install, dim3(kPsBlocks), dim3(kPsThreads), 0, s, a);
  return (int)hipGetLastError();
}

// ---- resident service loop (ps_serve) ----
long dtf_ps_ctl_bytes() { return (long)PSC_WORDS * 4; }
int dtf_ps_out_words() { return kPsOutWords; }
int dtf_ps_serve_init(void* ctl, int W, unsigned long long step, hipStream_t s) {
  if (!ctl || W < 1 || W > kPsMaxWorkers) return -2;
  hipLaunchKernelGGL(ps_serve_init, dim3(1), dim3(512), 0, s, (unsigned*)ctl, W, step);
  return (int)hipGetLastError();
}
int dtf_ps_ctl_set_dropped(void* ctl, unsigned mask, hipStream_t s) {
  hipLaunchKernelGGL(ps_ctl_set, dim3(1), dim3(1), 0, s, (unsigned*)ctl, (int)PSC_DROPPED, mask);
  return (int)hipGetLastError();
}
// layout9: {mb_flags, mb_hdr, mb_data, bx_flags, bx_hdr, bx_full_flags, bx_f32, bx_bf16, bx_full} byte offsets
// (parallel/async_push.py _Layout); boxes[w]: worker w's box as mapped into this process
int dtf_ps_serve(int adam, void* ctl, void* own, void* const* boxes, int W, int NB, int p, long n_total, long n_dense,
                 const long* layout9, int nr, const long* lo, const long* hi, void* w, void* m, void* v, const void* hyper,
                 unsigned adam_t0, unsigned long long training_iter, unsigned round0, double budget_s, double idle_s,
                 void* out, hipStream_t s) {
  if (!ranges_ok(nr, lo, hi, n_total) || W < 1 || W > kPsMaxWorkers || NB < 1 || (n_dense & 3) || n_dense < 0 || n_dense > n_total)
    return -2;
  if (!ctl || !own || !boxes || !out || (adam && (!m || !v))) return -3;
  PsServeArgs a;
  a.ctl = (unsigned*)ctl;
  a.own = (char*)own;
  for (int k = 0; k < kPsMaxWorkers; ++k) a.box[k] = k < W ? (char*)boxes[k] : nullptr;
  for (int k = 0; k < W; ++k)
    if (!a.box[k]) return -3;
  a.W = W;
  a.NB = NB;
  a.p = p;
  a.n_total = n_total;
  a.n_dense = n_dense;
  a.mb_flags = layout9[0]; a.mb_hdr = layout9[1]; a.mb_data = layout9[2];
  a.bx_flags = layout9[3]; a.bx_hdr = layout9[4]; a.bx_full_flags = layout9[5];
  a.bx_f32 = layout9[6]; a.bx_bf16 = layout9[7]; a.bx_full = layout9[8];
  a.r = make_ranges(nr, lo, hi);
  a.w = (float*)w;
  a.m = (float*)m;
  a.v = (float*)v;
  a.hyper = (const float*)hyper;
  a.adam_t0 = adam_t0;
  a.training_iter = training_iter;
  a.round0 = round0;
  a.budget_ticks = (long long)(budget_s * 1e8);
  a.idle_ticks = (long long)(idle_s * 1e8);
  a.out = (unsigned*)out;
  if (adam) hipLaunchKernelGGL(ps_serve<true>, dim3(kPsBlocks + 1), dim3(kPsThreads), 0, s, a);
  else hipLaunchKernelGGL(ps_serve<false>, dim3(kPsBlocks + 1), dim3(kPsThreads), 0, s, a);
  return (int)hipGetLastError();
}

}  // extern "C"
