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
//   ps p:   ps_poll (one wave, bounded spin) reports which workers have a complete message n_w waiting;
//           ps_apply_push applies it to the fp32 master (+ Adam moments) of its ranges and, in the same pass, pushes
//           what the worker's kernels read -- bf16 for wd1, fp32 for the rest -- into box(w).slot[n % NB] with the
//           header {global_step, stop} and flags[b] = n.  Arrival order = the order the poll finds them in.
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
  __threadfence_system();   // this thread's stores are visible system-wide ...
  __syncthreads();          // ... for every thread of the workgroup, before its flag goes up
  if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
    for (long c = blockIdx.x; c * kPsThreads < total4; c += kPsBlocks) {
      const long j = c * kPsThreads + threadIdx.x;
      if (j < total4) {
        const long off = ps_map(a.r, j);
        *reinterpret_cast<f32x4*>(a.dst + off) = *reinterpret_cast<const f32x4*>(a.grads + off);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const uint64_t v = (uint64_t)*a.version;
    a.hdr[0] = (unsigned)v;
    a.hdr[1] = (unsigned)(v >> 32);
    a.hdr[2] = a.kind;
  }
  ps_raise(a.flags, a.seq);
}

// ------------------------------------------------------------------------------------------ ps: who is waiting?
struct PsPollArgs {
  const unsigned* flags;     // mailbox flags [W][NB][kPsBlocks]
  const unsigned* hdr;       // mailbox headers [W][NB][4]
  int W, NB, start;
  unsigned expect[kPsMaxWorkers];   // next sequence number of worker w (0 = not watched)
  unsigned* out;             // host-visible [W][4]: {ready, kind, version lo, version hi}
  long long timeout_ticks;
};

// one wave: lane l checks flag l of a worker's slot; a message is complete when all 64 flags carry its number
__global__ __launch_bounds__(64) void ps_poll(PsPollArgs a) {
  const int lane = threadIdx.x;
  const long long t0 = wall_clock64();
  bool any = false;
  for (;;) {
    for (int i = 0; i < a.W; ++i) {
      const int w = (a.start + i) % a.W;
      const unsigned e = a.expect[w];
      bool ready = false;
      if (e) {
        const unsigned* f = a.flags + ((size_t)w * a.NB + e % a.NB) * kPsBlocks;
        ready = __ballot(__hip_atomic_load(f + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == e) == ~0ull;
      }
      if (ready) any = true;
      if (lane == 0) {
        const unsigned* h = a.hdr + ((size_t)w * a.NB + (e ? e % a.NB : 0)) * 4;
        a.out[4 * w + 0] = ready ? 1u : 0u;
        a.out[4 * w + 1] = ready ? h[2] : 0u;
        a.out[4 * w + 2] = ready ? h[0] : 0u;
        a.out[4 * w + 3] = ready ? h[1] : 0u;
      }
    }
    if (any || wall_clock64() - t0 > a.timeout_ticks) break;   // wave-uniform: every lane leaves together
    __builtin_amdgcn_s_sleep(8);
  }
  __threadfence_system();
}

// ------------------------------------------------------------------------------------------ ps: apply + respond
struct PsApplyArgs {
  float* w;                  // fp32 master arena
  float* m;                  // Adam moments (ADAM only)
  float* v;
  const float* hyper;        // [lr, beta1, beta2, eps, ...]
  const float* grad;         // mailbox slot, arena-indexed (apply != 0)
  PsRanges r;
  int apply;                 // 0: respond with the current parameters only (step budget used up, master request)
  float adam_t;              // ADAM: 1-based count of this update
  float* box_f32;            // worker's box slot, arena-indexed fp32
  bf16* box_bf16;            // worker's box slot for [0, n_dense) as bf16, or nullptr (fp32 configuration)
  long n_dense;
  unsigned* box_flags;       // [kPsBlocks]
  unsigned* box_hdr;         // [4]: {global_step lo, hi, stop, 0}
  float* full;               // worker's fp32 "master" area (arena-indexed) or nullptr
  unsigned* full_flags;
  const float* full_src;     // what goes into `full`: nullptr = the (updated) weights, else this arena (Adam m or v)
  unsigned adam_count;       // header word 3: number of Adam updates applied so far (the chief checkpoints it)
  unsigned seq, step_lo, step_hi, stop;
};

template <bool ADAM>
__global__ __launch_bounds__(kPsThreads) void ps_apply_push(PsApplyArgs a) {
  const float lr = a.hyper[0];
  float lr_t = 0.f, b1 = 0.f, b2 = 0.f, eps = 0.f;
  if (ADAM) {
    b1 = a.hyper[1]; b2 = a.hyper[2]; eps = a.hyper[3];
    lr_t = lr * sqrtf(1.f - powf(b2, a.adam_t)) / (1.f - powf(b1, a.adam_t));
  }
  const long total4 = ps_total4(a.r);
  for (long c = blockIdx.x; c * kPsThreads < total4; c += kPsBlocks) {
    const long j = c * kPsThreads + threadIdx.x;
    if (j >= total4) continue;
    const long off = ps_map(a.r, j);
    f32x4 wv = *reinterpret_cast<const f32x4*>(a.w + off);
    if (a.apply) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(a.grad + off);
      if (ADAM) {
        f32x4 mv = *reinterpret_cast<const f32x4*>(a.m + off), vv = *reinterpret_cast<const f32x4*>(a.v + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // TF AdamOptimizer, the arithmetic of optim.hip adam4
          mv[e] = fmaf(b1, mv[e], (1.f - b1) * g[e]);
          vv[e] = fmaf(b2, vv[e], (1.f - b2) * g[e] * g[e]);
          wv[e] -= lr_t * mv[e] / (sqrtf(vv[e]) + eps);
        }
        *reinterpret_cast<f32x4*>(a.m + off) = mv;
        *reinterpret_cast<f32x4*>(a.v + off) = vv;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) wv[e] = fmaf(-lr, g[e], wv[e]);   // un-averaged, as the reference applies them
      }
      *reinterpret_cast<f32x4*>(a.w + off) = wv;
    }
    if (a.box_bf16 && off < a.n_dense)
      *reinterpret_cast<bf16x4*>(a.box_bf16 + off) = (bf16x4){(bf16)wv[0], (bf16)wv[1], (bf16)wv[2], (bf16)wv[3]};
    else
      *reinterpret_cast<f32x4*>(a.box_f32 + off) = wv;
    if (a.full)
      *reinterpret_cast<f32x4*>(a.full + off) = a.full_src ? *reinterpret_cast<const f32x4*>(a.full_src + off) : wv;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.box_hdr[0] = a.step_lo;
    a.box_hdr[1] = a.step_hi;
    a.box_hdr[2] = a.stop;
    a.box_hdr[3] = a.adam_count;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    if (a.full) __hip_atomic_store(a.full_flags + blockIdx.x, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(a.box_flags + blockIdx.x, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
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
  if (!ps_wait(a.box_flags, a.seq, a.timeout_ticks, a.status)) return;
  if (a.full && !ps_wait(a.full_flags, a.seq, a.timeout_ticks, a.status)) return;
  const long total4 = ps_total4(a.r);
  for (long c = blockIdx.x; c * kPsThreads < total4; c += kPsBlocks) {
    const long j = c * kPsThreads + threadIdx.x;
    if (j >= total4) continue;
    const long off = ps_map(a.r, j);
    if (a.box_bf16 && off < a.n_dense)
      *reinterpret_cast<bf16x4*>(a.shadow + off) = *reinterpret_cast<const bf16x4*>(a.box_bf16 + off);
    else
      *reinterpret_cast<f32x4*>(a.params + off) = *reinterpret_cast<const f32x4*>(a.box_f32 + off);
    if (a.full)
      *reinterpret_cast<f32x4*>((a.full_dst ? a.full_dst : a.params) + off) = *reinterpret_cast<const f32x4*>(a.full + off);
  }
  if (a.owner && blockIdx.x == 0 && threadIdx.x == 0) {
    const uint64_t step = (uint64_t)a.box_hdr[0] | ((uint64_t)a.box_hdr[1] << 32);
    a.state[1] = (int64_t)step;
    *a.version = (int64_t)step;
    a.host_hdr[0] = a.box_hdr[0];
    a.host_hdr[1] = a.box_hdr[1];
    a.host_hdr[2] = a.box_hdr[2];
    a.host_hdr[3] = a.seq;
    a.host_hdr[4] = a.box_hdr[3];
    __threadfence_system();
  }
}

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

int dtf_ps_poll(const void* flags, const void* hdr, int W, int NB, int start, const unsigned* expect, void* out,
                double timeout_s, hipStream_t s) {
  if (W < 1 || W > kPsMaxWorkers || NB < 1 || start < 0) return -2;
  PsPollArgs a;
  a.flags = (const unsigned*)flags;
  a.hdr = (const unsigned*)hdr;
  a.W = W;
  a.NB = NB;
  a.start = start % W;
  for (int w = 0; w < kPsMaxWorkers; ++w) a.expect[w] = w < W ? expect[w] : 0u;
  a.out = (unsigned*)out;
  a.timeout_ticks = (long long)(timeout_s * 1e8);
  hipLaunchKernelGGL(ps_poll, dim3(1), dim3(64), 0, s, a);
  return (int)hipGetLastError();
}

int dtf_ps_apply_push(int adam, void* w, void* m, void* v, const void* hyper, const void* grad, int nr, const long* lo,
                      const long* hi, long n_total, int apply, double adam_t, void* box_f32, void* box_bf16, long n_dense,
                      void* box_flags, void* box_hdr, void* full, void* full_flags, const void* full_src, unsigned adam_count,
                      unsigned seq, unsigned long long step, unsigned stop, hipStream_t s) {
  if (!ranges_ok(nr, lo, hi, n_total) || seq == 0 || (n_dense & 3) || n_dense < 0 || n_dense > n_total) return -2;
  if (adam && (!m || !v)) return -3;
  if (apply && !grad) return -3;
  PsApplyArgs a;
  a.w = (float*)w;
  a.m = (float*)m;
  a.v = (float*)v;
  a.hyper = (const float*)hyper;
  a.grad = (const float*)grad;
  a.r = make_ranges(nr, lo, hi);
  a.apply = apply;
  a.adam_t = (float)adam_t;
  a.box_f32 = (float*)box_f32;
  a.box_bf16 = (bf16*)box_bf16;
  a.n_dense = n_dense;
  a.box_flags = (unsigned*)box_flags;
  a.box_hdr = (unsigned*)box_hdr;
  a.full = (float*)full;
  a.full_flags = (unsigned*)full_flags;
  a.full_src = (const float*)full_src;
  a.adam_count = adam_count;
  a.seq = seq;
  a.step_lo = (unsigned)step;
  a.step_hi = (unsigned)(step >> 32);
  a.stop = stop;
  if (adam) hipLaunchKernelGGL(ps_apply_push<true>, dim3(kPsBlocks), dim3(kPsThreads), 0, s, a);
  else hipLaunchKernelGGL(ps_apply_push<false>, dim3(kPsBlocks), dim3(kPsThreads), 0, s, a);
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

}  // extern "C"
