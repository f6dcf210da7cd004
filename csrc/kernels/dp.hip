// Data-parallel exchange fabric: the two messages of a synchronous step (reference .idea/MNISTDist.py:188 -- every
// worker pulls 13.1 MB of variables and pushes 13.1 MB of gradients through a parameter server per step) as peer-to-peer
// stores over the xGMI mesh, with every per-step quantity (sequence numbers, parities, verdicts) in DEVICE memory, so
// that the whole step is one HIP graph that replays with no host work at all.
//
//   dense (wd1, 12.8 MB):  reduce-scatter by peer stores -> rank-ordered sum of this rank's 1/world shard -> optimizer
//                          on the shard -> all-gather of what the fc1 kernels READ (bf16 compute copy; fp32 weights in
//                          the fp32 configuration), again by peer stores.  One kernel on a branch next to the conv
//                          backward (dp_dense_exchange).
//   tail (0.25 MB):        one-shot exchange (every rank stores its vector into a slot on every peer and sums the slots
//                          in rank order) + update of what it has just summed, in 64 fat workgroups (dp_tail_commit).
//
// Memory: ONE uncached (MTYPE_UC) allocation per rank, exported through hipIpc (DtfDpFabric, api.h).  Peers only
// ever store into it; nothing a local L2 may hold is written by a peer.
//   tail_stage [2][world][n_tail_max] f32      slot (parity, r) written by rank r only
//   tail_flags [2][world][64] u32              flag (parity, r, b) written by workgroup b of rank r only
//   dense_a    [2][world][csz] f32             slot (parity, q): rank q's contribution to THIS rank's shard
//   dense_b    [2][n_dense] bf16 | f32         all-gather landing area: shard q at q * csz
//   flag_a / flag_b [2][world][256] u32
//   tail_done / done_a / done_b [2][..] u32    local barrier words of the launch-wide verdicts (below)
// Two parities make back-to-back steps safe without any extra hand-shake: a rank can start step k+1 while a peer
// still reads step k, but it cannot finish k+1 -- and so cannot start k+2 -- before that peer has raised its k+1 flags,
// which it does after its step k kernels have completed.
//
// Sequence numbers live in device memory, one word PER WORKGROUP (seq_tail[b], seq_dense[b]): a workgroup reads its own
// word when it starts and advances it when it ends, so no workgroup ever reads a word another one writes and no
// launch argument changes from step to step.
//
// All-or-nothing (the reference's contract is should_stop() turning true for everybody, M:173 -- not divergent
// training): a wait that times out, or a peer that says it is poisoned, must not leave a half-applied step.
//   * every wait is bounded; the result of a workgroup's waits goes into a barrier word of its own, and every
//     workgroup reads ALL barrier words of the launch before anything is applied: the verdict is launch-wide;
//   * a bad verdict sets the sticky POISON word (ctl[0]).  Every kernel here checks it when it starts: a poisoned
//     rank changes no parameter, moment or step-state word any more, and raises its flags with the poison bit, so its
//     peers learn about it in the very next exchange instead of after a timeout;
//   * the dense exchange itself writes only derived data (the gradient sum of the shard, the bf16 compute copy).  The
//     master weights and moments of the shard are written by the TAIL kernel of the same step, after the tail's
//     verdict, which also covers "the dense exchange completed on every rank" (each rank's flags carry that bit): one
//     commit point per step.  A rank lost in the middle of a step leaves the survivors at the previous step or at this
//     one, never in between.
#include "api.h"
#include "common.hpp"
#include "optim_core.hpp"
#include "peer_store.hpp"

namespace dtf {

constexpr int kDpTailBlocks = 64;
constexpr int kDpMaxBlocks = 256;          // dense workgroups / commit workgroups
constexpr unsigned kDpPoison = 0x80000000u;
constexpr int kDpDoneWords = kDpTailBlocks + kDpMaxBlocks + 8;   // tail barrier words | commit "started" words | GO word
constexpr int kDpGoWord = kDpTailBlocks + kDpMaxBlocks;

__device__ __forceinline__ unsigned dp_next_seq(unsigned s) { return s + 1 == 0x7fffffffu ? 1u : s + 1; }   // parity keeps alternating

__device__ __forceinline__ unsigned dp_load(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void dp_store(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Bounded wait for *f == seq (poison bit ignored in the comparison).  0: arrived, 1: timed out, 2: arrived poisoned.
// Relaxed polls (an acquire per poll would invalidate caches in a loop that runs next to the conv kernels); the caller
// issues ONE acquire fence when all of its waits are over.
__device__ __forceinline__ int dp_wait(const unsigned* f, unsigned seq, long long timeout_ticks, int nap = 2) {
  const long long t0 = wall_clock64();
  for (;;) {
    const unsigned v = dp_load(f);
    if ((v & ~kDpPoison) == seq) return (v & kDpPoison) ? 2 : 0;
    if (wall_clock64() - t0 > timeout_ticks) return 1;
    if (nap > 2) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(2);
  }
}

// 16-byte poll of uncached protocol words, past every cache (sc0 sc1)
__device__ __forceinline__ u32x4 dp_load4(const unsigned* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// Bounded wait, by the FIRST WAVE of the workgroup, until words[0 .. n) all carry seq (n <= 256; words 16-byte
// aligned): lane l polls words 4l .. 4l+3 with one 16-byte load, so a whole barrier array is ONE wave-instruction per
// round.  (Every thread polling its own word -- n workgroups x n pollers on the same few hundred bytes of uncached
// memory -- made the barrier cost grow with the workgroup count: 41 / 67 / 121 us for the dense kernel with 64 / 128 /
// 256 workgroups, measured.)  Returns, in every lane of the first wave: 0 all arrived clean, |1 timed out, |2 poison.
__device__ __forceinline__ int dp_wait_all(const unsigned* words, int n, unsigned seq, long long timeout_ticks) {
  const int lane = threadIdx.x & 63, w0 = 4 * lane;
  const long long t0 = wall_clock64();
  int res = 0;
  for (;;) {
    bool here = true, poison = false;
    if (w0 < n) {
      const u32x4 v = dp_load4(words + w0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (w0 + e < n) {
          here = here && ((v[e] & ~kDpPoison) == seq);
          poison = poison || ((v[e] & ~kDpPoison) == seq && (v[e] & kDpPoison));
        }
    }
    if (__ballot(poison)) res |= 2;
    if (__ballot(!here) == 0) break;
    if (wall_clock64() - t0 > timeout_ticks) {
      res |= 1;
      break;
    }
    __builtin_amdgcn_s_sleep(4);
  }
  return res;
}

// Launch-wide verdict: workgroup b publishes its own result in done[b] and reads everybody's.  Every workgroup of the
// launch gets the same answer (0 good, else the OR of the reasons) unless a workgroup never shows up, which is a
// timeout for all of those that wait for it.  The barrier words live in uncached memory: no cache maintenance.
__device__ __forceinline__ int dp_verdict(unsigned* done, int nblk, int b, unsigned seq, int bad, long long timeout_ticks,
                                          int* sh) {
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) dp_store(done + b, seq | (bad ? kDpPoison : 0u));
    const int r = dp_wait_all(done, nblk, seq, timeout_ticks);
    if (threadIdx.x == 0) *sh = r;
  }
  __syncthreads();
  return *sh;
}

// dst[i] = src[i] for i in [lo + tid, hi) with stride nthr: four loads in flight per thread, then four stores
template <typename T>
__device__ __forceinline__ void dp_copy(T* dst, const T* src, long lo, long hi, int tid, int nthr) {
  for (long i0 = lo + tid; i0 < hi; i0 += 4L * nthr) {
    T v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[min(i0 + (long)u * nthr, hi - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + (long)u * nthr < hi) dst[i0 + (long)u * nthr] = v[u];
  }
}

// the same towards a peer's buffer (write-through stores)
__device__ __forceinline__ void dp_push16(f32x4* dst, const f32x4* src, long lo, long hi, int tid, int nthr) {
  for (long i0 = lo + tid; i0 < hi; i0 += 4L * nthr) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[min(i0 + (long)u * nthr, hi - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + (long)u * nthr < hi) dp_put16(dst + i0 + (long)u * nthr, v[u]);
  }
}

__device__ __forceinline__ char* dp_area(const DtfDpFabric& f, int r, long off) { return static_cast<char*>(f.base[r]) + off; }

template <typename AGT> struct AgStore;
template <> struct AgStore<bf16> {
  static __device__ __forceinline__ void put(bf16* p, const f32x4& w) { store_bf16x4(p, w[0], w[1], w[2], w[3]); }
  static __device__ __forceinline__ void put_peer(bf16* p, const f32x4& w) {
    const bf16x4 h = {(bf16)w[0], (bf16)w[1], (bf16)w[2], (bf16)w[3]};
    dp_put8(p, __builtin_bit_cast(u32x2, h));
  }
};
template <> struct AgStore<float> {
  static __device__ __forceinline__ void put(float* p, const f32x4& w) { *reinterpret_cast<f32x4*>(p) = w; }
  static __device__ __forceinline__ void put_peer(float* p, const f32x4& w) { dp_put16(p, w); }
};

// ---------------------------------------------------------------------------------------------
// Dense exchange.  Workgroup b owns slice b of EVERY shard:
//   A  store my gradient of shard p, slice b  -> dense_a(parity, me) on rank p          flag_a(parity, me, b) on p
//      wait flag_a(parity, q, b) for all q != me ; launch-wide verdict
//   1  sum the slots in rank order (my own contribution comes straight from the gradient arena) -> gsum;
//      optimizer arithmetic on the shard IN REGISTERS (the master copy is written at the commit, dp_tail_commit);
//      the new weights go out as bf16 / fp32: my own compute copy directly, dense_b(parity)[shard me] on every peer
//   B  flag_b(parity, me, b) on every peer ; wait flag_b(parity, q, b) ; install the foreign shards of the bf16
//      compute copy from my dense_b (fp32 configuration: the commit does that, the compute copy IS the master there);
//      launch-wide verdict -> ctl[2] = seq ("the dense exchange of this step completed here")
// ---------------------------------------------------------------------------------------------
template <bool ADAM, typename AGT>
__global__ __launch_bounds__(512) void dp_dense_exchange(DtfDpFabric f, const float* __restrict__ grads,
                                                         const float* __restrict__ w, const float* __restrict__ m,
                                                         const float* __restrict__ v, AGT* __restrict__ compute_copy,
                                                         const float* __restrict__ hyper, const int64_t* adam_t) {
  __shared__ int sh_bad, sh_verdict;
  const int tid = threadIdx.x, b = blockIdx.x, W = f.world, me = f.rank, nblk = gridDim.x;
  const unsigned seq = f.seq_dense[b];
  const int par = seq & 1u;
  const long csz = f.csz, nd = f.n_dense;
  const long c4 = csz / 4, per = (c4 + nblk - 1) / nblk;
  const long lo4 = b * per, hi4 = min(c4, lo4 + per);
  const bool poisoned = f.ctl[0] != 0;
  const bool mute = (unsigned)f.ctl[3] == seq;          // test hook: this step raises no flags
  auto dense_a = [&](int r) { return reinterpret_cast<float*>(dp_area(f, r, f.off_dense_a)); };
  auto dense_b = [&](int r) { return reinterpret_cast<AGT*>(dp_area(f, r, f.off_dense_b)); };
  auto flag_a = [&](int r) { return reinterpret_cast<unsigned*>(dp_area(f, r, f.off_flag_a)); };
  auto flag_b = [&](int r) { return reinterpret_cast<unsigned*>(dp_area(f, r, f.off_flag_b)); };
  unsigned* done_a = reinterpret_cast<unsigned*>(dp_area(f, me, f.off_done_a)) + (size_t)par * kDpMaxBlocks;
  unsigned* done_b = reinterpret_cast<unsigned*>(dp_area(f, me, f.off_done_b)) + (size_t)par * kDpMaxBlocks;
  const size_t fidx = ((size_t)par * W + me) * kDpMaxBlocks + b;      // my flag word on a peer

  if (tid == 0) sh_bad = poisoned ? 2 : 0;
  // ---- A: my contributions to the other ranks' shards ----
  if (!poisoned) {
#pragma unroll 1
    for (int k = 1; k < W; ++k) {
      // the workgroups start with different peers, so that all W - 1 links carry traffic at once (xGMI is a mesh of
      // point-to-point links: 64 workgroups walking the peers in the same order would load them one after the other)
      const int p = (me + 1 + (k - 1 + b) % (W - 1)) % W;
      const f32x4* src = reinterpret_cast<const f32x4*>(grads + (size_t)p * csz);
      f32x4* dst = reinterpret_cast<f32x4*>(dense_a(p) + ((size_t)par * W + me) * csz);
      dp_push16(dst, src, lo4, hi4, tid, 512);
    }
  }
  dp_puts_done();
  __syncthreads();
  if (tid < W && tid != me) {
    if (!mute) dp_raise(flag_a(tid) + fidx, seq | (poisoned ? kDpPoison : 0u));
    const int r = dp_wait(flag_a(me) + ((size_t)par * W + tid) * kDpMaxBlocks + b, seq, f.timeout_ticks);
    if (r) atomicOr(&sh_bad, r);
  }
  __syncthreads();
  int bad = dp_verdict(done_a, nblk, b, seq, sh_bad, f.timeout_ticks, &sh_verdict);
  if (bad) {
    // nobody applies anything; the peers hear about it through flag_b instead of waiting for their own timeouts
    if (tid < W && tid != me && !mute) dp_raise(flag_b(tid) + fidx, seq | kDpPoison);
    if (b == 0 && tid == 0) {
      f.ctl[0] = 1;
      f.ctl[1] |= bad;
    }
    if (tid == 0) f.seq_dense[b] = dp_next_seq(seq);
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");     // the slots the flags announced

  // ---- 1: rank-ordered sum, optimizer arithmetic, all-gather stores ----
  const float lr = hyper[0] * hyper[4];
  float lr_t = 0.f, b1 = 0.f, b2 = 0.f, eps = 0.f, gsc = 0.f;
  if (ADAM) {
    b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; gsc = hyper[4];
    const float t = (float)(*adam_t + 1);            // the counter is advanced at the commit
    lr_t = hyper[0] * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
  }
  const float* slots = dense_a(me) + (size_t)par * W * csz;
  const f32x4* own_g = reinterpret_cast<const f32x4*>(grads + (size_t)me * csz);
  const size_t shard0 = (size_t)me * csz;
  // four float4 per thread and array in flight (a plain load -> arithmetic -> store loop was a chain of memory round
  // trips); clamped addresses keep the loads unconditional.
  // TWO passes over the slice when there are peers: the first one only stores into the PEERS' buffers (uncached, never
  // in this L2), the second one -- after the fence and the flags -- recomputes the same values and stores the local,
  // cached results (gradient sum, own compute copy).  A system-scope release fence writes back every dirty line of the
  // L2: with 19 MB of freshly written local results in it, the 64 fences of this launch serialised into ~0.4 us each
  // (the kernel's time grew linearly with its workgroup count: 41 / 67 / 121 us for 64 / 128 / 256); fencing while
  // the L2 is still clean is what the tail kernel does, at 5 us for the whole launch.
  constexpr int kU = 4;
  auto pass = [&](const bool to_peers, const bool to_local) {
    for (long i0 = lo4 + tid; i0 < hi4; i0 += 512 * kU) {
      f32x4 sum[kU], wv[kU], mv[kU], vv[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const long i = min(i0 + u * 512L, hi4 - 1);
        sum[u] = (me == 0) ? own_g[i] : reinterpret_cast<const f32x4*>(slots)[i];
        for (int q = 1; q < W; ++q)
          sum[u] += (q == me) ? own_g[i] : reinterpret_cast<const f32x4*>(slots + (size_t)q * csz)[i];
        wv[u] = reinterpret_cast<const f32x4*>(w + shard0)[i];
        if (ADAM) {
          mv[u] = reinterpret_cast<const f32x4*>(m + shard0)[i];
          vv[u] = reinterpret_cast<const f32x4*>(v + shard0)[i];
        }
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const long i = i0 + u * 512L;
        if (i < hi4) {
          if (to_local) reinterpret_cast<f32x4*>(f.gsum)[i] = sum[u];
          if (ADAM) {
            adam4(wv[u], sum[u], mv[u], vv[u], lr_t, b1, b2, eps, gsc);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) wv[u][e] = fmaf(-lr, sum[u][e], wv[u][e]);
          }
          if (to_local && sizeof(AGT) == 2) AgStore<AGT>::put(compute_copy + shard0 + 4 * i, wv[u]);   // derived data: not gated
          if (to_peers) {
#pragma unroll 1
            for (int k = 1; k < W; ++k) {
              const int p = (me + k) % W;
              AgStore<AGT>::put_peer(dense_b(p) + (size_t)par * nd + shard0 + 4 * i, wv[u]);
            }
          }
        }
      }
    }
  };
  if (W == 1) {
    // nobody to exchange with: one pass, no fence, no flags (a one-rank job runs this only when it is asked to keep
    // the N > 1 kernels in the step, bench.py --no_collapse)
    pass(false, true);
    __syncthreads();
    bad = dp_verdict(done_b, nblk, b, seq, 0, f.timeout_ticks, &sh_verdict);
    if (b == 0 && tid == 0) {
      if (bad) {
        f.ctl[0] = 1;
        f.ctl[1] |= bad;
      } else {
        f.ctl[2] = (int)seq;
      }
    }
    if (tid == 0) f.seq_dense[b] = dp_next_seq(seq);
    return;
  }
  pass(true, false);
  dp_puts_done();
  __syncthreads();
  if (tid == 0) sh_bad = 0;
  __syncthreads();
  if (tid < W && tid != me && !mute) dp_raise(flag_b(tid) + fidx, seq);
  pass(false, true);          // the local results, while the peers' shards are on their way
  if (tid < W && tid != me) {
    const int r = dp_wait(flag_b(me) + ((size_t)par * W + tid) * kDpMaxBlocks + b, seq, f.timeout_ticks);
    if (r) atomicOr(&sh_bad, r);
  }
  __syncthreads();
  if (!sh_bad && sizeof(AGT) == 2) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    // install the foreign shards of the compute copy
    const AGT* mine = dense_b(me) + (size_t)par * nd;
#pragma unroll 1
    for (int k = 1; k < W; ++k) {
      const int q = (me + k) % W;
      const bf16x4* src = reinterpret_cast<const bf16x4*>(mine + (size_t)q * csz);
      bf16x4* dst = reinterpret_cast<bf16x4*>(compute_copy + (size_t)q * csz);
      dp_copy(dst, src, lo4, hi4, tid, 512);
    }
  }
  bad = dp_verdict(done_b, nblk, b, seq, sh_bad, f.timeout_ticks, &sh_verdict);
  if (b == 0 && tid == 0) {
    if (bad) {
      f.ctl[0] = 1;
      f.ctl[1] |= bad;
    } else {
      f.ctl[2] = (int)seq;
    }
  }
  if (tid == 0) f.seq_dense[b] = dp_next_seq(seq);
}

// ---------------------------------------------------------------------------------------------
// Tail exchange + commit.  Workgroups [0, 64): one-shot exchange of slice b of the (already slab-reduced) tail of the
// gradient arena, launch-wide verdict, then the update of exactly those elements (master, moments, bf16 shadow,
// out_wT, conv2 packs) and -- workgroup 0 -- the step state.  Workgroups [64, 64 + commit_blocks) (with_dense): wait
// for the same verdict and apply the optimizer to this rank's shard of the dense master weights from f.gsum; in the
// fp32 configuration they also install the foreign shards from dense_b (there the master IS the compute copy).
// The flags of a rank carry "poisoned, or my dense exchange of this step did not complete", so the verdict covers the
// whole step on every rank.
// Adam's step counter: every workgroup reads it when it starts (t = counter + 1) and workgroup 0 advances it at the
// very end -- after the barrier words show that every other workgroup of the launch has started.
// ---------------------------------------------------------------------------------------------
template <bool ADAM, typename AGT>
__global__ __launch_bounds__(256) void dp_tail_commit(FusedSgdArgs a, DtfDpFabric f, long tail_off, long dense_off,
                                                      int with_dense, int adam_slot, int adam_slot2) {
  __shared__ int sh_bad, sh_verdict;
  const int tid = threadIdx.x, b = blockIdx.x, W = f.world, me = f.rank;
  const int nall = gridDim.x;
  const unsigned seq = f.seq_tail[b];
  const int par = seq & 1u;
  const float lr = a.hyper[0] * a.hyper[4];
  float lr_t = 0.f, b1 = 0.f, b2 = 0.f, eps = 0.f, gsc = 0.f;
  if (ADAM) {
    b1 = a.hyper[1]; b2 = a.hyper[2]; eps = a.hyper[3]; gsc = a.hyper[4];
    const float t = (float)(*a.adam_t + 1);
    lr_t = a.hyper[0] * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
  }
  unsigned* done = reinterpret_cast<unsigned*>(dp_area(f, me, f.off_tail_done)) + (size_t)par * kDpDoneWords;

  if (b >= kDpTailBlocks) {
    // ---- dense commit workgroup ----
    // Everything it reads (the shard's gradient sum, master weights, moments) is final before this kernel starts, so
    // up to kPre float4 per thread are fetched BEFORE the wait for the verdict: the loads hide under the tail
    // exchange, and what follows the verdict is arithmetic and stores.
    constexpr int kPre = 6;
    const int c = b - kDpTailBlocks, nC = nall - kDpTailBlocks;
    const long c4 = f.csz / 4, stride = (long)nC * 256;
    const size_t shard0 = (size_t)dense_off + (size_t)me * f.csz;
    // with_dense == 2 (a one-rank job, any optimizer): no dense exchange ran -- the "sum" is this rank's own gradient,
    // still in the arena; these workgroups then also write the compute copy, and there is no verdict to wait for
    const bool local = with_dense == 2;
    const f32x4* gp = local ? reinterpret_cast<const f32x4*>(a.g + shard0) : reinterpret_cast<const f32x4*>(f.gsum);
    bf16* cp = (local && a.shadow) ? a.shadow + shard0 : nullptr;
    f32x4* wp = reinterpret_cast<f32x4*>(a.w + shard0);
    f32x4* mp = ADAM ? reinterpret_cast<f32x4*>(a.m + shard0) : nullptr;
    f32x4* vp = ADAM ? reinterpret_cast<f32x4*>(a.v + shard0) : nullptr;
    const long i0 = (long)c * 256 + tid;
    f32x4 gq[kPre], wq[kPre], mq[kPre], vq[kPre];
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const long i = min(i0 + u * stride, c4 - 1);     // clamped: the loads are unconditional
      gq[u] = gp[i];
      wq[u] = wp[i];
      if (ADAM) {
        mq[u] = mp[i];
        vq[u] = vp[i];
      }
    }
    if (tid == 0) {
      sh_verdict = 0;
      dp_store(done + b, seq);                      // "started: Adam's counter has been read"
    }
    // ONE thread per commit workgroup polls ONE word (tail workgroup 0 publishes the launch-wide verdict there): 256
    // workgroups polling all 64 barrier words made 16 K spinning threads on one memory channel, which slowed the
    // exchange they were waiting for (64 us for this kernel, measured)
    if (tid == 0 && !local) {
      __builtin_amdgcn_s_sleep(96);                // (~3 us) the verdict cannot be there before the exchange's round trips
      sh_verdict = dp_wait(done + kDpGoWord, seq, f.timeout_ticks, 16);
    }
    if (tid == 0 && local) sh_verdict = f.ctl[0] != 0 ? 2 : 0;
    __syncthreads();
    if (!sh_verdict) {
      auto apply = [&](long i, f32x4 gs, f32x4 wv, f32x4 mv, f32x4 vv) {
        if (ADAM) {
          adam4(wv, gs, mv, vv, lr_t, b1, b2, eps, gsc);
          mp[i] = mv;
          vp[i] = vv;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) wv[e] = fmaf(-lr, gs[e], wv[e]);
        }
        wp[i] = wv;
        if (cp) store_bf16x4(cp + 4 * i, wv[0], wv[1], wv[2], wv[3]);
      };
#pragma unroll
      for (int u = 0; u < kPre; ++u)
        if (i0 + u * stride < c4) apply(i0 + u * stride, gq[u], wq[u], mq[u], vq[u]);
      for (long i = i0 + kPre * stride; i < c4; i += stride) {
        f32x4 mv = {0.f, 0.f, 0.f, 0.f}, vv = mv;
        if (ADAM) {
          mv = mp[i];
          vv = vp[i];
        }
        apply(i, gp[i], wp[i], mv, vv);
      }
      if (sizeof(AGT) == 4 && W > 1) {
        const float* mine = reinterpret_cast<const float*>(dp_area(f, me, f.off_dense_b)) + (size_t)par * f.n_dense;
#pragma unroll 1
        for (int k = 1; k < W; ++k) {
          const int q = (me + k) % W;
          const f32x4* src = reinterpret_cast<const f32x4*>(mine + (size_t)q * f.csz);
          f32x4* dst = reinterpret_cast<f32x4*>(a.w + dense_off + (size_t)q * f.csz);
          dp_copy(dst, src, (long)c * 256, c4, tid, (int)stride);
        }
      }
    }
    if (tid == 0) f.seq_tail[b] = dp_next_seq(seq);
    return;
  }

  // ---- tail exchange workgroup ----
  const int n4 = (int)(f.n_tail / 4);
  const int per = (n4 + kDpTailBlocks - 1) / kDpTailBlocks;
  const int lo4 = b * per, hi4 = min(n4, lo4 + per);
  constexpr int kOwn = 2;                                        // float4 per thread of this workgroup's slice
  // this rank's own contribution and the master weights it will update: requested first, consumed after the verdict
  f32x4 own[kOwn], wold[kOwn];
#pragma unroll
  for (int u = 0; u < kOwn; ++u) {
    const int i = min(lo4 + tid + u * 256, n4 - 1);
    own[u] = reinterpret_cast<const f32x4*>(a.g + tail_off)[i];
    wold[u] = reinterpret_cast<const f32x4*>(a.w + tail_off)[i];
  }
  auto commit_slice = [&](const float* slots) {
#pragma unroll
    for (int u = 0; u < kOwn; ++u) {
      const int i = lo4 + tid + u * 256;
      if (i < hi4) {
        f32x4 gs = (me == 0 || !slots) ? own[u] : reinterpret_cast<const f32x4*>(slots)[i];
        for (int q = 1; q < W; ++q)
          gs += (q == me) ? own[u] : reinterpret_cast<const f32x4*>(slots + (size_t)q * f.n_tail_max)[i];
        tail_update4<ADAM>(a, tail_off + 4L * i, gs, wold[u], lr, lr_t, b1, b2, eps, gsc);
      }
    }
    // (a slice longer than kOwn * 256 float4 -- the widened model -- goes on from memory: the arena still holds this
    //  rank's own values until this very thread overwrites them)
    for (int i = lo4 + tid + kOwn * 256; i < hi4; i += 256) {
      const f32x4 mine_g = reinterpret_cast<const f32x4*>(a.g + tail_off)[i];
      f32x4 gs = (me == 0 || !slots) ? mine_g : reinterpret_cast<const f32x4*>(slots)[i];
      for (int q = 1; q < W; ++q)
        gs += (q == me) ? mine_g : reinterpret_cast<const f32x4*>(slots + (size_t)q * f.n_tail_max)[i];
      tail_update4<ADAM>(a, tail_off + 4L * i, gs, reinterpret_cast<const f32x4*>(a.w + tail_off)[i], lr, lr_t, b1, b2, eps, gsc);
    }
  };
  auto finish_step = [&]() {
    if (b == 0) {
      // every commit workgroup has started (and read Adam's counter) before the step state moves (one wave polls
      // their words 16 bytes per lane; one thread polling them in turn was 256 dependent uncached loads: 50 us)
      if (tid < 64) {
        const int r = nall > kDpTailBlocks ? dp_wait_all(done + kDpTailBlocks, nall - kDpTailBlocks, seq, f.timeout_ticks) : 0;
        if (tid == 0) sh_verdict = r;
      }
      __syncthreads();
      if (tid == 0) {
        if (sh_verdict) {         // cannot happen unless the launch itself is broken; never advance on a doubt
          f.ctl[0] = 1;
          f.ctl[1] |= 4;
        } else {
          advance_state(a.adv);
          if (ADAM) {
            a.adv.state[adam_slot] += 1;
            if (adam_slot2 >= 3 && adam_slot2 != adam_slot) a.adv.state[adam_slot2] += 1;
          }
        }
      }
    }
    if (tid == 0) f.seq_tail[b] = dp_next_seq(seq);
  };
  const bool poisoned = f.ctl[0] != 0;
  if (W == 1 && with_dense != 1) {
    // a one-rank job has no peer to hear from and no wait that could time out: the sum is this rank's own vector
    if (!poisoned) {
      commit_slice(nullptr);
      finish_step();
    } else if (tid == 0) {
      f.seq_tail[b] = dp_next_seq(seq);
    }
    return;
  }
  const bool dense_ok = with_dense != 1 || (unsigned)f.ctl[2] == seq;
  const bool mine_ok = !poisoned && dense_ok;
  const bool mute = (unsigned)f.ctl[3] == seq;
  auto stage = [&](int r) { return reinterpret_cast<float*>(dp_area(f, r, f.off_tail_stage)); };
  auto flags = [&](int r) { return reinterpret_cast<unsigned*>(dp_area(f, r, f.off_tail_flags)); };
  const size_t slot = ((size_t)par * W + me) * f.n_tail_max;
  if (tid == 0) sh_bad = mine_ok ? 0 : 2;
  if (mine_ok) {
#pragma unroll
    for (int u = 0; u < kOwn; ++u) {
      const int i = lo4 + tid + u * 256;
      if (i < hi4) {
#pragma unroll 1
        for (int k = 1; k < W; ++k) dp_put16(reinterpret_cast<f32x4*>(stage((me + k) % W) + slot) + i, own[u]);
      }
    }
    for (int i = lo4 + tid + kOwn * 256; i < hi4; i += 256) {
      const f32x4 val = reinterpret_cast<const f32x4*>(a.g + tail_off)[i];
#pragma unroll 1
      for (int k = 1; k < W; ++k) dp_put16(reinterpret_cast<f32x4*>(stage((me + k) % W) + slot) + i, val);
    }
  }
  dp_puts_done();
  __syncthreads();
  if (tid < W && tid != me) {
    if (!mute) dp_raise(flags(tid) + ((size_t)par * W + me) * kDpTailBlocks + b, seq | (mine_ok ? 0u : kDpPoison));
    const int r = dp_wait(flags(me) + ((size_t)par * W + tid) * kDpTailBlocks + b, seq, f.timeout_ticks);
    if (r) atomicOr(&sh_bad, r);
  }
  __syncthreads();
  const int bad = dp_verdict(done, kDpTailBlocks, b, seq, sh_bad, f.timeout_ticks, &sh_verdict);
  if (b == 0 && tid == 0 && with_dense == 1) dp_store(done + kDpGoWord, seq | (bad ? kDpPoison : 0u));   // the commit workgroups' cue
  if (bad) {
    if (b == 0 && tid == 0) {
      f.ctl[0] = 1;
      f.ctl[1] |= bad;
    }
    if (tid == 0) f.seq_tail[b] = dp_next_seq(seq);
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  commit_slice(stage(me) + (size_t)par * W * f.n_tail_max);
  finish_step();
}

}  // namespace dtf

using namespace dtf;

extern "C" {

static long align_up(long x, long a) { return (x + a - 1) / a * a; }

long dtf_dp_fabric_layout(DtfDpFabric* f) {
  if (f->world < 1 || f->world > 8 || (f->n_tail_max & 3) || f->n_dense % (8L * f->world) || (f->ag_bytes != 2 && f->ag_bytes != 4))
    return -1;
  f->csz = f->n_dense / f->world;
  const long W = f->world;
  long o = 0;
  f->off_tail_stage = o; o = align_up(o + 2 * W * f->n_tail_max * 4, 256);
  f->off_tail_flags = o; o = align_up(o + 2 * W * kDpTailBlocks * 4, 256);
  f->off_tail_done = o;  o = align_up(o + 2L * kDpDoneWords * 4, 256);
  f->off_dense_a = o;    o = align_up(o + (W > 1 ? 2 * W * f->csz * 4 : 16), 256);
  f->off_dense_b = o;    o = align_up(o + (W > 1 ? 2 * f->n_dense * (long)f->ag_bytes : 16), 256);
  f->off_flag_a = o;     o = align_up(o + 2 * W * kDpMaxBlocks * 4, 256);
  f->off_flag_b = o;     o = align_up(o + 2 * W * kDpMaxBlocks * 4, 256);
  f->off_done_a = o;     o = align_up(o + 2L * kDpMaxBlocks * 4, 256);
  f->off_done_b = o;     o = align_up(o + 2L * kDpMaxBlocks * 4, 256);
  return o;
}

static int fabric_ok(const DtfDpFabric* f) {
  if (!f || f->world < 1 || f->world > 8 || f->rank < 0 || f->rank >= f->world) return 0;
  if (f->dense_blocks < 1 || f->dense_blocks > kDpMaxBlocks || f->commit_blocks < 1 || f->commit_blocks > kDpMaxBlocks) return 0;
  if ((f->n_tail & 3) || f->n_tail > f->n_tail_max || f->csz * f->world > f->n_dense || (f->csz & 7)) return 0;   // (< : timing diagnostic, dp_fabric.py fake_world)
  if (!f->seq_tail || !f->seq_dense || !f->ctl || !f->gsum || f->timeout_ticks <= 0) return 0;
  for (int r = 0; r < f->world; ++r)
    if (!f->base[r]) return 0;
  return 1;
}

int dtf_dp_dense_exchange(const DtfDpFabric* f, int adam, const void* grads_dense, const void* w_dense,
                          const void* m_dense, const void* v_dense, void* compute_copy, const void* hyper,
                          const void* adam_t, hipStream_t s) {
  if (!fabric_ok(f)) return -2;
  if (adam && (!m_dense || !v_dense || !adam_t)) return -3;
  if (f->ag_bytes == 2 && !compute_copy) return -3;
  const dim3 grid(f->dense_blocks), block(512);
#define DP_DENSE(ADAM_, T_)                                                                                          \
  hipLaunchKernelGGL((dp_dense_exchange<ADAM_, T_>), grid, block, 0, s, *f, (const float*)grads_dense,              \
                     (const float*)w_dense, (const float*)m_dense, (const float*)v_dense, (T_*)compute_copy,        \
                     (const float*)hyper, (const int64_t*)adam_t)
  if (f->ag_bytes == 2) {
    if (adam) DP_DENSE(true, bf16); else DP_DENSE(false, bf16);
  } else {
    if (adam) DP_DENSE(true, float); else DP_DENSE(false, float);
  }
#undef DP_DENSE
  return (int)hipGetLastError();
}

int dtf_dp_tail_commit(const DtfDpFabric* f, int adam, int with_dense, void* w, void* g, void* m, void* v, void* shadow,
                       const void* hyper, void* wc2T, long wc2_off, int c1, int c2, void* out_wT, long out_w_off, int H,
                       void* state, int adam_slot, int adam_slot2, long cursor_inc, long step_inc, long drop_inc,
                       long tail_off, long dense_off, hipStream_t s) {
  if (!fabric_ok(f) || (tail_off & 3) || (wc2_off & 3) || (dense_off & 3) || !state) return -2;
  if (adam && (!m || !v || adam_slot < 3 || adam_slot > 6 || adam_slot2 > 6)) return -3;
  if (wc2T && ((c1 % 32) || (c2 % 32))) return -3;
  FusedSgdArgs a;
  for (int k = 0; k < 3; ++k) {
    a.s[k] = FusedSeg{nullptr, 0, 0, 0, 0};
    a.blocks_end[k] = 0;
  }
  a.count = 0;
  a.reduce_blocks = 0;
  a.w = (float*)w;
  a.g = (float*)g;
  a.shadow = (bf16*)shadow;
  a.hyper = (const float*)hyper;
  a.dense_lo4 = a.dense_hi4 = 0;
  a.out_wT = (float*)out_wT;
  a.out_w_off = out_w_off;
  a.H = H;
  a.wc2T = (bf16*)wc2T;
  a.wc2_off = wc2_off;
  a.c1 = c1;
  a.c2 = c2;
  a.adv = StepAdvance{(int64_t*)state, cursor_inc, step_inc, drop_inc, -1};
  a.m = (float*)m;
  a.v = (float*)v;
  a.adam_t = adam ? (const int64_t*)state + adam_slot : nullptr;
  const dim3 grid(kDpTailBlocks + (with_dense ? f->commit_blocks : 0)), block(256);
#define DP_TAIL(ADAM_, T_)                                                                                           \
  hipLaunchKernelGGL((dp_tail_commit<ADAM_, T_>), grid, block, 0, s, a, *f, tail_off, dense_off, with_dense, adam_slot, \
                     adam_slot2)
  if (f->ag_bytes == 2) {
    if (adam) DP_TAIL(true, bf16); else DP_TAIL(false, bf16);
  } else {
    if (adam) DP_TAIL(true, float); else DP_TAIL(false, float);
  }
#undef DP_TAIL
  return (int)hipGetLastError();
}

}  // extern "C"
