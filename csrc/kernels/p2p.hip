// One-shot all-reduce over peer-to-peer xGMI stores, for the small latency-bound "tail" message of the
// data-parallel step (0.25 MB: head + conv gradients).
//
// Why not RCCL for this one: the tail collective sits at the very end of backward and the next forward
// needs its result, so its latency is fully exposed; a ring/tree collective pays several hops of
// launch + protocol latency for 0.25 MB.  xGMI is a full point-to-point mesh, so every rank can simply
// PUSH its vector into a staging slot on every peer (7 links driven in parallel, 0.25 MB per link),
// raise a flag there, wait for the 7 flags raised in its own memory and sum the 8 slots locally in rank
// order -- one hop, deterministic, identical bits on every rank.  (The 12.8 MB dense message stays on RCCL:
// it is bandwidth-bound and hidden under backward.)
//
// Memory: every rank owns ONE uncached (MTYPE_UC) allocation, exported through hipIpc:
//   stage  [2 parities][world][n_max] float     slot (par, r) is written by rank r only
//   flags  [2 parities][world][kBlocks] u32     flag (par, r, b) is written by block b of rank r only
// Block b of a step moves / reduces elements [b*chunk, (b+1)*chunk) and only ever depends on block b of
// the peers, so there is no grid-wide synchronisation and no dependency between blocks of one rank.
// Two parities suffice: a rank can start step k+1 (writing parity (k+1)&1 on its peers) while a peer still
// reads step k, but it cannot finish k+1 -- and so cannot start k+2 -- before that peer has raised its k+1
// flags, which it does from its step k+1 launch, i.e. after its step k launch has completed.
// Every wait is bounded (wall clock): on a timeout the kernel records it in *status and returns, so a lost
// peer ends in an error on the host instead of a hung GPU.
#include "common.hpp"
#include <string.h>

namespace dtf {

constexpr int kP2PMaxWorld = 8;
constexpr int kP2PBlocks = 64;
constexpr int kP2PThreads = 256;

struct P2PArgs {
  float* data;                      // local vector, reduced in place
  float* stage[kP2PMaxWorld];       // stage base of every rank, as mapped into THIS process
  unsigned* flags[kP2PMaxWorld];    // flag base of every rank
  int* status;                      // local: set to 1 by a wait that timed out
  int world, rank, n, n_max;
  unsigned seq;                     // 1, 2, 3, ... (the same on every rank)
  long long timeout_ticks;          // wall_clock64 ticks (100 MHz)
};

__global__ __launch_bounds__(kP2PThreads) void p2p_allreduce_oneshot(P2PArgs a) {
  const int tid = threadIdx.x, b = blockIdx.x;
  const int par = a.seq & 1u;
  const int chunk4 = ((a.n + 3) / 4 + kP2PBlocks - 1) / kP2PBlocks;   // float4 per block
  const int lo4 = b * chunk4, hi4 = min((a.n + 3) / 4, lo4 + chunk4); // n_max is a multiple of 4: the tail quad fits
  const size_t slot = ((size_t)par * a.world + a.rank) * a.n_max;

  // ---- push this rank's slice into its slot on every rank (own memory included) ----
  const f32x4* src = reinterpret_cast<const f32x4*>(a.data);
  for (int i = lo4 + tid; i < hi4; i += kP2PThreads) {
    const f32x4 v = src[i];
#pragma unroll 1
    for (int k = 0; k < a.world; ++k) {
      const int p = (a.rank + k) % a.world;                 // start with oneself, then fan out over the links
      reinterpret_cast<f32x4*>(a.stage[p] + slot)[i] = v;
    }
  }
  __threadfence_system();    // the slice is visible system-wide ...
  __syncthreads();           // ... for every thread of the block, before any flag goes up
  if (tid < a.world)
    __hip_atomic_store(a.flags[tid] + ((size_t)par * a.world + a.rank) * kP2PBlocks + b, a.seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);

  // ---- wait for block b of every rank ----
  if (tid < a.world) {
    const unsigned* f = a.flags[a.rank] + ((size_t)par * a.world + tid) * kP2PBlocks + b;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
      if (wall_clock64() - t0 > a.timeout_ticks) {
        *a.status = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();

  // ---- sum the slots in rank order (same order, same bits on every rank) ----
  const float* mine = a.stage[a.rank] + (size_t)par * a.world * a.n_max;
  f32x4* dst = reinterpret_cast<f32x4*>(a.data);
  for (int i = lo4 + tid; i < hi4; i += kP2PThreads) {
    f32x4 s = reinterpret_cast<const f32x4*>(mine)[i];
    for (int q = 1; q < a.world; ++q) s += reinterpret_cast<const f32x4*>(mine + (size_t)q * a.n_max)[i];
    dst[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// Two-shot variant for the large (12.8 MB) dense message: reduce-scatter + all-gather, both as peer pushes.
// Rank r owns chunk r (n / world elements).  Block b works on slice b of EVERY chunk:
//   A  push   my data[chunk p][slice b]        -> stageA(par, me) on rank p          flagA(par, me, b) on p
//      wait   flagA(par, q, b) for all q ; sum stageA(par, q)[slice b] over q in rank order
//   B  push   that sum                         -> stageB(par)[chunk me][slice b] on every rank   flagB(par, me, b)
//      wait   flagB(par, q, b) for all q ; copy stageB(par)[chunk q][slice b] -> data, all q
// Every rank pushes (world-1)/world of the vector in each phase over its world-1 links at once; the result
// is assembled in uncached staging and copied into the (cached) gradient arena by its owner only, so no peer
// ever stores into memory that the local L2 may hold.  Same parity argument and bounded waits as above.
//   stageA [2][world][csz_max] | stageB [2][n_max] | flagA [2][world][kBlocks2] | flagB [2][world][kBlocks2]
// The block count is a launch parameter (<= kBlocks2, identical on all ranks): more blocks drive the links and the
// uncached staging harder but take CUs from the backward kernels this collective overlaps with.
// ---------------------------------------------------------------------------------------------
constexpr int kP2PBlocks2 = 256;   // flag slots per (parity, rank): the launch may use any block count up to this
constexpr int kP2PThreads2 = 512;

struct P2P2Args {
  float* data;
  char* base[kP2PMaxWorld];
  int* status;
  int world, rank, n, n_max;
  unsigned seq;
  long long timeout_ticks;
  int gather;   // 0: leave the result in stageB(parity) of this rank (the consumer reads it there) instead of copying it to data
};

__device__ __forceinline__ void p2p_wait_all(const unsigned* flags_row0, int stride, int world, unsigned seq,
                                             long long timeout_ticks, int* status) {
  if ((int)threadIdx.x < world) {
    const unsigned* f = flags_row0 + (size_t)threadIdx.x * stride;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      if (wall_clock64() - t0 > timeout_ticks) {
        *status = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(kP2PThreads2) void p2p_allreduce_twoshot(P2P2Args a) {
  const int tid = threadIdx.x, b = blockIdx.x, W = a.world;
  const int par = a.seq & 1u;
  const int csz = a.n / W, csz_max = a.n_max / W;       // n, n_max: multiples of 4 * world
  const int c4 = csz / 4;
  const int nblk = gridDim.x;                            // the same on every rank (host contract)
  const int per = (c4 + nblk - 1) / nblk;
  const int lo4 = b * per, hi4 = min(c4, lo4 + per);
  auto stageA = [&](int r) { return reinterpret_cast<float*>(a.base[r]); };
  auto stageB = [&](int r) { return reinterpret_cast<float*>(a.base[r]) + 2 * (size_t)a.n_max; };
  auto flagA = [&](int r) { return reinterpret_cast<unsigned*>(a.base[r] + 16 * (size_t)a.n_max); };
  auto flagB = [&](int r) { return flagA(r) + 2 * W * kP2PBlocks2; };

  // ---- A: scatter my chunks to their owners ----
#pragma unroll 1
  for (int k = 0; k < W; ++k) {
    const int p = (a.rank + k) % W;
    const f32x4* src = reinterpret_cast<const f32x4*>(a.data + (size_t)p * csz);
    f32x4* dst = reinterpret_cast<f32x4*>(stageA(p) + ((size_t)par * W + a.rank) * csz_max);
    for (int i = lo4 + tid; i < hi4; i += kP2PThreads2) dst[i] = src[i];
  }
  __threadfence_system();
  __syncthreads();
  if (tid < W)
    __hip_atomic_store(flagA(tid) + ((size_t)par * W + a.rank) * kP2PBlocks2 + b, a.seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  p2p_wait_all(flagA(a.rank) + (size_t)par * W * kP2PBlocks2 + b, kP2PBlocks2, W, a.seq, a.timeout_ticks, a.status);

  // ---- reduce my chunk (rank order) and B: broadcast it ----
  {
    const float* mine = stageA(a.rank) + (size_t)par * W * csz_max;
    for (int i = lo4 + tid; i < hi4; i += kP2PThreads2) {
      f32x4 s = reinterpret_cast<const f32x4*>(mine)[i];
      for (int q = 1; q < W; ++q) s += reinterpret_cast<const f32x4*>(mine + (size_t)q * csz_max)[i];
#pragma unroll 1
      for (int k = 0; k < W; ++k) {
        const int p = (a.rank + k) % W;
        reinterpret_cast<f32x4*>(stageB(p) + (size_t)par * a.n_max + (size_t)a.rank * csz)[i] = s;
      }
    }
  }
  __threadfence_system();
  __syncthreads();
  if (tid < W)
    __hip_atomic_store(flagB(tid) + ((size_t)par * W + a.rank) * kP2PBlocks2 + b, a.seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  p2p_wait_all(flagB(a.rank) + (size_t)par * W * kP2PBlocks2 + b, kP2PBlocks2, W, a.seq, a.timeout_ticks, a.status);

  // ---- gather: every chunk's slice b out of my staging into the arena ----
  if (!a.gather) return;
  for (int q = 0; q < W; ++q) {
    const f32x4* src = reinterpret_cast<const f32x4*>(stageB(a.rank) + (size_t)par * a.n_max + (size_t)q * csz);
    f32x4* dst = reinterpret_cast<f32x4*>(a.data + (size_t)q * csz);
    for (int i = lo4 + tid; i < hi4; i += kP2PThreads2) dst[i] = src[i];
  }
}

}  // namespace dtf

extern "C" {

// bytes needed by one rank for vectors of up to n_max floats (n_max rounded up to a multiple of 4 by the caller)
// (the flag area is sized for the fused data-parallel tail of optim.hip, 1024 workgroups; the one-shot kernel here
// uses the first kP2PBlocks slots per (parity, rank) with its own stride -- flags only ever match the current seq)
long dtf_p2p_bytes(int world, long n_max) {
  return 2L * world * n_max * 4 + 2L * world * 1024 * 4;
}

int dtf_p2p_alloc(long bytes, void** out) {
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) return (int)e;
  e = hipMemset(p, 0, (size_t)bytes);
  if (e != hipSuccess) return (int)e;
  e = hipDeviceSynchronize();
  *out = p;
  return (int)e;
}

int dtf_p2p_free(void* p) { return (int)hipFree(p); }

int dtf_p2p_export(void* p, char* handle64) {
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) return (int)e;
  static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
  memcpy(handle64, &h, 64);
  return 0;
}

int dtf_p2p_import(const char* handle64, void** out) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  return (int)hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess);
}

int dtf_p2p_unmap(void* p) { return (int)hipIpcCloseMemHandle(p); }

// bases[r]: rank r's allocation as mapped here (bases[rank] = the local one).  Reduces data[0..n) in place.
int dtf_p2p_allreduce(void* data, void* const* bases, int world, int rank, long n, long n_max, unsigned seq,
                      int* status, double timeout_s, hipStream_t s) {
  if (world < 1 || world > dtf::kP2PMaxWorld || rank < 0 || rank >= world) return -2;
  if (n < 0 || n > n_max || (n & 3) || (n_max & 3) || seq == 0) return -2;
  if ((reinterpret_cast<uintptr_t>(data) & 15) != 0) return -3;
  dtf::P2PArgs a;
  a.data = static_cast<float*>(data);
  for (int r = 0; r < dtf::kP2PMaxWorld; ++r) {
    char* base = static_cast<char*>(bases[r < world ? r : rank]);
    a.stage[r] = reinterpret_cast<float*>(base);
    a.flags[r] = reinterpret_cast<unsigned*>(base + 2L * world * n_max * 4);
  }
  a.status = status;
  a.world = world;
  a.rank = rank;
  a.n = (int)n;
  a.n_max = (int)n_max;
  a.seq = seq;
  a.timeout_ticks = (long long)(timeout_s * 1e8);
  hipLaunchKernelGGL(dtf::p2p_allreduce_oneshot, dim3(dtf::kP2PBlocks), dim3(dtf::kP2PThreads), 0, s, a);
  return (int)hipGetLastError();
}

long dtf_p2p2_bytes(int world, long n_max) {
  return 16L * n_max + 4L * world * dtf::kP2PBlocks2 * 4;
}

// Two-shot variant; n and n_max must be multiples of 4 * world.
int dtf_p2p2_allreduce(void* data, void* const* bases, int world, int rank, long n, long n_max, unsigned seq,
                       int* status, double timeout_s, int blocks, int gather, hipStream_t s) {
  if (blocks < 1 || blocks > dtf::kP2PBlocks2) return -2;
  if (world < 1 || world > dtf::kP2PMaxWorld || rank < 0 || rank >= world) return -2;
  if (n < 0 || n > n_max || n % (4 * world) || n_max % (4 * world) || seq == 0) return -2;
  if ((reinterpret_cast<uintptr_t>(data) & 15) != 0) return -3;
  dtf::P2P2Args a;
  a.data = static_cast<float*>(data);
  for (int r = 0; r < dtf::kP2PMaxWorld; ++r) a.base[r] = static_cast<char*>(bases[r < world ? r : rank]);
  a.status = status;
  a.world = world;
  a.rank = rank;
  a.n = (int)n;
  a.n_max = (int)n_max;
  a.seq = seq;
  a.timeout_ticks = (long long)(timeout_s * 1e8);
  a.gather = gather;
  hipLaunchKernelGGL(dtf::p2p_allreduce_twoshot, dim3(blocks), dim3(dtf::kP2PThreads2), 0, s, a);
  return (int)hipGetLastError();
}

}  // extern "C"
