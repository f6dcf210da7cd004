// Exact-fp32 GEMMs of the dense layers on the f32-input matrix cores (v_mfma_f32_32x32x2_f32), the
// --dtype fp32 counterpart of gemm.hip (reference .idea/MNISTDist.py:83-84 fc1 MatMul and the MatMul
// gradient ops autodiff generates from it).  Operands and results are fp32 in memory; no shadow copies.
//
//   nn (fc1 forward, split-K)  part[z][M,N] = A[M,Kz] . B[Kz,N]        A K-contiguous, B = wd1 [in,out] as TF stores it
//   nt (fc1 dgrad)             C[M,N]       = A[M,K]  . B[N,K]^T       both K-contiguous (B = wd1 itself)
//   tn (fc1 wgrad)             C[M,N]       = A[K,M]^T . B[K,N]        both K-slow (K = batch); optional fused SGD
//
// Tiling: 128x128x32, 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 64x32 = two 32x32 accumulators.
// Two waves per SIMD: one issues its 64-cycle MFMAs while the other reads fragments or waits at the barrier.
// LDS keeps each operand in its global orientation (coalesced 16-byte staging, no transposes):
//   K-contiguous: [row][32 + 4] floats -- a lane reads its row's four k (k = 8*kb + 4*h + e) with ONE
//                 ds_read_b128; the 36-float pitch makes the 16-lane groups of that read conflict-free;
//   K-slow:       [k][128] floats      -- the same four k are four ds_read_b32 (32 consecutive columns
//                 per half-wave: conflict-free).
// Register-prefetched double buffering, one barrier per K tile: a tile is 2048 cycles of MFMA per wave
// against ~150 cycles of LDS traffic, so this simple pipeline already leaves the matrix pipe as the bound.
#include <cstdlib>

#include "f32_common.hpp"
#include "head_wgrad_core.hpp"
#include "peer_store.hpp"
#include <hip/hip_ext.h>

// csrc/kernels/optim.hip: sums part[G][stride] slabs into out (the split-K dgrad below reuses it)
extern "C" int dtf_reduce_segments(int count, const void** part, void** out, const int* G, const int* n, const long* stride,
                                   hipStream_t s);

namespace dtf {

enum { F32_KCONTIG = 0, F32_KSLOW = 1 };
enum { F32_EPI_STORE = 0, F32_EPI_SGD = 1 };

// HEADW: the grid has hw_layers extra z layers whose workgroups compute the output-layer weight gradient
// (head_wgrad_core.hpp) -- dispatched after the tiles, into the CUs the fc1 dgrad's 200 tiles leave idle, instead of
// a launch of their own (the bf16 kernel set does the same, gemm.hip HEADW).
template <int LA, int LB, int EPI, bool HEADW = false, int BK = 32>
__global__ __launch_bounds__(512) void gemm_f32(const float* A, const float* Bm, float* __restrict__ C, int M, int N,
                                                int K, int lda, int ldb, int ldc, int tiles_per_split,
                                                const float* __restrict__ hyper,
                                                HeadWgradArgsT<float> hw = HeadWgradArgsT<float>{}, int hw_guests = 0,
                                                int hw_layers = 0) {
  constexpr int BM = 128, BN = 128, NT = 512;
  constexpr int RC = BM * BK / 4 / NT;   // 16-byte staging chunks per thread and operand
  constexpr int SL = BK / 4;             // 16-byte slots per K-contiguous row
  constexpr int PK = BK + 4;
  constexpr int A_FLOATS = (LA == F32_KCONTIG) ? BM * PK : BK * BM;
  constexpr int B_FLOATS = (LB == F32_KCONTIG) ? BN * PK : BK * BN;
  extern __shared__ __attribute__((aligned(16))) float smem_f32[];
  float* sA = smem_f32;                  // [2][A_FLOATS]
  float* sB = smem_f32 + 2 * A_FLOATS;   // [2][B_FLOATS]
  if constexpr (HEADW) {
    static_assert(2 * (A_FLOATS + B_FLOATS) >= head_wgrad_lds_floats<8>(), "head wgrad scratch must fit in the tile buffers");
    const int first_guest_layer = (int)gridDim.z - hw_layers;
    if ((int)blockIdx.z >= first_guest_layer) {
      const int id = (((int)blockIdx.z - first_guest_layer) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      if (id < hw_guests) head_wgrad_block<8, float>(hw, id % hw.nbx, id / hw.nbx, smem_f32);
      return;
    }
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int i32 = lane & 31, h = lane >> 5;
  int tile_x, tile_y;
  xcd_tile(gridDim.x, gridDim.y, tile_x, tile_y);
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int T_all = (K + BK - 1) / BK;
  const int t_begin = blockIdx.z * tiles_per_split;
  const int T = min(T_all, t_begin + tiles_per_split) - t_begin;

  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  // unconditional raw buffer loads: rows past the end of the matrix read as zero
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(A, (LA == F32_KCONTIG ? (size_t)M : (size_t)K) * lda * 4);
  const __amdgpu_buffer_rsrc_t rB = make_rsrc(Bm, (LB == F32_KCONTIG ? (size_t)N : (size_t)K) * ldb * 4);
  uint32_t va[RC], vb[RC];
  int la[RC], lb[RC];   // LDS float offsets of this thread's chunks per operand
#pragma unroll
  for (int j = 0; j < RC; ++j) {
    const int c = tid + j * NT;
    if (LA == F32_KCONTIG) {
      const int row = c / SL, slot = c % SL;
      va[j] = (uint32_t)(m0 + row) * (uint32_t)(lda * 4) + slot * 16 + (uint32_t)t_begin * (BK * 4);
      la[j] = row * PK + slot * 4;
    } else {
      const int row = c >> 5, slot = c & 31;
      va[j] = (uint32_t)(t_begin * BK + row) * (uint32_t)(lda * 4) + (uint32_t)(m0 + slot * 4) * 4;
      la[j] = row * BM + slot * 4;
    }
    if (LB == F32_KCONTIG) {
      const int row = c / SL, slot = c % SL;
      vb[j] = (uint32_t)(n0 + row) * (uint32_t)(ldb * 4) + slot * 16 + (uint32_t)t_begin * (BK * 4);
      lb[j] = row * PK + slot * 4;
    } else {
      const int row = c >> 5, slot = c & 31;
      vb[j] = (uint32_t)(t_begin * BK + row) * (uint32_t)(ldb * 4) + (uint32_t)(n0 + slot * 4) * 4;
      lb[j] = row * BN + slot * 4;
    }
  }
  const uint32_t stepA = (LA == F32_KCONTIG) ? BK * 4 : (uint32_t)lda * (BK * 4);
  const uint32_t stepB = (LB == F32_KCONTIG) ? BK * 4 : (uint32_t)ldb * (BK * 4);

  f32x4 ra[RC], rb[RC];
  auto gload = [&](int t) {
#pragma unroll
    for (int j = 0; j < RC; ++j) ra[j] = buf_load16f(rA, va[j], (uint32_t)t * stepA);
#pragma unroll
    for (int j = 0; j < RC; ++j) rb[j] = buf_load16f(rB, vb[j], (uint32_t)t * stepB);
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int j = 0; j < RC; ++j) *reinterpret_cast<f32x4*>(sA + buf * A_FLOATS + la[j]) = ra[j];
#pragma unroll
    for (int j = 0; j < RC; ++j) *reinterpret_cast<f32x4*>(sB + buf * B_FLOATS + lb[j]) = rb[j];
  };
  auto compute = [&](int buf) {
    const float* a = sA + buf * A_FLOATS;
    const float* b = sB + buf * B_FLOATS;
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      f32x4 av[2], bv;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = wm * 64 + mt * 32 + i32;
        if (LA == F32_KCONTIG) {
          av[mt] = *reinterpret_cast<const f32x4*>(a + row * PK + kb * 8 + 4 * h);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) av[mt][e] = a[(kb * 8 + 4 * h + e) * BM + row];
        }
      }
      {
        const int col = wn * 32 + i32;
        if (LB == F32_KCONTIG) {
          bv = *reinterpret_cast<const f32x4*>(b + col * PK + kb * 8 + 4 * h);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[e] = b[(kb * 8 + 4 * h + e) * BN + col];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma32f(av[mt][e], bv[e], acc[mt]);
    }
  };

  if (T > 0) {
    gload(0);
    swrite(0);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      // in flight during this tile's MFMAs.  Unconditional (exact wait counts): past the last tile the loads
      // return in-range data or zeros that are written to the idle buffer and never multiplied
      gload(t + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(t & 1);
      swrite((t + 1) & 1);
      __syncthreads();
    }
  }

  // C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * h
  float* Cz = C + (EPI == F32_EPI_STORE ? (size_t)blockIdx.z * M * ldc : 0);
  const int col = n0 + wn * 32 + i32;
  if (col < N) {
    if (EPI == F32_EPI_SGD) {
      // fused update of the fp32 weight: w -= lr * g straight from the accumulators (the gradient is never
      // written).  All weights of the lane are requested before the first store (a load/store chain otherwise).
      const float lr = hyper[0] * hyper[4];
      float w[2][16];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, M - 1);
          w[mt][r] = __builtin_nontemporal_load(Cz + (size_t)row * ldc + col);
        }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < M) Cz[(size_t)row * ldc + col] = fmaf(-lr, acc[mt][r], w[mt][r]);
        }
    } else {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < M) Cz[(size_t)row * ldc + col] = acc[mt][r];
        }
    }
  }
}

// ----------------------------------------------------------------------------------------------------
// gemm_f32_t112: the same products on 128 x 112 (WM: the eight waves split M) or 112 x 128 (the waves split N) tiles of
// v_mfma_f32_16x16x4_f32 fragments, wave tile 16 x 112 = seven accumulators.  What it is for: fc1's 3136 = 28 * 112
// features.  128-wide tiles cut them into 24.5 columns -- 200 workgroups for 256 CUs, one round in which 56 CUs have
// nothing to do -- 112-wide tiles into 28: 224 workgroups of 7/8 the work, the round 12.5 % shorter.
// (Finer tiles do not help: the round is as long as its longest CU, and 400 half-size workgroups still put two on one.)
//   K-contiguous operands: [row][32 + 4] floats as above (one ds_read_b128 = four k of a fragment row);
//   K-slow operands:       [k][112] or [k][128 + 16] floats: consecutive k rows land 48 / 16 floats apart modulo the
//                          64 banks, so the 4 x 16 lanes of a fragment read are conflict-free (see kslow_row).
// Per 16 k a wave issues 1 + 7 fragment reads for 28 MFMAs (896 cycles).
// ----------------------------------------------------------------------------------------------------
template <int LA, int LB, int EPI, bool WM, bool HEADW = false, int BK = 32>
__global__ __launch_bounds__(512) void gemm_f32_t112(const float* A, const float* Bm, float* __restrict__ C, int M, int N,
                                                     int K, int lda, int ldb, int ldc, int tiles_per_split,
                                                     const float* __restrict__ hyper,
                                                     HeadWgradArgsT<float> hw = HeadWgradArgsT<float>{}, int hw_guests = 0,
                                                     int hw_layers = 0) {
  constexpr int BM = WM ? 128 : 112, BN = WM ? 112 : 128, NT = 512;
  constexpr int FA = WM ? 1 : 7, FB = WM ? 7 : 1;   // fragments per wave
  constexpr int PK = BK + 4;
  constexpr int PA = (BM == 128) ? 144 : 112, PB = (BN == 128) ? 144 : 112;   // K-slow row pitches
  constexpr int A_FLOATS = (LA == F32_KCONTIG) ? BM * PK : BK * PA;
  constexpr int B_FLOATS = (LB == F32_KCONTIG) ? BN * PK : BK * PB;
  constexpr int A_CHUNKS = BM * BK / 4, B_CHUNKS = BN * BK / 4;   // 16-byte staging chunks per tile: 1024 / 896 at BK 32
  constexpr int RA = (A_CHUNKS + NT - 1) / NT, RB = (B_CHUNKS + NT - 1) / NT;   // staging rounds per thread
  constexpr int SL = BK / 4;                                                      // 16-byte slots per K-contiguous row
  extern __shared__ __attribute__((aligned(16))) float smem_f32[];
  float* sA = smem_f32;                  // [2][A_FLOATS]
  float* sB = smem_f32 + 2 * A_FLOATS;   // [2][B_FLOATS]
  if constexpr (HEADW) {
    static_assert(2 * (A_FLOATS + B_FLOATS) >= head_wgrad_lds_floats<8>(), "head wgrad scratch must fit in the tile buffers");
    const int first_guest_layer = (int)gridDim.z - hw_layers;
    if ((int)blockIdx.z >= first_guest_layer) {
      const int id = (((int)blockIdx.z - first_guest_layer) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      if (id < hw_guests) head_wgrad_block<8, float>(hw, id % hw.nbx, id / hw.nbx, smem_f32);
      return;
    }
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  int tile_x, tile_y;
  xcd_tile(gridDim.x, gridDim.y, tile_x, tile_y);
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int T_all = (K + BK - 1) / BK;
  const int t_begin = blockIdx.z * tiles_per_split;
  const int T = min(T_all, t_begin + tiles_per_split) - t_begin;

  f32x4 acc[FA][FB];
#pragma unroll
  for (int fa = 0; fa < FA; ++fa)
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) acc[fa][fb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // unconditional raw buffer loads: rows past the end of the matrix read as zero
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(A, (LA == F32_KCONTIG ? (size_t)M : (size_t)K) * lda * 4);
  const __amdgpu_buffer_rsrc_t rB = make_rsrc(Bm, (LB == F32_KCONTIG ? (size_t)N : (size_t)K) * ldb * 4);
  uint32_t va[RA], vb[RB];
  int la[RA], lb[RB];   // LDS float offsets of this thread's chunks per operand
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    {
      const int c = min(tid + j * NT, A_CHUNKS - 1);
      if (LA == F32_KCONTIG) {
        const int row = c / SL, slot = c % SL;
        va[j] = (uint32_t)(m0 + row) * (uint32_t)(lda * 4) + slot * 16 + (uint32_t)t_begin * (BK * 4);
        la[j] = row * PK + slot * 4;
      } else {
        const int row = c / (BM / 4), slot = c % (BM / 4);
        va[j] = (uint32_t)(t_begin * BK + row) * (uint32_t)(lda * 4) + (uint32_t)(m0 + slot * 4) * 4;
        la[j] = row * PA + slot * 4;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    {
      const int c = min(tid + j * NT, B_CHUNKS - 1);
      if (LB == F32_KCONTIG) {
        const int row = c / SL, slot = c % SL;
        vb[j] = (uint32_t)(n0 + row) * (uint32_t)(ldb * 4) + slot * 16 + (uint32_t)t_begin * (BK * 4);
        lb[j] = row * PK + slot * 4;
      } else {
        const int row = c / (BN / 4), slot = c % (BN / 4);
        vb[j] = (uint32_t)(t_begin * BK + row) * (uint32_t)(ldb * 4) + (uint32_t)(n0 + slot * 4) * 4;
        lb[j] = row * PB + slot * 4;
      }
    }
  }
  // the 112-row operand's last round is partial: threads past its end load a clamped chunk and drop it
  const uint32_t stepA = (LA == F32_KCONTIG) ? BK * 4 : (uint32_t)lda * (BK * 4);
  const uint32_t stepB = (LB == F32_KCONTIG) ? BK * 4 : (uint32_t)ldb * (BK * 4);

  f32x4 ra[RA], rb[RB];
  auto gload = [&](int t) {
#pragma unroll
    for (int j = 0; j < RA; ++j) ra[j] = buf_load16f(rA, va[j], (uint32_t)t * stepA);
#pragma unroll
    for (int j = 0; j < RB; ++j) rb[j] = buf_load16f(rB, vb[j], (uint32_t)t * stepB);
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int j = 0; j < RA; ++j)
      if ((j + 1) * NT <= A_CHUNKS || tid + j * NT < A_CHUNKS) *reinterpret_cast<f32x4*>(sA + buf * A_FLOATS + la[j]) = ra[j];
#pragma unroll
    for (int j = 0; j < RB; ++j)
      if ((j + 1) * NT <= B_CHUNKS || tid + j * NT < B_CHUNKS) *reinterpret_cast<f32x4*>(sB + buf * B_FLOATS + lb[j]) = rb[j];
  };
  // fragment element e of lane (i16, g) <-> k = 16 * kb + 4 * g + e, for A and B alike; with both operands K-slow
  // k = 16 * kb + 4 * e + g instead: the four lane groups of a ds_read_b32 then read CONSECUTIVE k rows (48 / 16 floats
  // apart modulo the banks) -- rows four apart are a multiple of 64 floats apart with either pitch, a 4-way conflict.
  constexpr bool KSS = (LA == F32_KSLOW && LB == F32_KSLOW);
  auto kslow_row = [&](int kb, int e) { return KSS ? kb * 16 + 4 * e + g : kb * 16 + 4 * g + e; };
  auto compute = [&](int buf) {
    const float* a = sA + buf * A_FLOATS;
    const float* b = sB + buf * B_FLOATS;
#pragma unroll
    for (int kb = 0; kb < BK / 16; ++kb) {
      f32x4 av[FA], bv[FB];
#pragma unroll
      for (int fa = 0; fa < FA; ++fa) {
        const int row = (WM ? wave : fa) * 16 + i16;
        if (LA == F32_KCONTIG) {
          av[fa] = *reinterpret_cast<const f32x4*>(a + row * PK + kb * 16 + 4 * g);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) av[fa][e] = a[kslow_row(kb, e) * PA + row];
        }
      }
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        const int col = (WM ? fb : wave) * 16 + i16;
        if (LB == F32_KCONTIG) {
          bv[fb] = *reinterpret_cast<const f32x4*>(b + col * PK + kb * 16 + 4 * g);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[fb][e] = b[kslow_row(kb, e) * PB + col];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int fa = 0; fa < FA; ++fa)
#pragma unroll
          for (int fb = 0; fb < FB; ++fb) acc[fa][fb] = mfma16f(av[fa][e], bv[fb][e], acc[fa][fb]);
    }
  };

  if (T > 0) {
    gload(0);
    swrite(0);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      gload(t + 1);   // in flight during this tile's MFMAs; unconditional (see gemm_f32)
      __builtin_amdgcn_sched_barrier(0);
      compute(t & 1);
      swrite((t + 1) & 1);
      __syncthreads();
    }
  }

  // C/D map of a 16x16 fragment: col = lane & 15, row = 4 * (lane >> 4) + r
  float* Cz = C + (EPI == F32_EPI_STORE ? (size_t)blockIdx.z * M * ldc : 0);
  if (EPI == F32_EPI_SGD) {
    // fused update of the fp32 weight (see gemm_f32): all 28 weights of the lane are requested before the first store
    const float lr = hyper[0] * hyper[4];
    f32x4 w[FA][FB];
#pragma unroll
    for (int fa = 0; fa < FA; ++fa)
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        const int col = min(n0 + (WM ? fb : wave) * 16 + i16, N - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = min(m0 + (WM ? wave : fa) * 16 + 4 * g + r, M - 1);
          w[fa][fb][r] = __builtin_nontemporal_load(Cz + (size_t)row * ldc + col);
        }
      }
#pragma unroll
    for (int fa = 0; fa < FA; ++fa)
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        const int col = n0 + (WM ? fb : wave) * 16 + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + (WM ? wave : fa) * 16 + 4 * g + r;
          if (row < M && col < N) Cz[(size_t)row * ldc + col] = fmaf(-lr, acc[fa][fb][r], w[fa][fb][r]);
        }
      }
  } else {
#pragma unroll
    for (int fa = 0; fa < FA; ++fa)
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        const int col = n0 + (WM ? fb : wave) * 16 + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + (WM ? wave : fa) * 16 + 4 * g + r;
          if (row < M && col < N) Cz[(size_t)row * ldc + col] = acc[fa][fb][r];
        }
      }
  }
}

// ----------------------------------------------------------------------------------------------------
// gemm_f32_sk: stream-K over the 128 x 128 x 32 kernel's tiles, for products whose tiles do not fill the chip --
// fc1 wgrad is 25 x 8 = 200 tiles for 256 CUs.  The (tile, k-tile) iterations are laid end to end -- 200 x T, T = K / 32
// -- and cut into one equal range per CU (25 T / 32 iterations: 25 at 1024 images).  A range spans at most two tiles:
// it finishes the tile it starts in (as that tile's OWNER, unless it started it), then begins the next one and hands
// that beginning over as a partial sum.  XCD-aligned: workgroup w = (blockIdx % 8) * 32 + blockIdx / 8, so the 32 ranges
// of an XCD are consecutive and cover exactly 25 whole tiles -- owner and contributors of a tile always sit on one XCD,
// and an owner only ever waits for LOWER-numbered workgroups of its own XCD (all of them resident: one workgroup per CU).
// Hand-over: the contributor writes its 64 KiB accumulator tile into its slot of an UNCACHED workspace with
// write-through stores, waits for them (vmcnt 0: no fence, see peer_store.hpp) and sets its flag; the owner -- which ran
// and published the part IT hands over before the part it finishes -- polls the flag(s), adds the partial
// sums in ascending-k order (a fixed order: results are deterministic), clears the flags for the next launch (graph
// replay needs no per-launch argument) and runs the normal epilogue.  Bounded polls: a broken launch computes
// garbage, it does not hang.
// ----------------------------------------------------------------------------------------------------
struct F32SkArgs {
  float* ws;         // [workgroups][128 * 128] partial tiles
  unsigned* flags;   // [workgroups], zero between launches
  int T, tiles_x, ipw;
};

template <int LA, int LB, int EPI>
__global__ __launch_bounds__(512) void gemm_f32_sk(const float* A, const float* Bm, float* __restrict__ C, int M, int N,
                                                   int K, int lda, int ldb, int ldc, const float* __restrict__ hyper,
                                                   F32SkArgs sk) {
  constexpr int BM = 128, BN = 128, BK = 32, NT = 512;
  constexpr int PK = BK + 4;
  constexpr int A_FLOATS = (LA == F32_KCONTIG) ? BM * PK : BK * BM;
  constexpr int B_FLOATS = (LB == F32_KCONTIG) ? BN * PK : BK * BN;
  extern __shared__ __attribute__((aligned(16))) float smem_f32[];
  float* sA = smem_f32;                  // [2][A_FLOATS]
  float* sB = smem_f32 + 2 * A_FLOATS;   // [2][B_FLOATS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int i32 = lane & 31, h = lane >> 5;
  const int nwg = gridDim.x;
  const int w = ((int)blockIdx.x % 8) * (nwg / 8) + (int)blockIdx.x / 8;
  const long g0 = (long)w * sk.ipw;
  const int t0 = (int)(g0 / sk.T), ka = (int)(g0 % sk.T);
  const int n1 = min(sk.T - ka, sk.ipw);   // iterations in the tile the range starts in
  const int n2 = sk.ipw - n1;              // ... and at the head of the next one

  const __amdgpu_buffer_rsrc_t rA = make_rsrc(A, (LA == F32_KCONTIG ? (size_t)M : (size_t)K) * lda * 4);
  const __amdgpu_buffer_rsrc_t rB = make_rsrc(Bm, (LB == F32_KCONTIG ? (size_t)N : (size_t)K) * ldb * 4);
  const uint32_t stepA = (LA == F32_KCONTIG) ? BK * 4 : (uint32_t)lda * (BK * 4);
  const uint32_t stepB = (LB == F32_KCONTIG) ? BK * 4 : (uint32_t)ldb * (BK * 4);

  // staging addresses of a segment (tile, first k-tile); the LDS offsets of a thread's two chunks are the same for all
  int la[2], lb[2];
  auto seg_addr = [&](int tile, int kbeg, uint32_t (&va)[2], uint32_t (&vb)[2]) {
    const int m0 = (tile / sk.tiles_x) * BM, n0 = (tile % sk.tiles_x) * BN;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = tid + j * NT;
      if (LA == F32_KCONTIG) {
        const int row = c >> 3, slot = c & 7;
        va[j] = (uint32_t)(m0 + row) * (uint32_t)(lda * 4) + slot * 16 + (uint32_t)kbeg * (BK * 4);
        la[j] = row * PK + slot * 4;
      } else {
        const int row = c >> 5, slot = c & 31;
        va[j] = (uint32_t)(kbeg * BK + row) * (uint32_t)(lda * 4) + (uint32_t)(m0 + slot * 4) * 4;
        la[j] = row * BM + slot * 4;
      }
      if (LB == F32_KCONTIG) {
        const int row = c >> 3, slot = c & 7;
        vb[j] = (uint32_t)(n0 + row) * (uint32_t)(ldb * 4) + slot * 16 + (uint32_t)kbeg * (BK * 4);
        lb[j] = row * PK + slot * 4;
      } else {
        const int row = c >> 5, slot = c & 31;
        vb[j] = (uint32_t)(kbeg * BK + row) * (uint32_t)(ldb * 4) + (uint32_t)(n0 + slot * 4) * 4;
        lb[j] = row * BN + slot * 4;
      }
    }
  };
  f32x4 ra[2], rb[2];
  auto gload = [&](const uint32_t (&va)[2], const uint32_t (&vb)[2], int t) {
#pragma unroll
    for (int j = 0; j < 2; ++j) ra[j] = buf_load16f(rA, va[j], (uint32_t)t * stepA);
#pragma unroll
    for (int j = 0; j < 2; ++j) rb[j] = buf_load16f(rB, vb[j], (uint32_t)t * stepB);
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(sA + buf * A_FLOATS + la[j]) = ra[j];
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(sB + buf * B_FLOATS + lb[j]) = rb[j];
  };
  f32x16 acc[2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  };
  auto compute = [&](int buf) {
    const float* a = sA + buf * A_FLOATS;
    const float* b = sB + buf * B_FLOATS;
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      f32x4 av[2], bv;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = wm * 64 + mt * 32 + i32;
        if (LA == F32_KCONTIG) {
          av[mt] = *reinterpret_cast<const f32x4*>(a + row * PK + kb * 8 + 4 * h);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) av[mt][e] = a[(kb * 8 + 4 * h + e) * BM + row];
        }
      }
      {
        const int col = wn * 32 + i32;
        if (LB == F32_KCONTIG) {
          bv = *reinterpret_cast<const f32x4*>(b + col * PK + kb * 8 + 4 * h);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[e] = b[(kb * 8 + 4 * h + e) * BN + col];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma32f(av[mt][e], bv[e], acc[mt]);
    }
  };

  // this thread's 32 accumulator values as eight 16-byte pieces of the slot: piece p of thread tid at (p * 512 + tid)
  auto put_partial = [&]() {
    f32x4* dst = reinterpret_cast<f32x4*>(sk.ws + (size_t)w * (BM * BN));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dp_put16(dst + (mt * 4 + q) * NT + tid, (f32x4){acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]});
    dp_puts_done();
    __syncthreads();
    if (tid == 0) uc_store(sk.flags + w, 1u);
  };
  auto epilogue = [&](int tile) {
    // C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * h
    const int m0 = (tile / sk.tiles_x) * BM, n0 = (tile % sk.tiles_x) * BN;
    const int col = n0 + wn * 32 + i32;
    if (col >= N) return;
    if (EPI == F32_EPI_SGD) {
      const float lr = hyper[0] * hyper[4];
      float wv[2][16];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, M - 1);
          wv[mt][r] = __builtin_nontemporal_load(C + (size_t)row * ldc + col);
        }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < M) C[(size_t)row * ldc + col] = fmaf(-lr, acc[mt][r], wv[mt][r]);
        }
    } else {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < M) C[(size_t)row * ldc + col] = acc[mt][r];
        }
    }
  };

  // ONE software pipeline over the range's iterations, the handed-over part FIRST (its owner then finds the partial sum
  // written, acknowledged and flagged long before it needs it, instead of every owner starting to wait when every
  // contributor starts to write): iterations [0, nS0) are the head of tile t0 + 1, the rest k-tiles ka.. of tile t0; the
  // first k-tile of the second part is requested under the last MFMAs of the first, so the switch costs no prologue.
  const bool finishes = ka + n1 == sk.T;
  const int nS0 = finishes ? n2 : 0, nI = nS0 + n1;
  uint32_t va0[2], vb0[2], va1[2], vb1[2];
  seg_addr(t0 + 1, 0, va0, vb0);
  seg_addr(t0, ka, va1, vb1);
  auto gload_it = [&](int it) {
    if (it < nS0) gload(va0, vb0, it);
    else gload(va1, vb1, it - nS0);   // (past the range: the tile's next k-tile or zeros, never multiplied)
  };
  zero_acc();
  gload_it(0);
  swrite(0);
  __syncthreads();
  for (int it = 0; it < nI; ++it) {
    gload_it(it + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(it & 1);
    if (it + 1 == nS0) {   // the head of tile t0 + 1 is complete: hand it over
      put_partial();
      zero_acc();
    }
    swrite((it + 1) & 1);
    __syncthreads();
  }
  if (!finishes) {   // the range lies inside one tile: a contributor only
    put_partial();
    return;
  }
  if (ka > 0) {
    // owner of tile t0: the k-tiles [0, ka) come from the ranges before this one, earliest first
    const int nc = (ka + sk.ipw - 1) / sk.ipw;
    if (tid == 0) {
      for (int j = nc; j >= 1; --j)
        for (int spin = 0; spin < (1 << 22) && uc_load(sk.flags + (w - j)) != 1u; ++spin) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    f32x4 sum[8];
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8) sum[p8] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int j = nc; j >= 1; --j) {
      const unsigned* src = reinterpret_cast<const unsigned*>(sk.ws + (size_t)(w - j) * (BM * BN));
      // all eight 16-byte pieces in flight, ONE wait (uc_load4 waits per load: eight dependent round trips to memory)
      f32x4 v[8];
      const unsigned* q0 = src + (size_t)tid * 4;
      asm volatile(
          "global_load_dwordx4 %0, %8, off sc0 sc1\n\t"
          "global_load_dwordx4 %1, %9, off sc0 sc1\n\t"
          "global_load_dwordx4 %2, %10, off sc0 sc1\n\t"
          "global_load_dwordx4 %3, %11, off sc0 sc1\n\t"
          "global_load_dwordx4 %4, %12, off sc0 sc1\n\t"
          "global_load_dwordx4 %5, %13, off sc0 sc1\n\t"
          "global_load_dwordx4 %6, %14, off sc0 sc1\n\t"
          "global_load_dwordx4 %7, %15, off sc0 sc1\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
          : "v"(q0), "v"(q0 + 1 * NT * 4), "v"(q0 + 2 * NT * 4), "v"(q0 + 3 * NT * 4), "v"(q0 + 4 * NT * 4),
            "v"(q0 + 5 * NT * 4), "v"(q0 + 6 * NT * 4), "v"(q0 + 7 * NT * 4)
          : "memory");
#pragma unroll
      for (int p8 = 0; p8 < 8; ++p8) sum[p8] = (j == nc) ? v[p8] : sum[p8] + v[p8];
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt][4 * q + e] = sum[mt * 4 + q][e] + acc[mt][4 * q + e];
    __syncthreads();   // every thread has read its pieces
    if (tid == 0)
      for (int j = nc; j >= 1; --j) uc_store(sk.flags + (w - j), 0u);
  }
  epilogue(t0);
}

// Which tiling a product gets: the rounds of workgroups it takes on this device times the work of one workgroup.
static int f32_gemm_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}
static int g_f32_t112_forced = -2;   // -1: by shape; 0 | 1: forced by DTF_F32_T112 (A/B runs) or dtf_gemm_f32_force_t112 (tests)
static bool f32_use_t112(int M, int N, bool wm) {
  if (g_f32_t112_forced == -2) {
    const char* v = getenv("DTF_F32_T112");
    g_f32_t112_forced = (v && *v) ? atoi(v) : -1;
  }
  if (g_f32_t112_forced >= 0) return g_f32_t112_forced != 0;
  // K-slow operands (TN, fc1 wgrad) pay four ds_read_b32 per fragment where a K-contiguous one pays one ds_read_b128:
  // with seven fragments per wave that eats the shorter round (72.6 / 71.5 us against the 128 x 128 kernel's 71.0)
  if (!wm) return false;
  const long cus = f32_gemm_cus();
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  const long t112 = wm ? (long)((M + 127) / 128) * ((N + 111) / 112) : (long)((M + 111) / 112) * ((N + 127) / 128);
  return ((t112 + cus - 1) / cus) * 112 < ((t128 + cus - 1) / cus) * 128;
}

static bool f32_bk64() {   // DTF_F32_BK64=0: 32-deep K tiles everywhere (A/B runs)
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("DTF_F32_BK64");
    v = (e && *e) ? (atoi(e) != 0) : 1;
  }
  return v != 0;
}

template <int LA, int LB, int EPI, bool WM, int BK = 32>
constexpr size_t f32_t112_lds() {
  constexpr int BM = WM ? 128 : 112, BN = WM ? 112 : 128;
  return (size_t)(2 * ((LA == F32_KCONTIG) ? BM * (BK + 4) : BK * (BM == 128 ? 144 : 112)) +
                  2 * ((LB == F32_KCONTIG) ? BN * (BK + 4) : BK * (BN == 128 ? 144 : 112))) * 4;
}

template <int LA, int LB, int EPI, bool WM, int BK = 32>
static int launch_gemm_f32_t112(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                const void* hyper, hipStream_t s, hipEvent_t done = nullptr) {
  constexpr size_t lds = f32_t112_lds<LA, LB, EPI, WM, BK>();
  constexpr int BM = WM ? 128 : 112, BN = WM ? 112 : 128;
  auto kern = gemm_f32_t112<LA, LB, EPI, WM, false, BK>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipExtLaunchKernelGGL(kern, dim3((N + BN - 1) / BN, (M + BM - 1) / BM, 1), dim3(512), lds, s, nullptr, done, 0,
                        (const float*)A, (const float*)B, (float*)C, M, N, K, lda, ldb, ldc, (K + 31) / 32,
                        (const float*)hyper, HeadWgradArgsT<float>{}, 0, 0);
  return (int)hipGetLastError();
}

template <int LA, int LB, int EPI, int BK = 32>
static int launch_gemm_f32(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int S,
                           int tps, const void* hyper, hipStream_t s, hipEvent_t done = nullptr) {
  constexpr int PK = BK + 4;
  constexpr size_t lds = (2 * ((LA == F32_KCONTIG) ? 128 * PK : BK * 128) + 2 * ((LB == F32_KCONTIG) ? 128 * PK : BK * 128)) * 4;
  auto kern = gemm_f32<LA, LB, EPI, false, BK>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  // done: bound to the kernel's own completion (see dtf_gemm_tn_ev in gemm.hip)
  hipExtLaunchKernelGGL(kern, dim3((N + 127) / 128, (M + 127) / 128, S), dim3(512), lds, s, nullptr, done, 0, (const float*)A,
                        (const float*)B, (float*)C, M, N, K, lda, ldb, ldc, tps, (const float*)hyper, HeadWgradArgsT<float>{}, 0, 0);
  return (int)hipGetLastError();
}

}  // namespace dtf

using namespace dtf;

extern "C" {

// fc1 forward, GEMM part: part[S][M][N] fp32 partial products of A[M,K] . B[K,N] (bias/ReLU/dropout run in the head
// kernel).  Returns the number of splits used (> 0) or a negative error; S <= 0 picks ~256 workgroups.
int dtf_gemm_f32_nn_partial(const void* A, const void* B, void* part, int M, int N, int K, int lda, int ldb, int S,
                            hipStream_t s) {
  if ((K & 31) || (lda & 3) || (ldb & 3) || (N & 3)) return -3;
  const int T = K / 32;
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  if (S <= 0) {
    S = (256 + tiles - 1) / tiles;
    if (S > 8) S = 8;
  }
  if (S > 32) S = 32;
  if (S > T) S = T;
  const int tps = (T + S - 1) / S;
  S = (T + tps - 1) / tps;
  const int rc = launch_gemm_f32<F32_KCONTIG, F32_KSLOW, F32_EPI_STORE>(A, B, part, M, N, K, lda, ldb, N, S, tps, nullptr, s);
  return rc == 0 ? S : -1;
}

// tiling of the NT / TN products: 1 = 112-wide tiles, 0 = 128 x 128, -1 = chosen by shape (the default)
int dtf_gemm_f32_force_t112(int mode) {
  g_f32_t112_forced = mode < 0 ? -1 : (mode != 0);
  return 0;
}

// C[M,N] = A[M,K] . B[N,K]^T  (fc1 dgrad: A = dh1, B = wd1)
int dtf_gemm_f32_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, hipStream_t s) {
  if ((K & 31) || (lda & 3) || (ldb & 3)) return -3;
  if (f32_use_t112(M, N, true)) {
    if (K % 64 == 0 && f32_bk64())
      return launch_gemm_f32_t112<F32_KCONTIG, F32_KCONTIG, F32_EPI_STORE, true, 64>(A, B, C, M, N, K, lda, ldb, ldc, nullptr, s);
    return launch_gemm_f32_t112<F32_KCONTIG, F32_KCONTIG, F32_EPI_STORE, true>(A, B, C, M, N, K, lda, ldb, ldc, nullptr, s);
  }
  return launch_gemm_f32<F32_KCONTIG, F32_KCONTIG, F32_EPI_STORE>(A, B, C, M, N, K, lda, ldb, ldc, 1, K / 32, nullptr, s);
}

// fc1 dgrad and the output-layer weight gradient in ONE launch (see gemm_f32 HEADW) while tiles x K splits + guests are
// a single round with CUs to spare; otherwise the two launches.  splitk_part != NULL and S > 1: the small-batch split-K
// product of dtf_gemm_f32_nt_splitk (partials summed into C by the slab-reduction kernel).
int dtf_head_f32_wgrad(const void* h1, const void* dlogits, const void* da, void* part, int B, int H, int KS, hipStream_t s);
int dtf_gemm_f32_nt_splitk(const void* A, const void* B, void* part, void* C, int M, int N, int K, int lda, int ldb,
                           int ldc, int S, hipStream_t s);
int dtf_gemm_f32_nt_head_wgrad(const void* A, const void* B, void* splitk_part, void* C, int M, int N, int K, int lda,
                               int ldb, int ldc, int S, const void* h1, const void* dlogits, const void* da, void* part,
                               int Bsz, int H, int KS, hipStream_t s) {
  if ((K & 31) || (lda & 3) || (ldb & 3)) return -3;
  if (KS <= 0) return -2;
  const int T = K / 32;
  if (S < 1 || !splitk_part) S = 1;
  if (S > T) S = T;
  const int tps = (T + S - 1) / S;
  S = (T + tps - 1) / tps;
  const bool t112 = S == 1 && f32_use_t112(M, N, true);
  const int gx = t112 ? (N + 111) / 112 : (N + 127) / 128, gy = (M + 127) / 128;
  const long tiles = (long)gx * gy, wgs = tiles * S;
  const int nbx = (H + kHeadCols - 1) / kHeadCols;
  const long guests = (long)nbx * KS;
  const int layers = (int)((guests + tiles - 1) / tiles);
  if (wgs < 128 || wgs + guests > 288 || (H % 4) || (S > 1 && (ldc != N || (long)M * N > 0x7fffffffL))) {
    int rc = dtf_head_f32_wgrad(h1, dlogits, da, part, Bsz, H, KS, s);
    if (rc) return rc;
    return dtf_gemm_f32_nt_splitk(A, B, splitk_part, C, M, N, K, lda, ldb, ldc, S, s);
  }
  HeadWgradArgsT<float> hw{(const float*)h1, (const float*)dlogits, (const float*)da, (float*)part, Bsz, H,
                           (Bsz + KS - 1) / KS, nbx};
  if (t112) {
    const bool bk64 = K % 64 == 0 && f32_bk64();
    const size_t lds112 = bk64 ? f32_t112_lds<F32_KCONTIG, F32_KCONTIG, F32_EPI_STORE, true, 64>()
                               : f32_t112_lds<F32_KCONTIG, F32_KCONTIG, F32_EPI_STORE, true, 32>();
    auto kern112 = bk64 ? gemm_f32_t112<F32_KCONTIG, F32_KCONTIG, F32_EPI_STORE, true, true, 64>
                        : gemm_f32_t112<F32_KCONTIG, F32_KCONTIG, F32_EPI_STORE, true, true, 32>;
    static bool attr112_set[2] = {false, false};
    if (!attr112_set[bk64]) {
      (void)hipFuncSetAttribute((const void*)kern112, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds112);
      attr112_set[bk64] = true;
    }
    hipLaunchKernelGGL(kern112, dim3(gx, gy, 1 + layers), dim3(512), lds112, s, (const float*)A, (const float*)B, (float*)C, M,
                       N, K, lda, ldb, ldc, tps, (const float*)nullptr, hw, (int)guests, layers);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  constexpr size_t lds = (size_t)(4 * 128 * 36) * 4;
  auto kern = gemm_f32<F32_KCONTIG, F32_KCONTIG, F32_EPI_STORE, true>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(gx, gy, S + layers), dim3(512), lds, s, (const float*)A, (const float*)B,
                     (float*)(S > 1 ? splitk_part : C), M, N, K, lda, ldb, S > 1 ? N : ldc, tps, (const float*)nullptr, hw,
                     (int)guests, layers);
  if (hipGetLastError() != hipSuccess) return -1;
  if (S <= 1) return 0;
  const void* parts[1] = {splitk_part};
  void* outs[1] = {C};
  int G[1] = {S}, n[1] = {M * N};
  long st[1] = {(long)M * N};
  return dtf_reduce_segments(1, parts, outs, G, n, st, s);
}

// The same product for a small batch: M / 128 * N / 128 output tiles do not fill 256 CUs (batch 100: 25 workgroups,
// each walking all of K -> 66 us measured), so K is cut into S ranges whose partial products go to part[S][M][N] and
// are summed into C (dense, ldc == N) by the slab-reduction kernel.  S <= 1 or part == NULL: the plain launch.
int dtf_gemm_f32_nt_splitk(const void* A, const void* B, void* part, void* C, int M, int N, int K, int lda, int ldb,
                           int ldc, int S, hipStream_t s) {
  if (S <= 1 || !part) return dtf_gemm_f32_nt(A, B, C, M, N, K, lda, ldb, ldc, s);
  if ((K & 31) || (lda & 3) || (ldb & 3) || ldc != N || (long)M * N > 0x7fffffffL) return -3;
  const int T = K / 32;
  if (S > T) S = T;
  const int tps = (T + S - 1) / S;
  S = (T + tps - 1) / tps;
  int rc = launch_gemm_f32<F32_KCONTIG, F32_KCONTIG, F32_EPI_STORE>(A, B, part, M, N, K, lda, ldb, N, S, tps, nullptr, s);
  if (rc != 0) return rc;
  const void* parts[1] = {part};
  void* outs[1] = {C};
  int G[1] = {S}, n[1] = {M * N};
  long st[1] = {(long)M * N};
  return dtf_reduce_segments(1, parts, outs, G, n, st, s);
}

// C[M,N] = A[K,M]^T . B[K,N]  (fc1 wgrad: A = flat, B = dh1, K = batch); hyper != NULL: C is the fp32 weight and the
// product its gradient, applied in place as w -= hyper[0] * hyper[4] * g (single-GPU SGD; the gradient is not stored)
int dtf_gemm_f32_tn_ev(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                       const void* hyper, hipEvent_t done, hipStream_t s) {
  if ((lda & 3) || (ldb & 3) || (M & 3) || (N & 3)) return -3;
  const int T = (K + 31) / 32;
  if (f32_use_t112(M, N, false)) {
    if (hyper) return launch_gemm_f32_t112<F32_KSLOW, F32_KSLOW, F32_EPI_SGD, false>(A, B, C, M, N, K, lda, ldb, ldc, hyper, s, done);
    return launch_gemm_f32_t112<F32_KSLOW, F32_KSLOW, F32_EPI_STORE, false>(A, B, C, M, N, K, lda, ldb, ldc, nullptr, s, done);
  }
  // (64-deep K tiles, launch_gemm_f32<.., 64>: 73.1 us against 71.2 for this product -- not the barriers, then)
  if (hyper) return launch_gemm_f32<F32_KSLOW, F32_KSLOW, F32_EPI_SGD>(A, B, C, M, N, K, lda, ldb, ldc, 1, T, hyper, s, done);
  return launch_gemm_f32<F32_KSLOW, F32_KSLOW, F32_EPI_STORE>(A, B, C, M, N, K, lda, ldb, ldc, 1, T, nullptr, s, done);
}
int dtf_gemm_f32_tn(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                    const void* hyper, hipStream_t s) {
  return dtf_gemm_f32_tn_ev(A, B, C, M, N, K, lda, ldb, ldc, hyper, nullptr, s);
}

// Stream-K workspace (gemm_f32_sk): uncached device memory, [256 partial tiles of 64 KiB | 256 flag words], zeroed.
// One per engine (two engines may run this product on two streams at once).
long dtf_gemm_f32_sk_workspace_bytes() { return 256L * 128 * 128 * 4 + 4096; }
void* dtf_gemm_f32_sk_workspace_alloc() {
  void* p = nullptr;
  const size_t bytes = (size_t)dtf_gemm_f32_sk_workspace_bytes();
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (hipMemset(p, 0, bytes) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p);
    return nullptr;
  }
  return p;
}
void dtf_gemm_f32_sk_workspace_free(void* p) {
  if (p) (void)hipFree(p);
}

// The same product as dtf_gemm_f32_tn_ev; ws != NULL (a workspace from dtf_gemm_f32_sk_workspace_alloc): stream-K over
// the tiles when they leave CUs idle and the iterations divide evenly (fc1 wgrad: 3136 x 1024, K = a multiple of 1024
// images), otherwise -- and with DTF_F32_TN_SK=0 -- the plain launch.
int dtf_gemm_f32_tn_sk(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                       const void* hyper, void* ws, hipEvent_t done, hipStream_t s) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* v = getenv("DTF_F32_TN_SK");
    enabled = (v && *v) ? (atoi(v) != 0) : 1;
  }
  const int cus = f32_gemm_cus();
  const int tx = (N + 127) / 128, ty = (M + 127) / 128, T = (K + 31) / 32;
  const long tiles = (long)tx * ty, total = tiles * T;
  const bool fits = ws && enabled && !(lda & 3) && !(ldb & 3) && !(M & 3) && !(N & 3) && cus == 256 && tiles < cus &&
                    tiles % 8 == 0 && total % cus == 0 && total / cus <= T && 4 * tiles >= 3 * cus;
  if (!fits) return dtf_gemm_f32_tn_ev(A, B, C, M, N, K, lda, ldb, ldc, hyper, done, s);
  F32SkArgs sk{(float*)ws, (unsigned*)((char*)ws + 256L * 128 * 128 * 4), T, tx, (int)(total / cus)};
  constexpr size_t lds = (size_t)(4 * 32 * 128) * 4;
#define SK_LAUNCH(EPI_)                                                                                             \
  do {                                                                                                              \
    auto kern = gemm_f32_sk<F32_KSLOW, F32_KSLOW, EPI_>;                                                            \
    static bool attr_set = false;                                                                                   \
    if (!attr_set) {                                                                                                \
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
      attr_set = true;                                                                                              \
    }                                                                                                               \
    hipExtLaunchKernelGGL(kern, dim3(cus), dim3(512), lds, s, nullptr, done, 0, (const float*)A, (const float*)B,   \
                          (float*)C, M, N, K, lda, ldb, ldc, (const float*)hyper, sk);                              \
  } while (0)
  if (hyper) SK_LAUNCH(F32_EPI_SGD);
  else SK_LAUNCH(F32_EPI_STORE);
#undef SK_LAUNCH
  return (int)hipGetLastError();
}

}  // extern "C"
