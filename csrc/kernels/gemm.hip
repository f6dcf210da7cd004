// bf16 MFMA GEMMs for the dense layers (reference .idea/MNISTDist.py:82-89:
// matmul(fc1, wd1)+bd1 -> relu -> dropout -> matmul(., out)+out_b, and the
// MatMul gradient ops autodiff generates from them).
//
//   gemm_nt : C[M,N] = A[M,K] . B[N,K]^T      (both operands K-contiguous)
//             forward  h1 = relu(flat . wd1 + bd1) [+ dropout]   (B = wd1^T shadow)
//             dgrad    dflat = dh1 . wd1^T                       (B = wd1 itself)
//   gemm_tn : C[M,N] = A[K,M]^T . B[K,N]      (both operands K-slow), fp32 out
//             wgrad    dwd1 = flat^T . dh1   -- K is the batch
//
// Tiling: 256 threads = 4 waves as 2x2, wave tile (BM/2)x(BN/2) of 16x16x32
// MFMAs, BK = 64, double-buffered LDS + two register sets (tile t+2 in flight while
// tile t is multiplied), one barrier per K tile.
//   NT: LDS rows are 128 B (64 bf16 of K); the 16-byte chunk index is XORed
//       with (row & 7) so that ds_read_b128 fragment reads are conflict-free.
//   TN: LDS keeps the global [K][M] orientation (coalesced staging) with a
//       +32 B row pad; fragments come from ds_read_b64_tr_b16, and element j of
//       lane group g is k = 16*(j>>2) + 4*g + (j&3) for BOTH operands (any
//       bijection of k is a valid GEMM), which makes the transposed reads
//       bank-conflict free.
#include "common.hpp"
#include <hip/hip_ext.h>
#include "head_wgrad_core.hpp"

namespace dtf {

enum { EPI_STORE = 0, EPI_BIAS_RELU_DROPOUT = 1, EPI_NOSTORE_DIAG = 2 };

#ifndef DTF_GEMM_DMA_DEFAULT
#define DTF_GEMM_DMA_DEFAULT 5
#endif

// ---------------------------------------------------------------------------------------------
// Shared software pipeline.  One K tile (BK = 64) per phase, double-buffered LDS, two register
// sets: the global loads of tile t+2 are issued at the head of phase t and written to LDS at the
// end of phase t+1, so they have two compute phases to land.
//
// What makes this work on the compiler we have (each point measured, see docs/ARCHITECTURE.md):
//  * loads are raw buffer loads (V# + per-thread byte offset computed once + scalar K offset):
//    unconditional, branch-free, and out-of-range rows read as zero.  A guarded `cond ? *p : 0`
//    load becomes a branch, the waitcnt pass then cannot count, and every wait in the loop
//    degrades to vmcnt(0) -- the full memory latency exposed in every phase;
//  * buffer loads are not sunk to their use the way loads through const __restrict__ pointers
//    are, and the sched_barrier after each issue keeps them at the head of the phase;
//  * no per-load address VGPRs, so no write-after-write waits against loads still in flight.
// 128x128 tiles run 8 waves (2 per SIMD): while one wave waits on LDS or at the barrier the other
// issues MFMAs.  A 4-wave tile measured ~1350 cycles per phase for 512 cycles of MFMA.
// ---------------------------------------------------------------------------------------------
// Diagnostic (-DDTF_TIMING=1, tools/conv_timeline.py run .. gemm): thread 0 of every workgroup stamps the
// 100 MHz wall clock: 0 start, 1 first tile in LDS, 2 main loop left, 3 epilogue done
#if DTF_TIMING
__device__ long long* g_gemm_timing = nullptr;
#define DTF_GSTAMP(slot)                                                                          \
  do {                                                                                            \
    if (threadIdx.x == 0 && g_gemm_timing)                                                        \
      g_gemm_timing[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define DTF_GSTAMP(slot) do {} while (0)
#endif

template <int A_CH, int B_CH, class GL, class SW, class CP>
__device__ __forceinline__ void gemm_pipeline(int T, GL&& gload, SW&& swrite, CP&& compute) {
  u32x4 ra0[A_CH], rb0[B_CH], ra1[A_CH], rb1[B_CH];
  DTF_GSTAMP(0);
  gload(ra0, rb0, 0);
  gload(ra1, rb1, 1);
  swrite(ra0, rb0, 0);
  __syncthreads();
  DTF_GSTAMP(1);
  for (int t = 0; t < T; t += 2) {
    gload(ra0, rb0, t + 2);  // past the end: in-range garbage or zeros, written to LDS but never multiplied
    __builtin_amdgcn_sched_barrier(0);
    compute(0);
    swrite(ra1, rb1, 1);
    __syncthreads();
    if (t + 1 >= T) break;
    gload(ra1, rb1, t + 3);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    swrite(ra0, rb0, 0);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA pipeline of the 128 x 128 x 8-wave kernels (round 2): `buffer_load_dwordx4 ... lds` writes the operand
// tiles straight into LDS -- no VGPR round trip and no ds_write_b128 on the LDS store path (13 cycles per
// wave-instruction, which together with the fragment reads made the register-staged loop LDS-bound at 2.3 x its MFMA
// time).  FOUR LDS buffers, tile t in buffer t % 4:
//   phase t:  issue the DMA of tile t+3 | read the fragments of tile t+1 | MFMAs on the fragments of tile t (read in
//             phase t-1) | s_waitcnt vmcnt(4) | s_barrier
// A DMA writes wave-uniform base + lane * 16 bytes, so the LDS images are linear per wave-instruction and every
// swizzle is applied to the SOURCE chunk a lane fetches.  Ordering (guide: read a staged buffer one phase AFTER the
// wait that retires it): vmcnt(4) at the end of phase t leaves only tile t+3's four DMAs of this wave in flight, so
// its parts of tile t+2 have landed; after the barrier every wave's have, and phase t+1 reads them.  Buffer (t+3) % 4
// held tile t-1, whose fragments were read in phase t-2 and consumed by the MFMAs of phase t-1: free at phase t.
// DMAs past the last tile read zeros (beyond the descriptor) or in-range data into buffers that are never multiplied.
// Every kernel using it stages exactly 2 + 2 sixteen-byte chunks per thread and tile (the 4 in vmcnt(4)).
// Measured (tools/scratch/gemm_tile_ab.py, fc1 dgrad shape): 13.8 -> 11.1 us; 45.0 -> 39.5 us at 4096 rows.
// ---------------------------------------------------------------------------------------------
// one 16-byte LDS-DMA per lane: LDS destination = lds (wave-uniform) + lane * 16, source = buffer offset voff + soff
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, const void* lds, uint32_t voff, uint32_t soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(3))) void lds_void;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 16, voff, soff, 0, 0);
#else
  (void)r; (void)lds; (void)voff; (void)soff;
#endif
}

// Transposed LDS read for the DMA-staged kernels.  Through the compiler's builtin, every ds_read_b64_tr_b16 that follows an
// LDS-DMA gets an `s_waitcnt vmcnt(0)` in front of it (the builtin is modelled as reading any memory, so it "may alias"
// the pending DMA writes): the prefetch depth collapses to zero and the kernel runs 10-15 % SLOWER than the register-
// staged one (measured).  As inline asm the read is invisible to that pass; gemm_pipeline_dma4 retires these reads
// itself with `s_waitcnt lgkmcnt(0)` at the end of the phase that issued them (they are consumed one phase later).
typedef __attribute__((ext_vector_type(4))) short dma_s16x4;
__device__ __forceinline__ dma_s16x4 lds_tr_read_b64(const bf16* p) {
  dma_s16x4 v;
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a));
  return v;
}

// The same for a plain 16-byte fragment read where the compiler orders it against the pending DMAs of its own array
// (split-K NN kernel: `s_waitcnt vmcnt(2)` in front of the A-fragment reads of every phase).
__device__ __forceinline__ bf16x8 lds_read_b128(const bf16* p) {
  u32x4 v;
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
  return __builtin_bit_cast(bf16x8, v);
}

template <int MT, int NTL>
struct GemmFrags {
  bf16x8 a[2][MT], b[2][NTL];
};

// `s_waitcnt lgkmcnt(0)` that the CONSUMERS of a fragment set depend on.  The fragment reads of the DMA kernels are
// inline asm (lds_tr_read_b64 / lds_read_b128), so the compiler does not know that their results arrive
// asynchronously; a bare `asm volatile("s_waitcnt ...")` orders only memory operations, and MFMAs -- pure register
// instructions -- may be scheduled above it (seen: the three-buffer forward variant returned garbage).  Passing every
// fragment register through the asm as "+v" makes each MFMA on them data-dependent on the wait.
__device__ __forceinline__ void frags_ready(GemmFrags<4, 2>& f) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[0][2]), "+v"(f.a[0][3]), "+v"(f.a[1][0]), "+v"(f.a[1][1]),
                 "+v"(f.a[1][2]), "+v"(f.a[1][3]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1])
               :
               : "memory");
}

template <class FR, class DM, class LF, class MM>
__device__ __forceinline__ void gemm_pipeline_dma4(int T, DM&& dma, LF&& load_frags, MM&& mma) {
  FR f0, f1;
  DTF_GSTAMP(0);
  // No tile past the last one is fetched: a workgroup of the split-K forward walks 13 tiles, and three useless ones
  // were +23 % of its operand traffic (and a drain of their latency at the end).  Where nothing new is issued the
  // wait is vmcnt(0): whatever is still in flight is exactly the tile the next phase reads.
  dma(0, 0);
  if (T > 1) dma(1, 1);
  if (T > 2) {
    dma(2, 2);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tiles 0 and 1 landed (this wave's parts)
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  DTF_GSTAMP(1);
  load_frags(f0, 0);
  frags_ready(f0);   // (fragment reads issued as inline asm: see lds_tr_read_b64)
#pragma unroll 1
  for (int t = 0; t < T; t += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (t + u < T) {
        const bool more = t + u + 3 < T;
        if (more) dma(t + u + 3, (u + 3) % 4);
        if (t + u + 1 < T) load_frags((u & 1) ? f0 : f1, (u + 1) % 4);
        mma((u & 1) ? f1 : f0);
        frags_ready((u & 1) ? f0 : f1);   // the set just read is complete before anything of the next phase uses it
        if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  // (nothing is in flight here; the last barrier above also orders the LDS for an epilogue that reuses it)
}

// HEADW: the launch is a 1-D grid of the GEMM's tiles FOLLOWED by hw.nbx * KS workgroups that compute the
// output-layer weight gradient (head_wgrad_core.hpp) -- independent work that fills the CUs this GEMM
// leaves idle (fc1 dgrad: 200 tiles on 256 CUs) instead of paying for a launch of its own.
// WAVES_M x (NW / WAVES_M) waves.  The 256 x 128 tile runs 4 x 2 waves of 64 x 64: 8 fragment reads per 16 MFMAs
// instead of the 128 x 128 tile's 6 per 8, and 1.5 instead of 2 staged bytes per MAC -- the 128 x 128 loop is bound
// by the LDS (ds_write_b128 store path + fragment reads), not by the MFMA pipe.
// DMA: LDS-DMA staging through gemm_pipeline_dma4 (128 x 128 x 8 waves only) instead of the register-staged pipeline.
template <int BM, int BN, int NW, int EPI, bool HEADW = false, int WAVES_M = 2, bool DMA = false>
__global__ __launch_bounds__(NW * 64, DTF_GEMM_MINW) void gemm_nt_bf16(const bf16* A, const bf16* Bm, bf16* __restrict__ C,
                                                        const float* __restrict__ bias, int M, int N, int K,
                                                        int lda, int ldb, int ldc, DropoutArgs drop,
                                                        HeadWgradArgs hw = HeadWgradArgs{}) {
  constexpr int BK = 64, NT = NW * 64;
  constexpr int WAVES_N = NW / WAVES_M;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MT = WM / 16, NTL = WN / 16;
  constexpr int A_CH = BM * 8 / NT, B_CH = BN * 8 / NT;  // 16-byte chunks per thread
  constexpr int NBUF = DMA ? 4 : 2;
  static_assert(!DMA || (A_CH == 2 && B_CH == 2), "gemm_pipeline_dma4 counts 2 + 2 DMAs per thread and tile");
  __shared__ __attribute__((aligned(16))) struct {   // one object: the head wgrad workgroups use it as one scratch area
    bf16 a[NBUF][BM * BK];
    bf16 b[NBUF][BN * BK];
  } sm;
  auto& sA = sm.a;
  auto& sB = sm.b;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int tile_x, tile_y;
  if constexpr (HEADW) {
    static_assert(sizeof(sm) >= sizeof(float) * head_wgrad_lds_floats<NW>(), "head wgrad scratch must fit in the tile buffers");
    const int gx = (N + BN - 1) / BN, gy = (M + BM - 1) / BM;
    const int extra = (int)blockIdx.x - gx * gy;
    if (extra >= 0) {
      DTF_GSTAMP(0);
      head_wgrad_block<NW>(hw, extra % hw.nbx, extra / hw.nbx, reinterpret_cast<float*>(&sA[0][0]));
      DTF_GSTAMP(3);
      return;
    }
    xcd_tile_linear((int)blockIdx.x, gx, gy, tile_x, tile_y);
  } else {
    xcd_tile(gridDim.x, gridDim.y, tile_x, tile_y);
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int i16 = lane & 15, g = lane >> 4;

  f32x4 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t rA = make_rsrc(A, (size_t)M * lda * 2), rB = make_rsrc(Bm, (size_t)N * ldb * 2);
  uint32_t va[A_CH], vb[B_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    int c = tid + i * NT, row = c >> 3, slot = c & 7;
    va[i] = (uint32_t)(m0 + row) * (uint32_t)(lda * 2) + slot * 16;
  }
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    int c = tid + i * NT, row = c >> 3, slot = c & 7;
    vb[i] = (uint32_t)(n0 + row) * (uint32_t)(ldb * 2) + slot * 16;
  }
  auto gload = [&](u32x4 (&ra)[A_CH], u32x4 (&rb)[B_CH], int t) {
    const uint32_t so = (uint32_t)t * (BK * 2);
#pragma unroll
    for (int i = 0; i < A_CH; ++i) ra[i] = buf_load16(rA, va[i], so);
#pragma unroll
    for (int i = 0; i < B_CH; ++i) rb[i] = buf_load16(rB, vb[i], so);
  };
  auto swrite = [&](const u32x4 (&ra)[A_CH], const u32x4 (&rb)[B_CH], int buf) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      int c = tid + i * NT, row = c >> 3, slot = c & 7;
      *reinterpret_cast<u32x4*>(sA[buf] + row * BK + ((slot ^ (row & 7)) << 3)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      int c = tid + i * NT, row = c >> 3, slot = c & 7;
      *reinterpret_cast<u32x4*>(sB[buf] + row * BK + ((slot ^ (row & 7)) << 3)) = rb[i];
    }
  };
  auto compute = [&](int buf) {
    bf16x8 af[2][MT], bfv[2][NTL];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        int row = wm * WM + i * 16 + i16;
        af[kk][i] = *reinterpret_cast<const bf16x8*>(sA[buf] + row * BK + (((kk * 4 + g) ^ (row & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        int row = wn * WN + j * 16 + i16;
        bfv[kk][j] = *reinterpret_cast<const bf16x8*>(sB[buf] + row * BK + (((kk * 4 + g) ^ (row & 7)) << 3));
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = mfma16(af[kk][i], bfv[kk][j], acc[i][j]);
  };

  if constexpr (DMA) {
    // LDS chunk position (row, p) of a lane <- global chunk p ^ (row & 7) of that row: the same swizzled image
    uint32_t da[A_CH], db[B_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int c = tid + i * NT, row = c >> 3, p = c & 7;
      da[i] = (uint32_t)(m0 + row) * (uint32_t)(lda * 2) + ((p ^ (row & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int c = tid + i * NT, row = c >> 3, p = c & 7;
      db[i] = (uint32_t)(n0 + row) * (uint32_t)(ldb * 2) + ((p ^ (row & 7)) << 4);
    }
    auto dma = [&](int t, int buf) {
      const uint32_t so = (uint32_t)t * (BK * 2);
#pragma unroll
      for (int i = 0; i < A_CH; ++i)
        dma16(rA, sA[buf] + (wave * 64 + i * NT) * 8, da[i], so);
#pragma unroll
      for (int i = 0; i < B_CH; ++i)
        dma16(rB, sB[buf] + (wave * 64 + i * NT) * 8, db[i], so);
    };
    typedef GemmFrags<MT, NTL> FR;
    auto load_frags = [&](FR& f, int buf) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          int row = wm * WM + i * 16 + i16;
          f.a[kk][i] = *reinterpret_cast<const bf16x8*>(sA[buf] + row * BK + (((kk * 4 + g) ^ (row & 7)) << 3));
        }
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          int row = wn * WN + j * 16 + i16;
          f.b[kk][j] = *reinterpret_cast<const bf16x8*>(sB[buf] + row * BK + (((kk * 4 + g) ^ (row & 7)) << 3));
        }
      }
    };
    auto mma = [&](const FR& f) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NTL; ++j) acc[i][j] = mfma16(f.a[kk][i], f.b[kk][j], acc[i][j]);
    };
    gemm_pipeline_dma4<FR>(K / BK, dma, load_frags, mma);
  } else {
    gemm_pipeline<A_CH, B_CH>(K / BK, gload, swrite, compute);
  }
  DTF_GSTAMP(2);

  uint64_t drop_off = 0;
  const bool do_drop = (EPI == EPI_BIAS_RELU_DROPOUT) && drop.keep_prob < 1.0f;
  if (do_drop) drop_off = drop.state ? (uint64_t)drop.state[2] : 0;
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int n = n0 + wn * WN + j * 16 + i16;
    if (n >= N) continue;
    float bv = 0.f;
    if (EPI == EPI_BIAS_RELU_DROPOUT) bv = bias[n];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mrow = m0 + wm * WM + i * 16 + 4 * g + r;
        if (mrow >= M) continue;
        float v = acc[i][j][r];
        if (EPI == EPI_BIAS_RELU_DROPOUT) {
          v = fmaxf(v + bv, 0.f);
          if (do_drop) v *= dropout_scale(drop, drop_off, (uint64_t)mrow * (uint64_t)N + n);
        }
        if (EPI == EPI_NOSTORE_DIAG) { if (v != v) C[(size_t)mrow * ldc + n] = (bf16)v; }  // timing diagnostic only
        else C[(size_t)mrow * ldc + n] = (bf16)v;
      }
  }
  DTF_GSTAMP(3);
}

// C[M,N] (fp32) = sum_k A[k][m] * B[k][n];  A: [K][lda], B: [K][ldb]
// SGD = true: C is the fp32 MASTER WEIGHT and the product is its gradient; the epilogue applies
// w -= lr * g in place and writes the bf16 compute copy -- the fc1 weight gradient (98 % of the
// model's gradient bytes) then never goes through memory.  Single-GPU SGD only: with data
// parallelism the gradient has to exist for the all-reduce.
// G16 = true: C is a bf16 matrix (the gradient goes out in bf16: the opt-in 16-bit dense collective of
// parallel/sync_dp.py, half the bytes on the links).
// DMA (128 x 128 x 8 waves): LDS-DMA staging through gemm_pipeline_dma4; the K-slow images are then unpadded 256-byte
// rows with their 32-byte windows XORed by (row & 7) (the eight rows of a transposed read's lane group land in eight
// different windows, which is what the +32-byte pitch does for the register-staged image).
template <int BM, int BN, int NW, bool SGD = false, bool G16 = false, bool DMA = false>
__global__ __launch_bounds__(NW * 64, DTF_GEMM_MINW) void gemm_tn_bf16(const bf16* A, const bf16* Bm, float* __restrict__ C,
                                                        int M, int N, int K, int lda, int ldb, int ldc,
                                                        bf16* __restrict__ shadow = nullptr,
                                                        const float* __restrict__ hyper = nullptr) {
  constexpr int BK = 64, NT = NW * 64;
  constexpr int PA = DMA ? BM : BM + 16, PB = DMA ? BN : BN + 16;  // row pitch in elements (+32 B when register-staged)
  constexpr int NBUF = DMA ? 4 : 2;
  constexpr int WAVES_N = NW / 2;
  constexpr int WM = BM / 2, WN = BN / WAVES_N;
  constexpr int MT = WM / 16, NTL = WN / 16;
  constexpr int A_CH = BK * (BM / 8) / NT, B_CH = BK * (BN / 8) / NT;
  static_assert(!DMA || (A_CH == 2 && B_CH == 2 && BM == 128 && BN == 128), "gemm_pipeline_dma4: 2 + 2 DMAs per thread and tile");
  __shared__ __attribute__((aligned(16))) bf16 sA[NBUF][BK * PA];
  __shared__ __attribute__((aligned(16))) bf16 sB[NBUF][BK * PB];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int tile_x, tile_y;
  xcd_tile(gridDim.x, gridDim.y, tile_x, tile_y);
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int i16 = lane & 15, g = lane >> 4;
  const int qr = i16 >> 2, pc = i16 & 3;

  f32x4 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // rows past K read as zero (beyond the descriptor); columns past M/N wrap into the next row and
  // only feed output rows/columns that are never stored
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(A, (size_t)K * lda * 2), rB = make_rsrc(Bm, (size_t)K * ldb * 2);
  uint32_t va[A_CH], vb[B_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    int c = tid + i * NT, row = c / (BM / 8), slot = c % (BM / 8);
    va[i] = (uint32_t)row * (uint32_t)(lda * 2) + (uint32_t)(m0 + slot * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    int c = tid + i * NT, row = c / (BN / 8), slot = c % (BN / 8);
    vb[i] = (uint32_t)row * (uint32_t)(ldb * 2) + (uint32_t)(n0 + slot * 8) * 2;
  }
  const uint32_t stepA = (uint32_t)lda * (BK * 2), stepB = (uint32_t)ldb * (BK * 2);
  auto gload = [&](u32x4 (&ra)[A_CH], u32x4 (&rb)[B_CH], int t) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) ra[i] = buf_load16(rA, va[i], (uint32_t)t * stepA);
#pragma unroll
    for (int i = 0; i < B_CH; ++i) rb[i] = buf_load16(rB, vb[i], (uint32_t)t * stepB);
  };
  auto swrite = [&](const u32x4 (&ra)[A_CH], const u32x4 (&rb)[B_CH], int buf) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      int c = tid + i * NT, row = c / (BM / 8), slot = c % (BM / 8);
      *reinterpret_cast<u32x4*>(sA[buf] + row * PA + slot * 8) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      int c = tid + i * NT, row = c / (BN / 8), slot = c % (BN / 8);
      *reinterpret_cast<u32x4*>(sB[buf] + row * PB + slot * 8) = rb[i];
    }
  };
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto tr_frag = [&](const bf16* base, int pitch, int kk, int col0) -> bf16x8 {
    // rows kk*32 + 4g + qr (+16), columns col0 + 4*pc .. +3
    const bf16* p0 = base + (kk * 32 + 4 * g + qr) * pitch + col0 + 4 * pc;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 16 * pitch));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto compute = [&](int buf) {
    bf16x8 af[2][MT], bfv[2][NTL];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < MT; ++i) af[kk][i] = tr_frag(sA[buf], PA, kk, wm * WM + i * 16);
#pragma unroll
      for (int j = 0; j < NTL; ++j) bfv[kk][j] = tr_frag(sB[buf], PB, kk, wn * WN + j * 16);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = mfma16(af[kk][i], bfv[kk][j], acc[i][j]);
  };

  if constexpr (DMA) {
    uint32_t da[A_CH], db[B_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      // LDS chunk (row, p) of 16 per row <- global chunk p ^ ((row & 7) << 1)
      const int c = tid + i * NT, row = c >> 4, p = c & 15;
      da[i] = (uint32_t)row * (uint32_t)(lda * 2) + (uint32_t)m0 * 2 + ((p ^ ((row & 7) << 1)) << 4);
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int c = tid + i * NT, row = c >> 4, p = c & 15;
      db[i] = (uint32_t)row * (uint32_t)(ldb * 2) + (uint32_t)n0 * 2 + ((p ^ ((row & 7) << 1)) << 4);
    }
    auto dma = [&](int t, int buf) {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) dma16(rA, sA[buf] + (wave * 64 + i * NT) * 8, da[i], (uint32_t)t * stepA);
#pragma unroll
      for (int i = 0; i < B_CH; ++i) dma16(rB, sB[buf] + (wave * 64 + i * NT) * 8, db[i], (uint32_t)t * stepB);
    };
    // rows kk*32 + 4g + qr (+16): key = (4g + qr) & 7 for both; column window col0 / 16
    const int key = (4 * g + qr) & 7;
    auto tr_frag_sw = [&](const bf16* base, int kk, int col0) -> bf16x8 {
      const bf16* p0 = base + (kk * 32 + 4 * g + qr) * BM + (((col0 >> 4) ^ key) << 4) + 4 * pc;
      const dma_s16x4 lo = lds_tr_read_b64(p0), hi = lds_tr_read_b64(p0 + 16 * BM);
      s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8, v);
    };
    typedef GemmFrags<MT, NTL> FR;
    auto load_frags = [&](FR& f, int buf) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < MT; ++i) f.a[kk][i] = tr_frag_sw(sA[buf], kk, wm * WM + i * 16);
#pragma unroll
        for (int j = 0; j < NTL; ++j) f.b[kk][j] = tr_frag_sw(sB[buf], kk, wn * WN + j * 16);
      }
    };
    auto mma = [&](const FR& f) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NTL; ++j) acc[i][j] = mfma16(f.a[kk][i], f.b[kk][j], acc[i][j]);
    };
    gemm_pipeline_dma4<FR>((K + BK - 1) / BK, dma, load_frags, mma);
  } else {
    gemm_pipeline<A_CH, B_CH>((K + BK - 1) / BK, gload, swrite, compute);
  }
  DTF_GSTAMP(2);

  if constexpr (SGD) {
    // Read-modify-write of the master weight straight from the C/D layout would be 4-byte accesses
    // scattered over rows (measured: +8 us).  The gradient tile goes through LDS instead (the
    // pipeline buffers are dead: rows 0..63 into sA, rows 64..127 into sB, pitch 132 floats so the
    // four row groups of a store land on different banks) and every thread then updates float4s.
    static_assert(BM == 128 && BN == 128 && NW == 8, "SGD epilogue is written for the 128x128x8-wave tile");
    constexpr int GP = BN + 4;
    static_assert(64 * GP * 4 <= BK * PA * 2 * NBUF, "gradient half-tile must fit an operand's LDS buffers");
    // All eight master-weight float4s of this thread are requested up front, before the gradient tile is
    // even staged: C is loaded AND stored below, so the compiler keeps "load k+1" behind "store k" and the
    // update was a chain of eight dependent memory round trips (clamped addresses: the loads are unconditional).
    constexpr int NK = BM * BN / 4 / NT;
    f32x4 wreg[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int c = tid + k * NT, row = c / (BN / 4), c4 = c % (BN / 4);
      const int mrow = min(m0 + row, M - 1), n = min(n0 + c4 * 4, N - 4);
      wreg[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(C + (size_t)mrow * ldc + n));
    }
    float* gt = reinterpret_cast<float*>(wm == 0 ? &sA[0][0] : &sB[0][0]);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) gt[(i * 16 + 4 * g + r) * GP + wn * WN + j * 16 + i16] = acc[i][j][r];
    __syncthreads();
    const float lr = hyper[0] * hyper[4];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int c = tid + k * NT, row = c / (BN / 4), c4 = c % (BN / 4);
      const int mrow = m0 + row, n = n0 + c4 * 4;
      if (mrow < M && n < N) {
        const float* gsrc = reinterpret_cast<const float*>(row < 64 ? &sA[0][0] : &sB[0][0]) + (row & 63) * GP + c4 * 4;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gsrc);
        const size_t o = (size_t)mrow * ldc + n;
        f32x4 w = wreg[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = fmaf(-lr, gv[e], w[e]);
        *reinterpret_cast<f32x4*>(C + o) = w;
        bf16x4 sh = {(bf16)w[0], (bf16)w[1], (bf16)w[2], (bf16)w[3]};
        *reinterpret_cast<bf16x4*>(shadow + o) = sh;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      const int n = n0 + wn * WN + j * 16 + i16;
      if (n >= N) continue;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mrow = m0 + wm * WM + i * 16 + 4 * g + r;
          if (mrow < M) {
            if (G16) reinterpret_cast<bf16*>(C)[(size_t)mrow * ldc + n] = (bf16)acc[i][j][r];
            else C[(size_t)mrow * ldc + n] = acc[i][j][r];
          }
        }
    }
  }
  DTF_GSTAMP(3);
}

// C_part[z][M,N] (fp32) = A[M, Kz] . B[Kz, N]  -- "NN": A K-contiguous (activations), B in its
// native [K][N] layout (e.g. wd1 [in,out] exactly as TF stores it: no transposed weight copy).
// Split-K over blockIdx.z fills the machine when M*N alone gives too few tiles (fc1 forward at
// batch 1024 is only 8x8 tiles of 128^2).  A fragments: ds_read_b128 from the XOR-swizzled image of
// gemm_nt.  B fragments: ds_read_b64_tr_b16 with rows 8g+q / 8g+4+q so that element j of lane
// group g is k = 8g + j on BOTH operands; the two 128-byte halves of every row with bit 3 set are
// swapped, which makes these transposed reads bank-conflict free (rows r and r+8 would collide).
// DMA (128 x 128 x 8 waves): LDS-DMA staging through gemm_pipeline_dma4.  The B image is then UNPADDED 256-byte rows
// whose 32-byte windows are XORed with key(r) = (r & 3) | ((r >> 3) & 1) << 2: the eight rows a 32-lane group of a
// transposed read touches ({0..3} and {8..11} of a 16-row block, same column window) land in eight different windows
// (what the +32-byte row padding and the half swap achieve in the register-staged image; a DMA cannot write a padded
// image, one wave-instruction covers four whole rows).
template <int BM, int BN, int NW, bool DMA = false>
__global__ __launch_bounds__(NW * 64, DTF_GEMM_MINW) void gemm_nn_splitk_bf16(const bf16* A, const bf16* Bm,
                                                               float* __restrict__ Cp, int M, int N, int K,
                                                               int lda, int ldb, int tiles_per_split) {
  static_assert(BN == 128, "the half-swap swizzle below assumes 256-byte B rows");
  constexpr int BK = 64, NT = NW * 64;
  constexpr int PB = DMA ? BN : BN + 16;
  constexpr int NBUF = DMA ? 4 : 2;
  constexpr int WAVES_N = NW / 2;
  constexpr int WM = BM / 2, WN = BN / WAVES_N;
  constexpr int MT = WM / 16, NTL = WN / 16;
  constexpr int A_CH = BM * 8 / NT, B_CH = BK * (BN / 8) / NT;
  static_assert(!DMA || (A_CH == 2 && B_CH == 2), "gemm_pipeline_dma4 counts 2 + 2 DMAs per thread and tile");
  __shared__ __attribute__((aligned(16))) bf16 sA[NBUF][BM * BK];
  __shared__ __attribute__((aligned(16))) bf16 sB[NBUF][BK * PB];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int tile_x, tile_y;
  xcd_tile(gridDim.x, gridDim.y, tile_x, tile_y);
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int i16 = lane & 15, g = lane >> 4;
  const int qr = i16 >> 2, pc = i16 & 3;
  const int T_all = (K + BK - 1) / BK;
  const int t_begin = blockIdx.z * tiles_per_split;
  const int T = min(T_all, t_begin + tiles_per_split) - t_begin;

  f32x4 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // A columns past K wrap into the next row, but meet B rows past K, which read as zero
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(A, (size_t)M * lda * 2), rB = make_rsrc(Bm, (size_t)K * ldb * 2);
  uint32_t va[A_CH], vb[B_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    int c = tid + i * NT, row = c >> 3, slot = c & 7;
    va[i] = (uint32_t)(m0 + row) * (uint32_t)(lda * 2) + slot * 16 + (uint32_t)t_begin * (BK * 2);
  }
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    int c = tid + i * NT, row = c / (BN / 8), slot = c % (BN / 8);
    vb[i] = (uint32_t)(t_begin * BK + row) * (uint32_t)(ldb * 2) + (uint32_t)(n0 + slot * 8) * 2;
  }
  const uint32_t stepB = (uint32_t)ldb * (BK * 2);
  auto gload = [&](u32x4 (&ra)[A_CH], u32x4 (&rb)[B_CH], int t) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) ra[i] = buf_load16(rA, va[i], (uint32_t)t * (BK * 2));
#pragma unroll
    for (int i = 0; i < B_CH; ++i) rb[i] = buf_load16(rB, vb[i], (uint32_t)t * stepB);
  };
  auto swrite = [&](const u32x4 (&ra)[A_CH], const u32x4 (&rb)[B_CH], int buf) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      int c = tid + i * NT, row = c >> 3, slot = c & 7;
      *reinterpret_cast<u32x4*>(sA[buf] + row * BK + ((slot ^ (row & 7)) << 3)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      int c = tid + i * NT, row = c / (BN / 8), slot = c % (BN / 8);
      int col = (slot * 8) ^ (((row >> 3) & 1) << 6);   // swap the 64-element halves on rows with bit 3
      *reinterpret_cast<u32x4*>(sB[buf] + row * PB + col) = rb[i];
    }
  };
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto compute = [&](int buf) {
    bf16x8 af[2][MT], bfv[2][NTL];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        int row = wm * WM + i * 16 + i16;
        af[kk][i] = *reinterpret_cast<const bf16x8*>(sA[buf] + row * BK + (((kk * 4 + g) ^ (row & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        // rows kk*32 + 8g + qr and +4; bit 3 of the row is g & 1 for both
        const int r0 = kk * 32 + 8 * g + qr;
        const int col = (wn * WN + j * 16 + 4 * pc) ^ ((g & 1) << 6);
        const bf16* p0 = sB[buf] + r0 * PB + col;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * PB));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bfv[kk][j] = __builtin_bit_cast(bf16x8, v);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = mfma16(af[kk][i], bfv[kk][j], acc[i][j]);
  };

  if constexpr (DMA) {
    uint32_t da[A_CH], db[B_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int c = tid + i * NT, row = c >> 3, p = c & 7;
      da[i] = (uint32_t)(m0 + row) * (uint32_t)(lda * 2) + ((p ^ (row & 7)) << 4) + (uint32_t)t_begin * (BK * 2);
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      // LDS chunk (row, p) of 16 per row <- global chunk p ^ (key(row) << 1)
      const int c = tid + i * NT, row = c >> 4, p = c & 15;
      const int key = (row & 3) | (((row >> 3) & 1) << 2);
      db[i] = (uint32_t)(t_begin * BK + row) * (uint32_t)(ldb * 2) + (uint32_t)n0 * 2 + ((p ^ (key << 1)) << 4);
    }
    auto dma = [&](int t, int buf) {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) dma16(rA, sA[buf] + (wave * 64 + i * NT) * 8, da[i], (uint32_t)t * (BK * 2));
#pragma unroll
      for (int i = 0; i < B_CH; ++i) dma16(rB, sB[buf] + (wave * 64 + i * NT) * 8, db[i], (uint32_t)t * stepB);
    };
    typedef GemmFrags<MT, NTL> FR;
    auto load_frags = [&](FR& f, int buf) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          int row = wm * WM + i * 16 + i16;
          f.a[kk][i] = lds_read_b128(sA[buf] + row * BK + (((kk * 4 + g) ^ (row & 7)) << 3));
        }
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          // rows kk*32 + 8g + qr and +4: key = qr | (g & 1) << 2 for both; column window (wn * WN + j * 16) / 16
          const int r0 = kk * 32 + 8 * g + qr;
          const int win = ((wn * WN) / 16 + j) ^ (qr | ((g & 1) << 2));
          const bf16* p0 = sB[buf] + r0 * PB + win * 16 + 4 * pc;
          const dma_s16x4 lo = lds_tr_read_b64(p0), hi = lds_tr_read_b64(p0 + 4 * PB);
          s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          f.b[kk][j] = __builtin_bit_cast(bf16x8, v);
        }
      }
    };
    auto mma = [&](const FR& f) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NTL; ++j) acc[i][j] = mfma16(f.a[kk][i], f.b[kk][j], acc[i][j]);
    };
    if (T > 0) gemm_pipeline_dma4<FR>(T, dma, load_frags, mma);
  } else {
    if (T > 0) gemm_pipeline<A_CH, B_CH>(T, gload, swrite, compute);
  }
  DTF_GSTAMP(2);

  float* C = Cp + (size_t)blockIdx.z * M * N;
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int n = n0 + wn * WN + j * 16 + i16;
    if (n >= N) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mrow = m0 + wm * WM + i * 16 + 4 * g + r;
        if (mrow < M) C[(size_t)mrow * N + n] = acc[i][j][r];
      }
  }
  DTF_GSTAMP(3);
}

// out[M,N] bf16 = dropout(relu(sum_z part[z] + bias))  -- the fused epilogue of the split-K forward
__global__ __launch_bounds__(256) void splitk_bias_relu_dropout(const float* __restrict__ part,
                                                                const float* __restrict__ bias,
                                                                bf16* __restrict__ out, long MN, int N, int S,
                                                                DropoutArgs drop) {
  const bool do_drop = drop.keep_prob < 1.0f;
  const uint64_t drop_off = (do_drop && drop.state) ? (uint64_t)drop.state[2] : 0;
  for (long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x; i4 * 4 < MN; i4 += (long)gridDim.x * blockDim.x) {
    f32x4 a = reinterpret_cast<const f32x4*>(part)[i4];
    for (int z = 1; z < S; ++z) a += reinterpret_cast<const f32x4*>(part + (size_t)z * MN)[i4];
    const int n = (int)((i4 * 4) % N);
    f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = fmaxf(a[e] + bv[e], 0.f);
      if (do_drop) v *= dropout_scale(drop, drop_off, (uint64_t)(i4 * 4 + e));
      o[e] = (bf16)v;
    }
    reinterpret_cast<bf16x4*>(out)[i4] = o;
  }
}

// Tiled, batched bf16 transpose: out[z][c][r] = in[z][r][c]  (weight shadows in the
// K-contiguous layout the NT kernels want)
__global__ __launch_bounds__(256) void transpose_bf16(const bf16* __restrict__ in, bf16* __restrict__ out,
                                                      int R, int Cn) {
  __shared__ bf16 t[64][66];
  in += (size_t)blockIdx.z * R * Cn;
  out += (size_t)blockIdx.z * R * Cn;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    int r = r0 + i, c = c0 + tx;
    t[i][tx] = (r < R && c < Cn) ? in[(size_t)r * Cn + c] : (bf16)0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    int c = c0 + i, r = r0 + tx;
    if (r < R && c < Cn) out[(size_t)c * R + r] = t[tx][i];
  }
}

}  // namespace dtf

using namespace dtf;

#if DTF_TIMING
extern "C" int dtf_gemm_set_timing_buf(void* p) {
  long long* q = (long long*)p;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dtf::g_gemm_timing), &q, sizeof(q));
}
#endif

// Which 128 x 128 kernels stage through LDS-DMA (gemm_pipeline_dma4): bit 0 NT (fc1 dgrad), bit 1 NN (split-K forward),
// bit 2 TN (fc1 wgrad).  Stand-alone every one of them is 15-28 % faster than its register-staged twin; what counts is
// the STEP, and there the chip's clock management couples the kernels (a denser GEMM lowers the clock the conv kernels
// run at: rocprof shows them 7 % slower next to the DMA GEMMs).  Measured per mask on two devices, interleaved runs
// (tools/scratch/gemm_dma_mask_ab.sh): at 1024 images NT + TN = -3.1 / -3.4 % step time, with the forward as well +1 %;
// at 2048 images and up (backward fork/join on side streams: the GEMMs overlap other kernels, and a 128 KiB workgroup
// cannot share a CU with them) TN is +3.6 % and NT neutral.  Default: NT and TN, only while the batch dimension is at
// most 1024.  DTF_GEMM_DMA=<mask> forces a mask for every shape (A/B runs).
static int gemm_dma_forced() {
  static int mask = -2;
  if (mask == -2) {
    const char* v = getenv("DTF_GEMM_DMA");
    mask = (v && *v) ? (atoi(v) & 7) : -1;
  }
  return mask;
}
static bool gemm_dma_use(int bit, int batch_dim) {
  const int forced = gemm_dma_forced();
  if (forced >= 0) return (forced & bit) != 0;
  return (DTF_GEMM_DMA_DEFAULT & bit) != 0 && batch_dim <= 1024;
}
// The split-K forward: not by default (+1 % on the 1024-image step of the 3136 -> 1024 layer, see above), but the widened
// model's 12544 -> 4096 layer is long enough per workgroup for the deeper pipeline to pay: step 1.294 -> 1.286 ms.
static bool gemm_dma_use_nn(int M, int N, int K) {
  const int forced = gemm_dma_forced();
  if (forced >= 0) return (forced & 2) != 0;
  return M <= 1024 && (long)N * K >= (1L << 25);
}

extern "C" {

// C[M,N] bf16 = A[M,K] . B[N,K]^T ; epi 0 = store, 1 = bias+relu+dropout
int dtf_gemm_nt(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int lda,
                int ldb, int ldc, int epi, float keep_prob, uint64_t seed, const void* state, int tile,
                hipStream_t s) {
  if ((K & 63) || (lda & 7) || (ldb & 7)) return -3;  // the unguarded loader needs whole K tiles
  DropoutArgs d{keep_prob, seed, (const int64_t*)state};
  if (tile == 0) tile = ((long)((M + 127) / 128) * ((N + 127) / 128) >= 192) ? 128 : 64;
  // 256 x 128 tiles once they fill the machine 1.5 times over (measured on MI355X: M 4096, N 3136, K 1024: 42.2 vs 49.1 us;
  // M 1024, N 12544, K 4096: 134 vs 149 us; at M 1024, N 3136 their 100 tiles leave 60 % of the CUs idle: 21.9 vs 14.0 us)
  // (the LDS-DMA 128 x 128 kernel beats both register-staged tiles at these shapes -- M 1024, N 12544, K 4096: 121 vs
  //  134 / 138 us -- so the 256 x 128 tile is for the shapes that DMA staging is not used for)
  if (tile == 128 && epi == 0 && (long)((M + 255) / 256) * ((N + 127) / 128) >= 384 && !gemm_dma_use(1, M)) tile = 256;
  // 128 x 128 tiles stage through LDS-DMA (gemm_pipeline_dma4); 64 x 64 and 256 x 128 through registers
#define NT_CASE(BMN, E)                                                                               \
  if (BMN == 128 && gemm_dma_use(1, M))                                                               \
    hipLaunchKernelGGL((gemm_nt_bf16<BMN, BMN, BMN / 16, E, false, 2, BMN == 128>),                   \
                       dim3((N + BMN - 1) / BMN, (M + BMN - 1) / BMN), dim3(BMN * 4), 0, s, (const bf16*)A, \
                       (const bf16*)B, (bf16*)C, (const float*)bias, M, N, K, lda, ldb, ldc, d, HeadWgradArgs{}); \
  else NT_CASE_REG(BMN, E)
#define NT_CASE_REG(BMN, E)                                                                           \
  hipLaunchKernelGGL((gemm_nt_bf16<BMN, BMN, BMN / 16, E>),                                           \
                     dim3((N + BMN - 1) / BMN, (M + BMN - 1) / BMN), dim3(BMN * 4), 0, s, (const bf16*)A, \
                     (const bf16*)B, (bf16*)C, (const float*)bias, M, N, K, lda, ldb, ldc, d, HeadWgradArgs{})
  if (tile == 128 && epi == 0) NT_CASE(128, EPI_STORE);
  else if (tile == 129 && epi == 0) NT_CASE_REG(128, EPI_STORE);   // register-staged pipeline regardless of the default
  else if (tile == 130 && epi == 0)                                // LDS-DMA pipeline regardless of the default
    hipLaunchKernelGGL((gemm_nt_bf16<128, 128, 8, EPI_STORE, false, 2, true>), dim3((N + 127) / 128, (M + 127) / 128), dim3(512), 0,
                       s, (const bf16*)A, (const bf16*)B, (bf16*)C, (const float*)bias, M, N, K, lda, ldb, ldc, d, HeadWgradArgs{});
  else if (tile == 128 && epi == 1) NT_CASE(128, EPI_BIAS_RELU_DROPOUT);
  else if (tile == 64 && epi == 0) NT_CASE(64, EPI_STORE);
  else if (tile == 64 && epi == 1) NT_CASE(64, EPI_BIAS_RELU_DROPOUT);
  else if (tile == 256 && epi == 0)
    hipLaunchKernelGGL((gemm_nt_bf16<256, 128, 8, EPI_STORE, false, 4>), dim3((N + 127) / 128, (M + 255) / 256), dim3(512), 0, s,
                       (const bf16*)A, (const bf16*)B, (bf16*)C, (const float*)bias, M, N, K, lda, ldb, ldc, d, HeadWgradArgs{});
  else if (tile == 128 && epi == 2) NT_CASE(128, EPI_NOSTORE_DIAG);
  else if (tile == 64 && epi == 2) NT_CASE(64, EPI_NOSTORE_DIAG);
  else return -1;
#undef NT_CASE
#undef NT_CASE_REG
  return (int)hipGetLastError();
}

// fc1 dgrad (C = A . B^T, plain store) and the output-layer weight gradient in ONE launch: the GEMM's
// tiles plus ceil(H/64) * KS reduction workgroups.  Falls back to two launches when the GEMM is too small
// for the 128x128 / 512-thread configuration the fused kernel is built for, or has more than one round of tiles.
int dtf_head_wgrad(const void* h1, const void* dlogits, const void* da, void* part, int B, int H, int KS,
                   hipStream_t s);
int dtf_gemm_nt_head_wgrad(const void* A, const void* Bw, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                           const void* h1, const void* dlogits, const void* da, void* part, int Bsz, int H, int KS,
                           hipStream_t s) {
  if ((K & 63) || (lda & 7) || (ldb & 7)) return -3;
  if (KS <= 0) return -2;
  const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  // One launch only while the GEMM is a single round of tiles with CUs to spare (192..256 tiles on 256 CUs).
  // With more tiles the reduction workgroups -- dispatched after the last tile -- become a serial tail
  // (measured: -3 % at 4096 images, -5 % at 8192), and there the separate kernel packs 4 workgroups per CU.
  // Less than a round of 128 x 128 tiles (the 64 x 64 x 4-wave tile, several workgroups per CU): the same single
  // launch -- at batch 100 a launch of its own is 4.5 us of a 76 us step; measured -4..-6.5 % on the step from 32 to
  // 896 images (whether the tiles are 64 x 64 or 128 x 128 makes no difference between 576 and 960 images).
  const long tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64);
  const long guests = (long)((H + kHeadCols - 1) / kHeadCols) * KS;
  if (tiles < 192 && H % 4 == 0) {
    HeadWgradArgs hw{(const bf16*)h1, (const float*)dlogits, (const bf16*)da, (float*)part, Bsz, H, (Bsz + KS - 1) / KS,
                     (H + kHeadCols - 1) / kHeadCols};
    DropoutArgs d{1.0f, 0, nullptr};
    hipLaunchKernelGGL((gemm_nt_bf16<64, 64, 4, EPI_STORE, true>), dim3((unsigned)(tiles64 + guests)), dim3(256), 0, s,
                       (const bf16*)A, (const bf16*)Bw, (bf16*)C, (const float*)nullptr, M, N, K, lda, ldb, ldc, d, hw);
    return (int)hipGetLastError();
  }
  if (tiles < 192 || tiles > 256) {
    int rc = dtf_head_wgrad(h1, dlogits, da, part, Bsz, H, KS, s);
    if (rc) return rc;
    return dtf_gemm_nt(A, Bw, C, nullptr, M, N, K, lda, ldb, ldc, 0, 1.0f, 0, nullptr, 0, s);
  }
  if (H % 4) return -3;
  HeadWgradArgs hw{(const bf16*)h1, (const float*)dlogits, (const bf16*)da, (float*)part, Bsz, H, (Bsz + KS - 1) / KS,
                   (H + kHeadCols - 1) / kHeadCols};
  DropoutArgs d{1.0f, 0, nullptr};
  if (gemm_dma_use(1, M))
    hipLaunchKernelGGL((gemm_nt_bf16<128, 128, 8, EPI_STORE, true, 2, true>), dim3((unsigned)(tiles + (long)hw.nbx * KS)),
                       dim3(512), 0, s, (const bf16*)A, (const bf16*)Bw, (bf16*)C, (const float*)nullptr, M, N, K, lda, ldb,
                       ldc, d, hw);
  else
    hipLaunchKernelGGL((gemm_nt_bf16<128, 128, 8, EPI_STORE, true>), dim3((unsigned)(tiles + (long)hw.nbx * KS)), dim3(512),
                       0, s, (const bf16*)A, (const bf16*)Bw, (bf16*)C, (const float*)nullptr, M, N, K, lda, ldb, ldc, d, hw);
  return (int)hipGetLastError();
}

// C[M,N] fp32 = A[K,M]^T . B[K,N]
// done != NULL: the event is bound to the kernel itself (its own completion signal), which is what a
// hipEventRecord behind it would say, minus the marker packet that record puts into the stream -- measured as a
// ~7 us hole in the compute stream of the data-parallel step, whose fc1 wgrad hands the gradient to the comm stream.
int dtf_gemm_tn_ev(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                   int tile, hipEvent_t done, hipStream_t s) {
  if ((lda & 7) || (ldb & 7) || (M & 7) || (N & 7)) return -3;
  if (tile == 0) tile = ((long)((M + 127) / 128) * ((N + 127) / 128) >= 192) ? 128 : 64;
  // tile codes 129 / 130: the 128 x 128 tile with register staging / LDS-DMA staging regardless of the default (tests, A/B)
  if (tile == 128 || tile == 129 || tile == 130)
    hipExtLaunchKernelGGL(((tile == 130 || (tile == 128 && gemm_dma_use(4, K))) ? gemm_tn_bf16<128, 128, 8, false, false, true>
                                                                                : gemm_tn_bf16<128, 128, 8>),
                          dim3((N + 127) / 128, (M + 127) / 128), dim3(512), 0, s, nullptr,
                          done, 0, (const bf16*)A, (const bf16*)B, (float*)C, M, N, K, lda, ldb, ldc, (bf16*)nullptr,
                          (const float*)nullptr);
  else if (tile == 64)
    hipExtLaunchKernelGGL((gemm_tn_bf16<64, 64, 4>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, s, nullptr, done, 0,
                          (const bf16*)A, (const bf16*)B, (float*)C, M, N, K, lda, ldb, ldc, (bf16*)nullptr,
                          (const float*)nullptr);
  else
    return -1;
  return (int)hipGetLastError();
}
// The same product stored as bf16 (C: bf16 [M][ldc]); 128 x 128 tiles
int dtf_gemm_tn_g16_ev(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                       hipEvent_t done, hipStream_t s) {
  if ((lda & 7) || (ldb & 7) || (M & 7) || (N & 7)) return -3;
  hipExtLaunchKernelGGL((gemm_dma_use(4, K) ? gemm_tn_bf16<128, 128, 8, false, true, true> : gemm_tn_bf16<128, 128, 8, false, true>),
                        dim3((N + 127) / 128, (M + 127) / 128), dim3(512), 0, s,
                        nullptr, done, 0, (const bf16*)A, (const bf16*)B, (float*)C, M, N, K, lda, ldb, ldc, (bf16*)nullptr,
                        (const float*)nullptr);
  return (int)hipGetLastError();
}
int dtf_gemm_tn(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                int tile, hipStream_t s) {
  return dtf_gemm_tn_ev(A, B, C, M, N, K, lda, ldb, ldc, tile, nullptr, s);
}

// W[M,N] -= lr * (A[K,M]^T . B[K,N]), shadow = bf16(W); lr = hyper[0] * hyper[4]  (see gemm_tn_bf16<SGD>)
int dtf_gemm_tn_sgd(const void* A, const void* B, void* W, void* shadow, const void* hyper, int M, int N, int K,
                    int lda, int ldb, int ldw, hipStream_t s) {
  if ((lda & 7) || (ldb & 7) || (M & 7) || (N & 7) || (ldw & 3) || !shadow || !hyper) return -3;
  hipLaunchKernelGGL((gemm_dma_use(4, K) ? gemm_tn_bf16<128, 128, 8, true, false, true> : gemm_tn_bf16<128, 128, 8, true>),
                     dim3((N + 127) / 128, (M + 127) / 128), dim3(512), 0, s,
                     (const bf16*)A, (const bf16*)B, (float*)W, M, N, K, lda, ldb, ldw, (bf16*)shadow,
                     (const float*)hyper);
  return (int)hipGetLastError();
}

// fc1 forward: out[M,N] bf16 = dropout(relu(A[M,K] . B[K,N] + bias)); part must hold S*M*N floats.
// S = 0 picks a split that gives ~256 workgroups.
int dtf_gemm_nn_fwd(const void* A, const void* B, void* part, void* out, const void* bias, int M, int N, int K,
                    int lda, int ldb, int S, float keep_prob, uint64_t seed, const void* state, hipStream_t s) {
  if ((K & 7) || (lda & 7) || (ldb & 7) || (N & 7)) return -3;
  const int T = (K + 63) / 64;
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  if (S <= 0) {
    S = (256 + tiles - 1) / tiles;
    if (S > 8) S = 8;
  }
  if (S > T) S = T;
  const int tps = (T + S - 1) / S;
  S = (T + tps - 1) / tps;
  hipLaunchKernelGGL((gemm_dma_use_nn(M, N, K) ? gemm_nn_splitk_bf16<128, 128, 8, true> : gemm_nn_splitk_bf16<128, 128, 8>),
                     dim3((N + 127) / 128, (M + 127) / 128, S), dim3(512), 0, s,
                     (const bf16*)A, (const bf16*)B, (float*)part, M, N, K, lda, ldb, tps);
  DropoutArgs d{keep_prob, seed, (const int64_t*)state};
  const long MN = (long)M * N;
  long blocks = (MN / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_bias_relu_dropout, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)part,
                     (const float*)bias, (bf16*)out, MN, N, S, d);
  return (int)hipGetLastError();
}

// fc1 forward, GEMM part only: part[S][M][N] fp32 partial products (the bias/ReLU/dropout epilogue is
// fused into the head kernel, dtf_head_fused_fwd_bwd).  Returns the number of splits used (> 0) or a
// negative error.
int dtf_gemm_nn_partial(const void* A, const void* B, void* part, int M, int N, int K, int lda, int ldb, int S,
                        hipStream_t s) {
  if ((K & 7) || (lda & 7) || (ldb & 7) || (N & 7)) return -3;
  const int T = (K + 63) / 64;
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  if (S <= 0) {
    S = (256 + tiles - 1) / tiles;
    if (S > 8) S = 8;
  }
  if (S > T) S = T;
  const int tps = (T + S - 1) / S;
  S = (T + tps - 1) / tps;
  hipLaunchKernelGGL((gemm_dma_use_nn(M, N, K) ? gemm_nn_splitk_bf16<128, 128, 8, true> : gemm_nn_splitk_bf16<128, 128, 8>),
                     dim3((N + 127) / 128, (M + 127) / 128, S), dim3(512), 0, s,
                     (const bf16*)A, (const bf16*)B, (float*)part, M, N, K, lda, ldb, tps);
  return hipGetLastError() == hipSuccess ? S : -1;
}

// max number of K splits dtf_gemm_nn_fwd may use (sizes the partial buffer)
int dtf_gemm_nn_max_splits() { return 8; }

int dtf_transpose_bf16(const void* in, void* out, int batch, int R, int C, hipStream_t s) {
  hipLaunchKernelGGL(transpose_bf16, dim3((C + 63) / 64, (R + 63) / 64, batch), dim3(256), 0, s,
                     (const bf16*)in, (bf16*)out, R, C);
  return (int)hipGetLastError();
}

}  // extern "C"
