// Second conv block in exact fp32 (--dtype fp32): 5x5 SAME stride-1 convolution over 14x14 maps as an
// implicit GEMM on the f32-input matrix cores, NHWC fp32 activations, fp32 HWIO weights
// (reference .idea/MNISTDist.py:52-62,76-78 and the Conv2DBackprop{Input,Filter} / MaxPoolGrad / ReluGrad /
// BiasAddGrad ops autodiff generates for them).  Same decomposition ideas as the bf16 kernels of conv2.hip,
// re-tiled for the fp32 instruction shapes and for a kernel whose bound is the (16x slower) matrix pipe:
//
// conv5x5_igemm_f32 (forward and dgrad), v_mfma_f32_16x16x4_f32, ONE image per 4-wave workgroup:
//   M = the image's 49 pooling windows x 4 positions, rows ordered m = 4*window + q, so the four accumulator
//   registers of a lane ARE one pooling window (bias + ReLU + max-pool + argmax in registers) -- 12 M tiles of
//   4 windows and the 49th window on the 4x4x1 instruction (below): no padding.
//   The zero-haloed 18x18 input tile sits in LDS as [pos][32 ch + 4 pad] floats; the A fragment of tap (ky,kx),
//   k block kb is one ds_read_b128 of four consecutive channels at a shifted position -- no im2col.
//   Weights come from a fragment-ordered fp32 pack (pack_conv2_weights_f32): the B fragment of (tap, 16-k block,
//   16-wide n tile) is one coalesced 1 KiB read, prefetched one step ahead.  A wave owns one 16-channel n tile
//   and all M tiles: 48 + 4 MFMAs (1568 cycles) per k block against 13 LDS reads and one 16-byte load.
//   Forward: 64 output channels = 4 waves.  Dgrad (N = 32 input channels): 2 n tiles x 2 K halves (the waves of a
//   pair take alternate k blocks and are summed through LDS once); the tile is staged through the max-pool /
//   ReLU gradient (pooled gradient scattered to the saved argmax position) and taps are read flipped.
//
// conv5x5_wgrad_f32, v_mfma_f32_32x32x2_f32: dW[tap][ci][co] = sum_pos X[pos + tap][ci] * dY[pos][co]
//   M = 32 input channels, N = 32-wide halves of 64 output channels, K = positions (two per MFMA).  A workgroup
//   owns ALL 25 taps x 32 ci x 64 co of its image range: 50 (tap, co half) accumulator tiles over 4 waves
//   (13/13/12/12), 208 accumulator registers per wave, one wave per SIMD.  Per position pair a wave issues one
//   B read (dY, shared by its 13 taps) and 13 A reads (X at 13 shifted positions), all conflict-free
//   ds_read_b32, for 13 MFMAs (832 cycles).  The next image's global loads are in flight during the k-loop.
//   Split-K over images into per-workgroup slabs, summed in fixed order by reduce_segments / the fused tail.
#include <cstdlib>
#include <type_traits>

#include "f32_common.hpp"

namespace dtf {

constexpr int kF32TileW = 18;
constexpr int kF32TilePos = 18 * 18;
constexpr int kF32Pitch = 36;   // floats per position in the igemm tile (32 channels + 4: spreads b128 reads over banks)
constexpr int kF32TileBytes = kF32TilePos * kF32Pitch * 4;   // 46656

enum { F32_MODE_FWD = 0, F32_MODE_DGRAD = 1 };

// -DDTF_F32_TIMING=1 (tools/conv_f32_timeline.py): wave 0 of every workgroup stamps the 100 MHz wall clock into
// g_f32_timing[workgroup][16]: 0 start; per image n: 1+4n at the staging barrier, 2+4n tile staged, 3+4n left the k-loop,
// 4+4n stores issued.  The release build has no trace of it.
#if DTF_F32_TIMING
__device__ long long* g_f32_timing;
#define F32_STAMP(slot)                                                                                            \
  do {                                                                                                             \
    if (g_f32_timing && threadIdx.x == 0 && (slot) < 16)                                                           \
      g_f32_timing[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define F32_STAMP(slot) do { } while (0)
#endif

// ----------------------------------------------------------------------------------------------------
//   in   FWD  : x     [B,14,14,KTOT] fp32          DGRAD: gpool [B,7,7,KTOT] fp32 + in_arg [B,7,7,KTOT] u8
//   wk   fragment-ordered pack for this direction (see pack_conv2_weights_f32)
//   out  FWD  : pooled [B,7,7,NCTOT] fp32 + out_arg u8      DGRAD: dx [B,14,14,NCTOT] fp32
// grid = (B, NCTOT / NCW, 12 / MT): MT = full (4-window) M tiles per workgroup.  12 is the whole image; a small batch
// (fewer workgroups than CUs: the reference's batch 100) cuts the image into 2 or 3 M ranges (MT = 6 / 4), each
// workgroup staging the same input tile and running its share of the MFMAs -- the kernel is bound by the per-SIMD
// matrix pipe, so what matters is how many SIMDs work, not how much is staged twice.
//
// The 49th window: a 13th 16-row tile would hold ONE real window, 4 rows of 16.  With every lane's row clamped to window
// 48 -- position q = l16 & 3, k = 4g + e -- that tile's A fragment is exactly the A operand of a 4x4x1 MFMA whose
// sixteen blocks (block = lane >> 2 = 4g + n/4) are [4 channels n/4] x [k group g], and the B fragment (k = 4g + e,
// column n) fits the same way.  So the window runs as v_mfma_f32_4x4x1 on the registers a 16x16x4 would have used, in
// a quarter of its matrix-pipe time: the accumulator of lane (g, n) is the partial sum over ITS k group, rows = the
// four positions, and the four groups are added by two cross-lane steps per image.  An image costs 12.25 tiles, not 13
// (-11 us per kernel at 1024 images).  Every M range runs it (branch-free k-loop); the last range stores it.
// ----------------------------------------------------------------------------------------------------
constexpr int kF32RedBytes = 2 * 13 * 64 * 16;   // dgrad: the K-half partner's 13 accumulator tiles per n tile

template <int KTOT, int NCTOT, int MODE, int MT>
__global__ __launch_bounds__(256, 2) void conv5x5_igemm_f32(const float* __restrict__ in, const uint8_t* __restrict__ in_arg,
                                                         const float* __restrict__ wk, const float* __restrict__ bias,
                                                         float* __restrict__ out, uint8_t* __restrict__ out_arg, int B) {
  constexpr int KC = 32;
  constexpr int NCW = (MODE == F32_MODE_FWD) ? 64 : 32;   // output channels per workgroup
  constexpr int NTN = NCW / 16;                           // n tiles (one per wave column)
  constexpr int KSPLIT = 4 / NTN;                         // K halves (dgrad)
  constexpr int NS = 25 * (KC / 16) / KSPLIT;             // k-block steps per wave and K chunk
  static_assert(KTOT % KC == 0 && NCTOT % NCW == 0, "channel counts must be multiples of the tile");
  static_assert(12 % MT == 0, "the 12 full M tiles are cut into equal ranges");
  constexpr int MTW = MT;   // index of the window-48 fragment / accumulator behind the MT full tiles
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [324][36] (+ the dgrad reduction area)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = wave % NTN, ksp = wave / NTN;
  const int ng = blockIdx.y;
  const int t0 = blockIdx.z * MT;   // first M tile of this workgroup
  const int l16 = lane & 15, g = lane >> 4;

  // A row of this lane in M tile t: window 4t + (l16 >> 2), position q = l16 & 3; entry MT: window 48
  int aoff[MT + 1];
#pragma unroll
  for (int t = 0; t <= MT; ++t) {
    const int wI = t < MT ? 4 * (t0 + t) + (l16 >> 2) : 48, q = l16 & 3;
    const int pos = (2 * (wI / 7) + (q >> 1)) * kF32TileW + 2 * (wI % 7) + (q & 1);
    aoff[t] = pos * kF32Pitch + 4 * g;
  }

  // zero once: the halo stays zero, the interior is completely rewritten by every (image, K chunk)
  for (int i = tid; i < kF32TileBytes / 16; i += 256) reinterpret_cast<f32x4*>(tile)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- staging: (image, 32-channel K chunk) units; the global loads of unit u+1 are issued before the k-loop of
  // unit u and written to LDS after it (persistent workgroups: the grid is ~2 per CU and walks the batch) ----
  constexpr int NCH = (MODE == F32_MODE_FWD) ? 196 * 8 : 49 * 8;   // 16-byte chunks per unit
  constexpr int ROUNDS = (NCH + 255) / 256;
  f32x4 sv[ROUNDS];
  unsigned scd[ROUNDS];
  auto issue_loads = [&](int b, int kc0) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int c = min(tid + r * 256, NCH - 1);
      const size_t off = ((size_t)b * (MODE == F32_MODE_FWD ? 196 : 49) + (c >> 3)) * KTOT + kc0 + (c & 7) * 4;
      sv[r] = *reinterpret_cast<const f32x4*>(in + off);
      if (MODE == F32_MODE_DGRAD) scd[r] = *reinterpret_cast<const unsigned*>(in_arg + off);
    }
  };
  auto write_tile = [&]() {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int c = tid + r * 256;
      const int pos = c >> 3, ch = c & 7;
      if (c < NCH) {
        if (MODE == F32_MODE_FWD) {
          const int tp = (pos / 14 + 2) * kF32TileW + pos % 14 + 2;
          *reinterpret_cast<f32x4*>(tile + tp * kF32Pitch + ch * 4) = sv[r];
        } else {
          // max-pool / ReLU gradient: the pooled gradient goes to the saved argmax position of its window
          f32x4 o[4];
          expand_pool_grad_f32(sv[r], scd[r], o);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int tp = (2 * (pos / 7) + (q >> 1) + 2) * kF32TileW + 2 * (pos % 7) + (q & 1) + 2;
            *reinterpret_cast<f32x4*>(tile + tp * kF32Pitch + ch * 4) = o[q];
          }
        }
      }
    }
  };

  int b = blockIdx.x;
  F32_STAMP(0);
  if (b < B) issue_loads(b, 0);
  [[maybe_unused]] int img_n = 0;
#pragma unroll 1
  for (; b < B; b += gridDim.x) {
    f32x4 acc[MT + 1];
#pragma unroll
    for (int t = 0; t <= MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int kc0 = 0; kc0 < KTOT; kc0 += KC) {
      F32_STAMP(1 + 4 * img_n);
      __syncthreads();   // zero fill done / previous unit's k-loop (and reduction reads) finished
      write_tile();
      __syncthreads();
      F32_STAMP(2 + 4 * img_n);
      {   // next unit's loads fly during this unit's k-loop
        const bool more_k = kc0 + KC < KTOT;
        const int nb = more_k ? b : b + (int)gridDim.x;
        if (nb < B) issue_loads(nb, more_k ? kc0 + KC : 0);
      }
      __builtin_amdgcn_sched_barrier(0);

      // ---- k-loop: step s of this wave <-> (tap, 16-k block kb) ----
      const float* bbase = wk + ((size_t)(kc0 / 16) * (NCTOT / 16) + ng * NTN + ntile) * 256 + lane * 4;
      auto step_tap = [&](int s) { return KSPLIT == 1 ? (s >> 1) : s; };
      auto step_kb = [&](int s) { return KSPLIT == 1 ? (s & 1) : ksp; };
      auto load_b = [&](int s) -> f32x4 {
        const int tap = step_tap(s), kb = step_kb(s);
        const int wtap = (MODE == F32_MODE_DGRAD) ? 24 - tap : tap;
        return *reinterpret_cast<const f32x4*>(bbase + ((size_t)wtap * (KTOT / 16) + kb) * (NCTOT / 16) * 256);
      };
      auto do_step = [&](int s, const f32x4& bf) {
        const int tap = step_tap(s), kb = step_kb(s);
        const float* ap = tile + ((tap / 5) * kF32TileW + tap % 5) * kF32Pitch + kb * 16;
        f32x4 a[MT + 1];
#pragma unroll
        for (int t = 0; t <= MT; ++t) a[t] = *reinterpret_cast<const f32x4*>(ap + aoff[t]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t] = mfma16f(a[t][e], bf[e], acc[t]);
          acc[MTW] = mfma4f(a[MTW][e], bf[e], acc[MTW]);   // window 48
        }
      };
      // B fragment of step s+1 requested before the MFMAs of step s; unconditional (the last step re-loads itself)
      f32x4 b0 = load_b(0), b1;
#pragma unroll 1
      for (int s = 0; s < NS; s += 2) {
        b1 = load_b(min(s + 1, NS - 1));
        __builtin_amdgcn_sched_barrier(0);
        do_step(s, b0);
        if (s + 1 >= NS) break;
        b0 = load_b(min(s + 2, NS - 1));
        __builtin_amdgcn_sched_barrier(0);
        do_step(s + 1, b1);
      }
    }

    F32_STAMP(3 + 4 * img_n);
    // ---- K halves summed through LDS (dgrad); its own area behind the tile, so the tile's zero halo survives ----
    if (KSPLIT == 2) {
      f32x4* red = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(tile) + kF32TileBytes);   // [ntile][MT + 1][64 lanes]
      if (ksp == 1) {
#pragma unroll
        for (int t = 0; t <= MT; ++t) red[(ntile * (MT + 1) + t) * 64 + lane] = acc[t];
      }
      __syncthreads();
      // every wave adds (branch-free: a conditional update of the accumulators costs a second register copy of
      // them); only the ksp == 0 wave of a pair stores its result
#pragma unroll
      for (int t = 0; t <= MT; ++t) acc[t] += red[(ntile * (MT + 1) + t) * 64 + lane];
    }

    // window 48: sum of the four k groups (lanes n, n + 16, n + 32, n + 48); every lane gets the total
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[MTW][r];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      acc[MTW][r] = v;
    }
    const bool own_w48 = blockIdx.z == gridDim.z - 1;

    // ---- epilogue.  C/D map: lane holds rows 4g + r -> window 4t + g, q = r; column l16 ----
    // (the lane's row group is made opaque per image: the store addresses are then recomputed here instead of being
    // hoisted out of the image loop as ~50 loop-invariant registers that spill around the k-loop)
    int ge = g;
    asm volatile("" : "+v"(ge));
    const int n = ng * NCW + ntile * 16 + l16;
    if (MODE == F32_MODE_FWD) {
      const float bv = bias[n];
#pragma unroll
      for (int t = 0; t <= MT; ++t) {
        const int wI = t < MT ? 4 * (t0 + t) + ge : 48;
        if (t < MT || (own_w48 && ge == 0)) {
          const f32x4 v = acc[t];
          float best = v[0];
          int bi = 0;
          if (v[1] > best) { best = v[1]; bi = 1; }
          if (v[2] > best) { best = v[2]; bi = 2; }
          if (v[3] > best) { best = v[3]; bi = 3; }
          const float o = fmaxf(best + bv, 0.f);
          const size_t off = ((size_t)b * 49 + wI) * NCTOT + n;
          out[off] = o;
          out_arg[off] = (uint8_t)(o > 0.f ? bi : (int)kPoolDead);
        }
      }
    } else if (ksp == 0) {
#pragma unroll
      for (int t = 0; t <= MT; ++t) {
        const int wI = t < MT ? 4 * (t0 + t) + ge : 48;
        if (t < MT || (own_w48 && ge == 0)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int y = 2 * (wI / 7) + (r >> 1), x = 2 * (wI % 7) + (r & 1);
            out[((size_t)b * 196 + y * 14 + x) * NCTOT + n] = acc[t][r];
          }
        }
      }
    }
    F32_STAMP(4 + 4 * img_n);
    ++img_n;
  }
}

// ----------------------------------------------------------------------------------------------------
// Fragment-ordered fp32 packs of w = HWIO [25][cin][cout] for conv5x5_igemm_f32: one contiguous 1 KiB block per
// (tap, 16-wide k block, 16-wide n tile); lane l = 16*g + n holds its four k = 4g .. 4g+3 consecutively.
//   fwd   (k = ci, n = co): F[((tap*(cin/16)  + ci/16)*(cout/16) + co/16)*256 + (((ci%16)/4)*16 + co%16)*4 + ci%4]
//   dgrad (k = co, n = ci): D[((tap*(cout/16) + co/16)*(cin/16)  + ci/16)*256 + (((co%16)/4)*16 + ci%16)*4 + co%4]
// packed = [F | D], 25*cin*cout floats each.
// ----------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ long conv2_f32_pack_fwd_index(int tap, int ci, int co, int cin, int cout) {
  return (((long)tap * (cin / 16) + ci / 16) * (cout / 16) + co / 16) * 256 + (((ci % 16) / 4) * 16 + co % 16) * 4 + ci % 4;
}
__host__ __device__ __forceinline__ long conv2_f32_pack_dgrad_index(int tap, int ci, int co, int cin, int cout) {
  return (((long)tap * (cout / 16) + co / 16) * (cin / 16) + ci / 16) * 256 + (((co % 16) / 4) * 16 + ci % 16) * 4 + co % 4;
}

__global__ __launch_bounds__(256) void pack_conv2_weights_f32(const float* __restrict__ w, float* __restrict__ packed,
                                                              int cin, int cout) {
  const long n = 25L * cin * cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i / (cin * cout)), rem = (int)(i % (cin * cout));
    const int ci = rem / cout, co = rem % cout;
    const float v = w[i];
    packed[conv2_f32_pack_fwd_index(tap, ci, co, cin, cout)] = v;
    packed[n + conv2_f32_pack_dgrad_index(tap, ci, co, cin, cout)] = v;
  }
}

// ----------------------------------------------------------------------------------------------------
// wgrad.  grid = (G image ranges, (CIN/32) * (COUT/64) channel tiles, ceil(13 / NJ) tap ranges)
//   NJ = taps per wave: 13 (all of the wave's taps) or 7 -- the small-batch launch, where G alone leaves CUs idle:
//   two workgroups stage the same image range and each runs half of the taps into the same slab.
//   x [B,14,14,CIN] fp32, gpool [B,7,7,COUT] fp32, arg [B,7,7,COUT] u8
//   part [G][25*CIN*COUT + COUT] fp32 slabs (HWIO order, then the bias gradient)
// ----------------------------------------------------------------------------------------------------
constexpr int kWgfXFloats = kF32TilePos * 32;   // zero-haloed input tile [324][32 ci]
constexpr int kWgfDyFloats = 196 * 64;          // max-pool/ReLU-expanded output gradient [196][64 co]
constexpr int kWgfBytes = (kWgfXFloats + kWgfDyFloats) * 4;   // 91648

template <int CIN, int COUT, int NJ>
__global__ __launch_bounds__(256, 1) void conv5x5_wgrad_f32(const float* __restrict__ x, const float* __restrict__ gpool,
                                                            const uint8_t* __restrict__ arg, float* __restrict__ part,
                                                            int B, int imgs_per_block) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  float* xt = wsm;
  float* dyt = wsm + kWgfXFloats;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cohalf = wave & 1, tpar = wave >> 1;   // this wave's taps: tpar, tpar + 2, ...
  const int cic = blockIdx.y % (CIN / 32), coc = blockIdx.y / (CIN / 32);
  const int i32 = lane & 31, h = lane >> 5;
  const int b_begin = blockIdx.x * imgs_per_block, b_end = min(B, b_begin + imgs_per_block);

  f32x16 acc[NJ];
#pragma unroll
  for (int t = 0; t < NJ; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};   // bias gradient of channels 4*(tid & 15) .. +3 over this thread's staging items

  for (int i = tid; i < kWgfXFloats / 4; i += 256) reinterpret_cast<f32x4*>(xt)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int XCH = 196 * 8, XR = (XCH + 255) / 256;     // 16-byte chunks of the input tile / rounds per thread
  constexpr int DCH = 49 * 16, DR = (DCH + 255) / 256;     // (pooled position, 4-channel chunk) items
  f32x4 xr[XR], gr[DR];
  unsigned cr[DR];
  // raw buffer loads: one 32-bit per-thread offset per round (computed once) + a scalar per-image offset, instead of
  // a 64-bit address pair per load -- this kernel has 208 accumulator registers per lane to keep out of scratch
  const __amdgpu_buffer_rsrc_t rX = make_rsrc(x, (size_t)B * 196 * CIN * 4);
  const __amdgpu_buffer_rsrc_t rG = make_rsrc(gpool, (size_t)B * 49 * COUT * 4);
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(arg, (size_t)B * 49 * COUT);
  uint32_t vx[XR], vd[DR];
#pragma unroll
  for (int r = 0; r < XR; ++r) {
    const int c = min(tid + r * 256, XCH - 1);
    vx[r] = (uint32_t)((c >> 3) * CIN + cic * 32 + (c & 7) * 4) * 4u;
  }
#pragma unroll
  for (int r = 0; r < DR; ++r) {
    const int c = min(tid + r * 256, DCH - 1);
    vd[r] = (uint32_t)((c >> 4) * COUT + coc * 64 + (c & 15) * 4);
  }
  auto prefetch = [&](int b) {
#pragma unroll
    for (int r = 0; r < XR; ++r) xr[r] = buf_load16f(rX, vx[r], (uint32_t)b * (196u * CIN * 4u));
#pragma unroll
    for (int r = 0; r < DR; ++r) {
      gr[r] = buf_load16f(rG, vd[r] * 4u, (uint32_t)b * (49u * COUT * 4u));
      cr[r] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rA, (int)vd[r], (int)((uint32_t)b * (49u * COUT)), 0);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int r = 0; r < XR; ++r) {
      const int c = tid + r * 256;
      const int pos = c >> 3, ch = c & 7;
      const int tp = (pos / 14 + 2) * kF32TileW + pos % 14 + 2;
      if (c < XCH) *reinterpret_cast<f32x4*>(xt + tp * 32 + ch * 4) = xr[r];
    }
#pragma unroll
    for (int r = 0; r < DR; ++r) {
      const int c = tid + r * 256;
      const int pp = c >> 4, c4 = c & 15;
      if (c < DCH) {
        f32x4 o[4];
        expand_pool_grad_f32(gr[r], cr[r], o);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int pos = (2 * (pp / 7) + (q >> 1)) * 14 + 2 * (pp % 7) + (q & 1);
          *reinterpret_cast<f32x4*>(dyt + pos * 64 + c4 * 4) = o[q];
        }
        bsum += (o[0] + o[1]) + (o[2] + o[3]);
      }
    }
  };
  // one position pair (two k) per MFMA: lane half h takes position 2*xj + h of row y.  Two fragment sets: the 14
  // LDS reads of pair xj+1 are issued before the 13 MFMAs of pair xj (one wave per SIMD: nothing else hides them).
  auto kloop = [&](auto TP) {
    constexpr int tp_ = decltype(TP)::value;   // first tap of this wave in this workgroup; the others follow in twos
    auto load = [&](const float* pa, const float* pb, int xj, float (&av)[NJ], float& bv) {
      bv = pb[2 * xj * 64];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int tap = (tp_ + 2 * j) < 25 ? (tp_ + 2 * j) : 23;   // taps past the 25th are discarded repeats
        av[j] = pa[(2 * xj + (tap / 5) * kF32TileW + tap % 5) * 32];
      }
    };
    auto mma = [&](const float (&av)[NJ], float bv) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] = mfma32f(av[j], bv, acc[j]);
    };
#pragma unroll 1
    for (int y = 0; y < 14; ++y) {
      const float* pa = xt + (y * kF32TileW + h) * 32 + i32;
      const float* pb = dyt + (y * 14 + h) * 64 + cohalf * 32 + i32;
      float a0[NJ], a1[NJ], b0, b1;
      load(pa, pb, 0, a0, b0);
#pragma unroll
      for (int xj = 0; xj < 6; xj += 2) {
        load(pa, pb, xj + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        load(pa, pb, xj + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      mma(a0, b0);
    }
  };

  // The whole image loop and the slab write are instantiated per tap parity (wave-uniform branch taken once): with
  // the branch around the k-loop only, the accumulators of the two instantiations met in phi nodes that the
  // register allocator kept in VGPRs, i.e. 208 v_accvgpr moves each way per image and spills in the staging code.
  float* slab = part + (size_t)blockIdx.x * (25 * CIN * COUT + COUT);
  auto body = [&](auto TP) {
    constexpr int tp_ = decltype(TP)::value;
    if (b_begin < b_end) prefetch(b_begin);
    for (int b = b_begin; b < b_end; ++b) {
      __syncthreads();   // zero fill done / previous image's k-loop finished
      stage();
      __syncthreads();
      if (b + 1 < b_end) prefetch(b + 1);   // in flight during this image's k-loop
      __builtin_amdgcn_sched_barrier(0);
      kloop(TP);
    }
    // ---- slab: dW rows ci = (r & 3) + 8 * (r >> 2) + 4 * h, column co = i32 of this wave's half ----
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int tap = tp_ + 2 * j;
      if (tap < 25) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = cic * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          slab[((size_t)tap * CIN + ci) * COUT + coc * 64 + cohalf * 32 + i32] = acc[j][r];
        }
      }
    }
  };
  // (first tap = tap parity + 2 * NJ * tap range: a compile-time constant per instantiation, so the tap offsets stay
  // immediates of the LDS reads)
  if (NJ == 13 || blockIdx.z == 0) {
    if (tpar == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
  } else {
    if (tpar == 0) body(std::integral_constant<int, 2 * (NJ % 13)>{});
    else body(std::integral_constant<int, 2 * (NJ % 13) + 1>{});
  }
  // bias gradient: thread t holds channels 4 * (t & 15) .. +3; fixed-order sum of the 16 threads per chunk
  __syncthreads();
  f32x4* red = reinterpret_cast<f32x4*>(wsm);
  red[tid] = bsum;
  __syncthreads();
  if (cic == 0 && blockIdx.z == 0 && tid < 64) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k * 16 + (tid >> 2)][tid & 3];
    slab[25 * CIN * COUT + coc * 64 + tid] = s;
  }
}

}  // namespace dtf

using namespace dtf;

static int f32_cu_count() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

template <int KTOT, int NCTOT, int MODE, int MT>
static int launch_igemm_f32_mt(const void* in, const void* in_arg, const void* wk, const void* bias, void* out,
                               void* out_arg, int B, hipStream_t s) {
  auto kern = conv5x5_igemm_f32<KTOT, NCTOT, MODE, MT>;
  // 80 KiB per workgroup = exactly two per CU (the dgrad reduction area fits in the slack): the grid is persistent,
  // two workgroups per CU walking the batch, so there is no partial last round of workgroups
  constexpr int lds = 80 * 1024;
  static_assert(kF32TileBytes + kF32RedBytes <= lds, "tile + reduction area must fit");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  constexpr int NCW = (MODE == F32_MODE_FWD) ? 64 : 32;
  constexpr int groups = NCTOT / NCW;
  int gx = (2 * f32_cu_count() + groups - 1) / groups;
#if DTF_F32_TIMING
  if (const char* v = getenv("DTF_F32_WG_PER_CU")) gx = (atoi(v) * f32_cu_count() + groups - 1) / groups;   // timeline experiments
#endif
  if (gx > B) gx = B;
  hipLaunchKernelGGL(kern, dim3(gx, groups, 12 / MT), dim3(256), lds, s, (const float*)in,
                     (const uint8_t*)in_arg, (const float*)wk, (const float*)bias, (float*)out, (uint8_t*)out_arg, B);
  return (int)hipGetLastError();
}

// M ranges per image: as many (up to 3) as keep the launch at one workgroup per CU.  More would put two workgroups'
// waves on the same SIMDs, whose matrix pipe is what bounds this kernel.  DTF_F32_MSPLIT=1|2|3 overrides (tools).
static int f32_msplit(int wgs) {
  static int forced = -1;
  if (forced < 0) {
    const char* v = getenv("DTF_F32_MSPLIT");
    forced = v ? atoi(v) : 0;
  }
  if (forced >= 1 && forced <= 3) return forced;
  const int cus = f32_cu_count();
  return 3 * wgs <= cus ? 3 : (2 * wgs <= cus ? 2 : 1);
}

template <int KTOT, int NCTOT, int MODE>
static int launch_igemm_f32(const void* in, const void* in_arg, const void* wk, const void* bias, void* out, void* out_arg,
                            int B, hipStream_t s) {
  constexpr int groups = NCTOT / ((MODE == F32_MODE_FWD) ? 64 : 32);
  switch (f32_msplit(B * groups)) {
    case 3: return launch_igemm_f32_mt<KTOT, NCTOT, MODE, 4>(in, in_arg, wk, bias, out, out_arg, B, s);
    case 2: return launch_igemm_f32_mt<KTOT, NCTOT, MODE, 6>(in, in_arg, wk, bias, out, out_arg, B, s);
    default: return launch_igemm_f32_mt<KTOT, NCTOT, MODE, 12>(in, in_arg, wk, bias, out, out_arg, B, s);
  }
}

template <int CIN, int COUT, int NJ>
static int launch_wgrad_f32_nj(const void* x, const void* gpool, const void* arg, void* part, int B, int G, hipStream_t s) {
  auto kern = conv5x5_wgrad_f32<CIN, COUT, NJ>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kWgfBytes);
    attr_set = true;
  }
  const int ipb = (B + G - 1) / G;
  hipLaunchKernelGGL(kern, dim3(G, (CIN / 32) * (COUT / 64), (13 + NJ - 1) / NJ), dim3(256), kWgfBytes, s, (const float*)x,
                     (const float*)gpool, (const uint8_t*)arg, (float*)part, B, ipb);
  return (int)hipGetLastError();
}

template <int CIN, int COUT>
static int launch_wgrad_f32(const void* x, const void* gpool, const void* arg, void* part, int B, int G, hipStream_t s) {
  // two tap ranges when that still leaves one workgroup per CU (see f32_msplit)
  if (f32_msplit(G * (CIN / 32) * (COUT / 64)) >= 2) return launch_wgrad_f32_nj<CIN, COUT, 7>(x, gpool, arg, part, B, G, s);
  return launch_wgrad_f32_nj<CIN, COUT, 13>(x, gpool, arg, part, B, G, s);
}

extern "C" {

#if DTF_F32_TIMING
int dtf_conv2_f32_set_timing_buf(void* p) {
  long long* q = (long long*)p;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dtf::g_f32_timing), &q, sizeof(q));
}
#endif

// packed (2 * 25*cin*cout fp32) = [fwd pack | dgrad pack] of the fp32 HWIO weights w
int dtf_conv2_f32_pack_weights(const void* w, void* packed, int cin, int cout, hipStream_t s) {
  if ((cin % 16) || (cout % 16)) return -3;
  hipLaunchKernelGGL(pack_conv2_weights_f32, dim3((25 * cin * cout + 255) / 256), dim3(256), 0, s, (const float*)w,
                     (float*)packed, cin, cout);
  return (int)hipGetLastError();
}

// x [B,14,14,cin] -> pooled [B,7,7,cout] fp32 + arg u8; wpack = fwd pack
int dtf_conv2_f32_fwd(const void* x, const void* wpack, const void* bias, void* pooled, void* arg, int B, int cin, int cout,
                      hipStream_t s) {
  if (B <= 0) return -2;
  if (cin == 32 && cout == 64) return launch_igemm_f32<32, 64, F32_MODE_FWD>(x, nullptr, wpack, bias, pooled, arg, B, s);
  if (cin == 128 && cout == 256) return launch_igemm_f32<128, 256, F32_MODE_FWD>(x, nullptr, wpack, bias, pooled, arg, B, s);
  return -1;
}

// gpool [B,7,7,cout] (+ arg code) -> dx [B,14,14,cin]; wpack = dgrad pack (second half of dtf_conv2_f32_pack_weights)
int dtf_conv2_f32_dgrad(const void* gpool, const void* arg, const void* wpack, void* dx, int B, int cin, int cout,
                        hipStream_t s) {
  if (B <= 0) return -2;
  if (cin == 32 && cout == 64) return launch_igemm_f32<64, 32, F32_MODE_DGRAD>(gpool, arg, wpack, nullptr, dx, nullptr, B, s);
  if (cin == 128 && cout == 256) return launch_igemm_f32<256, 128, F32_MODE_DGRAD>(gpool, arg, wpack, nullptr, dx, nullptr, B, s);
  return -1;
}

// part [G][25*cin*cout + cout] fp32 slabs
int dtf_conv2_f32_wgrad(const void* x, const void* gpool, const void* arg, void* part, int B, int cin, int cout, int G,
                        hipStream_t s) {
  if (G <= 0 || B <= 0) return -2;
  if (cin == 32 && cout == 64) return launch_wgrad_f32<32, 64>(x, gpool, arg, part, B, G, s);
  if (cin == 128 && cout == 256) return launch_wgrad_f32<128, 256>(x, gpool, arg, part, B, G, s);
  return -1;
}

}  // extern "C"
