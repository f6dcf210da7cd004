// 5x5 SAME stride-1 convolution over 14x14 feature maps as an implicit GEMM on
// the CDNA4 bf16 matrix cores (v_mfma_f32_16x16x32_bf16), NHWC activations.
//
// Replaces the TF ops the reference graph instantiates for its second conv
// block (reference .idea/MNISTDist.py:52-62,76-78): Conv2D + BiasAdd + Relu +
// MaxPool forward, and Conv2DBackpropInput / Conv2DBackpropFilter + ReluGrad +
// MaxPoolGrad + BiasAddGrad backward.
//
// Forward / dgrad share one kernel (conv5x5_igemm):
//   M = positions of IMGS images, N = NC output channels, K = 25 taps x KC.
//   The (18x18 halo-padded) input tile of each image lives in LDS; the A
//   fragment of tap (ky,kx) is the same LDS image read at a shifted address,
//   so no im2col is ever materialised.  The 16 rows of an MFMA tile are
//   ordered m = 4*window + q (q = 2*dy+dx inside a 2x2 pooling window), which
//   puts every pooling window in the 4 accumulator registers of ONE lane:
//   bias + ReLU + 2x2 max-pool (+argmax) is a pure in-register epilogue.
//   MODE_DGRAD stages the tile through the max-pool/ReLU gradient instead
//   (scatter of the pooled gradient to the saved argmax position) and reads
//   the taps flipped.
//
// wgrad (conv5x5_wgrad): dW[tap][ci][co] = sum_pos X[pos+tap][ci] * dY[pos][co]
//   M = input channels, N = output channels, K = positions.  Both the X tile [pos][ci] and the
//   max-pool/ReLU-expanded dY tile [pos][co] are K-slow in LDS, so both fragments come from the
//   gfx950 transposing LDS read (ds_read_b64_tr_b16); dY is expanded in LDS only, never in
//   memory.  Split-K over images across workgroups; partial slabs are summed by
//   reduce_segments (optim.hip).
#include "conv1_core.hpp"
#include <type_traits>
#include "conv2_diag.hpp"   // DTF_ABLATED / DTF_STAMP / DTF_LAP... : empty in the shipped library

// Tile configuration of the 32->64 layer: KTOT, KC, NC, IMGS, NW, MODE, min waves/SIMD
// (overridable so tools/conv_ablate.py can sweep it)
#ifndef DTF_FWD_CFG
#define DTF_FWD_CFG 32, 32, 64, 2, 2, MODE_FWD, 4
#endif
#ifndef DTF_DGRAD_CFG
#define DTF_DGRAD_CFG 64, 32, 32, 2, 1, MODE_DGRAD, 3
#endif

// ... and of the widened 128->256 layer (NC is the layer's whole N: the launch has no channel groups).  dgrad walks its
// 256 reduction channels in 64-channel chunks: a 47 KiB tile lets three 4-wave workgroups share a CU where the
// 128-channel tile (89 KiB) allowed one -- dgrad 286 -> ~215 us, step 1.378 -> 1.296 ms on one device
// (tools/scratch/ab_wide_cfg.sh; 32-channel chunks 1.300, 8 waves per image 1.348 / 1.294 with 128 / 64 channels, two
// images per workgroup 1.31-1.33; forward: 64- or 32-channel chunks and 4 waves per image are neutral or worse).
#ifndef DTF_WIDE_FWD_CFG
#define DTF_WIDE_FWD_CFG 128, 128, 256, 1, 4, MODE_FWD
#endif
#ifndef DTF_WIDE_DGRAD_CFG
#define DTF_WIDE_DGRAD_CFG 256, 64, 128, 1, 2, MODE_DGRAD
#endif

namespace dtf {

constexpr int kTileW = 18;          // 14 + 2*2 halo
constexpr int kTilePos = 328;       // 18*18 = 324 (+4 slack for the dummy window's reads)

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

// ----------------------------------------------------------------------------
// Forward / dgrad
//   in   MODE_FWD  : x    [B,14,14,KC] bf16
//        MODE_DGRAD: gpool[B,7,7,KC] bf16 (gradient wrt the pooled output),
//                    arg  [B,7,7,KC] u8 argmax code (kPoolDead where the pooled activation is 0)
//   wk   fragment-ordered ("packed") weights, see pack_conv2_weights: one contiguous 1 KiB block
//        per (tap, 32-wide k block, 16-wide n tile), lane l's 8 k-values at l*16 bytes, so a B
//        fragment load is one fully coalesced 1 KiB read.  (From a plain [n][k] matrix the same
//        load touches 16 rows x 4 quarter-lines; the texture path handles it line by line and
//        those loads alone cost dgrad 11 of 34 us -- tools/conv_ablate.py.)
//   out  MODE_FWD  : pooled [B,7,7,NC] bf16 + argmax [B,7,7,NC] u8
//        MODE_DGRAD: dx [B,14,14,NC] bf16
// ----------------------------------------------------------------------------
// KTOT = reduction channels of the layer, KC = channels resident in LDS at a time (KTOT is
// walked in KC chunks when the full tile would not fit the 160 KiB LDS).
constexpr int igemm_tile_w(int kc) { return kc == 32 ? 20 : kTileW; }
constexpr int igemm_tile_pos(int kc) { return kc == 32 ? 18 * 20 : kTilePos; }

// Fused first conv block (FUSE1, forward only, 1 -> 32 channels): instead of loading the layer
// input x = pool1 output from memory, the workgroup computes it -- dataset row lookup (batch sampler),
// 28x28 image -> conv1 + bias + ReLU + max-pool on the matrix cores (conv1_core.hpp) -- and writes each
// pooled row both to global memory (p1 / arg1, which backward needs) and straight into the LDS tile
// of this kernel.  Removes one launch and the write -> read round trip of p1 from the forward path.
struct Conv1Fused {
  const bf16* images;   // dataset [n][784]
  const int* idx;       // optional gather (used when smp.state == nullptr)
  SampleArgs smp;
  const float* w1;      // fp32 HWIO [25][32]
  const float* b1;
  bf16* p1;             // [B,14,14,32]
  uint8_t* arg1;
};

// The same fusion in backward (FUSE1 with MODE_DGRAD): the dgrad result dx = gradient wrt pool1 is
// consumed only by the conv1 filter gradient, so it never leaves the workgroup.  After the k-loop the
// tile memory is re-used for dx (bf16 [IMGS][196][32]), the five shifted copies of the input image and
// half of the pool/ReLU-expanded gradient tile at a time (14 rows, LDS budget), and the 4 waves run
// conv1_core.hpp's wgrad K blocks over the workgroup's images; the workgroup writes one [26][32] slab.
// Removes a launch and the 12.8 MB write + read of dx.
struct Conv1WgradFused {
  const bf16* images;     // dataset [n][784]
  const int* idx;         // batch gather
  const uint8_t* arg1;    // [B,14,14,32] argmax codes of pool1
  float* part;            // [gridDim.x][26 * 32]
};
constexpr int kFusedC1wDyRows = 14;

template <int KTOT, int KC, int NC, int IMGS, int NW, int MODE, int MINW, bool FUSE1 = false>
__global__ __launch_bounds__(IMGS * 2 * NW * 64, MINW) void conv5x5_igemm(
    const bf16* __restrict__ in, const uint8_t* __restrict__ in_arg, const bf16* __restrict__ wk,
    const float* __restrict__ bias, bf16* __restrict__ out, uint8_t* __restrict__ out_arg,
    int B, Conv1Fused c1f, Conv1WgradFused c1w) {
  static_assert(!FUSE1 || MODE != MODE_FWD || (KTOT == 32 && KC == 32 && (NW == 2 || NW == 4)),
                "the fused conv1 stage feeds the 32-channel forward tile, 4 or 8 waves per image");
  static_assert(!FUSE1 || MODE != MODE_DGRAD || (NC == 32 && IMGS * 2 * NW == 4),
                "the fused conv1 wgrad stage takes 32-channel dx and runs on 4 waves");
  constexpr int NTHREADS = IMGS * 2 * NW * 64;
  constexpr int NT = NC / (16 * NW);   // n-tiles per wave
  constexpr int KSTEPS = KC / 32;      // k-steps per tap
  constexpr int CH8 = KC / 8;          // 16-byte chunks per position
  constexpr int KP = (KC % 64 == 0) ? KC + 8 : KC;  // LDS position pitch: +16 B turns the 4-way b128 conflict of 128-B pitches into 2-way
  // KC == 32 (64-byte positions): rows of 20 positions and the 16-byte chunk index XORed with
  // (pos >> 1) & 3 make every ds_read_b128 A-fragment read bank-conflict free for all 25 taps
  // (18-wide rows without the swizzle are 2-way conflicted on every read; tools/lds_bank_search.py).
  // 20 is a multiple of 4, so the swizzle of a lane is the same for all 7 row pairs of a step.
  constexpr bool SWZ = (KC == 32);
  constexpr int TW = igemm_tile_w(KC), TPOS = igemm_tile_pos(KC);
  static_assert(NC % (16 * NW) == 0, "NC must split evenly over NW waves");
  static_assert(KC % 32 == 0 && KTOT % KC == 0, "KC must be a multiple of 32 dividing KTOT");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* tile = reinterpret_cast<bf16*>(smem);  // [IMGS][TPOS][KP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w_img = wave / (2 * NW);
  const int w_h = (wave / NW) & 1;
  const int w_n = wave % NW;
  const int img0 = blockIdx.x * IMGS;
  DTF_STAMP(0);
  DTF_STAMP_CYCLES(6);
  DTF_STAGGER_HERE();

  // Fused conv1 stage: the dependent chain cursor -> dataset row -> image pixels (two memory round trips)
  // and the filter fragments are requested BEFORE the LDS zero fill, which then runs under their latency
  // (tools/conv_timeline.py: 3.8 us from workgroup start to "images in LDS" with the loads after the fill).
  constexpr int TPI = 2 * NW * 64;              // threads per image
  constexpr int PX_ROUNDS = (392 + TPI - 1) / TPI;   // 392 pixel pairs per image
  unsigned c1_px[PX_ROUNDS] = {};
  Conv1Frags c1_frags;
  if constexpr (FUSE1 && MODE == MODE_FWD) {
    const int b = img0 + w_img, tid_img = tid % TPI;
    const SampleCursor cur = sample_cursor(c1f.smp);
    c1_frags = conv1_load_frags(c1f.w1, c1f.b1, 32, 0, lane);
    if (b < B) {
      const size_t row = sample_row(c1f.smp, cur, c1f.idx, b, tid_img == 0);
      const unsigned* src = reinterpret_cast<const unsigned*>(c1f.images + row * 784);
#pragma unroll
      for (int r = 0; r < PX_ROUNDS; ++r) c1_px[r] = src[min(tid_img + r * TPI, 391)];   // 2 pixels (one dword) per load
    }
  }
  // ---- zero the whole tile (halo + slack), then fill the interior ----
  {
    u32x4 z = {0u, 0u, 0u, 0u};
    u32x4* t4 = reinterpret_cast<u32x4*>(smem);
    constexpr int N16 = (IMGS * TPOS * KP * 2 + ((FUSE1 && MODE == MODE_FWD) ? IMGS * kC1ImgElems * 2 : 0)) / 16;   // + conv1 images (zero halo)
    for (int i = tid; i < N16; i += NTHREADS) t4[i] = z;
  }
  // lane -> (row m of the 16-row tile, k-group g)
  const int m = lane & 15, g = lane >> 4;
  const int wl = m >> 2, q = m & 3;
  // position of this lane's A row for m-tile 0 (row pair 0), tap (0,0), in tile coordinates
  const int a_pos = (q >> 1) * TW + 2 * (4 * w_h + wl) + (q & 1);
  const bf16* a_base = tile + ((size_t)w_img * TPOS + a_pos) * KP + (SWZ ? 0 : g * 8);

  f32x4 acc[7][NT];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // MODE_DGRAD staging items: (image, pooled position, 8-channel chunk); thread t owns items t, t + NTHREADS, ..
  // Their global loads are issued one K chunk AHEAD (before the k-loop of the previous chunk, before the zero
  // fill for the first one): as two load -> expand -> write trips after the barrier they were two dependent
  // memory round trips, 2 us per chunk with the matrix pipe idle (tools/conv_timeline.py run .. fused_dgrad).
  constexpr int DG_NCHUNK = IMGS * 49 * CH8;
  constexpr int DG_ROUNDS = (MODE == MODE_DGRAD) ? (DG_NCHUNK + NTHREADS - 1) / NTHREADS : 1;
  u32x4 dg_g[DG_ROUNDS];
  u32x2 dg_a[DG_ROUNDS];
  const __amdgpu_buffer_rsrc_t dg_rg = make_rsrc(in, MODE == MODE_DGRAD ? (size_t)B * 49 * KTOT * 2 : 0);
  const __amdgpu_buffer_rsrc_t dg_ra = make_rsrc(in_arg, MODE == MODE_DGRAD ? (size_t)B * 49 * KTOT : 0);
  auto dgrad_prefetch = [&](int kc) {
#pragma unroll
    for (int r = 0; r < DG_ROUNDS; ++r) {
      const int c = min(tid + r * NTHREADS, DG_NCHUNK - 1);
      const int ch = c % CH8, pp = (c / CH8) % 49, im = c / (CH8 * 49);
      const uint32_t off = (uint32_t)(((size_t)min(img0 + im, B - 1) * 49 + pp) * KTOT + kc + ch * 8);
      if DTF_ABLATED(8) {
        dg_g[r] = (u32x4){0u, 0u, 0u, 0u};
        dg_a[r] = (u32x2){0x04040404u, 0x04040404u};
      } else {
        dg_g[r] = buf_load16(dg_rg, off * 2, 0);
        dg_a[r] = buf_load8(dg_ra, off, 0);
      }
    }
  };
  if constexpr (MODE == MODE_DGRAD) {
    dgrad_prefetch(0);
    __builtin_amdgcn_sched_barrier(0);
  }

#pragma unroll 1
  for (int kc0 = 0; kc0 < KTOT; kc0 += KC) {
  __syncthreads();  // zero fill done / previous chunk fully consumed
  if constexpr (FUSE1 && MODE == MODE_FWD) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    constexpr int TILE_BYTES = IMGS * TPOS * KP * 2;
    bf16* img_all = reinterpret_cast<bf16*>(smem + TILE_BYTES);                                   // [IMGS][32 x 96]
    bf16* orow_all = reinterpret_cast<bf16*>(smem + TILE_BYTES + IMGS * kC1ImgElems * 2);         // [waves][16 x 32]
    uint8_t* arow_all = reinterpret_cast<uint8_t*>(orow_all + IMGS * 2 * NW * 512);                // [waves][16 x 32]
    bf16* img_w = img_all + w_img * kC1ImgElems;
    bf16* orow = orow_all + wave * 512;
    uint8_t* arow = arow_all + wave * 512;
    const int b = img0 + w_img;
    const int tid_img = tid % TPI, wi = wave % (2 * NW);
    const Conv1Frags& frags = c1_frags;
    if (b < B) {
#pragma unroll
      for (int r = 0; r < PX_ROUNDS; ++r) {
        const int i = tid_img + r * TPI;
        if (i < 392) reinterpret_cast<unsigned*>(img_w)[((i / 14 + 2) * kC1Pitch + 2) / 2 + i % 14] = c1_px[r];
      }
    }
    __syncthreads();
    DTF_STAMP(6);   // (fused forward only: overwrites the start clock) input images in LDS
    DTF_LAPS(tr_mma = 0, tr_copy = 0);
    for (int rp = wi; rp < 14; rp += 2 * NW) {
      conv1_pooled_row(img_w, rp, frags, orow, arow, lane);
      DTF_LAP(tr_mma);
      __builtin_amdgcn_wave_barrier();
      if (b < B) {
        // 14 px x 32 ch: 56 x 16 B of bf16 -> global p1 AND this kernel's LDS tile; 28 x 16 B of argmax codes
        const size_t obase = ((size_t)b * 196 + rp * 14) * 32;
        if (lane < 56) {
          const u4 chunk = reinterpret_cast<const u4*>(orow)[lane];
          reinterpret_cast<u4*>(c1f.p1 + obase)[lane] = chunk;
          const int px = lane >> 2, ch = lane & 3;
          const int tp = (rp + 2) * TW + (px + 2);
          *reinterpret_cast<u4*>(tile + ((size_t)w_img * TPOS + tp) * KP + (ch ^ ((tp >> 1) & 3)) * 8) = chunk;
        }
        if (lane < 28) reinterpret_cast<u4*>(c1f.arg1 + obase)[lane] = reinterpret_cast<const u4*>(arow)[lane];
      }
      __builtin_amdgcn_wave_barrier();
      DTF_LAP(tr_copy);
    }
    DTF_PUT(8, tr_mma);    // wave 0: 4 rows (0, 4, 8, 12)
    DTF_PUT(9, tr_copy);
  } else if (MODE == MODE_FWD) {
    // all loads of a thread are issued before the first LDS write (one memory latency per
    // block, not one per chunk); indices past the end are clamped and their write is skipped
    constexpr int NCHUNK = IMGS * 196 * CH8;
    constexpr int ROUNDS = (NCHUNK + NTHREADS - 1) / NTHREADS;
    u32x4 v[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      int c = min(tid + r * NTHREADS, NCHUNK - 1);
      int ch = c % CH8, pos = (c / CH8) % 196, im = c / (CH8 * 196);
      int b = min(img0 + im, B - 1);
      v[r] = DTF_ABLATED(8) ? (u32x4){0u, 0u, 0u, 0u}
                              : *reinterpret_cast<const u32x4*>(in + ((size_t)b * 196 + pos) * KTOT + kc0 + ch * 8);
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      int c = tid + r * NTHREADS;
      int ch = c % CH8, pos = (c / CH8) % 196, im = c / (CH8 * 196);
      int y = pos / 14, x = pos % 14;
      int tp = (y + 2) * TW + (x + 2);
      if (c < NCHUNK && img0 + im < B)
        *reinterpret_cast<u32x4*>(tile + ((size_t)im * TPOS + tp) * KP + (SWZ ? (ch ^ ((tp >> 1) & 3)) : ch) * 8) = v[r];
    }
  } else {
    // max-pool / ReLU gradient: route g (prefetched above) to the saved argmax position.
#pragma unroll
    for (int r = 0; r < DG_ROUNDS; ++r) {
      const int c = tid + r * NTHREADS;
      const int ch = c % CH8, pp = (c / CH8) % 49, im = c / (CH8 * 49);
      if (c < DG_NCHUNK && img0 + im < B) {
        const int py = pp / 7, px = pp % 7;
        u32x4 o[4];
        expand_pool_grad(dg_g[r], dg_a[r], o);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          int tp = (2 * py + (qq >> 1) + 2) * TW + (2 * px + (qq & 1) + 2);
          *reinterpret_cast<u32x4*>(tile + ((size_t)im * TPOS + tp) * KP + (SWZ ? (ch ^ ((tp >> 1) & 3)) : ch) * 8) = o[qq];
        }
      }
    }
    // the next chunk's loads fly during this chunk's k-loop
    if (kc0 + KC < KTOT) {
      dgrad_prefetch(kc0 + KC);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();
  DTF_STAMP(1);

  // ---- main loop over this chunk ----
  // B fragment: column n = lane&15 of n-tile nt, k = kc0 + 8g..8g+7
  const bf16* b_base = wk + ((size_t)(kc0 / 32) * (NC / 16) + w_n * NT) * 512 + lane * 8;

  bf16x8 bcur[NT], bnxt[NT];
  auto load_b = [&](bf16x8 (&dst)[NT], int step) {
    int tap = step / KSTEPS, ks = step % KSTEPS;
    int wtap = (MODE == MODE_DGRAD) ? (24 - tap) : tap;
#pragma unroll
    for (int j = 0; j < NT; ++j)
      dst[j] = *reinterpret_cast<const bf16x8*>(b_base + (((size_t)wtap * (KTOT / 32) + ks) * (NC / 16) + j) * 512);
  };
  constexpr int NSTEPS = 25 * KSTEPS;
  auto do_step = [&](int step, const bf16x8 (&bf)[NT]) {
    const int tap = step / KSTEPS, ks = step % KSTEPS;
    const int ky = tap / 5, kx = tap % 5;
    const bf16* ap = a_base + (ky * TW + kx) * KP + ks * 32;
    if (SWZ) ap += (g ^ (((a_pos + ky * TW + kx) >> 1) & 3)) * 8;
    bf16x8 a[7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
      a[i] = *reinterpret_cast<const bf16x8*>(DTF_ABLATED(2) ? a_base + i * 8 : ap + i * (2 * TW * KP));
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if DTF_ABLATED(4) { asm volatile("" ::"v"(a[i]), "v"(bf[j])); }
        else acc[i][j] = mfma16(a[i], bf[j], acc[i][j]);
      }
  };
  if constexpr (NSTEPS <= 50) {
    // Fully unrolled with a ring of D fragment sets: the B fragments of step s+D-1 are requested
    // before the MFMAs of step s.  One step is only ~250 cycles of MFMA, an L2 hit costs more, and a
    // distance-1 prefetch left ~10 us of exposed load latency in dgrad (tools/conv_ablate.py).
    // Loads are unconditional (past the end they re-load the last step): exact vmcnt waits.
    constexpr int D = 4;
    bf16x8 ring[D][NT];
#pragma unroll
    for (int p = 0; p < D - 1; ++p) load_b(ring[p], p < NSTEPS ? p : NSTEPS - 1);
#pragma unroll
    for (int step = 0; step < NSTEPS; ++step) {
      if (!DTF_ABLATED(1)) load_b(ring[(step + D - 1) % D], step + D - 1 < NSTEPS ? step + D - 1 : NSTEPS - 1);
      __builtin_amdgcn_sched_barrier(0);
      do_step(step, ring[DTF_ABLATED(1) ? 0 : step % D]);
    }
  } else {
    // B fragments of step s+1 are loaded (unconditionally: the last step re-loads itself, so every
    // wait is an exact count) before the MFMAs of step s; two register sets, loop unrolled by two
    load_b(bcur, 0);
#pragma unroll 1
    for (int step = 0; step < NSTEPS; step += 2) {
      if (!DTF_ABLATED(1)) load_b(bnxt, min(step + 1, NSTEPS - 1));
      __builtin_amdgcn_sched_barrier(0);
      do_step(step, bcur);
      if (step + 1 >= NSTEPS) break;
      if (!DTF_ABLATED(1)) load_b(bcur, min(step + 2, NSTEPS - 1));
      __builtin_amdgcn_sched_barrier(0);
      do_step(step + 1, DTF_ABLATED(1) ? bcur : bnxt);
    }
  }
  }  // K chunks

  // ---- epilogue: results go through LDS (the tile is dead by now) so that global memory sees
  // linear 16-byte stores; straight from the C/D layout they would be 2-byte (1-byte for argmax)
  // stores scattered over positions ----
  // C/D map: lane holds rows 4*(lane>>4)+r -> window (lane>>4), q = r; col = lane&15
  const int ewin = 4 * w_h + g;  // window index along x (0..7; 7 is the dummy window)
  DTF_STAMP(2);   // this wave's k-loop issued (MFMAs may still drain)
  __syncthreads();
  DTF_STAMP(3);   // every wave's k-loop done
  if (MODE == MODE_FWD) {
    bf16* ot = reinterpret_cast<bf16*>(smem);                      // [IMGS][49][NC]
    uint8_t* oa = reinterpret_cast<uint8_t*>(smem) + IMGS * 49 * NC * 2;  // [IMGS][49][NC]
    if (ewin < 7) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = w_n * NT * 16 + j * 16 + m;
        const float bv = bias[n];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          f32x4 v = acc[i][j];
          float best = v[0];
          int bi = 0;
          if (v[1] > best) { best = v[1]; bi = 1; }
          if (v[2] > best) { best = v[2]; bi = 2; }
          if (v[3] > best) { best = v[3]; bi = 3; }
          float o = fmaxf(best + bv, 0.f);
          const int off = ((w_img * 7 + i) * 7 + ewin) * NC + n;
          ot[off] = (bf16)o;
          oa[off] = (uint8_t)(o > 0.f ? bi : (int)kPoolDead);
        }
      }
    }
    __syncthreads();
    DTF_STAMP(4);
    if (!DTF_ABLATED(16)) {
      const int nimg = min(IMGS, B - img0);
      u32x4* gp = reinterpret_cast<u32x4*>(out + (size_t)img0 * 49 * NC);
      for (int c = tid; c < nimg * 49 * NC / 8; c += NTHREADS) gp[c] = reinterpret_cast<const u32x4*>(ot)[c];
      u32x4* ga = reinterpret_cast<u32x4*>(out_arg + (size_t)img0 * 49 * NC);
      for (int c = tid; c < nimg * 49 * NC / 16; c += NTHREADS) ga[c] = reinterpret_cast<const u32x4*>(oa)[c];
    }
    DTF_STAMP(5);
    DTF_STAMP_CYCLES(7);
  } else {
    bf16* ot = reinterpret_cast<bf16*>(smem);  // [IMGS][196][NC]
    if (ewin < 7) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = w_n * NT * 16 + j * 16 + m;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          f32x4 v = acc[i][j];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int y = 2 * i + (r >> 1), x = 2 * ewin + (r & 1);
            ot[((w_img * 14 + y) * 14 + x) * NC + n] = (bf16)v[r];
          }
        }
      }
    }
    if constexpr (FUSE1) {
      // ---- conv1 filter/bias gradient of this workgroup's images, dx taken from LDS ----
      constexpr int OT_BYTES = IMGS * 196 * NC * 2;
      bf16* cp = reinterpret_cast<bf16*>(smem + OT_BYTES);
      bf16* dyt = reinterpret_cast<bf16*>(smem + OT_BYTES + kCpBytes);
      const int nimg = min(IMGS, B - img0);
      // every global load of this phase up front: 2 pixel pairs per image, and the argmax codes of the
      // (image, tile half, round) items this thread expands
      unsigned px_r[IMGS][2];
      u32x2 ar_r[IMGS][2][2];
#pragma unroll
      for (int im = 0; im < IMGS; ++im) {
        const int b = min(img0 + im, B - 1);
        const unsigned* src = reinterpret_cast<const unsigned*>(c1w.images + (size_t)c1w.idx[b] * 784);
#pragma unroll
        for (int r = 0; r < 2; ++r) px_r[im][r] = src[min(tid + r * 256, 391)];
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int c = min(tid + r * 256, 98 * 4 - 1);
            ar_r[im][half][r] = *reinterpret_cast<const u32x2*>(c1w.arg1 + ((size_t)b * 196 + half * 98 + (c >> 2)) * 32 + (c & 3) * 8);
          }
      }
      // the copy halos and the x = 28..31 pad of the gradient tile must be zero (this memory held the conv2 tile)
      for (int i = tid; i < (kCpBytes + kFusedC1wDyRows * kDy1RowBytes) / 16; i += NTHREADS)
        reinterpret_cast<u32x4*>(smem + OT_BYTES)[i] = (u32x4){0u, 0u, 0u, 0u};
      const Conv1WgradLane L = conv1w_lane(lane);
      f32x4 acc1[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      __syncthreads();   // dx complete in ot, zero fill done
      DTF_STAMP(4);
#pragma unroll
      for (int im = 0; im < IMGS; ++im) {
        if (im < nimg) {   // uniform
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int i = tid + r * 256;
            if (i < 392) conv1w_write_pair(cp, i, px_r[im][r]);
          }
#pragma unroll
          for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const int c = tid + r * 256;
              if (c < 98 * 4) {
                const int pp = half * 98 + (c >> 2), c8 = c & 3;
                const u32x4 gv = *reinterpret_cast<const u32x4*>(ot + ((size_t)im * 196 + pp) * NC + c8 * 8);
                conv1w_expand(dyt, pp / 14, pp % 14, c8, gv, ar_r[im][half][r], half * kFusedC1wDyRows);
              }
            }
            __syncthreads();
            for (int kb = half * 14 + wave; kb < half * 14 + 14; kb += 4)
              conv1w_kblock(cp, dyt, kb, half * kFusedC1wDyRows, L, lane, acc1);
            __syncthreads();
          }
        }
      }
      conv1w_reduce_store(acc1, reinterpret_cast<float*>(smem + OT_BYTES), c1w.part + (size_t)blockIdx.x * 26 * 32, 32, 0,
                          wave, lane);
      DTF_STAMP(5);
    } else {
      __syncthreads();
      if (!DTF_ABLATED(16)) {
        const int nimg = min(IMGS, B - img0);
        u32x4* gp = reinterpret_cast<u32x4*>(out + (size_t)img0 * 196 * NC);
        for (int c = tid; c < nimg * 196 * NC / 8; c += NTHREADS) gp[c] = reinterpret_cast<const u32x4*>(ot)[c];
      }
    }
  }
}

// ----------------------------------------------------------------------------
// Weight packing for conv5x5_igemm.  w: bf16 HWIO [25][cin][cout] (the shadow arena).
//   fwd  pack (k = ci, n = co): F[((tap*(cin/32)  + ci/32)*(cout/16) + co/16)*512 + (((ci%32)/8)*16 + co%16)*8 + ci%8]
//   dgrad pack (k = co, n = ci): D[((tap*(cout/32) + co/32)*(cin/16)  + ci/16)*512 + (((co%32)/8)*16 + ci%16)*8 + co%8]
// packed = [F | D], 25*cin*cout elements each.  The fused SGD tail (optim.hip) writes the same
// layout directly; this kernel serves init / restore / the multi-GPU and Adam paths.
// ----------------------------------------------------------------------------
__host__ __device__ __forceinline__ long conv2_pack_fwd_index(int tap, int ci, int co, int cin, int cout) {
  return (((long)tap * (cin / 32) + ci / 32) * (cout / 16) + co / 16) * 512 + (((ci % 32) / 8) * 16 + co % 16) * 8 + ci % 8;
}
__host__ __device__ __forceinline__ long conv2_pack_dgrad_index(int tap, int ci, int co, int cin, int cout) {
  return (((long)tap * (cout / 32) + co / 32) * (cin / 16) + ci / 16) * 512 + (((co % 32) / 8) * 16 + ci % 16) * 8 + co % 8;
}

__global__ __launch_bounds__(256) void pack_conv2_weights(const bf16* __restrict__ w, bf16* __restrict__ packed,
                                                          int cin, int cout) {
  const long n = 25L * cin * cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i / (cin * cout)), rem = (int)(i % (cin * cout));
    const int ci = rem / cout, co = rem % cout;
    const bf16 v = w[i];
    packed[conv2_pack_fwd_index(tap, ci, co, cin, cout)] = v;
    packed[n + conv2_pack_dgrad_index(tap, ci, co, cin, cout)] = v;
  }
}

// ----------------------------------------------------------------------------
// wgrad.  grid = (G image chunks, (C_IN/16) * (C_OUT/32) output tiles)
//   x      [B,14,14,C_IN] bf16   layer input
//   gpool  [B,7,7,C_OUT]  bf16   gradient wrt pooled output
//   arg    [B,7,7,C_OUT]  u8     saved argmax code (kPoolDead where the pooled activation is 0)
//   part   [G][25*C_IN*C_OUT + C_OUT] fp32 partial slabs (HWIO order, then bias)
// A block owns ALL 25 taps of a 16(ci) x 32(co) output tile, so the two LDS images it stages per
// image -- the halo-padded input [pos][16 ci] and the max-pool/ReLU-expanded output gradient
// [row pair][k][32 co] (k = 16*dy + x) -- are each staged by only C_OUT/32 resp. C_IN/16 blocks.
// Both images are K-slow, so A and B fragments come from ds_read_b64_tr_b16 (the input image
// is conflict-free as is; the 32-byte halves of the gradient image are XOR-swizzled by k>>2).
// Waves split the taps (7/6/6/6): a wave holds 7 taps x 2 n-tiles of accumulators and each B
// fragment feeds every tap.
// Element j of lane group g  <->  dy = g>>1, x = 4*(g&1) + (j&3) + 8*(j>>2).
// ----------------------------------------------------------------------------
// NT = 16-wide co tiles per wave (output tile 16 ci x 16*NT co).  NT = 2 is the 32/64 model's tile (4 tiles x 64
// image chunks fill the chip); NT = 4 reads 11 fragments per 28 MFMAs instead of 9 per 14 (-39 % LDS bytes per
// MFMA, the bound of this kernel) at the price of half as many tiles: the widened model has enough of them.
constexpr int kWgXTileBytes = kTilePos * 16 * 2;   // 10496
constexpr int wg_dy_tile_bytes(int nt) { return 7 * 32 * 16 * nt * 2; }          // NT 2: 14336, NT 4: 28672
constexpr int wg_red_rows(int nt) { return 7 * nt * 4 + nt; }                    // accumulators + bias sums per lane
constexpr int wg_red_bytes(int nt) { return 4 * wg_red_rows(nt) * 64 * 4; }      // end-of-kernel sum of the two k halves
constexpr int wg_lds_bytes(int nt) {
  return wg_red_bytes(nt) > 2 * (kWgXTileBytes + wg_dy_tile_bytes(nt)) ? wg_red_bytes(nt)
                                                                       : 2 * (kWgXTileBytes + wg_dy_tile_bytes(nt));
}

// 8 waves = (tw: tap group, taps tw, tw+4, ..) x (kh: alternate row pairs).  Two waves per SIMD let
// the transposed LDS reads of one wave run under the MFMAs of the other; the global loads of image
// b+1 are issued before the k-loop of image b (register prefetch) and expanded/written to LDS after
// it.  The two kh halves are summed through LDS once, at the end.
template <int C_IN, int C_OUT, int NT>
__global__ __launch_bounds__(512, NT == 2 ? 2 : 1) void conv5x5_wgrad(
    const bf16* __restrict__ x, const bf16* __restrict__ gpool,
    const uint8_t* __restrict__ arg, float* __restrict__ part, int B, int imgs_per_block) {
  constexpr int CI_T = C_IN / 16;
  constexpr int CO_W = 16 * NT;        // output channels per workgroup
  constexpr int CO_CH = CO_W / 8;      // 16-byte chunks per gradient row
  constexpr int RED_ROWS = wg_red_rows(NT);
  static_assert(NT == 2 || NT == 4, "2 or 4 co tiles per wave");
  static_assert(C_OUT % CO_W == 0, "C_OUT must be a multiple of the workgroup's co tile");
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // wg_lds_bytes(NT)
  // two image buffers, each: xt [328][16] | dyt [7][32][CO_W] (32-B n-tile chunks swizzled over the 256-B bank row)
  constexpr int kWgDyTileBytes = wg_dy_tile_bytes(NT);
  constexpr int kBufElems = (kWgXTileBytes + kWgDyTileBytes) / 2;
  bf16* xt = reinterpret_cast<bf16*>(smem);
  bf16* dyt = reinterpret_cast<bf16*>(smem + kWgXTileBytes);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tw = wave & 3, kh = wave >> 2;
  const int ci_t = blockIdx.y % CI_T, co_t = blockIdx.y / CI_T;
  const int g = lane >> 4, i16 = lane & 15;
  const int qr = i16 >> 2, pc = i16 & 3;
  const bool do_bias = (ci_t == 0) && (tw == 0);

  const int b_begin = blockIdx.x * imgs_per_block;
  const int b_end = min(B, b_begin + imgs_per_block);
  // (timing build: slots per workgroup (gridDim.x * blockIdx.y + blockIdx.x), 16 each: wave 0 -> 0..7, wave 7 -> 8..15:
  //  0 start, 1 loop entered, 2 loop left, 3 reduced, 4 sum(wait for prefetched loads), 5 sum(k-loop), 6 sum(stage), 7 sum(barrier wait))
  DTF_TIMING_ONLY(long long* tb = (g_timing_buf && (lane == 0) && (wave == 0 || wave == 7))
                      ? g_timing_buf + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (wave ? 8 : 0)) : nullptr;)
  DTF_LAPS(t_k = 0, t_s = 0, t_w = 0, t_v = 0);
  DTF_TIMING_ONLY(if (tb) tb[0] = dtf_lap_prev_;)

  f32x4 acc[7][NT];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bsum[j] = 0.f;

  // zero once: halo, slack and the x = 14,15 dummy columns stay zero
  for (int i = tid; i < 2 * (kWgXTileBytes + kWgDyTileBytes) / 16; i += 512)
    reinterpret_cast<u32x4*>(smem)[i] = (u32x4){0u, 0u, 0u, 0u};

  const int tr_pos = (g >> 1) * kTileW + 4 * (g & 1) + qr;   // A: row pair 0, tap (0,0), first read
  const int kkrow = (g >> 1) * 16 + 4 * (g & 1) + qr;        // B: k row of the first read
  // 8 / NT gradient rows span the 256-B bank row, and the 8 rows a 32-lane group reads must land in 8 different
  // 32-byte chunks of it: chunk = n tile XOR (row / (8 / NT)) mod NT.  The +8 read has the same swizzle.
  auto dy_swz = [](int kk) { return (kk / (8 / NT)) & (NT - 1); };
  const int bsw = dy_swz(kkrow);
  // element offsets of this wave's taps (tw + 4t; the 7th tap of groups 1..3 does not exist and is
  // clamped to tap 24: computed branch-free and discarded, so the k-loop has no control flow)
  int tap_off[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int tap = min(tw + 4 * t, 24);
    tap_off[t] = ((tap / 5) * kTileW + tap % 5) * 16;
  }
  const bf16* a_lane = xt + tr_pos * 16 + pc * 4;

  // staging roles: threads 0..391 own one 16-byte chunk of the input tile, threads 0..195 one
  // (pooled position, 8-channel chunk) item of the gradient expansion
  const bool has_x = tid < 196 * 2, has_dy = tid < 49 * CO_CH;
  const int x_ch = tid & 1, x_pos = tid >> 1;
  const int x_tp = (x_pos / 14 + 2) * kTileW + (x_pos % 14 + 2);
  const int d_c8 = tid % CO_CH, d_pp = tid / CO_CH;
  u32x4 xr = {0u, 0u, 0u, 0u};
  u32x4 gv_r = {0u, 0u, 0u, 0u};
  u32x2 ar_r = {0u, 0u};
  auto prefetch = [&](int b) {
    if (has_x && !DTF_ABLATED(32))
      xr = *reinterpret_cast<const u32x4*>(x + ((size_t)b * 196 + x_pos) * C_IN + ci_t * 16 + x_ch * 8);
    if (has_dy && !DTF_ABLATED(128)) {
      size_t off = ((size_t)b * 49 + d_pp) * C_OUT + co_t * CO_W + d_c8 * 8;
      gv_r = *reinterpret_cast<const u32x4*>(gpool + off);
      ar_r = *reinterpret_cast<const u32x2*>(arg + off);
    }
  };
  auto stage = [&](int boff) {
    if (has_x && !DTF_ABLATED(32)) *reinterpret_cast<u32x4*>(xt + boff + x_tp * 16 + x_ch * 8) = xr;
    if (has_dy && !DTF_ABLATED(128)) {
      const int r = d_pp / 7, w = d_pp % 7;
      u32x4 o[4];
      expand_pool_grad(gv_r, ar_r, o);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int kk = (q >> 1) * 16 + 2 * w + (q & 1);
        int slot = (((d_c8 >> 1) ^ dy_swz(kk)) << 1) | (d_c8 & 1);
        *reinterpret_cast<u32x4*>(dyt + boff + (r * 32 + kk) * CO_W + slot * 8) = o[q];
      }
    }
  };

  if (b_begin < b_end) prefetch(b_begin);
  __syncthreads();  // zero fill done
  if (b_begin < b_end) stage(0);
  if (b_begin + 1 < b_end) prefetch(b_begin + 1);
  DTF_TIMING_ONLY(if (tb) tb[1] = wall_clock64();)
  for (int b = b_begin; b < b_end; ++b) {
    const int boff = ((b - b_begin) & 1) * kBufElems;
    // image b is staged in buffer boff; every wave is done reading the other buffer (image b-1),
    // which this iteration refills with image b+1
    DTF_LAP_START();
    __syncthreads();
    DTF_LAP(t_w);
    // Image b+1 goes into the other buffer BEFORE this wave's k-loop: the staging waves (0..3: gradient
    // expansion, ~0.6 us of VALU + LDS writes per image) then work while their SIMD partners (waves 4..7)
    // already run MFMAs, and run their own k-loop while the partners drain -- with the staging after the
    // k-loop all eight waves competed for the matrix pipe first and it then sat idle during the staging
    // (tools/conv_timeline.py: 9.1 us of staging on the critical path of a 31 us kernel).
    DTF_LAP_AFTER("s_waitcnt vmcnt(0)", t_v);
    if (!DTF_WGRAD_STAGE_LAST && b + 1 < b_end) {
      stage(kBufElems - boff);
      if (b + 2 < b_end) prefetch(b + 2);
    }
    DTF_LAP(t_s);
    // k-loop over this wave's row pairs, software-pipelined: the 18 transposed reads of step r+2
    // are in flight while the 14 MFMAs of step r issue (two fragment sets, loop unrolled by two)
    const bf16* a_img = a_lane + boff;
    const bf16* b_img = dyt + boff + kkrow * CO_W + pc * 4;
    auto load_frags = [&](int r, bf16x8 (&af)[7], bf16x8 (&bf)[NT]) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf16* p0 = b_img + r * 32 * CO_W + ((j ^ bsw) << 4);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 8 * CO_W));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bf[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        const bf16* p0 = a_img + 2 * r * kTileW * 16 + tap_off[t];
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 8 * 16));
        s16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        af[t] = __builtin_bit_cast(bf16x8, av);
      }
    };
    auto mma_frags = [&](const bf16x8 (&af)[7], const bf16x8 (&bf)[NT]) {
      if (do_bias) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[j] += (float)bf[j][e];
      }
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        if DTF_ABLATED(64) {
          asm volatile("" ::"v"(af[t]), "v"(bf[0]), "v"(bf[1]));
        } else {
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[t][j] = mfma16(af[t], bf[j], acc[t][j]);
        }
      }
    };
    if (NT == 4) {
      // 112 accumulator registers per lane leave room for ONE fragment set (two spill: 153 registers to scratch):
      // no software pipeline here; the SIMD's other wave runs its 28 MFMAs (448 cycles) while this one waits for
      // its 22 transposed reads
      bf16x8 af0[7], bf0[NT];
      const int r0 = (kh + b) & 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = r0 + 2 * i;
        if (r < 7) {
          load_frags(r, af0, bf0);
          __builtin_amdgcn_sched_barrier(0);
          mma_frags(af0, bf0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if (!DTF_ABLATED(256)) {
      // 4 (r = 0,2,4,6) or 3 (r = 1,3,5) steps, straight-line so every wait count is exact
      bf16x8 af0[7], bf0[NT], af1[7], bf1[NT];
      const int r = (kh + b) & 1;
      load_frags(r, af0, bf0);
      load_frags(r + 2, af1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      mma_frags(af0, bf0);
      __builtin_amdgcn_sched_barrier(0);
      load_frags(r + 4, af0, bf0);
      __builtin_amdgcn_sched_barrier(0);
      mma_frags(af1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      if (r == 0) {
        load_frags(6, af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
        mma_frags(af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        mma_frags(af1, bf1);
      } else {
        mma_frags(af0, bf0);
      }
    }
    DTF_LAP_AFTER("s_nop 0", t_k);
    if (DTF_WGRAD_STAGE_LAST && b + 1 < b_end) {
      stage(kBufElems - boff);
      if (b + 2 < b_end) prefetch(b + 2);
    }
  }
  DTF_TIMING_ONLY(if (tb) { tb[2] = wall_clock64(); tb[5] = t_k; tb[6] = t_s; tb[7] = t_w; tb[4] = t_v; })

  // ---- sum the two kh halves through LDS; kh = 0 waves write their taps: rows ci, cols co ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);  // [tw][RED_ROWS][64 lanes]
  if (kh == 1) {
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(tw * RED_ROWS + (t * NT + j) * 4 + r) * 64 + lane] = acc[t][j][r];
#pragma unroll
    for (int j = 0; j < NT; ++j) red[(tw * RED_ROWS + 28 * NT + j) * 64 + lane] = bsum[j];
  }
  __syncthreads();
  DTF_TIMING_ONLY(if (tb) tb[3] = wall_clock64();)
  if (kh == 1) return;
  float* slab = part + (size_t)blockIdx.x * (25 * C_IN * C_OUT + C_OUT);
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int tap = tw + 4 * t;
    if (tap < 25) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int ci = ci_t * 16 + 4 * g + r;
          int co = co_t * CO_W + j * 16 + i16;
          float v = acc[t][j][r] + red[(tw * RED_ROWS + (t * NT + j) * 4 + r) * 64 + lane];
          if (DTF_ABLATED(512) && v != 12345.f) continue;
          slab[((size_t)tap * C_IN + ci) * C_OUT + co] = v;
        }
    }
  }
  if (do_bias) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float s = bsum[j] + red[(tw * RED_ROWS + 28 * NT + j) * 64 + lane];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (g == 0) slab[25 * C_IN * C_OUT + co_t * CO_W + j * 16 + i16] = s;
    }
  }
}


// ----------------------------------------------------------------------------
// wgrad, producer / consumer form (the 32 -> 64 model's launch).  Same output tile (16 ci x 32 co x 25 taps), same
// LDS images and fragment reads as conv5x5_wgrad<.., NT = 2>; what differs is who does what:
//   * waves 8..11 are PRODUCERS (one per SIMD): they fetch the two images of round p+2 into registers, expand /
//     write those of round p+1 into the other pair of LDS buffers and meet the MFMA waves at the barrier of the
//     round.  Their VALU / LDS-write instructions issue beside the MFMAs of the SIMD's other two waves.  In
//     conv5x5_wgrad the MFMA waves stage themselves, and the stamps (profiles/r1_conv2_fwd_timeline.txt) show
//     3-8 us of staging + 2.6-4.4 us of barrier skew per wave on top of a 14-18 us k-loop.
//   * a round is TWO images: MFMA wave (tw, kh) runs all seven row pairs of image 2p + kh for taps tw, tw+4, ...
//     -- a straight-line, branch-free 98-MFMA body (the 4-or-3-step alternation of conv5x5_wgrad needs a branch
//     with the accumulators live across it), half as many barriers, one fragment-latency bubble per 7 k-steps.
//   * twelve waves leave 168 registers per lane: ONE rolling set of A fragments (tap t of the next k-step is
//     fetched right behind the two MFMAs that consumed tap t of this one) and two sets of B fragments.
//   * rounds are handed over through two LDS progress counters, not s_barrier: three pairs of image buffers, the
//     producers run up to two rounds ahead, and an MFMA wave never waits for another MFMA wave (with a barrier per
//     round every wave waited 0.8 us per round for the slowest of the eight: stamps of the barrier version).
// LDS: 6 image buffers (146 KiB), one workgroup per CU.
// ----------------------------------------------------------------------------
constexpr int kPcBufBytes = kWgXTileBytes + wg_dy_tile_bytes(2);   // 24832
constexpr int kPcPairs = 3;                                        // rounds in flight (pairs of image buffers)
constexpr int kPcFlagOff = 2 * kPcPairs * kPcBufBytes;             // two progress counters behind the buffers
constexpr int kPcLdsBytes = kPcFlagOff + 64;                       // 149056 of the CU's 163840
static_assert(kPcFlagOff >= wg_red_bytes(2), "the end-of-kernel reduction aliases the image buffers");

// (round hand-over through per-wave LDS progress words: pc_signal / pc_wait_all_ge, common.hpp)

// Tap chains.  The A fragment F(y, kx) = input rows (y, y + 1) shifted by kx serves tap (ky, kx) at row pair r whenever
// 2 r + ky = y: taps (0, kx), (2, kx), (4, kx) of consecutive row pairs read the SAME fragments, and so do (1, kx),
// (3, kx).  An MFMA wave therefore owns whole chains -- E_kx = {(0,kx), (2,kx), (4,kx)} and O_kx = {(1,kx), (3,kx)} --
// keeps a chain's 3 (2) fragments in registers and fetches ONE new fragment per chain and k-step (the one the chain's
// last tap needs next), into the slot of the fragment that just died: 2-3 A fragments per k-step instead of 7, and no
// tap slot that is computed and discarded (25 taps = 6 + 6 + 7 + 6).  Roles of the four waves of an image:
//   0: E0 E1 (+ the bias row: ones x dY in slot 6)   1: E2 E3   2: E4 O0 O1   3: O2 O3 O4
// (14, 12, 14, 12 tiles per k-step.)  The kh = 1 wave of SIMD tw takes role tw + 1, so every SIMD multiplies 26 tiles
// per k-step pair (28 before).
struct WgChain {
  int len, ky0, kx;
};
constexpr int wg_role_chains(int role) { return role < 2 ? 2 : 3; }
constexpr WgChain wg_role_chain(int role, int c) {
  return role == 0 ? WgChain{3, 0, c}
         : role == 1 ? WgChain{3, 0, 2 + c}
         : role == 2 ? (c == 0 ? WgChain{3, 0, 4} : WgChain{2, 1, c - 1})
                     : WgChain{2, 1, 2 + c};
}
constexpr int wg_role_base(int role, int c) {
  int b = 0;
  for (int i = 0; i < c; ++i) b += wg_role_chain(role, i).len;
  return b;
}

template <int C_IN, int C_OUT>
__global__ __launch_bounds__(768) void conv5x5_wgrad_pc(const bf16* __restrict__ x, const bf16* __restrict__ gpool,
                                                        const uint8_t* __restrict__ arg, float* __restrict__ part, int B,
                                                        int imgs_per_block) {
  constexpr int NT = 2, CI_T = C_IN / 16, CO_W = 16 * NT, CO_CH = CO_W / 8;
  constexpr int kBufElems = kPcBufBytes / 2;
  static_assert(C_OUT % CO_W == 0 && C_IN % 16 == 0, "tile shape");
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // kPcLdsBytes
  bf16* xt = reinterpret_cast<bf16*>(smem);                     // buffer i: xt + i * kBufElems  [328][16]
  bf16* dyt = reinterpret_cast<bf16*>(smem + kWgXTileBytes);    //           dyt + i * kBufElems [7][32][CO_W]
  unsigned* staged = reinterpret_cast<unsigned*>(smem + kPcFlagOff);   // [4]
  unsigned* done = staged + 4;                                         // [8]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_prod = wave >= 8;
  const int tw = wave & 3, kh = (wave >> 2) & 1;
  const int ci_t = blockIdx.y % CI_T, co_t = blockIdx.y / CI_T;
  const int b_begin = blockIdx.x * imgs_per_block;
  const int b_end = min(B, b_begin + imgs_per_block);
  const int rounds = (b_end - b_begin + 1) >> 1;
  auto dy_swz = [](int kk) { return (kk / (8 / NT)) & (NT - 1); };

  // (timing build: 16 slots per workgroup, wave 0 -> 0..7, wave 8 -> 8..15: 0 start, 1 loop entered, 2 loop left,
  //  3 reduced, 4 sum(wait for prefetched loads), 5 sum(k-loop), 6 sum(stage), 7 sum(barrier wait))
  //  32 slots per workgroup here: 16 + w = when MFMA wave w left its loop, 24 + w = its summed wait)
  DTF_TIMING_ONLY(long long* tb_wg = g_timing_buf ? g_timing_buf + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 32 : nullptr;)
  DTF_TIMING_ONLY(long long* tb = (tb_wg && (lane == 0) && (wave == 0 || wave == 8)) ? tb_wg + (wave ? 8 : 0) : nullptr;)
  DTF_LAPS(t_k = 0, t_s = 0, t_w = 0, t_v = 0);
  DTF_TIMING_ONLY(if (tb) tb[0] = dtf_lap_prev_;)
  // Zero what no round writes: the halo (and slack) positions of the six input tiles, the x = 14, 15 dummy k rows of the
  // six gradient tiles, the `done` words -- 36 KiB, not the whole 146 KiB.  The MFMA waves do it (512 threads) while
  // the producers already stage round 0: the two touch disjoint bytes, and `staged` needs no zero (it is stored, not
  // added to, and first read after the barrier that follows the producers' first store).
  auto zero_fill = [&]() {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int c = tid; c < 2 * kPcPairs * 2 * kTilePos; c += 512) {
      const int buf = c / (2 * kTilePos), rem = c - buf * (2 * kTilePos), pos = rem >> 1;
      const int row = pos / kTileW, col = pos - row * kTileW;
      if (!(row >= 2 && row < 16 && col >= 2 && col < 16)) *reinterpret_cast<u32x4*>(xt + buf * kBufElems + pos * 16 + (rem & 1) * 8) = z;
    }
    for (int c = tid; c < 2 * kPcPairs * 7 * 4 * CO_CH; c += 512) {
      const int buf = c / (28 * CO_CH), rem = c - buf * (28 * CO_CH), rr = rem / CO_CH, ch = rem - rr * CO_CH;
      const int kk = (rr >> 2) * 32 + 14 + (rr & 1) + 16 * ((rr >> 1) & 1);
      *reinterpret_cast<u32x4*>(dyt + buf * kBufElems + kk * CO_W + ch * 8) = z;
    }
    if (tid < 8) done[tid] = 0u;
  };

  if (is_prod) {
    // ---- producers: per round 2 x 392 input chunks (13 wave-rows of 64) and 2 x 196 gradient items (7 rows) ----
    // A gradient item (8 pooled gradients + codes -> four 16-byte window positions) costs about five input chunks
    // (a 16-byte copy): waves 8..10 take two gradient rows + three input rows each, wave 11 the last gradient row
    // (8 items) + four input rows.  Everything that does not change from round to round -- the item's offset in
    // its image, its LDS destination -- is worked out once, here: a lone wave per SIMD issues a dependent VALU
    // instruction every 5-8 cycles beside two MFMA waves, and the first version of this loop (index arithmetic per
    // item per round) took 2.3 us per round, longer than the 2.0 us of MFMA work it was meant to hide behind.
    __builtin_amdgcn_s_setprio(3);   // win the SIMD's issue arbitration against its two MFMA waves
    constexpr int X_ITEMS = 196 * 2, DY_ITEMS = 49 * CO_CH, XR = 4, DR = 2;
    constexpr int kXImg = 196 * C_IN, kDyImg = 49 * C_OUT;   // elements per image
    const int pw = wave - 8;
    int xsrc[XR], xdst[XR], xim[XR], dsrc[DR], ddst[DR], dim_[DR];   // src: element offset in the image; dst: in the buffer pair
#pragma unroll
    for (int k = 0; k < XR; ++k) {
      const int row = pw < 3 ? 3 * pw + k : 9 + k;
      const int it = row * 64 + lane;
      const bool ok = (pw == 3 || k < 3) && it < 2 * X_ITEMS;
      const int im = it >= X_ITEMS, w = it - im * X_ITEMS, x_pos = w >> 1;
      xim[k] = im;
      xsrc[k] = ok ? x_pos * C_IN + ci_t * 16 + (w & 1) * 8 : -1;
      xdst[k] = im * kBufElems + ((x_pos / 14 + 2) * kTileW + (x_pos % 14 + 2)) * 16 + (w & 1) * 8;
    }
#pragma unroll
    for (int k = 0; k < DR; ++k) {
      const int row = pw < 3 ? 2 * pw + k : 6;
      const int it = row * 64 + lane;
      const bool ok = (pw < 3 || k == 0) && it < 2 * DY_ITEMS;
      const int im = it >= DY_ITEMS, w = it - im * DY_ITEMS;
      const int d_c8 = w % CO_CH, d_pp = w / CO_CH, r = d_pp / 7, c = d_pp % 7;
      const int slot = (((d_c8 >> 1) ^ dy_swz(2 * c)) << 1) | (d_c8 & 1);   // the same for the four window positions
      dim_[k] = im;
      dsrc[k] = ok ? d_pp * C_OUT + co_t * CO_W + d_c8 * 8 : -1;
      ddst[k] = im * kBufElems + (r * 32 + 2 * c) * CO_W + slot * 8;
    }
    u32x4 xr[XR], gv_r[DR];
    u32x2 ar_r[DR];
    auto prefetch = [&](int p, bool with_x = true) {
      const int b0 = b_begin + 2 * p;
      const int second = (b0 + 1 < b_end) ? 1 : 0;   // odd image count: the last round re-reads its only image
      const bf16* xg = x + (size_t)b0 * kXImg;
      const bf16* gg = gpool + (size_t)b0 * kDyImg;
      const uint8_t* ag = arg + (size_t)b0 * kDyImg;
#pragma unroll
      for (int k = 0; k < XR; ++k)
        if (with_x && xsrc[k] >= 0 && !DTF_ABLATED(32)) xr[k] = *reinterpret_cast<const u32x4*>(xg + xsrc[k] + (xim[k] & second) * kXImg);
#pragma unroll
      for (int k = 0; k < DR; ++k)
        if (dsrc[k] >= 0 && !DTF_ABLATED(128)) {
          const int off = dsrc[k] + (dim_[k] & second) * kDyImg;
          gv_r[k] = *reinterpret_cast<const u32x4*>(gg + off);
          ar_r[k] = *reinterpret_cast<const u32x2*>(ag + off);
        }
    };
    auto stage = [&](int p, bool with_x = true) {
      bf16* xp = xt + (p % kPcPairs) * 2 * kBufElems;
      bf16* dp = dyt + (p % kPcPairs) * 2 * kBufElems;
#pragma unroll
      for (int k = 0; k < XR; ++k)
        if (with_x && xsrc[k] >= 0 && !DTF_ABLATED(32)) *reinterpret_cast<u32x4*>(xp + xdst[k]) = xr[k];
#pragma unroll
      for (int k = 0; k < DR; ++k)
        if (dsrc[k] >= 0 && !DTF_ABLATED(128)) {
          u32x4 o[4];
          expand_pool_grad(gv_r[k], ar_r[k], o);
#pragma unroll
          for (int q = 0; q < 4; ++q)   // k row (q >> 1) * 16 + 2c + (q & 1): same swizzle slot (see dy_swz)
            *reinterpret_cast<u32x4*>(dp + ddst[k] + ((q >> 1) * 16 + (q & 1)) * CO_W) = o[q];
        }
    };
    if (rounds > 0) {   // round 0: the gradient tiles only, the (idle) MFMA waves copy the input tiles
      prefetch(0, false);
      stage(0, false);
      if (rounds > 1) prefetch(1);
    }
    pc_signal(staged + pw, 1u, lane);
    __syncthreads();   // the MFMA waves' zero fill is done; they find round 0 staged
    DTF_TIMING_ONLY(if (tb) tb[1] = wall_clock64();)
    for (int q = 1; q < rounds; ++q) {
      // pair q % 3 was last read in round q - 3
      DTF_LAP_START();
      if (q >= kPcPairs) pc_wait_all_ge(done, 7, (unsigned)(q - kPcPairs + 1), lane);
      DTF_LAP(t_w);
      DTF_LAP_AFTER("s_waitcnt vmcnt(0)", t_v);
      stage(q);
      if (q + 1 < rounds) prefetch(q + 1);
      pc_signal(staged + pw, (unsigned)(q + 1), lane);
      DTF_LAP_AFTER("s_waitcnt lgkmcnt(0)", t_s);
    }
    DTF_TIMING_ONLY(if (tb) { tb[2] = wall_clock64(); tb[5] = t_k; tb[6] = t_s; tb[7] = t_w; tb[4] = t_v; })
    __syncthreads();
    __syncthreads();
    DTF_TIMING_ONLY(if (tb) tb[3] = wall_clock64();)
    return;
  }

  // ---- MFMA waves ----
  const int g = lane >> 4, i16 = lane & 15;
  const int qr = i16 >> 2, pc = i16 & 3;
  {
    // round 0's input tiles (2 x 392 chunks of 16 bytes): requested before the zero fill, stored after it
    u32x4 x0[2];
    int dst0[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int it = tid + k * 512, im = it >= 392, w = it - im * 392, x_pos = w >> 1;
      dst0[k] = -1;
      if (it < 784 && b_begin + im < b_end && !DTF_ABLATED(32)) {
        dst0[k] = im * kBufElems + ((x_pos / 14 + 2) * kTileW + (x_pos % 14 + 2)) * 16 + (w & 1) * 8;
        x0[k] = *reinterpret_cast<const u32x4*>(x + ((size_t)(b_begin + im) * 196 + x_pos) * C_IN + ci_t * 16 + (w & 1) * 8);
      }
    }
    zero_fill();
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (dst0[k] >= 0) *reinterpret_cast<u32x4*>(xt + dst0[k]) = x0[k];
  }
  const int tr_pos = (g >> 1) * kTileW + 4 * (g & 1) + qr;   // A: row pair 0, tap (0,0), first read
  const int kkrow = (g >> 1) * 16 + 4 * (g & 1) + qr;        // B: k row of the first read
  const int bsw = dy_swz(kkrow);
  const bf16* a_lane = xt + tr_pos * 16 + pc * 4;
  const bf16* b_lane = dyt + kkrow * CO_W + pc * 4;
  const int role = (tw + kh) & 3;   // SIMD tw hosts roles tw and tw + 1: 14 + 12 tiles per k-step pair on every SIMD

  __syncthreads();   // zero fill done
  DTF_TIMING_ONLY(if (tb) tb[1] = wall_clock64();)
  DTF_TIMING_ONLY(t_v = clock64();)   // MFMA wave: slots 4 / 6 = shader-clock cycles at loop entry / exit
  // Everything from here to the slab write is instantiated per role (the chain shapes fix which register a tap reads
  // at which k-step; a run-time role would put the accumulators behind a join).
  // The k-loop is ONE software pipeline over all rounds: the fragments of the next round's first k-step roll in behind
  // the MFMAs of this round's last one, so a round boundary costs no fragment-latency bubble and no exposed poll -- the
  // progress words of round p + 1 are read during k-step 3 of round p and looked at after k-step 4.  Seven k-steps per
  // round flip the parity of the two B sets (hence the body instantiated for both) and leave the chains' slot rotation
  // where it started (6 = 0 mod 3 and mod 2).  The last round prefetches from its own buffers (never used): every round
  // issues the same instruction stream and the compiler's wait counts stay exact.
  auto run = [&](auto role_c) {
    constexpr int ROLE = decltype(role_c)::value;
    constexpr int NCH = wg_role_chains(ROLE);
    constexpr bool BIAS = ROLE == 0;   // slot 6: ones x dY = the column sums of the gradient tile = the bias gradient
    f32x4 acc[7][NT];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 af[7], bq[2][NT];
    const unsigned one2 = 0x3F803F80u;   // bf16 (1.0, 1.0)
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (u32x4){one2, one2, one2, one2});
    auto load_b = [&](const bf16* b_img, int r, bf16x8 (&bf)[NT]) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf16* p0 = b_img + r * 32 * CO_W + ((j ^ bsw) << 4);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 8 * CO_W));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bf[j] = __builtin_bit_cast(bf16x8, v);
      }
    };
    auto load_a = [&](const bf16* a_img, int y, int kx, int slot) {   // F(y, kx) -> af[slot]
      const bf16* p0 = a_img + (y * kTileW + kx) * 16;
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 8 * 16));
      s16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      af[slot] = __builtin_bit_cast(bf16x8, av);
    };
    int pair = 0;   // p % kPcPairs
    auto round_body = [&](auto par_c, int p) {
      constexpr int PAR = decltype(par_c)::value;
      // the two MFMA waves of a SIMD take turns at being preferred by its issue arbiter (oldest first otherwise: the
      // kh = 1 waves then fall 0.25 us behind per round and the whole workgroup waits for them at the end)
      if ((PAR ^ kh) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
      const bool has_next = p + 1 < rounds;
      const int npair = pair + 1 == kPcPairs ? 0 : pair + 1;
      const int boff = (pair * 2 + kh) * kBufElems;
      const int noff = ((has_next ? npair : pair) * 2 + kh) * kBufElems;
      const bf16* a_img = a_lane + boff;
      const bf16* b_img = b_lane + boff;
      const bf16* a_nxt = a_lane + noff;
      const bf16* b_nxt = b_lane + noff;
      const unsigned need = (unsigned)(has_next ? p + 2 : p + 1);
      unsigned seen = 0;
#pragma unroll
      for (int r = 0; r < 7; ++r) {
        bf16x8(&bf)[NT] = bq[(r + PAR) & 1];
        if (r == 5) {   // before the first read of the next round's buffers
          if (__builtin_amdgcn_ballot_w64(seen < need) != 0ull) {
            DTF_LAP(t_k);
            pc_wait_all_ge(staged, 3, need, lane);
            DTF_LAP(t_w);
          }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const WgChain ch = wg_role_chain(ROLE, c);
          const int base = wg_role_base(ROLE, c);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            if (j < ch.len) {
              const int slot = base + (r + j) % ch.len;   // holds F(2 r + ky0 + 2 j, kx)
              if DTF_ABLATED(64) {
                asm volatile("" ::"v"(af[slot]), "v"(bf[0]), "v"(bf[1]));
              } else {
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[base + j][n] = mfma16(af[slot], bf[n], acc[base + j][n]);
              }
              if (!DTF_ABLATED(1024)) {
                if (r < 6) {
                  // the chain's oldest fragment died with tap j = 0: its slot takes what the last tap reads next step
                  if (j == 0) load_a(a_img, 2 * r + ch.ky0 + 2 * ch.len, ch.kx, slot);
                } else {
                  load_a(a_nxt, ch.ky0 + 2 * j, ch.kx, slot);   // next round, k-step 0: slot (0 + j) % len = this one
                }
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        if (BIAS && !DTF_ABLATED(64)) {
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[6][n] = mfma16(ones, bf[n], acc[6][n]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!DTF_ABLATED(1024)) {
          if (r + 2 < 7) load_b(b_img, r + 2, bf);
          else load_b(b_nxt, r + 2 - 7, bf);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (r == 3) {
          seen = __hip_atomic_load(staged + (lane & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      pc_signal(done + wave, (unsigned)(p + 1), lane);   // every read of this round's buffers has been waited for by its MFMA
      pair = npair;
    };
    if (rounds > 0 && !DTF_ABLATED(256)) {
      pc_wait_all_ge(staged, 3, 1u, lane);
      DTF_LAP(t_w);
      const int boff = kh * kBufElems;
      load_b(b_lane + boff, 0, bq[0]);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const WgChain ch = wg_role_chain(ROLE, c);
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (j < ch.len) load_a(a_lane + boff, ch.ky0 + 2 * j, ch.kx, wg_role_base(ROLE, c) + j);
      }
      load_b(b_lane + boff, 1, bq[1]);
      __builtin_amdgcn_sched_barrier(0);
      // an odd image count leaves the kh = 1 waves without an image in the last round: they stop before it (the
      // producers never wait for a round's `done` words after the last three rounds)
      const int my_rounds = (b_end - b_begin - kh + 1) >> 1;
      for (int p = 0; p < my_rounds; p += 2) {
        round_body(std::integral_constant<int, 0>{}, p);
        if (p + 1 < my_rounds) round_body(std::integral_constant<int, 1>{}, p + 1);
      }
    }

    // ---- sum the two images' waves of a role through LDS; the kh = 0 wave writes the role's taps: rows ci, cols co ----
    DTF_LAP_AFTER("s_nop 0", t_k);
    DTF_TIMING_ONLY(t_s = clock64();)
    DTF_TIMING_ONLY(if (tb) { tb[2] = wall_clock64(); tb[5] = t_k; tb[6] = t_s; tb[7] = t_w; tb[4] = t_v; })
    DTF_TIMING_ONLY(if (tb_wg && lane == 0) { tb_wg[16 + wave] = wall_clock64(); tb_wg[24 + wave] = t_w; })
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);  // [role][7 * NT tiles][64 lanes]: one 16-byte store / load per tile
    if (kh == 1) {
#pragma unroll
      for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) red[(ROLE * 7 * NT + t * NT + j) * 64 + lane] = acc[t][j];
    }
    __syncthreads();
    DTF_TIMING_ONLY(if (tb) tb[3] = wall_clock64();)
    if (kh == 1) return;
    float* slab = part + (size_t)blockIdx.x * (25 * C_IN * C_OUT + C_OUT);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const WgChain ch = wg_role_chain(ROLE, c);
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (j < ch.len) {
          const int t = wg_role_base(ROLE, c) + j, tap = 5 * (ch.ky0 + 2 * j) + ch.kx;
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const f32x4 o = red[(ROLE * 7 * NT + t * NT + n) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ci = ci_t * 16 + 4 * g + e;
              const int co = co_t * CO_W + n * 16 + i16;
              slab[((size_t)tap * C_IN + ci) * C_OUT + co] = acc[t][n][e] + o[e];
            }
          }
        }
    }
    if (BIAS && ci_t == 0 && g == 0) {   // row 0 of the ones product: lanes 0..15, element 0
#pragma unroll
      for (int n = 0; n < NT; ++n)
        slab[25 * C_IN * C_OUT + co_t * CO_W + n * 16 + i16] = acc[6][n][0] + red[(ROLE * 7 * NT + 6 * NT + n) * 64 + lane][0];
    }
  };
  switch (role) {
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    default: run(std::integral_constant<int, 3>{}); break;
  }
}

}  // namespace dtf

// ---------------------------------------------------------------------------
// C ABI launchers
// ---------------------------------------------------------------------------
using namespace dtf;

#define DTF_LAUNCH_CHECK() (int)hipGetLastError()

#if DTF_TIMING
extern "C" int dtf_conv2_set_timing_buf(void* p) {
  long long* q = (long long*)p;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dtf::g_timing_buf), &q, sizeof(q));
}
#endif

template <int KTOT, int KC, int NC, int IMGS, int NW, int MODE, int MINW = 1, bool FUSE1 = false>
static int launch_igemm(const void* in, const void* in_arg, const void* wk,
                        const void* bias, void* out, void* out_arg, int B, hipStream_t s,
                        Conv1Fused c1f = Conv1Fused{}, Conv1WgradFused c1w = Conv1WgradFused{}) {
  constexpr int threads = IMGS * 2 * NW * 64;
  constexpr int KP = (KC % 64 == 0) ? KC + 8 : KC;
  // fused conv1: + zero-haloed input images + per-wave pooled-row staging (bf16 row + argmax row)
  constexpr size_t tile_lds = (size_t)IMGS * igemm_tile_pos(KC) * KP * 2;
  constexpr size_t fwd_extra = (size_t)IMGS * kC1ImgElems * 2 + (size_t)IMGS * 2 * NW * (1024 + 512);
  // fused conv1 wgrad: dx + shifted image copies + half a gradient tile, aliased over the (dead) conv2 tile
  constexpr size_t c1w_lds = (size_t)IMGS * 196 * NC * 2 + kCpBytes + (size_t)kFusedC1wDyRows * kDy1RowBytes;
  // the epilogue stages the results in the (dead) tile: [IMGS][49][NC] bf16 + argmax bytes forward, [IMGS][196][NC] bf16
  // dgrad -- with a K chunk narrower than the layer's N that can be MORE than the tile (wide dgrad, 64-channel chunks:
  // 47 KiB of tile, 49 KiB of dx)
  constexpr size_t epi_lds = MODE == MODE_FWD ? (size_t)IMGS * 49 * NC * 3 : (size_t)IMGS * 196 * NC * 2;
  constexpr size_t plain_lds = tile_lds > epi_lds ? tile_lds : epi_lds;
  constexpr size_t lds = !FUSE1 ? plain_lds : (MODE == MODE_FWD ? tile_lds + fwd_extra : (tile_lds > c1w_lds ? tile_lds : c1w_lds));
  auto kern = conv5x5_igemm<KTOT, KC, NC, IMGS, NW, MODE, MINW, FUSE1>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  int grid = (B + IMGS - 1) / IMGS;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, (const bf16*)in, (const uint8_t*)in_arg,
                     (const bf16*)wk, (const float*)bias, (bf16*)out,
                     (uint8_t*)out_arg, B, c1f, c1w);
  return DTF_LAUNCH_CHECK();
}

template <int C_IN, int C_OUT, int NT>
static int launch_wgrad(const void* x, const void* gpool, const void* arg, void* part, int B, int G, hipStream_t s) {
  auto kern = conv5x5_wgrad<C_IN, C_OUT, NT>;
  constexpr int lds = wg_lds_bytes(NT);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const int ipb = (B + G - 1) / G;
  hipLaunchKernelGGL(kern, dim3(G, (C_IN / 16) * (C_OUT / (16 * NT))), dim3(512), lds, s, (const bf16*)x, (const bf16*)gpool,
                     (const uint8_t*)arg, (float*)part, B, ipb);
  return DTF_LAUNCH_CHECK();
}

template <int C_IN, int C_OUT>
static int launch_wgrad_pc(const void* x, const void* gpool, const void* arg, void* part, int B, int G, hipStream_t s) {
  auto kern = conv5x5_wgrad_pc<C_IN, C_OUT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kPcLdsBytes);
    attr_set = true;
  }
  const int ipb = (B + G - 1) / G;
  hipLaunchKernelGGL(kern, dim3(G, (C_IN / 16) * (C_OUT / 32)), dim3(768), kPcLdsBytes, s, (const bf16*)x, (const bf16*)gpool,
                     (const uint8_t*)arg, (float*)part, B, ipb);
  return DTF_LAUNCH_CHECK();
}

static int conv2_small_batch() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("DTF_IMGS1_MAX_B");
    v = (e && *e) ? atoi(e) : 640;
    if (v < 0) v = 0;
  }
  return v;
}
extern "C" {

// packed (2 * 25*cin*cout bf16) = [fwd pack | dgrad pack] of the bf16 HWIO weights w
int dtf_conv2_pack_weights(const void* w, void* packed, int cin, int cout, hipStream_t s) {
  if ((cin % 32) || (cout % 32)) return -3;
  hipLaunchKernelGGL(pack_conv2_weights, dim3((25 * cin * cout + 255) / 256), dim3(256), 0, s, (const bf16*)w,
                     (bf16*)packed, cin, cout);
  return DTF_LAUNCH_CHECK();
}

// Forward: x[B,14,14,cin] -> pooled[B,7,7,cout], arg[B,7,7,cout]; wT = fwd pack (dtf_conv2_pack_weights)
int dtf_conv2_fwd(const void* x, const void* wT, const void* bias, void* pooled, void* arg, int B,
                  int cin, int cout, hipStream_t s) {
  if (cin == 32 && cout == 64)
    return launch_igemm<DTF_FWD_CFG>(x, nullptr, wT, bias, pooled, arg, B, s);
  if (cin == 128 && cout == 256)
    return launch_igemm<DTF_WIDE_FWD_CFG>(x, nullptr, wT, bias, pooled, arg, B, s);
  return -1;
}

// Fused forward of both conv blocks of the 1 -> 32 -> 64 model (see Conv1Fused): images[n][784] (+ idx gather,
// or the on-device batch sampler when state != NULL: rows from the sample cursor, published in idx_out)
// -> p1/arg1 [B,14,14,32] and p2/arg2 [B,7,7,64].
int dtf_conv12_fwd(const void* images, const void* idx, void* idx_out, const void* state, unsigned n, unsigned hb,
                   unsigned seed, int shuffle, const void* w1, const void* b1, void* p1, void* arg1, const void* wT,
                   const void* bias2, void* p2, void* arg2, int B, hipStream_t s) {
  if (state && (!idx_out || n == 0 || (1ull << (2 * hb)) < n)) return -3;
  Conv1Fused c{(const bf16*)images, (const int*)idx, SampleArgs{(const int64_t*)state, (int*)idx_out, n, hb, seed, shuffle},
               (const float*)w1, (const float*)b1, (bf16*)p1, (uint8_t*)arg1};
  // up to conv2_small_batch() images: ONE image per 8-wave workgroup (4-way n split), twice the workgroups and half
  // the serial work of each -- at batch 100 the two-image kernel is 50 workgroups on 256 CUs, 17 us of latency chain.
  // Measured with the matching dgrad launch (step, one device): batch 32..256 -11..-13 %, 512 -4 %, 640 -1.4 %,
  // 768 +-0, 1024 +5 % (two images per workgroup share the B fragments of a step)
  if (B <= conv2_small_batch())
    return launch_igemm<32, 32, 64, 1, 4, MODE_FWD, 4, true>(nullptr, nullptr, wT, bias2, p2, arg2, B, s, c);
  return launch_igemm<32, 32, 64, 2, 2, MODE_FWD, 4, true>(nullptr, nullptr, wT, bias2, p2, arg2, B, s, c);
}

// dgrad: gpool[B,7,7,cout] (+arg code) -> dx[B,14,14,cin]; w = dgrad pack (second half of dtf_conv2_pack_weights)
int dtf_conv2_dgrad(const void* gpool, const void* arg, const void* w, void* dx,
                    int B, int cin, int cout, hipStream_t s) {
  if (cin == 32 && cout == 64)
    return launch_igemm<DTF_DGRAD_CFG>(gpool, arg, w, nullptr, dx, nullptr, B, s);
  if (cin == 128 && cout == 256)
    return launch_igemm<DTF_WIDE_DGRAD_CFG>(gpool, arg, w, nullptr, dx, nullptr, B, s);
  return -1;
}

// Fused conv2 dgrad -> conv1 filter/bias gradient of the 1 -> 32 -> 64 model (see Conv1WgradFused): dx is
// never written.  part1 holds G slabs of 26 * 32 floats, one per workgroup: G = (B + 1) / 2 selects two images per
// workgroup, G = B one image per workgroup (4 waves, 2-way n split: the small-batch launch, see dtf_conv12_fwd).
int dtf_conv2_dgrad_conv1_wgrad_g(const void* gpool, const void* arg2, const void* w, const void* images,
                                  const void* idx, const void* arg1, void* part1, int B, int G, hipStream_t s) {
  if (!idx) return -3;
  Conv1WgradFused c{(const bf16*)images, (const int*)idx, (const uint8_t*)arg1, (float*)part1};
  if (G == B && B > 1)
    return launch_igemm<64, 32, 32, 1, 2, MODE_DGRAD, 3, true>(gpool, arg2, w, nullptr, nullptr, nullptr, B, s, Conv1Fused{}, c);
  if (G != (B + 1) / 2) return -3;
  return launch_igemm<64, 32, 32, 2, 1, MODE_DGRAD, 3, true>(gpool, arg2, w, nullptr, nullptr, nullptr, B, s, Conv1Fused{}, c);
}
int dtf_conv2_dgrad_conv1_wgrad(const void* gpool, const void* arg2, const void* w, const void* images,
                                const void* idx, const void* arg1, void* part1, int B, hipStream_t s) {
  return dtf_conv2_dgrad_conv1_wgrad_g(gpool, arg2, w, images, idx, arg1, part1, B, (B + 1) / 2, s);
}
// batch size up to which the fused kernels of the 32/64 model take one image per workgroup (DTF_IMGS1_MAX_B overrides;
// engine.py reads the same variable to size the conv1 slabs)
int dtf_conv2_small_batch(void) { return conv2_small_batch(); }

// wgrad partial slabs: part [G][25*cin*cout + cout].  nt = 16-wide co tiles per wave of conv5x5_wgrad (2 or 4);
// 0 = the default launch: the producer / consumer kernel for the 32/64 model (DTF_C2W_PC=0: conv5x5_wgrad<2>),
// conv5x5_wgrad<4> for the widened one (32 output tiles x G); DTF_C2W_NT=2|4 forces conv5x5_wgrad (tools).
int dtf_conv2_wgrad_nt(const void* x, const void* gpool, const void* arg, void* part, int B, int cin, int cout, int G,
                       int nt, hipStream_t s) {
  if (G <= 0 || B <= 0) return -2;
  if (nt == 0) {
    static int forced = -1, pc = -1;
    if (forced < 0) {
      const char* v = getenv("DTF_C2W_NT");
      forced = v ? atoi(v) : 0;
      v = getenv("DTF_C2W_PC");
      pc = (v && *v) ? atoi(v) : 1;
    }
    if (forced == 2 || forced == 4) nt = forced;
    else if (cin == 32 && cout == 64 && pc) return launch_wgrad_pc<32, 64>(x, gpool, arg, part, B, G, s);
    else if (cin == 128 && cout == 256 && pc) return launch_wgrad_pc<128, 256>(x, gpool, arg, part, B, G, s);
    else nt = cin >= 128 ? 4 : 2;
  }
  if (nt != 2 && nt != 4) return -3;
  if (cin == 32 && cout == 64)
    return nt == 4 ? launch_wgrad<32, 64, 4>(x, gpool, arg, part, B, G, s) : launch_wgrad<32, 64, 2>(x, gpool, arg, part, B, G, s);
  if (cin == 128 && cout == 256)
    return nt == 4 ? launch_wgrad<128, 256, 4>(x, gpool, arg, part, B, G, s)
                   : launch_wgrad<128, 256, 2>(x, gpool, arg, part, B, G, s);
  return -1;
}
int dtf_conv2_wgrad(const void* x, const void* gpool, const void* arg, void* part,
                    int B, int cin, int cout, int G, hipStream_t s) {
  return dtf_conv2_wgrad_nt(x, gpool, arg, part, B, cin, cout, G, 0, s);
}

}  // extern "C"
