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
//   M = 32 input channels, N = 64 output channels, K = positions.  The X tile
//   is [pos][ci] in LDS (K-slow), so A fragments come from the gfx950
//   transposing LDS read (ds_read_b64_tr_b16); the dY fragment is built in
//   registers straight from the pooled gradient + argmax (never expanded in
//   memory).  Split-K over images across workgroups; partial slabs are summed
//   by reduce_partials (optim.hip).
#include "common.hpp"

// Timing-ablation hook (tools/conv_ablate.py builds diagnostic copies with -DDTF_ABLATE=<bits>;
// the shipped library always has 0): 1 = no B global loads in the k-loop, 2 = no A LDS reads,
// 4 = no MFMAs, 8 = no staging loads, 16 = no epilogue stores.
#ifndef DTF_ABLATE
#define DTF_ABLATE 0
#endif

namespace dtf {

constexpr int kTileW = 18;          // 14 + 2*2 halo
constexpr int kTilePos = 328;       // 18*18 = 324 (+4 slack for the dummy window's reads)

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

// ----------------------------------------------------------------------------
// Forward / dgrad
//   in   MODE_FWD  : x    [B,14,14,KC] bf16
//        MODE_DGRAD: gpool[B,7,7,KC] bf16 (gradient wrt the pooled output),
//                    arg  [B,7,7,KC] u8, pooled [B,7,7,KC] bf16 (fwd output)
//   wk   [25][NC][KC] bf16  (K contiguous; fwd: transposed shadow of HWIO,
//                            dgrad: the HWIO tensor itself, taps read flipped)
//   out  MODE_FWD  : pooled [B,7,7,NC] bf16 + argmax [B,7,7,NC] u8
//        MODE_DGRAD: dx [B,14,14,NC] bf16
// ----------------------------------------------------------------------------
// KTOT = reduction channels of the layer, KC = channels resident in LDS at a time (KTOT is
// walked in KC chunks when the full tile would not fit the 160 KiB LDS).
template <int KTOT, int KC, int NC, int IMGS, int NW, int MODE, int MINW>
__global__ __launch_bounds__(IMGS * 2 * NW * 64, MINW) void conv5x5_igemm(
    const bf16* __restrict__ in, const uint8_t* __restrict__ in_arg,
    const bf16* __restrict__ in_pooled, const bf16* __restrict__ wk,
    const float* __restrict__ bias, bf16* __restrict__ out, uint8_t* __restrict__ out_arg,
    int B) {
  constexpr int NTHREADS = IMGS * 2 * NW * 64;
  constexpr int NT = NC / (16 * NW);   // n-tiles per wave
  constexpr int KSTEPS = KC / 32;      // k-steps per tap
  constexpr int CH8 = KC / 8;          // 16-byte chunks per position
  constexpr int KP = (KC % 64 == 0) ? KC + 8 : KC;  // LDS position pitch: +16 B turns the 4-way b128 conflict of 128-B pitches into 2-way
  static_assert(NC % (16 * NW) == 0, "NC must split evenly over NW waves");
  static_assert(KC % 32 == 0 && KTOT % KC == 0, "KC must be a multiple of 32 dividing KTOT");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* tile = reinterpret_cast<bf16*>(smem);  // [IMGS][kTilePos][KP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int w_img = wave / (2 * NW);
  const int w_h = (wave / NW) & 1;
  const int w_n = wave % NW;
  const int img0 = blockIdx.x * IMGS;

  // ---- zero the whole tile (halo + slack), then fill the interior ----
  {
    u32x4 z = {0u, 0u, 0u, 0u};
    u32x4* t4 = reinterpret_cast<u32x4*>(smem);
    constexpr int N16 = IMGS * kTilePos * KP * 2 / 16;
    for (int i = tid; i < N16; i += NTHREADS) t4[i] = z;
  }
  // lane -> (row m of the 16-row tile, k-group g)
  const int m = lane & 15, g = lane >> 4;
  const int wl = m >> 2, q = m & 3;
  // position of this lane's A row for m-tile 0 (row pair 0), tap (0,0), in tile coordinates
  const int a_pos = (q >> 1) * kTileW + 2 * (4 * w_h + wl) + (q & 1);
  const bf16* a_base = tile + ((size_t)w_img * kTilePos + a_pos) * KP + g * 8;

  f32x4 acc[7][NT];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int kc0 = 0; kc0 < KTOT; kc0 += KC) {
  __syncthreads();  // zero fill done / previous chunk fully consumed
  if (MODE == MODE_FWD) {
    constexpr int NCHUNK = IMGS * 196 * CH8;
    for (int c = tid; c < NCHUNK; c += NTHREADS) {
      int ch = c % CH8;
      int pos = (c / CH8) % 196;
      int im = c / (CH8 * 196);
      int b = img0 + im;
      if (b < B && !(DTF_ABLATE & 8)) {
        u32x4 v = *reinterpret_cast<const u32x4*>(in + ((size_t)b * 196 + pos) * KTOT + kc0 + ch * 8);
        int y = pos / 14, x = pos % 14;
        int tp = (y + 2) * kTileW + (x + 2);
        *reinterpret_cast<u32x4*>(tile + ((size_t)im * kTilePos + tp) * KP + ch * 8) = v;
      }
    }
  } else {
    // max-pool / ReLU gradient: route g to the saved argmax position.
    constexpr int NCHUNK = IMGS * 49 * CH8;
    for (int c = tid; c < NCHUNK; c += NTHREADS) {
      int ch = c % CH8;
      int pp = (c / CH8) % 49;
      int im = c / (CH8 * 49);
      int b = img0 + im;
      if (b < B) {
        size_t off = ((size_t)b * 49 + pp) * KTOT + kc0 + ch * 8;
        bf16x8 gq = *reinterpret_cast<const bf16x8*>(in + off);
        bf16x8 p = *reinterpret_cast<const bf16x8*>(in_pooled + off);
        u32x2 a2 = *reinterpret_cast<const u32x2*>(in_arg + off);
        int py = pp / 7, px = pp % 7;
        bf16x8 o[4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          unsigned a = ((e < 4 ? a2[0] : a2[1]) >> ((e & 3) * 8)) & 0xffu;
          bool live = (float)p[e] > 0.f;
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) o[qq][e] = (live && a == (unsigned)qq) ? gq[e] : (bf16)0.f;
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          int tp = (2 * py + (qq >> 1) + 2) * kTileW + (2 * px + (qq & 1) + 2);
          *reinterpret_cast<bf16x8*>(tile + ((size_t)im * kTilePos + tp) * KP + ch * 8) = o[qq];
        }
      }
    }
  }
  __syncthreads();

  // ---- main loop over this chunk ----
  // B fragment: column n = lane&15 of n-tile nt, k = kc0 + 8g..8g+7
  const bf16* b_base = wk + ((size_t)(w_n * NT * 16 + m)) * KTOT + kc0 + g * 8;

  bf16x8 bcur[NT], bnxt[NT];
  auto load_b = [&](bf16x8 (&dst)[NT], int step) {
    int tap = step / KSTEPS, ks = step % KSTEPS;
    int wtap = (MODE == MODE_DGRAD) ? (24 - tap) : tap;
#pragma unroll
    for (int j = 0; j < NT; ++j)
      dst[j] = *reinterpret_cast<const bf16x8*>(b_base + ((size_t)wtap * NC + j * 16) * KTOT + ks * 32);
  };
  load_b(bcur, 0);
  constexpr int NSTEPS = 25 * KSTEPS;
#pragma unroll 1
  for (int step = 0; step < NSTEPS; ++step) {
    if (!(DTF_ABLATE & 1) && step + 1 < NSTEPS) load_b(bnxt, step + 1);
    const int tap = step / KSTEPS, ks = step % KSTEPS;
    const int ky = tap / 5, kx = tap % 5;
    const bf16* ap = a_base + (ky * kTileW + kx) * KP + ks * 32;
    bf16x8 a[7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
      a[i] = *reinterpret_cast<const bf16x8*>((DTF_ABLATE & 2) ? a_base + i * 8 : ap + i * (2 * kTileW * KP));
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (DTF_ABLATE & 4) { asm volatile("" ::"v"(a[i]), "v"(bcur[j])); }
        else acc[i][j] = mfma16(a[i], bcur[j], acc[i][j]);
      }
    if (!(DTF_ABLATE & 1)) {
#pragma unroll
      for (int j = 0; j < NT; ++j) bcur[j] = bnxt[j];
    }
  }
  }  // K chunks

  // ---- epilogue ----
  const int b = img0 + w_img;
  if (b >= B) return;
  // C/D map: lane holds rows 4*(lane>>4)+r -> window (lane>>4), q = r; col = lane&15
  const int ewin = 4 * w_h + g;  // window index along x (0..7; 7 is the dummy window)
  if (ewin >= 7) return;
  if (MODE == MODE_FWD) {
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
        size_t off = (((size_t)b * 7 + i) * 7 + ewin) * NC + n;
        if ((DTF_ABLATE & 16) && o != -1.f) continue;
        out[off] = (bf16)o;
        out_arg[off] = (uint8_t)bi;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = w_n * NT * 16 + j * 16 + m;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        f32x4 v = acc[i][j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int y = 2 * i + (r >> 1), x = 2 * ewin + (r & 1);
          out[(((size_t)b * 14 + y) * 14 + x) * NC + n] = (bf16)v[r];
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------
// wgrad.  grid = (G image chunks, 5 tap rows ky, (C_IN/32)*(C_OUT/64) tiles)
//   x      [B,14,14,C_IN] bf16   layer input
//   gpool  [B,7,7,C_OUT]  bf16   gradient wrt pooled output
//   arg    [B,7,7,C_OUT]  u8     saved argmax
//   pooled [B,7,7,C_OUT]  bf16   forward pooled output (ReLU mask = pooled > 0)
//   part   [G][25*C_IN*C_OUT + C_OUT] fp32 partial slabs (HWIO order, then bias)
// Per image the block stages two LDS images: the halo-padded input tile
// [pos][32 ci] and the max-pool/ReLU-expanded output gradient [row pair][k][64 co]
// (k = 16*dy + x, the K order of a row pair).  Both are K-slow, so A and B
// fragments both come from ds_read_b64_tr_b16; 32-byte sub-slots are XOR
// swizzled so every transposed read is bank-conflict free.
// 4 waves = (mi: 16 input channels) x (kh: alternate row pairs); a wave owns
// 5 taps x 4 n-tiles = 20 accumulator tiles, so each A fragment feeds 4 MFMAs
// and each B fragment 5.  The two kh halves are summed through LDS at the end.
// Element j of lane group g  <->  dy = g>>1, x = 4*(g&1) + (j&3) + 8*(j>>2).
// ----------------------------------------------------------------------------
constexpr int kWgXTileBytes = kTilePos * 32 * 2;   // 20992
constexpr int kWgDyTileBytes = 7 * 32 * 64 * 2;    // 28672

template <int C_IN, int C_OUT>
__global__ __launch_bounds__(256) void conv5x5_wgrad(
    const bf16* __restrict__ x, const bf16* __restrict__ gpool,
    const uint8_t* __restrict__ arg, const bf16* __restrict__ pooled,
    float* __restrict__ part, int B, int imgs_per_block) {
  constexpr int CI_T = C_IN / 32;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* xt = reinterpret_cast<bf16*>(smem);                  // [328][32], halves swizzled
  bf16* dyt = reinterpret_cast<bf16*>(smem + kWgXTileBytes);  // [7][32][64], 32-B slots swizzled

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mi = wave & 1, kh = wave >> 1;
  const int ky = blockIdx.y;
  const int ci_t = blockIdx.z % CI_T, co_t = blockIdx.z / CI_T;
  const int g = lane >> 4, i16 = lane & 15;
  const int qr = i16 >> 2, pc = i16 & 3;
  const bool do_bias = (ky == 0) && (ci_t == 0) && (mi == 0);

  const int b_begin = blockIdx.x * imgs_per_block;
  const int b_end = min(B, b_begin + imgs_per_block);

  f32x4 acc[5][4];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  // zero once: halo, slack and the x = 14,15 dummy columns stay zero
  for (int i = tid; i < (kWgXTileBytes + kWgDyTileBytes) / 16; i += 256)
    reinterpret_cast<u32x4*>(smem)[i] = (u32x4){0u, 0u, 0u, 0u};

  const int tr_pos = (ky + (g >> 1)) * kTileW + 4 * (g & 1) + qr;  // A: row pair 0, kx 0, first read
  const int kkrow = (g >> 1) * 16 + 4 * (g & 1) + qr;              // B: k row of the first read
  const int bsw = (kkrow >> 1) & 3;                                // its slot swizzle (same for +8)

  for (int b = b_begin; b < b_end; ++b) {
    __syncthreads();  // previous image fully consumed (and the zero fill done)
    for (int c = tid; c < 196 * 4; c += 256) {
      int ch = c & 3, pos = c >> 2;
      u32x4 v = *reinterpret_cast<const u32x4*>(x + ((size_t)b * 196 + pos) * C_IN + ci_t * 32 + ch * 8);
      int y = pos / 14, xx = pos % 14;
      int tp = (y + 2) * kTileW + (xx + 2);
      int chs = ch ^ (((tp >> 2) & 1) << 1);  // swap 32-byte halves on odd 4-position groups
      *reinterpret_cast<u32x4*>(xt + tp * 32 + chs * 8) = v;
    }
    for (int c = tid; c < 49 * 8; c += 256) {
      int c8 = c & 7, pp = c >> 3;
      int r = pp / 7, w = pp % 7;
      size_t off = ((size_t)b * 49 + pp) * C_OUT + co_t * 64 + c8 * 8;
      bf16x8 gv = *reinterpret_cast<const bf16x8*>(gpool + off);
      bf16x8 pv = *reinterpret_cast<const bf16x8*>(pooled + off);
      u32x2 a2 = *reinterpret_cast<const u32x2*>(arg + off);
      bf16x8 o[4];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        unsigned a = ((e < 4 ? a2[0] : a2[1]) >> ((e & 3) * 8)) & 0xffu;
        bool live = (float)pv[e] > 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q][e] = (live && a == (unsigned)q) ? gv[e] : (bf16)0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int kk = (q >> 1) * 16 + 2 * w + (q & 1);
        int slot = (((c8 >> 1) ^ ((kk >> 1) & 3)) << 1) | (c8 & 1);
        *reinterpret_cast<bf16x8*>(dyt + (r * 32 + kk) * 64 + slot * 8) = o[q];
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int r = (kh + b) & 1; r < 7; r += 2) {
      bf16x8 bfrag[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16* p0 = dyt + (r * 32 + kkrow) * 64 + ((j ^ bsw) << 4) + pc * 4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 8 * 64));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bfrag[j] = __builtin_bit_cast(bf16x8, v);
      }
      if (do_bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[j] += (float)bfrag[j][e];
      }
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) {
        const int p0 = tr_pos + 2 * r * kTileW + kx;
        const int p1 = p0 + 8;
        const int h0 = mi ^ ((p0 >> 2) & 1), h1 = mi ^ ((p1 >> 2) & 1);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xt + p0 * 32 + h0 * 16 + pc * 4));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xt + p1 * 32 + h1 * 16 + pc * 4));
        s16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bf16x8 afrag = __builtin_bit_cast(bf16x8, av);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[kx][j] = mfma16(afrag, bfrag[j], acc[kx][j]);
      }
    }
  }

  // ---- sum the two kh halves through LDS, then write the slab ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);  // [mi][84][64 lanes]
  if (kh == 1) {
#pragma unroll
    for (int kx = 0; kx < 5; ++kx)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(mi * 84 + (kx * 4 + j) * 4 + r) * 64 + lane] = acc[kx][j][r];
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(mi * 84 + 80 + j) * 64 + lane] = bsum[j];
  }
  __syncthreads();
  if (kh == 1) return;
  float* slab = part + (size_t)blockIdx.x * (25 * C_IN * C_OUT + C_OUT);
#pragma unroll
  for (int kx = 0; kx < 5; ++kx)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int ci = ci_t * 32 + mi * 16 + 4 * g + r;
        int co = co_t * 64 + j * 16 + i16;
        slab[((size_t)(ky * 5 + kx) * C_IN + ci) * C_OUT + co] =
            acc[kx][j][r] + red[(mi * 84 + (kx * 4 + j) * 4 + r) * 64 + lane];
      }
  if (do_bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = bsum[j] + red[(mi * 84 + 80 + j) * 64 + lane];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (g == 0) slab[25 * C_IN * C_OUT + co_t * 64 + j * 16 + i16] = s;
    }
  }
}

}  // namespace dtf

// ---------------------------------------------------------------------------
// C ABI launchers
// ---------------------------------------------------------------------------
using namespace dtf;

#define DTF_LAUNCH_CHECK() (int)hipGetLastError()

template <int KTOT, int KC, int NC, int IMGS, int NW, int MODE, int MINW = 1>
static int launch_igemm(const void* in, const void* in_arg, const void* in_pooled, const void* wk,
                        const void* bias, void* out, void* out_arg, int B, hipStream_t s) {
  constexpr int threads = IMGS * 2 * NW * 64;
  constexpr int KP = (KC % 64 == 0) ? KC + 8 : KC;
  constexpr size_t lds = (size_t)IMGS * kTilePos * KP * 2;
  auto kern = conv5x5_igemm<KTOT, KC, NC, IMGS, NW, MODE, MINW>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  int grid = (B + IMGS - 1) / IMGS;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, (const bf16*)in, (const uint8_t*)in_arg,
                     (const bf16*)in_pooled, (const bf16*)wk, (const float*)bias, (bf16*)out,
                     (uint8_t*)out_arg, B);
  return DTF_LAUNCH_CHECK();
}

extern "C" {

// Forward: x[B,14,14,cin] -> pooled[B,7,7,cout], arg[B,7,7,cout]; wT [25][cout][cin]
int dtf_conv2_fwd(const void* x, const void* wT, const void* bias, void* pooled, void* arg, int B,
                  int cin, int cout, hipStream_t s) {
  if (cin == 32 && cout == 64)
    return launch_igemm<32, 32, 64, 1, 2, MODE_FWD, 4>(x, nullptr, nullptr, wT, bias, pooled, arg, B, s);
  if (cin == 128 && cout == 256)
    return launch_igemm<128, 128, 256, 1, 4, MODE_FWD>(x, nullptr, nullptr, wT, bias, pooled, arg, B, s);
  return -1;
}

// dgrad: gpool[B,7,7,cout] (+arg, pooled) -> dx[B,14,14,cin]; w is HWIO [25][cin][cout]
int dtf_conv2_dgrad(const void* gpool, const void* arg, const void* pooled, const void* w, void* dx,
                    int B, int cin, int cout, hipStream_t s) {
  if (cin == 32 && cout == 64)
    return launch_igemm<64, 32, 32, 1, 2, MODE_DGRAD, 4>(gpool, arg, pooled, w, nullptr, dx, nullptr, B, s);
  if (cin == 128 && cout == 256)
    return launch_igemm<256, 128, 128, 1, 2, MODE_DGRAD>(gpool, arg, pooled, w, nullptr, dx, nullptr, B, s);
  return -1;
}

// wgrad partial slabs: part [G][25*cin*cout + cout]
int dtf_conv2_wgrad(const void* x, const void* gpool, const void* arg, const void* pooled, void* part,
                    int B, int cin, int cout, int G, hipStream_t s) {
  if (G <= 0) return -2;
  int ipb = (B + G - 1) / G;
  if (cin == 32 && cout == 64) {
    hipLaunchKernelGGL((conv5x5_wgrad<32, 64>), dim3(G, 5, 1), dim3(256), kWgXTileBytes + kWgDyTileBytes, s, (const bf16*)x,
                       (const bf16*)gpool, (const uint8_t*)arg, (const bf16*)pooled, (float*)part, B, ipb);
    return DTF_LAUNCH_CHECK();
  }
  if (cin == 128 && cout == 256) {
    hipLaunchKernelGGL((conv5x5_wgrad<128, 256>), dim3(G, 5, 16), dim3(256), kWgXTileBytes + kWgDyTileBytes, s, (const bf16*)x,
                       (const bf16*)gpool, (const uint8_t*)arg, (const bf16*)pooled, (float*)part, B, ipb);
    return DTF_LAUNCH_CHECK();
  }
  return -1;
}

}  // extern "C"
