// First conv block in exact fp32 (--dtype fp32): 5x5 SAME conv with ONE input channel + bias + ReLU +
// 2x2 max-pool + argmax code (reference .idea/MNISTDist.py:71-73), and its filter/bias gradient.
//
// With C_in = 1 the layer has K = 25: at fp32 the matrix cores run at exactly the (packed) vector rate, 64 FLOP/clk/SIMD.
//   * forward: implicit GEMM on v_mfma_f32_16x16x4_f32 with K padded to 28 (see the kernel): what the matrix core
//     saves is instruction issue, not FLOPs; conv + bias + ReLU + pool + argmax never leave registers.
//   * wgrad: the pooled gradient reaches exactly ONE of the four positions of a window (the saved argmax), so
//     dW[tap][c] += g * x[argmax position + tap] is 25 FMAs per (window, channel) instead of 100 -- the
//     max-pool/ReLU gradient is never expanded.  Lanes gather their 25 pixels from LDS at their own argmax
//     position (at most 4 distinct addresses per half-wave; the 36-float row pitch keeps them on different
//     banks).  Per-workgroup [26][C] slabs (25 taps + the bias row), summed by reduce_segments / the fused tail:
//     deterministic, no atomics.
#include "conv1_core.hpp"
#include "f32_common.hpp"

namespace dtf {

constexpr int kC1fPitch = 36;                 // floats per zero-haloed image row (columns 2..29 hold pixels)
constexpr int kC1fElems = 32 * kC1fPitch;     // rows 2..29 hold pixels

// pixels 4*i4 .. 4*i4+3 of the 28x28 image -> the haloed LDS image (8-byte aligned: two b64 stores)
__device__ __forceinline__ void c1f_store_px(float* img, int i4, const f32x4& v) {
  const int y = i4 / 7, x = (i4 % 7) * 4;
  float* d = img + (y + 2) * kC1fPitch + 2 + x;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  *reinterpret_cast<f32x2*>(d) = (f32x2){v[0], v[1]};
  *reinterpret_cast<f32x2*>(d + 2) = (f32x2){v[2], v[3]};
}

// grid = (min(B, ..), C/32, NS), 256 threads = 4 waves.  w: fp32 HWIO [25][C].  gridDim.z = NS workgroups share an
// image's 49 M tiles (small batches: one workgroup per image is 100 workgroups on 256 CUs at batch 100).
//
// On the matrix cores (v_mfma_f32_16x16x4_f32), K = 25 taps padded to 28: at fp32 they are no faster than packed VALU
// FMAs (64 FLOP/clk/SIMD either way), but an MFMA needs ONE instruction per 2048 FLOP where the stencil needs one per
// 256 and a patch of 36 registers.  Measured against the packed-FMA stencil it replaced (lane = two channels, 100
// v_pk_fma_f32 per window; itself 29.0 -> 24.2 us against the one-channel-per-lane stencil): kernel 24.2 -> 21.7 us,
// step unchanged at 1024 images and -1.2 % at 100 -- 10.7 us of matrix pipe per SIMD instead of 14 us of VALU issue,
// and a workgroup's life (sampler, image fetch, 13 tiles per wave, stores) that neither version overlaps any better
// with fewer, fatter workgroups (grids of 768 / 512 / 256: +0 / +1 / +6 us).  M = the image's 196 pooling windows x 4 positions, rows m = 4 * window + q
// as in the conv2 kernels, so the four accumulator registers of a lane are one window: 49 M tiles, no padding.
//   A fragment (step s): lane (row i, k group g) reads pixel [position of row i + tap 4s + g] from the zero-haloed LDS
//     image -- one ds_read_b32 per step and tile, its tap offset a per-lane constant; taps 25..27 read any pixel
//     (their weights are zero).
//   B fragments: w[tap 4s + g][channel], 7 steps x 2 n tiles = 14 registers, loaded once per workgroup.
// A tile is 7 reads + 14 MFMAs (448 cycles) + ~25 VALU instructions of epilogue; tile t of an image goes to wave
// (t + blockIdx.x) & 3, so the co-resident workgroups of a CU do not all leave their 13th tile on the same SIMD.
__global__ __launch_bounds__(256) void conv1_pool_fwd_f32(const float* __restrict__ images, const int* __restrict__ idx,
                                                          SampleArgs smp, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ pooled,
                                                          uint8_t* __restrict__ arg, int B, int C) {
  __shared__ __attribute__((aligned(16))) float img[kC1fElems];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int c0 = blockIdx.y * 32 + j;
  const int wslot = (wave + (int)blockIdx.x) & 3;

  const SampleCursor cur = sample_cursor(smp);
  float bw[7][2];
  int tapoff[7];
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int k = 4 * s + g;
    tapoff[s] = k < 25 ? (k / 5) * kC1fPitch + k % 5 : 0;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bw[s][nt] = k < 25 ? w[k * C + c0 + 16 * nt] : 0.f;
  }
  const float bv[2] = {bias[c0], bias[c0 + 16]};
  // row i = lane & 15 of a tile: window 4t + (i >> 2), position q = i & 3
  const int wi = j >> 2, q = j & 3;
  for (int i = tid; i < kC1fElems / 4; i += 256) reinterpret_cast<f32x4*>(img)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 px = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int b) {
    const size_t row = sample_row(smp, cur, idx, b, tid == 0 && blockIdx.y == 0 && blockIdx.z == 0);
    if (tid < 196) px = reinterpret_cast<const f32x4*>(images + row * 784)[tid];
  };
  if ((int)blockIdx.x < B) fetch(blockIdx.x);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();   // previous image consumed / zero fill done
    if (tid < 196) c1f_store_px(img, tid, px);
    __syncthreads();
    if (b + (int)gridDim.x < B) fetch(b + gridDim.x);   // next image's pixels fly during this one's tiles
    for (int t = wslot + 4 * (int)blockIdx.z; t < 49; t += 4 * (int)gridDim.z) {
      const int W = 4 * t + wi, py = W / 14, pxw = W % 14;
      const float* ap = img + (2 * py + (q >> 1)) * kC1fPitch + 2 * pxw + (q & 1);
      float a[7];
#pragma unroll
      for (int s = 0; s < 7; ++s) a[s] = ap[tapoff[s]];
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma16f(a[s], bw[s][nt], acc[nt]);
      // C/D: lane (column j, group g) holds rows 4g + r = window 4t + g, position q = r
      const size_t off = ((size_t)b * 196 + 4 * t + g) * C + c0;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 v = acc[nt];
        float best = v[0];
        int bi = 0;
        if (v[1] > best) { best = v[1]; bi = 1; }
        if (v[2] > best) { best = v[2]; bi = 2; }
        if (v[3] > best) { best = v[3]; bi = 3; }
        const float o = fmaxf(best + bv[nt], 0.f);
        pooled[off + 16 * nt] = o;
        arg[off + 16 * nt] = (uint8_t)(o > 0.f ? bi : (int)kPoolDead);
      }
    }
  }
}

// grid = (G, C/32): workgroup (g, chunk) sums images g, g + G, ... into part[g][26 * C] (its 32-channel chunk).
__global__ __launch_bounds__(256) void conv1_wgrad_f32(const float* __restrict__ images, const int* __restrict__ idx,
                                                       const float* __restrict__ gpool, const uint8_t* __restrict__ arg,
                                                       float* __restrict__ part, int B, int C) {
  __shared__ __attribute__((aligned(16))) float img[kC1fElems];
  __shared__ float red[8][26][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cl = lane & 31, c = blockIdx.y * 32 + cl;
  const int hw = wave * 2 + (lane >> 5);
  constexpr int NW = 25;   // windows per half-wave and image: ceil(196 / 8)

  float dw[25], db = 0.f;
#pragma unroll
  for (int t = 0; t < 25; ++t) dw[t] = 0.f;
  for (int i = tid; i < kC1fElems / 4; i += 256) reinterpret_cast<f32x4*>(img)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 px = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int b) {
    const size_t row = idx ? (size_t)idx[b] : (size_t)b;
    if (tid < 196) px = reinterpret_cast<const f32x4*>(images + row * 784)[tid];
  };
  if ((int)blockIdx.x < B) fetch(blockIdx.x);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    // every pooled gradient / argmax code of this half-wave's windows is requested up front (one memory latency
    // per image, not one per window)
    float gv[NW];
    unsigned cv[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int win = hw + 8 * k;
      const size_t off = ((size_t)b * 196 + min(win, 195)) * C + c;
      gv[k] = gpool[off];
      cv[k] = win < 196 ? (unsigned)arg[off] : kPoolDead;
    }
    __syncthreads();   // previous image consumed / zero fill done
    if (tid < 196) c1f_store_px(img, tid, px);
    __syncthreads();
    if (b + (int)gridDim.x < B) fetch(b + gridDim.x);
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int win = min(hw + 8 * k, 195);
      const int py = win / 14, pxw = win % 14;
      const bool alive = cv[k] < kPoolDead;
      const float g = alive ? gv[k] : 0.f;
      const int q = alive ? (int)cv[k] : 0;
      // argmax position (y, x) in image coordinates; tap (ky, kx) reads haloed pixel (y + ky, x + kx)
      const float* base = img + (2 * py + (q >> 1)) * kC1fPitch + 2 * pxw + (q & 1);
#pragma unroll
      for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) dw[ky * 5 + kx] = fmaf(g, base[ky * kC1fPitch + kx], dw[ky * 5 + kx]);
      db += g;
    }
  }
#pragma unroll
  for (int t = 0; t < 25; ++t) red[hw][t][cl] = dw[t];
  red[hw][25][cl] = db;
  __syncthreads();
  float* slab = part + (size_t)blockIdx.x * 26 * C + blockIdx.y * 32;
  for (int o = tid; o < 26 * 32; o += 256) {
    const int m = o >> 5, cc = o & 31;
    float s = red[0][m][cc];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][m][cc];
    slab[m * C + cc] = s;
  }
}

__global__ __launch_bounds__(256) void u8_to_f32_unit(const uint8_t* __restrict__ src, float* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = (float)src[i] / 255.0f;   // TF's input_data: float32(x) * (1/255) up to rounding; idx pixels are exact either way
}

}  // namespace dtf

using namespace dtf;

extern "C" {

static int conv1_f32_fwd_launch(const void* images, const void* idx, SampleArgs smp, const void* w, const void* bias,
                                void* pooled, void* arg, int B, int cout, int grid, hipStream_t s) {
  if (cout % 32) return -1;
  if (grid <= 0) grid = B < 2048 ? B : 2048;
  // window split: fill ~256 CUs when the batch alone does not
  const int per_img = cout / 32;
  int ns = 256 / (grid * per_img > 0 ? grid * per_img : 1);
  ns = ns < 1 ? 1 : (ns > 4 ? 4 : ns);
  hipLaunchKernelGGL(conv1_pool_fwd_f32, dim3(grid, cout / 32, ns), dim3(256), 0, s, (const float*)images, (const int*)idx, smp,
                     (const float*)w, (const float*)bias, (float*)pooled, (uint8_t*)arg, B, cout);
  return (int)hipGetLastError();
}

// images fp32 [n][784] (+ idx gather) -> pooled fp32 [B,14,14,cout], arg u8
int dtf_conv1_f32_fwd(const void* images, const void* idx, const void* w, const void* bias, void* pooled, void* arg, int B,
                      int cout, int grid, hipStream_t s) {
  return conv1_f32_fwd_launch(images, idx, SampleArgs{nullptr, nullptr, 0, 0, 0, 0}, w, bias, pooled, arg, B, cout, grid, s);
}

// rows drawn from the device sample cursor state[0] inside the kernel and published in idx_out[B]
int dtf_conv1_f32_fwd_sampled(const void* images, void* idx_out, const void* state, unsigned n, unsigned hb, unsigned seed,
                              int shuffle, const void* w, const void* bias, void* pooled, void* arg, int B, int cout,
                              int grid, hipStream_t s) {
  if (!state || !idx_out || n == 0 || (1ull << (2 * hb)) < n) return -3;
  return conv1_f32_fwd_launch(images, nullptr, SampleArgs{(const int64_t*)state, (int*)idx_out, n, hb, seed, shuffle}, w, bias,
                              pooled, arg, B, cout, grid, s);
}

// part must hold grid * 26 * cout floats
int dtf_conv1_f32_wgrad(const void* images, const void* idx, const void* gpool, const void* arg, void* part, int B, int cout,
                        int grid, hipStream_t s) {
  if (grid <= 0) return -2;
  if (cout % 32) return -1;
  hipLaunchKernelGGL(conv1_wgrad_f32, dim3(grid, cout / 32), dim3(256), 0, s, (const float*)images, (const int*)idx,
                     (const float*)gpool, (const uint8_t*)arg, (float*)part, B, cout);
  return (int)hipGetLastError();
}

int dtf_u8_to_f32_unit(const void* src, void* dst, long n, hipStream_t s) {
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(u8_to_f32_unit, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, (const uint8_t*)src,
                     (float*)dst, n);
  return (int)hipGetLastError();
}

}  // extern "C"
