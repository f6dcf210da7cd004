// First conv block of the deep-MNIST net: 5x5 SAME conv with ONE input channel
// (reference .idea/MNISTDist.py:71-73: conv2d(x, wc1, bc1) -> maxpool2d).
//
// C_in = 1 makes this layer degenerate for the matrix cores (K = 25), so it is
// a direct LDS stencil on the vector ALUs: the 28x28 image sits in LDS with a
// 2-pixel zero halo, each lane owns one output channel (25 weights in
// registers) and walks pooling windows; conv + bias + ReLU + 2x2 max-pool and
// the argmax byte are produced in registers, so the [B,28,28,C] pre-pool tensor
// never exists in memory.  The batch gather (dataset row = idx[b]) is fused
// into the image load: there is no host input pipeline and no batch copy.
//
// Backward needs only the filter/bias gradient (the layer input is data):
//   dW[tap][c] = sum_{b, window} x[b, argmax_pos + tap] * g[b, window, c]
// with g masked by the ReLU (pooled > 0).  Each block accumulates a private
// [26][C] slab over its images; reduce_partials sums the slabs.
#include "common.hpp"

namespace dtf {

constexpr int kImgW = 28;
constexpr int kPadW = 33;  // 32 (28 + 2*2 halo) + 1 pad column

// images: [N,784] bf16, idx: [B] dataset rows (nullptr -> identity), w: [25][C] fp32 (HWIO, I = 1)
template <int C>
__global__ __launch_bounds__(256) void conv1_pool_fwd(const bf16* __restrict__ images,
                                                      const int* __restrict__ idx,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      bf16* __restrict__ pooled,
                                                      uint8_t* __restrict__ arg, int B) {
  constexpr int PG = 256 / C;  // window groups per block
  __shared__ float img[32 * kPadW];
  const int tid = threadIdx.x;
  const int c = tid % C, pg = tid / C;

  float wr[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) wr[t] = w[t * C + c];
  const float bv = bias[c];

  for (int i = tid; i < 32 * kPadW; i += 256) img[i] = 0.f;

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    const size_t row = idx ? (size_t)idx[b] : (size_t)b;
    const bf16* src = images + row * 784;
    for (int i = tid; i < 392; i += 256) {  // 2 pixels per thread-iteration
      bf16x2 v = *reinterpret_cast<const bf16x2*>(src + 2 * i);
      int p = 2 * i, y = p / kImgW, x = p % kImgW;
      img[(y + 2) * kPadW + x + 2] = (float)v[0];
      img[(y + 2) * kPadW + x + 3] = (float)v[1];
    }
    __syncthreads();
    for (int pp = pg; pp < 196; pp += PG) {
      const int py = pp / 14, px = pp % 14;
      const float* base = img + (2 * py) * kPadW + 2 * px;  // top-left of the 6x6 patch
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        float v[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) v[s] = base[r * kPadW + s];
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
          if (r < 5) {
            a0 = fmaf(v[kx], wr[r * 5 + kx], a0);
            a1 = fmaf(v[kx + 1], wr[r * 5 + kx], a1);
          }
          if (r >= 1) {
            a2 = fmaf(v[kx], wr[(r - 1) * 5 + kx], a2);
            a3 = fmaf(v[kx + 1], wr[(r - 1) * 5 + kx], a3);
          }
        }
      }
      float best = a0;
      int bi = 0;
      if (a1 > best) { best = a1; bi = 1; }
      if (a2 > best) { best = a2; bi = 2; }
      if (a3 > best) { best = a3; bi = 3; }
      const size_t off = ((size_t)b * 196 + pp) * C + c;
      pooled[off] = (bf16)fmaxf(best + bv, 0.f);
      arg[off] = (uint8_t)bi;
    }
  }
}

// part: [gridDim.x][26*C] fp32 (25 filter taps then the bias row)
template <int C>
__global__ __launch_bounds__(256) void conv1_wgrad(const bf16* __restrict__ images,
                                                   const int* __restrict__ idx,
                                                   const bf16* __restrict__ gpool,
                                                   const uint8_t* __restrict__ arg,
                                                   const bf16* __restrict__ pooled,
                                                   float* __restrict__ part, int B) {
  constexpr int PG = 256 / C;
  __shared__ float img[32 * kPadW];
  __shared__ float red[26 * 256];
  const int tid = threadIdx.x;
  const int c = tid % C, pg = tid / C;

  float acc[26];
#pragma unroll
  for (int t = 0; t < 26; ++t) acc[t] = 0.f;
  for (int i = tid; i < 32 * kPadW; i += 256) img[i] = 0.f;

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    const size_t row = idx ? (size_t)idx[b] : (size_t)b;
    const bf16* src = images + row * 784;
    for (int i = tid; i < 392; i += 256) {
      bf16x2 v = *reinterpret_cast<const bf16x2*>(src + 2 * i);
      int p = 2 * i, y = p / kImgW, x = p % kImgW;
      img[(y + 2) * kPadW + x + 2] = (float)v[0];
      img[(y + 2) * kPadW + x + 3] = (float)v[1];
    }
    __syncthreads();
    for (int pp = pg; pp < 196; pp += PG) {
      const size_t off = ((size_t)b * 196 + pp) * C + c;
      const float pv = (float)pooled[off];
      const float gv = pv > 0.f ? (float)gpool[off] : 0.f;
      const int a = arg[off];
      const int py = pp / 14, px = pp % 14;
      const float* base = img + (2 * py + (a >> 1)) * kPadW + 2 * px + (a & 1);
#pragma unroll
      for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) acc[ky * 5 + kx] = fmaf(base[ky * kPadW + kx], gv, acc[ky * 5 + kx]);
      acc[25] += gv;
    }
  }
  // reduce the PG window groups of the block
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 26; ++t) red[t * 256 + tid] = acc[t];
  __syncthreads();
  float* slab = part + (size_t)blockIdx.x * 26 * C;
  for (int o = tid; o < 26 * C; o += 256) {
    int t = o / C, cc = o % C;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PG; ++k) s += red[t * 256 + k * C + cc];
    slab[o] = s;
  }
}

}  // namespace dtf

using namespace dtf;

extern "C" {

int dtf_conv1_fwd(const void* images, const void* idx, const void* w, const void* bias, void* pooled,
                  void* arg, int B, int cout, int grid, hipStream_t s) {
  if (grid <= 0) grid = B < 2048 ? B : 2048;
  if (cout == 32)
    hipLaunchKernelGGL((conv1_pool_fwd<32>), dim3(grid), dim3(256), 0, s, (const bf16*)images,
                       (const int*)idx, (const float*)w, (const float*)bias, (bf16*)pooled, (uint8_t*)arg, B);
  else if (cout == 128)
    hipLaunchKernelGGL((conv1_pool_fwd<128>), dim3(grid), dim3(256), 0, s, (const bf16*)images,
                       (const int*)idx, (const float*)w, (const float*)bias, (bf16*)pooled, (uint8_t*)arg, B);
  else
    return -1;
  return (int)hipGetLastError();
}

// part must hold grid * 26 * cout floats
int dtf_conv1_wgrad(const void* images, const void* idx, const void* gpool, const void* arg,
                    const void* pooled, void* part, int B, int cout, int grid, hipStream_t s) {
  if (grid <= 0) return -2;
  if (cout == 32)
    hipLaunchKernelGGL((conv1_wgrad<32>), dim3(grid), dim3(256), 0, s, (const bf16*)images, (const int*)idx,
                       (const bf16*)gpool, (const uint8_t*)arg, (const bf16*)pooled, (float*)part, B);
  else if (cout == 128)
    hipLaunchKernelGGL((conv1_wgrad<128>), dim3(grid), dim3(256), 0, s, (const bf16*)images, (const int*)idx,
                       (const bf16*)gpool, (const uint8_t*)arg, (const bf16*)pooled, (float*)part, B);
  else
    return -1;
  return (int)hipGetLastError();
}

}  // extern "C"
