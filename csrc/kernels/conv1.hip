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
// Backward needs only the filter/bias gradient (the layer input is data); it IS
// GEMM-shaped (K = all positions of the batch), so it runs on the matrix cores --
// see conv1_wgrad below.  Each block accumulates a private [26][C] slab over its
// images; reduce_segments sums the slabs.
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

// Filter/bias gradient on the matrix cores.
//   dW[tap][c] = sum_{b,pos} x[b, pos + tap] * dY[b, pos, c]
// as a GEMM with M = 32 rows (25 taps, row 25 = all-ones -> bias gradient, rest
// zero), N = 32 channels, K = positions.  A K block is 2 image rows x 16 columns;
// a lane's 8 consecutive k are 8 consecutive pixels of one row, so with five
// pre-shifted LDS copies of the padded image (copy kx holds column x + kx) every
// A fragment is ONE aligned ds_read_b128.  dY is the max-pool/ReLU-expanded
// gradient, staged [pos][32 c] in LDS and consumed with ds_read_b64_tr_b16.
// grid = (G, C/32); part: [G][26*C] fp32 (25 filter taps then the bias row).
constexpr int kCpPitch = 40;                  // elements per copy row (80 B: conflict-free taps)
constexpr int kCpStride = 32 * kCpPitch + 8;  // elements per shifted copy
constexpr int kCpBytes = 5 * kCpStride * 2;   // 12880
constexpr int kDy1Bytes = 28 * 32 * 32 * 2;   // 57344

template <int C>
__global__ __launch_bounds__(256) void conv1_wgrad(const bf16* __restrict__ images,
                                                   const int* __restrict__ idx,
                                                   const bf16* __restrict__ gpool,
                                                   const uint8_t* __restrict__ arg,
                                                   const bf16* __restrict__ pooled,
                                                   float* __restrict__ part, int B) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* cp = reinterpret_cast<bf16*>(smem);
  bf16* dyt = reinterpret_cast<bf16*>(smem + kCpBytes);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, i16 = lane & 15, qr = i16 >> 2, pc = i16 & 3;
  const int cchunk = blockIdx.y;

  // A rows of this lane for the two m-tiles
  int a_off[2];   // element offset of (copy kx, row ky) or -1 = constant fragment
  bf16x8 a_const[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int m = mt * 16 + i16;
    a_off[mt] = (m < 25) ? (m % 5) * kCpStride + (m / 5) * kCpPitch : -1;
    bf16 cv = (m == 25) ? (bf16)1.0f : (bf16)0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) a_const[mt][e] = cv;
  }

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int i = tid; i < (kCpBytes + kDy1Bytes) / 16; i += 256)
    reinterpret_cast<u32x4*>(smem)[i] = (u32x4){0u, 0u, 0u, 0u};

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    const size_t row = idx ? (size_t)idx[b] : (size_t)b;
    const bf16* src = images + row * 784;
    for (int e = tid; e < 5 * 1024; e += 256) {
      int s = e >> 10, r = (e >> 5) & 31, c = e & 31;
      int y = r - 2, x = c + s - 2;
      bf16 v = (y >= 0 && y < 28 && x >= 0 && x < 28) ? src[y * 28 + x] : (bf16)0.f;
      cp[s * kCpStride + r * kCpPitch + c] = v;
    }
    for (int c = tid; c < 196 * 4; c += 256) {
      int c8 = c & 3, pp = c >> 2;
      int py = pp / 14, px = pp % 14;
      size_t off = ((size_t)b * 196 + pp) * C + cchunk * 32 + c8 * 8;
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
        int y = 2 * py + (q >> 1), x = 2 * px + (q & 1);
        int slot = (((c8 >> 1) ^ ((x >> 3) & 1)) << 1) | (c8 & 1);
        *reinterpret_cast<bf16x8*>(dyt + (y * 32 + x) * 32 + slot * 8) = o[q];
      }
    }
    __syncthreads();
    for (int kb = wave; kb < 28; kb += 4) {
      const int yrow = 2 * (kb >> 1) + (g >> 1);
      const int xb = (kb & 1) * 16 + 8 * (g & 1);
      bf16x8 bfrag[2], afrag[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16* p0 = dyt + (yrow * 32 + xb + qr) * 32 + ((j ^ ((xb >> 3) & 1)) << 4) + pc * 4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * 32));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bfrag[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        // every lane reads (constant rows read a valid dummy address) and then selects
        int off = (a_off[mt] >= 0 ? a_off[mt] : 0) + yrow * kCpPitch + xb;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(cp + off);
        afrag[mt] = a_off[mt] >= 0 ? v : a_const[mt];
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[mt][j] = mfma16(afrag[mt], bfrag[j], acc[mt][j]);
    }
  }

  // cross-wave reduction, then rows m < 26 go to the slab
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);  // [4 waves][16][64]
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 16 + (mt * 2 + j) * 4 + r) * 64 + lane] = acc[mt][j][r];
  __syncthreads();
  if (wave != 0) return;
  float* slab = part + (size_t)blockIdx.x * 26 * C;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int e = ((mt * 2 + j) * 4 + r) * 64 + lane;
        float s = (red[e] + red[16 * 64 + e]) + (red[32 * 64 + e] + red[48 * 64 + e]);
        int m = mt * 16 + 4 * g + r;
        if (m < 26) slab[m * C + cchunk * 32 + j * 16 + i16] = s;
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
  constexpr int lds = kCpBytes + kDy1Bytes;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv1_wgrad<32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)conv1_wgrad<128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  if (cout == 32)
    hipLaunchKernelGGL((conv1_wgrad<32>), dim3(grid, 1), dim3(256), lds, s, (const bf16*)images, (const int*)idx,
                       (const bf16*)gpool, (const uint8_t*)arg, (const bf16*)pooled, (float*)part, B);
  else if (cout == 128)
    hipLaunchKernelGGL((conv1_wgrad<128>), dim3(grid, 4), dim3(256), lds, s, (const bf16*)images, (const int*)idx,
                       (const bf16*)gpool, (const uint8_t*)arg, (const bf16*)pooled, (float*)part, B);
  else
    return -1;
  return (int)hipGetLastError();
}

}  // extern "C"
