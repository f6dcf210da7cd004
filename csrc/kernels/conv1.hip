// First conv block of the deep-MNIST net: 5x5 SAME conv with ONE input channel
// (reference .idea/MNISTDist.py:71-73: conv2d(x, wc1, bc1) -> maxpool2d).
//
// C_in = 1 gives a tiny reduction dimension (K = 25), but the layer is still run on the matrix
// cores by padding the tap axis to 8 per filter row: the 28x28 image sits in LDS with a zero halo,
// fragments are plain aligned reads of consecutive pixels, and conv + bias + ReLU + 2x2 max-pool
// + the argmax byte are produced in registers, so the [B,28,28,C] pre-pool tensor never exists in
// memory.  The batch gather (dataset row = idx[b]) is fused into the image load: there is no host
// input pipeline and no batch copy.
//
// Backward needs only the filter/bias gradient (the layer input is data); it IS
// GEMM-shaped (K = all positions of the batch), so it runs on the matrix cores --
// see conv1_wgrad below.  Each block accumulates a private [26][C] slab over its
// images; reduce_segments sums the slabs.
#include "conv1_core.hpp"

namespace dtf {

// Forward on the matrix cores.
//   M = output positions, N = 32 channels, K = taps ordered k = 8*ky + kx (kx < 5 carry weights,
//   kx = 5..7 are zero columns), i.e. 64 = 2 MFMA k-steps of which the second only holds ky = 4.
//   An A fragment (8 consecutive k of one row) is then 8 CONSECUTIVE PIXELS of image row y + ky.
//   M tiles take every second pixel: tile E holds x = 0,2,..,30 and tile O holds x = 1,3,..,31 of one
//   output row, so (a) both fragments come from the same five aligned ds_read_b32 of pixels
//   [2m, 2m+10) -- O is E shifted by one pixel with v_alignbit -- and (b) pooled column px = m sits in
//   the SAME accumulator register of the four tiles {E,O} x {row y, row y+1}: bias + ReLU + 2x2
//   max-pool + argmax are in-register, and one lane produces four pooled columns of its channel.
//   The pooled row is staged through wave-private LDS and leaves as 16-byte stores.
// grid-stride over images; 4 waves split the 14 row pairs.  w: fp32 HWIO [25][C] (rounded to bf16).
template <int C>
__global__ __launch_bounds__(256) void conv1_pool_fwd(const bf16* __restrict__ images,
                                                      const int* __restrict__ idx, SampleArgs smp,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      bf16* __restrict__ pooled,
                                                      uint8_t* __restrict__ arg, int B) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  __shared__ __attribute__((aligned(16))) bf16 img[kC1ImgElems];            // 6 KB, zero halo
  __shared__ __attribute__((aligned(16))) bf16 orow[4][16 * 32];            // per wave: pooled row [px][32 ch]
  __shared__ __attribute__((aligned(16))) uint8_t arow[4][16 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cchunk = blockIdx.y;   // 32-channel chunk

  // The image row of this workgroup depends on the device sample cursor (load -> 64-bit divide ->
  // Feistel -> image load): start that chain first, the weight loads below overlap with it.
  const SampleCursor cur = sample_cursor(smp);
  const Conv1Frags frags = conv1_load_frags(w, bias, C, cchunk, lane);

  for (int i = tid; i < kC1ImgElems / 8; i += 256) reinterpret_cast<u4*>(img)[i] = (u4){0u, 0u, 0u, 0u};

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();   // previous image consumed / zero fill done
    const size_t row = sample_row(smp, cur, idx, b, tid == 0 && blockIdx.y == 0);
    const unsigned* src = reinterpret_cast<const unsigned*>(images + row * 784);
    for (int i = tid; i < 392; i += 256) {   // 2 pixels (one dword) per thread-iteration
      int y = i / 14, x2 = i % 14;
      reinterpret_cast<unsigned*>(img)[((y + 2) * kC1Pitch + 2) / 2 + x2] = src[i];
    }
    __syncthreads();
    for (int rp = wave; rp < 14; rp += 4) {
      conv1_pooled_row(img, rp, frags, orow[wave], arow[wave], lane);
      __builtin_amdgcn_wave_barrier();
      // 14 px x 32 ch: 896 B of bf16 = 56 x 16 B, 448 B of argmax = 28 x 16 B
      const size_t obase = ((size_t)b * 196 + rp * 14) * C + cchunk * 32;
      if (C == 32) {
        if (lane < 56)
          reinterpret_cast<u4*>(pooled + obase)[lane] = reinterpret_cast<const u4*>(orow[wave])[lane];
        if (lane < 28)
          reinterpret_cast<u4*>(arg + obase)[lane] = reinterpret_cast<const u4*>(arow[wave])[lane];
      } else {
        // C > 32: a pooled pixel's 32-channel chunk is 64 B (4 x 16 B) at stride C
        if (lane < 56) {
          int px = lane >> 2, c4 = lane & 3;
          *reinterpret_cast<u4*>(pooled + obase + (size_t)px * C + c4 * 8) =
              reinterpret_cast<const u4*>(orow[wave])[lane];
        }
        if (lane < 28) {
          int px = lane >> 1, c2 = lane & 1;
          *reinterpret_cast<u4*>(arg + obase + (size_t)px * C + c2 * 16) =
              reinterpret_cast<const u4*>(arow[wave])[lane];
        }
      }
      __builtin_amdgcn_wave_barrier();
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
                                                   float* __restrict__ part, int B) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* cp = reinterpret_cast<bf16*>(smem);
  bf16* dyt = reinterpret_cast<bf16*>(smem + kCpBytes);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  // Staging.  Every global load of image b+1 (2 pixel pairs of the input, 4 x (8 gradients + 8
  // argmax codes) of the pooled gradient per thread) is issued before the k-loop of image b and
  // only consumed after it: the loop never waits on memory.  The five shifted input copies are
  // written from the pixel pairs directly (an even shift keeps the pair dword-aligned, an odd one
  // splits it into two 2-byte stores), not gathered pixel by pixel from global memory.
  constexpr int DY_ROUNDS = (196 * 4 + 255) / 256;   // 4
  unsigned px_r[2] = {0u, 0u};
  u32x4 gv_r[DY_ROUNDS];
  u32x2 ar_r[DY_ROUNDS];
  auto prefetch = [&](int b) {
    const size_t row = idx ? (size_t)idx[b] : (size_t)b;
    const unsigned* src = reinterpret_cast<const unsigned*>(images + row * 784);
#pragma unroll
    for (int r = 0; r < 2; ++r) px_r[r] = src[min(tid + r * 256, 391)];
#pragma unroll
    for (int r = 0; r < DY_ROUNDS; ++r) {
      const int c = min(tid + r * 256, 196 * 4 - 1);
      const size_t off = ((size_t)b * 196 + (c >> 2)) * C + cchunk * 32 + (c & 3) * 8;
      gv_r[r] = *reinterpret_cast<const u32x4*>(gpool + off);
      ar_r[r] = *reinterpret_cast<const u32x2*>(arg + off);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int i = tid + r * 256;   // pixel pair (y, 2*x2), (y, 2*x2 + 1)
      if (i < 392) {
        const int y = i / 14, x = 2 * (i % 14);
        const unsigned v = px_r[r];
        bf16* rowp = cp + (y + 2) * kCpPitch;
#pragma unroll
        for (int sft = 0; sft < 5; ++sft) {
          const int c = x - sft + 2;   // column of pixel x in copy sft (copy holds x = c + sft - 2)
          bf16* dst = rowp + sft * kCpStride + c;
          if ((sft & 1) == 0) {
            if (c >= 0 && c < 31) *reinterpret_cast<unsigned*>(dst) = v;
          } else {
            if (c >= 0) reinterpret_cast<unsigned short*>(dst)[0] = (unsigned short)(v & 0xffffu);
            if (c + 1 >= 0 && c + 1 < 32) reinterpret_cast<unsigned short*>(dst)[1] = (unsigned short)(v >> 16);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < DY_ROUNDS; ++r) {
      const int c = tid + r * 256;
      if (c < 196 * 4) {
        const int c8 = c & 3, pp = c >> 2;
        const int py = pp / 14, px = pp % 14;
        u32x4 o[4];
        expand_pool_grad(gv_r[r], ar_r[r], o);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int y = 2 * py + (q >> 1), x = 2 * px + (q & 1);
          int slot = (((c8 >> 1) ^ ((x >> 3) & 1)) << 1) | (c8 & 1);
          *reinterpret_cast<u32x4*>(dyt + (y * 32 + x) * 32 + slot * 8) = o[q];
        }
      }
    }
  };

  if ((int)blockIdx.x < B) prefetch(blockIdx.x);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();   // previous image consumed / zero fill done
    stage();
    __syncthreads();
    if (b + (int)gridDim.x < B) prefetch(b + gridDim.x);
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

static int conv1_fwd_launch(const void* images, const void* idx, SampleArgs smp, const void* w, const void* bias,
                            void* pooled, void* arg, int B, int cout, int grid, hipStream_t s) {
  if (grid <= 0) grid = B < 1024 ? B : 1024;
  if (cout == 32)
    hipLaunchKernelGGL((conv1_pool_fwd<32>), dim3(grid, 1), dim3(256), 0, s, (const bf16*)images,
                       (const int*)idx, smp, (const float*)w, (const float*)bias, (bf16*)pooled, (uint8_t*)arg, B);
  else if (cout == 128)
    hipLaunchKernelGGL((conv1_pool_fwd<128>), dim3(grid, 4), dim3(256), 0, s, (const bf16*)images,
                       (const int*)idx, smp, (const float*)w, (const float*)bias, (bf16*)pooled, (uint8_t*)arg, B);
  else
    return -1;
  return (int)hipGetLastError();
}

int dtf_conv1_fwd(const void* images, const void* idx, const void* w, const void* bias, void* pooled,
                  void* arg, int B, int cout, int grid, hipStream_t s) {
  return conv1_fwd_launch(images, idx, SampleArgs{nullptr, nullptr, 0, 0, 0, 0}, w, bias, pooled, arg, B, cout, grid, s);
}

// conv1 forward on the next training batch of the resident dataset: rows are drawn from the
// device sample cursor state[0] inside the kernel and written to idx_out[B]
int dtf_conv1_fwd_sampled(const void* images, void* idx_out, const void* state, unsigned n, unsigned hb,
                          unsigned seed, int shuffle, const void* w, const void* bias, void* pooled, void* arg,
                          int B, int cout, int grid, hipStream_t s) {
  if (!state || !idx_out || n == 0 || (1ull << (2 * hb)) < n) return -3;
  return conv1_fwd_launch(images, nullptr, SampleArgs{(const int64_t*)state, (int*)idx_out, n, hb, seed, shuffle}, w,
                          bias, pooled, arg, B, cout, grid, s);
}

// part must hold grid * 26 * cout floats
int dtf_conv1_wgrad(const void* images, const void* idx, const void* gpool, const void* arg, void* part, int B, int cout, int grid, hipStream_t s) {
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
                       (const bf16*)gpool, (const uint8_t*)arg, (float*)part, B);
  else if (cout == 128)
    hipLaunchKernelGGL((conv1_wgrad<128>), dim3(grid, 4), dim3(256), lds, s, (const bf16*)images, (const int*)idx,
                       (const bf16*)gpool, (const uint8_t*)arg, (float*)part, B);
  else
    return -1;
  return (int)hipGetLastError();
}

}  // extern "C"
