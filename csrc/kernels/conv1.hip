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
constexpr int kDy1Bytes = 28 * kDy1RowBytes;   // 57344: the whole 28-row gradient tile

template <int C>
__global__ __launch_bounds__(256) void conv1_wgrad(const bf16* __restrict__ images,
                                                   const int* __restrict__ idx,
                                                   const bf16* __restrict__ gpool,
                                                   const uint8_t* __restrict__ arg,
                                                   float* __restrict__ part, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* cp = reinterpret_cast<bf16*>(smem);
  bf16* dyt = reinterpret_cast<bf16*>(smem + kCpBytes);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cchunk = blockIdx.y;
  const Conv1WgradLane L = conv1w_lane(lane);

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
  // written from the pixel pairs directly, not gathered pixel by pixel from global memory.
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
      const int i = tid + r * 256;
      if (i < 392) conv1w_write_pair(cp, i, px_r[r]);
    }
#pragma unroll
    for (int r = 0; r < DY_ROUNDS; ++r) {
      const int c = tid + r * 256;
      if (c < 196 * 4) conv1w_expand(dyt, (c >> 2) / 14, (c >> 2) % 14, c & 3, gv_r[r], ar_r[r], 0);
    }
  };

  if ((int)blockIdx.x < B) prefetch(blockIdx.x);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();   // previous image consumed / zero fill done
    stage();
    __syncthreads();
    if (b + (int)gridDim.x < B) prefetch(b + gridDim.x);
    for (int kb = wave; kb < 28; kb += 4) conv1w_kblock(cp, dyt, kb, 0, L, lane, acc);
  }

  // cross-wave reduction, then rows m < 26 go to the slab
  __syncthreads();
  conv1w_reduce_store(acc, reinterpret_cast<float*>(smem), part + (size_t)blockIdx.x * 26 * C, C, cchunk, wave, lane);
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
