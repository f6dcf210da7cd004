// Classifier head: logits GEMM (N = 10) fused with softmax cross-entropy
// forward AND backward, the accuracy metric, and the ReLU/dropout gradient of
// fc1 (reference .idea/MNISTDist.py:89 logits, :148 reduce_mean(softmax_xent),
// :152-153 argmax-equality accuracy; TF's SoftmaxCrossEntropyWithLogits kernel
// also emits softmax - y, which is what this kernel hands to backward).
//
// N = 10 is far too skinny for the matrix cores, so one wave owns one batch row:
// lanes stride the hidden dimension with 16-byte bf16 loads, reduce the 10
// partial dot products across the wave, and every lane then holds the full
// logits vector.  The row's loss, correctness bit, dlogits and the gradient
// wrt the fc1 pre-activation (da = (h1 > 0) * (dlogits . W^T) / keep) are all
// produced from registers in the same launch; h1 is stored post-ReLU and
// post-dropout, so "h1 > 0" is exactly relu'(a) * dropout_mask.
//
// head_wgrad reduces over the batch: dW_out = h1^T . dlogits, db_out, and the
// fc1 bias gradient db_d1 = colsum(da), as deterministic split-K slabs.
#include "common.hpp"

namespace dtf {

constexpr int kCls = 10;

// labels: int32 dataset labels; idx: optional gather (label = labels[idx[b]])
__global__ __launch_bounds__(256) void head_fwd_bwd(
    const bf16* __restrict__ h1, const float* __restrict__ w, const float* __restrict__ bias,
    const int* __restrict__ labels, const int* __restrict__ idx, float* __restrict__ logits_out,
    float* __restrict__ loss_rows, float* __restrict__ correct_rows, float* __restrict__ dlogits,
    bf16* __restrict__ da, int B, int H, float grad_scale, float inv_keep, int backward) {
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;

  for (int b = wave_global; b < B; b += nwaves) {
    const bf16* hrow = h1 + (size_t)b * H;
    float z[kCls];
#pragma unroll
    for (int c = 0; c < kCls; ++c) z[c] = 0.f;
    for (int j0 = lane * 8; j0 < H; j0 += 512) {
      bf16x8 hv = *reinterpret_cast<const bf16x8*>(hrow + j0);
      const float* wp = w + (size_t)j0 * kCls;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float hf = (float)hv[e];
#pragma unroll
        for (int c = 0; c < kCls; ++c) z[c] = fmaf(hf, wp[e * kCls + c], z[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < kCls; ++c) z[c] = wave_sum(z[c]) + bias[c];

    const int label = labels[idx ? idx[b] : b];
    float zmax = z[0];
    int amax = 0;
#pragma unroll
    for (int c = 1; c < kCls; ++c)
      if (z[c] > zmax) { zmax = z[c]; amax = c; }
    float p[kCls], se = 0.f;
#pragma unroll
    for (int c = 0; c < kCls; ++c) { p[c] = __expf(z[c] - zmax); se += p[c]; }
    const float inv = 1.f / se;
    float zl = 0.f;
#pragma unroll
    for (int c = 0; c < kCls; ++c) zl = (c == label) ? z[c] : zl;
    float dl[kCls];
#pragma unroll
    for (int c = 0; c < kCls; ++c) dl[c] = (p[c] * inv - (c == label ? 1.f : 0.f)) * grad_scale;

    if (lane == 0) {
      loss_rows[b] = __logf(se) + zmax - zl;
      correct_rows[b] = (amax == label) ? 1.f : 0.f;
    }
    if (logits_out && lane < kCls) {
      float v = 0.f;
#pragma unroll
      for (int c = 0; c < kCls; ++c) v = (c == lane) ? z[c] : v;
      logits_out[(size_t)b * kCls + lane] = v;
    }
    if (!backward) continue;
    if (lane < kCls) {
      float v = 0.f;
#pragma unroll
      for (int c = 0; c < kCls; ++c) v = (c == lane) ? dl[c] : v;
      dlogits[(size_t)b * kCls + lane] = v;
    }
    bf16* drow = da + (size_t)b * H;
    for (int j0 = lane * 8; j0 < H; j0 += 512) {
      bf16x8 hv = *reinterpret_cast<const bf16x8*>(hrow + j0);
      const float* wp = w + (size_t)j0 * kCls;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < kCls; ++c) s = fmaf(dl[c], wp[e * kCls + c], s);
        o[e] = (bf16)(((float)hv[e] > 0.f) ? s * inv_keep : 0.f);
      }
      *reinterpret_cast<bf16x8*>(drow + j0) = o;
    }
  }
}

// grid = (H/64, KS).  slab layout per k-split: [H*10 dW_out][12: db_out padded][H db_d1]
__global__ __launch_bounds__(256) void head_wgrad(const bf16* __restrict__ h1,
                                                  const float* __restrict__ dlogits,
                                                  const bf16* __restrict__ da, float* __restrict__ part,
                                                  int B, int H, int rows_per_split) {
  __shared__ float red[4][64][kCls + 1];
  const int jc = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jc;
  const int b0 = blockIdx.y * rows_per_split;
  const int b1 = min(B, b0 + rows_per_split);
  float acc[kCls + 1];
#pragma unroll
  for (int c = 0; c <= kCls; ++c) acc[c] = 0.f;
  if (j < H) {
    for (int b = b0 + sub; b < b1; b += 4) {
      const float hf = (float)h1[(size_t)b * H + j];
      const float* dl = dlogits + (size_t)b * kCls;
#pragma unroll
      for (int c = 0; c < kCls; ++c) acc[c] = fmaf(hf, dl[c], acc[c]);
      acc[kCls] += (float)da[(size_t)b * H + j];
    }
  }
#pragma unroll
  for (int c = 0; c <= kCls; ++c) red[sub][jc][c] = acc[c];
  __syncthreads();
  float* slab = part + (size_t)blockIdx.y * ((size_t)H * kCls + 12 + H);
  if (sub == 0 && j < H) {
#pragma unroll
    for (int c = 0; c <= kCls; ++c) {
      float s = red[0][jc][c] + red[1][jc][c] + red[2][jc][c] + red[3][jc][c];
      if (c < kCls) slab[(size_t)j * kCls + c] = s;
      else slab[(size_t)H * kCls + 12 + j] = s;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 12) {
    float s = 0.f;
    if (threadIdx.x < kCls)
      for (int b = b0; b < b1; ++b) s += dlogits[(size_t)b * kCls + threadIdx.x];
    slab[(size_t)H * kCls + threadIdx.x] = s;
  }
}

}  // namespace dtf

using namespace dtf;

extern "C" {

int dtf_head_fwd_bwd(const void* h1, const void* w, const void* bias, const void* labels, const void* idx,
                     void* logits_out, void* loss_rows, void* correct_rows, void* dlogits, void* da, int B,
                     int H, float grad_scale, float inv_keep, int backward, hipStream_t s) {
  if (H % 8) return -3;
  int blocks = (B + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(head_fwd_bwd, dim3(blocks), dim3(256), 0, s, (const bf16*)h1, (const float*)w,
                     (const float*)bias, (const int*)labels, (const int*)idx, (float*)logits_out,
                     (float*)loss_rows, (float*)correct_rows, (float*)dlogits, (bf16*)da, B, H, grad_scale,
                     inv_keep, backward);
  return (int)hipGetLastError();
}

// part: [KS][H*10 + 12 + H] fp32
int dtf_head_wgrad(const void* h1, const void* dlogits, const void* da, void* part, int B, int H, int KS,
                   hipStream_t s) {
  if (KS <= 0) return -2;
  int rps = (B + KS - 1) / KS;
  hipLaunchKernelGGL(head_wgrad, dim3((H + 63) / 64, KS), dim3(256), 0, s, (const bf16*)h1,
                     (const float*)dlogits, (const bf16*)da, (float*)part, B, H, rps);
  return (int)hipGetLastError();
}

}  // extern "C"
