// Output-layer weight gradient as a device function, so that it can run either as its own kernel
// (head.hip) or as extra workgroups of the fc1 dgrad GEMM launch (gemm.hip: the 128x128-tile GEMM
// has 200 workgroups on 256 CUs, and this reduction is ~2 us of work that otherwise costs a whole
// launch of its own).
//   dW_out = h1^T . dlogits, db_out = sum_b dlogits, db_d1 = sum_b da   (reference .idea/MNISTDist.py:89,149)
// Work split: block (bx, by) owns 64 hidden columns and one batch split; its NSUB waves take rows
// b0 + sub, b0 + sub + NSUB, ...  Slab layout per split: [H*10 dW_out][12: db_out padded][H db_d1].
#pragma once
#include "common.hpp"

namespace dtf {

struct HeadWgradArgs {
  const bf16* h1;
  const float* dlogits;
  const bf16* da;
  float* part;
  int B, H, rows_per_split;
  int nbx;  // blocks along the hidden dimension = ceil(H / 64)
};

constexpr int kHeadCls = 10;
template <int NSUB>
constexpr int head_wgrad_lds_floats() { return NSUB * 64 * (kHeadCls + 1); }

// red: NSUB * 64 * 11 floats of LDS
template <int NSUB>
__device__ __forceinline__ void head_wgrad_block(const HeadWgradArgs& a, int bx, int by, float* red_raw) {
  constexpr int kCls = kHeadCls;
  float (*red)[64][kCls + 1] = reinterpret_cast<float (*)[64][kCls + 1]>(red_raw);
  const bf16* __restrict__ h1 = a.h1;
  const bf16* __restrict__ da = a.da;
  const float* __restrict__ dlogits = a.dlogits;
  const int B = a.B, H = a.H;
  const int jc = threadIdx.x & 63, sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform row phase
  const int j = bx * 64 + jc;
  const int b0 = by * a.rows_per_split;
  const int b1 = min(B, b0 + a.rows_per_split);
  float acc[kCls + 1];
#pragma unroll
  for (int c = 0; c <= kCls; ++c) acc[c] = 0.f;
  if (j < H) {
    // four rows per trip: their activation loads are issued together (a one-row loop was a chain of
    // dependent-latency trips), and the dlogits row address is wave-uniform -> scalar loads
    int b = b0 + sub;
    for (; b + 3 * NSUB < b1; b += 4 * NSUB) {
      float hf[4], dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        hf[u] = (float)h1[(size_t)(b + NSUB * u) * H + j];
        dv[u] = (float)da[(size_t)(b + NSUB * u) * H + j];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* dl = dlogits + (size_t)(b + NSUB * u) * kCls;
#pragma unroll
        for (int c = 0; c < kCls; ++c) acc[c] = fmaf(hf[u], dl[c], acc[c]);
        acc[kCls] += dv[u];
      }
    }
    for (; b < b1; b += NSUB) {
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
  float* slab = a.part + (size_t)by * ((size_t)H * kCls + 12 + H);
  if (sub == 0 && j < H) {
#pragma unroll
    for (int c = 0; c <= kCls; ++c) {
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < NSUB; ++u) s += red[u][jc][c];
      if (c < kCls) slab[(size_t)j * kCls + c] = s;
      else slab[(size_t)H * kCls + 12 + j] = s;
    }
  }
  if (bx == 0 && threadIdx.x < 12) {
    float s = 0.f;
    if (threadIdx.x < kCls)
      for (int b = b0; b < b1; ++b) s += dlogits[(size_t)b * kCls + threadIdx.x];
    slab[(size_t)H * kCls + threadIdx.x] = s;
  }
}

}  // namespace dtf
