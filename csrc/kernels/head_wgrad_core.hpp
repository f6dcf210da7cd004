// Output-layer weight gradient as a device function, so that it can run either as its own kernel
// (head.hip) or as extra workgroups of the fc1 dgrad GEMM launch (gemm.hip: the 128x128-tile GEMM
// has 200 workgroups on 256 CUs, and this reduction is ~2 us of work that otherwise costs a whole
// launch of its own).
//   dW_out = h1^T . dlogits, db_out = sum_b dlogits, db_d1 = sum_b da   (reference .idea/MNISTDist.py:89,149)
// Work split: block (bx, by) owns 256 hidden columns (4 per lane: 8-byte loads) and one batch split; its NSUB
// waves take rows b0 + sub, b0 + sub + NSUB, ... and are summed by a tree through LDS.  Slab layout per split:
// [H*10 dW_out][12: db_out padded][H db_d1].  (64 columns per block meant 512 latency-bound workgroups of 2.4 us
// for the 1024-image step -- more CU time than the fc1 dgrad launch leaves idle; tools/conv_timeline.py run .. gemm.)
#pragma once
#include "common.hpp"

namespace dtf {

// AT = element type of the activations h1 / da: bf16 (mixed-precision path) or float (--dtype fp32)
template <typename AT>
struct HeadWgradArgsT {
  const AT* h1;
  const float* dlogits;
  const AT* da;
  float* part;
  int B, H, rows_per_split;
  int nbx;  // blocks along the hidden dimension = ceil(H / 256)
};
typedef HeadWgradArgsT<bf16> HeadWgradArgs;

// four consecutive activations as fp32
__device__ __forceinline__ f32x4 load_act4(const bf16* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ f32x4 load_act4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void store_act4(bf16* p, const f32x4& v) {
  *reinterpret_cast<bf16x4*>(p) = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
}
__device__ __forceinline__ void store_act4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }

constexpr int kHeadCls = 10;
constexpr int kHeadCols = 256;   // hidden columns per workgroup
template <int NSUB>
constexpr int head_wgrad_lds_floats() { return (NSUB / 2) * ((kHeadCls + 1) * kHeadCols + 16); }

// red: head_wgrad_lds_floats<NSUB>() floats of LDS.  H must be a multiple of 4.
template <int NSUB, typename AT = bf16>
__device__ __forceinline__ void head_wgrad_block(const HeadWgradArgsT<AT>& a, int bx, int by, float* red) {
  static_assert(NSUB == 4 || NSUB == 8, "tree reduction over 4 or 8 waves");
  constexpr int kCls = kHeadCls;
  constexpr int kWaveFloats = (kCls + 1) * kHeadCols + 16;   // per-wave slot in red: [11][256] + 16 for db_out
  const AT* __restrict__ h1 = a.h1;
  const AT* __restrict__ da = a.da;
  const float* __restrict__ dlogits = a.dlogits;
  const int B = a.B, H = a.H;
  const int lane = threadIdx.x & 63, sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform row phase
  const int j = bx * kHeadCols + 4 * lane;
  const bool live = j < H;
  const int jl = live ? j : 0;   // dead lanes load column 0 (unconditional loads) and never store
  const int b0 = by * a.rows_per_split;
  const int b1 = min(B, b0 + a.rows_per_split);
  f32x4 acc[kCls + 1];
  float dbo[kCls];   // wave-uniform: column sums of this wave's dlogits rows
#pragma unroll
  for (int c = 0; c <= kCls; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < kCls; ++c) dbo[c] = 0.f;
  auto row = [&](const f32x4& hf, const f32x4& dv, int b) {
    const float* dl = dlogits + (size_t)b * kCls;   // wave-uniform address -> scalar loads
#pragma unroll
    for (int c = 0; c < kCls; ++c) {
      const float d = dl[c];
      dbo[c] += d;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[c][e] = fmaf(hf[e], d, acc[c][e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[kCls][e] += dv[e];
  };
  // four rows per trip: their activation loads are issued together (a one-row loop was a chain of
  // dependent-latency trips)
  int b = b0 + sub;
  for (; b + 3 * NSUB < b1; b += 4 * NSUB) {
    f32x4 hv[4], dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      hv[u] = load_act4(h1 + (size_t)(b + NSUB * u) * H + jl);
      dv[u] = load_act4(da + (size_t)(b + NSUB * u) * H + jl);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) row(hv[u], dv[u], b + NSUB * u);
  }
  for (; b < b1; b += NSUB) {
    const f32x4 hv = load_act4(h1 + (size_t)b * H + jl);
    const f32x4 dv = load_act4(da + (size_t)b * H + jl);
    row(hv, dv, b);
  }
  // tree over the waves: the upper half of the live waves writes, the lower half adds (fixed order)
#pragma unroll
  for (int half = NSUB / 2; half >= 1; half >>= 1) {
    if (sub >= half && sub < 2 * half) {
      float* w = red + (size_t)(sub - half) * kWaveFloats;
#pragma unroll
      for (int c = 0; c <= kCls; ++c) *reinterpret_cast<f32x4*>(w + c * kHeadCols + 4 * lane) = acc[c];
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < kCls; ++c) w[(kCls + 1) * kHeadCols + c] = dbo[c];
      }
    }
    __syncthreads();
    if (sub < half) {
      const float* w = red + (size_t)sub * kWaveFloats;
#pragma unroll
      for (int c = 0; c <= kCls; ++c) acc[c] += *reinterpret_cast<const f32x4*>(w + c * kHeadCols + 4 * lane);
#pragma unroll
      for (int c = 0; c < kCls; ++c) dbo[c] += w[(kCls + 1) * kHeadCols + c];
    }
    __syncthreads();
  }
  float* slab = a.part + (size_t)by * ((size_t)H * kCls + 12 + H);
  if (sub == 0 && live) {
    // dW_out rows j..j+3 are 40 consecutive floats: element 10*e + c
    float* dst = slab + (size_t)j * kCls;
#pragma unroll
    for (int k = 0; k < kCls; ++k) {
      f32x4 o;
#pragma unroll
      for (int m = 0; m < 4; ++m) o[m] = acc[(4 * k + m) % kCls][(4 * k + m) / kCls];
      *reinterpret_cast<f32x4*>(dst + 4 * k) = o;
    }
    *reinterpret_cast<f32x4*>(slab + (size_t)H * kCls + 12 + j) = acc[kCls];
  }
  if (bx == 0 && sub == 0 && lane < 12) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < kCls; ++c) s = (lane == c) ? dbo[c] : s;
    slab[(size_t)H * kCls + lane] = s;
  }
}

}  // namespace dtf
