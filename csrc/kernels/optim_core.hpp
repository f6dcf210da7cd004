// Device helpers shared by the optimizer kernels (optim.hip) and the data-parallel exchange kernels (dp.hip):
// step-state advance, the SGD / Adam arithmetic on one float4, and the update of four tail elements together with
// every copy derived from them (bf16 shadow, transposed output weight, the two conv2 fragment packs).
#pragma once
#include "common.hpp"

namespace dtf {

// hyper[0]=lr [1]=beta1 [2]=beta2 [3]=eps [4]=grad_scale (e.g. 1/world) [5]=weight_decay
// state[0]=sample cursor  [1]=global_step  [2]=dropout offset  [3..]=adam t counters (one per
// independently updated arena range, so range updates on different streams never share a word)
struct StepAdvance {
  int64_t* state;       // nullptr -> no advance
  int64_t cursor_inc;
  int64_t step_inc;
  int64_t drop_inc;
  int adam_slot;        // >= 3 -> state[adam_slot] += 1
};

__device__ __forceinline__ void advance_state(const StepAdvance& a) {
  if (a.state && blockIdx.x == 0 && threadIdx.x == 0) {
    // Only the counters this launch actually advances are touched: "+= 0" is still a read-modify-write, and the
    // data-parallel Adam step runs two of these on different streams at once (the dense range's counter bump on the
    // comm stream, the tail kernel's cursor / step advance on the compute stream) -- an untouched "+= 0" there wrote
    // the OLD cursor / global step / dropout offset back over the other stream's increment (seen as a step that
    // trained on the previous batch again, 2 runs in 8 of the two-rank Adam replay test).
    if (a.cursor_inc) a.state[0] += a.cursor_inc;
    if (a.step_inc) a.state[1] += a.step_inc;
    if (a.drop_inc) a.state[2] += a.drop_inc;
    if (a.adam_slot >= 3) a.state[a.adam_slot] += 1;
  }
}

__device__ __forceinline__ void store_bf16x4(bf16* p, float a, float b, float c, float d) {
  bf16x4 v = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
  *reinterpret_cast<bf16x4*>(p) = v;
}

// four consecutive gradient values: fp32, or bf16 (the 16-bit dense collective of parallel/sync_dp.py)
__device__ __forceinline__ f32x4 load_grad4(const float* g, long i) { return reinterpret_cast<const f32x4*>(g)[i]; }
__device__ __forceinline__ f32x4 load_grad4(const bf16* g, long i) {
  const bf16x4 v = reinterpret_cast<const bf16x4*>(g)[i];
  return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

struct FusedSeg {
  const float* part;  // [G][stride]
  int G, n;
  long stride;
  long arena_off;     // float offset of the segment in the parameter / gradient arenas
};
struct FusedSgdArgs {
  FusedSeg s[3];
  int blocks_end[3];
  int count, reduce_blocks;
  float* w;
  float* g;
  bf16* shadow;
  const float* hyper;
  long dense_lo4, dense_hi4;   // dense range in float4 units
  float* out_wT;               // fp32 [10][H] transpose of the output weight for head_fwd_bwd (nullptr: none)
  long out_w_off;
  int H;
  bf16* wc2T;                  // [fwd pack | dgrad pack] of the wc2 shadow (nullptr: none); with shadow == nullptr
                               // (the fp32 configuration) it points at the FP32 packs of conv2_f32.hip instead
  long wc2_off;
  int c1, c2;
  StepAdvance adv;
  // Adam variant only: moment arenas and the step counter of this update.  Every workgroup reads *adam_t, so it
  // cannot be advanced inside this kernel: a "last workgroup to finish" counter was measured at ~35 ns per
  // workgroup (3000 serialised same-address atomics = +100 us), so a 1-thread kernel bumps it BEFORE the launch.
  float* m;
  float* v;
  const int64_t* adam_t;
};

// TF AdamOptimizer on one float4 (same arithmetic as adam_update)
__device__ __forceinline__ void adam4(f32x4& wv, const f32x4& gv, f32x4& mv, f32x4& vv, float lr_t, float b1, float b2,
                                      float eps, float gs) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float ge = gv[e] * gs;
    mv[e] = fmaf(b1, mv[e], (1.f - b1) * ge);
    vv[e] = fmaf(b2, vv[e], (1.f - b2) * ge * ge);
    wv[e] -= lr_t * mv[e] / (sqrtf(vv[e]) + eps);
  }
}

// Update of the four tail elements at arena offset e0 with the (final) gradient gs: gradient arena, optimizer,
// master weight, bf16 shadow, and the derived copies (out_wT, the two conv2 packs).  Shared by the fused tails.
template <bool ADAM>
__device__ __forceinline__ void tail_update4(const FusedSgdArgs& a, long e0, f32x4 gs, f32x4 wv, float lr, float lr_t,
                                             float b1, float b2, float eps, float gsc) {
  *reinterpret_cast<f32x4*>(a.g + e0) = gs;
  if (ADAM) {
    f32x4 mv = *reinterpret_cast<const f32x4*>(a.m + e0), vv = *reinterpret_cast<const f32x4*>(a.v + e0);
    adam4(wv, gs, mv, vv, lr_t, b1, b2, eps, gsc);
    *reinterpret_cast<f32x4*>(a.m + e0) = mv;
    *reinterpret_cast<f32x4*>(a.v + e0) = vv;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) wv[e] = fmaf(-lr, gs[e], wv[e]);
  }
  *reinterpret_cast<f32x4*>(a.w + e0) = wv;
  if (a.shadow) store_bf16x4(a.shadow + e0, wv[0], wv[1], wv[2], wv[3]);
  // (32-bit index arithmetic: with `long` operands the divisions below are 64-bit software divisions, a few hundred
  // instructions on the one wave of the workgroup that runs the update)
  const long ro64 = e0 - a.out_w_off;
  if (a.out_wT && ro64 >= 0 && ro64 < 10L * a.H) {
    const unsigned ro = (unsigned)ro64;
#pragma unroll
    for (int e = 0; e < 4; ++e) a.out_wT[(size_t)((ro + e) % 10u) * a.H + (ro + e) / 10u] = wv[e];
  }
  const long r64 = e0 - a.wc2_off;
  const long nw = 25L * a.c1 * a.c2;
  if (a.wc2T && r64 >= 0 && r64 < nw && !a.shadow) {
    // fp32 configuration (no bf16 shadow): wc2T holds the fp32 fragment packs of conv2_f32.hip
    // (pack_conv2_weights_f32): [fwd pack | dgrad pack], one lane slot = four consecutive k
    float* pk = reinterpret_cast<float*>(a.wc2T);
    const unsigned r = (unsigned)r64, cc = (unsigned)(a.c1 * a.c2), c1 = (unsigned)a.c1, c2 = (unsigned)a.c2;
    const unsigned tap = r / cc, rem = r - tap * cc;
    const unsigned ci = rem / c2, co = rem - ci * c2;
    const long f0 = (((long)tap * (c1 / 16) + ci / 16) * (c2 / 16) + co / 16) * 256 + (((ci % 16) / 4) * 16 + co % 16) * 4 + ci % 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) pk[f0 + 4 * e] = wv[e];   // co+e -> next lane slot (co % 16 + e <= 15)
    const long d0 = (((long)tap * (c2 / 16) + co / 16) * (c1 / 16) + ci / 16) * 256 + (((co % 16) / 4) * 16 + ci % 16) * 4;
    *reinterpret_cast<f32x4*>(pk + nw + d0) = wv;          // co .. co+3 = the four k of one lane slot
  } else if (a.wc2T && r64 >= 0 && r64 < nw) {
    // HWIO element (tap, ci, co..co+3) -> the two fragment-ordered packs of conv2.hip
    // (pack_conv2_weights): [fwd pack | dgrad pack]
    const unsigned r = (unsigned)r64, cc = (unsigned)(a.c1 * a.c2), c1 = (unsigned)a.c1, c2 = (unsigned)a.c2;
    const unsigned tap = r / cc, rem = r - tap * cc;
    const unsigned ci = rem / c2, co = rem - ci * c2;
    const long f0 = (((long)tap * (c1 / 32) + ci / 32) * (c2 / 16) + co / 16) * 512 + (((ci % 32) / 8) * 16 + co % 16) * 8 + ci % 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) a.wc2T[f0 + 8 * e] = (bf16)wv[e];   // co+e -> next lane slot (co % 16 + e <= 15)
    const long d0 = (((long)tap * (c2 / 32) + co / 32) * (c1 / 16) + ci / 16) * 512 + (((co % 32) / 8) * 16 + ci % 16) * 8 + co % 8;
    store_bf16x4(a.wc2T + nw + d0, wv[0], wv[1], wv[2], wv[3]);       // 4 consecutive k of one lane slot
  }
}

}  // namespace dtf
