// Parameter update over the flat parameter arena (reference
// .idea/MNISTDist.py:149: GradientDescentOptimizer(lr).minimize(cost,
// global_step) -> ApplyGradientDescent x8 + AssignAdd(global_step) on the PS).
//
// All trainable tensors live in ONE fp32 arena (and their gradients in a
// mirror arena), so the whole model is updated by a single launch with 16-byte
// loads/stores; the same pass emits the bf16 compute copy of every weight and
// bumps the device-resident step state (global_step, sample cursor, dropout
// offset) so a captured HIP graph can be replayed without host involvement.
// Hyper-parameters are read from device memory for the same reason.
//
// reduce_segments sums the deterministic split-K slabs the wgrad kernels write
// (no float atomics anywhere in backward -> bitwise reproducible gradients).
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

// n must be a multiple of 4 (arena tensors are padded to 4 floats)
template <typename GT>
__global__ __launch_bounds__(256) void sgd_update(float* __restrict__ w, const GT* __restrict__ g,
                                                  bf16* __restrict__ shadow, const float* __restrict__ hyper,
                                                  long n4, StepAdvance adv) {
  const float lr = hyper[0] * hyper[4];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 wv = reinterpret_cast<const f32x4*>(w)[i];
    f32x4 gv = load_grad4(g, i);
    wv[0] = fmaf(-lr, gv[0], wv[0]);
    wv[1] = fmaf(-lr, gv[1], wv[1]);
    wv[2] = fmaf(-lr, gv[2], wv[2]);
    wv[3] = fmaf(-lr, gv[3], wv[3]);
    reinterpret_cast<f32x4*>(w)[i] = wv;
    if (shadow) store_bf16x4(shadow + 4 * i, wv[0], wv[1], wv[2], wv[3]);
  }
  advance_state(adv);
}

// TF AdamOptimizer semantics: lr_t = lr * sqrt(1-b2^t)/(1-b1^t); w -= lr_t * m / (sqrt(v) + eps)
template <typename GT>
__global__ __launch_bounds__(256) void adam_update(float* __restrict__ w, const GT* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   bf16* __restrict__ shadow, const float* __restrict__ hyper,
                                                   long n4, const int64_t* __restrict__ tptr) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], gs = hyper[4];
  const float t = (float)(tptr ? *tptr + 1 : 1);
  const float lr_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 wv = reinterpret_cast<const f32x4*>(w)[i];
    f32x4 gv = load_grad4(g, i);
    f32x4 mv = reinterpret_cast<const f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<const f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ge = gv[e] * gs;
      mv[e] = fmaf(b1, mv[e], (1.f - b1) * ge);
      vv[e] = fmaf(b2, vv[e], (1.f - b2) * ge * ge);
      wv[e] -= lr_t * mv[e] / (sqrtf(vv[e]) + eps);
    }
    reinterpret_cast<f32x4*>(w)[i] = wv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (shadow) store_bf16x4(shadow + 4 * i, wv[0], wv[1], wv[2], wv[3]);
  }
  // *tptr is read by every block above, so it is bumped by a trailing 1-thread kernel
  // (dtf_adam_update) rather than by whichever block ends first.
}

__global__ void advance_state_kernel(StepAdvance adv) { advance_state(adv); }

__global__ __launch_bounds__(256) void cast_bf16(const float* __restrict__ w, bf16* __restrict__ shadow,
                                                 long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 wv = reinterpret_cast<const f32x4*>(w)[i];
    store_bf16x4(shadow + 4 * i, wv[0], wv[1], wv[2], wv[3]);
  }
}

struct ReduceSeg {
  const float* part;  // [G][stride]
  float* out;         // [n]
  int G;
  int n;              // multiple of 4
  long stride;        // multiple of 4
};
struct ReduceSegs {
  ReduceSeg s[4];
  int blocks_end[4];  // exclusive prefix of blocks per segment
  int count;
};

// 256 threads = 16 float4 columns x 16 slab groups; fixed summation order -> deterministic.
__global__ __launch_bounds__(256) void reduce_segments(ReduceSegs segs) {
  __shared__ f32x4 red[16][17];
  int k = 0, blk = blockIdx.x;
#pragma unroll
  for (int t = 0; t < 3; ++t)
    if (k == t && t + 1 < segs.count && (int)blockIdx.x >= segs.blocks_end[t]) { k = t + 1; blk = blockIdx.x - segs.blocks_end[t]; }
  const ReduceSeg sg = segs.s[k];
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const long i4 = (long)blk * 16 + col;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  if (i4 * 4 < sg.n) {
    const float* p = sg.part + i4 * 4;
    int gI = grp;
    for (; gI + 112 < sg.G; gI += 128) {   // eight loads in flight; same accumulation order as the loop below
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(gI + 16 * u) * sg.stride);
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        a0 += v[u];
        a1 += v[u + 1];
      }
    }
    for (; gI + 16 < sg.G; gI += 32) {
      f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)gI * sg.stride);
      f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (size_t)(gI + 16) * sg.stride);
      a0 += v0;
      a1 += v1;
    }
    if (gI < sg.G) a0 += *reinterpret_cast<const f32x4*>(p + (size_t)gI * sg.stride);
  }
  red[grp][col] = a0 + a1;
  __syncthreads();
  if (grp == 0 && i4 * 4 < sg.n) {
    f32x4 s = red[0][col];
#pragma unroll
    for (int t = 1; t < 16; ++t) s += red[t][col];
    *reinterpret_cast<f32x4*>(sg.out + i4 * 4) = s;
  }
}

// ---------------------------------------------------------------------------------------------
// Single-GPU SGD step tail in ONE launch: slab reduction + parameter update + bf16 shadow + the
// K-contiguous conv2 weight copy (+ step-state advance).  Replaces reduce_segments (x2), sgd_update
// and transpose_bf16: on this chip every dependent launch costs ~2 us before it does any work and
// the three small kernels were latency chains of 5-6 us each.
// Blocks [0, reduce_blocks) own 16 float4 columns of a slab-backed segment (head | conv2 | conv1):
// they sum the slabs in the fixed order of reduce_segments (bitwise identical gradients), store the
// reduced gradient (so the gradient arena stays inspectable) and apply the update to exactly those
// elements.  The remaining blocks stream the dense range whose gradient is already final (wd1).
// ---------------------------------------------------------------------------------------------
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

// One-shot peer-to-peer exchange state of the data-parallel tail (p2p_tail_allreduce_update below; protocol in p2p.hip).
constexpr int kTailFlagSlots = 1024;   // flag slots per (parity, rank): p2p.hip sizes its allocation for this
struct TailP2P {
  float* stage[8];      // [2][world][n_max] of every rank, as mapped here
  unsigned* flags[8];   // [2][world][kTailFlagSlots]
  int* status;
  int world, rank, n_max;
  unsigned seq;
  long long timeout_ticks;
  long tail_off;        // arena float offset of the first tail element (slot index = arena offset - tail_off)
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

template <bool ADAM>
__global__ __launch_bounds__(256) void fused_reduce_sgd(FusedSgdArgs a) {
  __shared__ f32x4 red[16][17];
  const float lr = a.hyper[0] * a.hyper[4];
  float lr_t = 0.f, b1 = 0.f, b2 = 0.f, eps = 0.f, gsc = 0.f;
  if (ADAM) {
    b1 = a.hyper[1]; b2 = a.hyper[2]; eps = a.hyper[3]; gsc = a.hyper[4];
    const float t = (float)*a.adam_t;   // already advanced for this update by the 1-thread kernel launched before
    lr_t = a.hyper[0] * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
  }
  if ((int)blockIdx.x < a.reduce_blocks) {
    int k = 0, blk = blockIdx.x;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (k == t && t + 1 < a.count && (int)blockIdx.x >= a.blocks_end[t]) { k = t + 1; blk = blockIdx.x - a.blocks_end[t]; }
    const FusedSeg sg = a.s[k];
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long i4 = (long)blk * 16 + col;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    const long e0 = sg.arena_off + i4 * 4;
    f32x4 wv = {0.f, 0.f, 0.f, 0.f};
    if (i4 * 4 < sg.n) {
      const float* p = sg.part + i4 * 4;
      // the master weight is requested together with the slabs (not after the reduction)
      if (grp == 0) wv = *reinterpret_cast<const f32x4*>(a.w + e0);
      int gI = grp;
      // eight, then four loads in flight per thread; the accumulation order (a0: gI, gI+32, ..; a1: gI+16, ..)
      // is reduce_segments' order, so the sums stay bitwise identical.  (The conv1 segment has 512 slabs at
      // 1024 images: two loads per trip made its 13 workgroups a chain of 15 dependent memory round trips.)
      for (; gI + 112 < sg.G; gI += 128) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(gI + 16 * u) * sg.stride);
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          a0 += v[u];
          a1 += v[u + 1];
        }
      }
      if (gI + 48 < sg.G) {
        f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)gI * sg.stride);
        f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (size_t)(gI + 16) * sg.stride);
        f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (size_t)(gI + 32) * sg.stride);
        f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (size_t)(gI + 48) * sg.stride);
        a0 += v0;
        a1 += v1;
        a0 += v2;
        a1 += v3;
        gI += 64;
      }
      for (; gI + 16 < sg.G; gI += 32) {
        f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)gI * sg.stride);
        f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (size_t)(gI + 16) * sg.stride);
        a0 += v0;
        a1 += v1;
      }
      if (gI < sg.G) a0 += *reinterpret_cast<const f32x4*>(p + (size_t)gI * sg.stride);
    }
    red[grp][col] = a0 + a1;
    __syncthreads();
    f32x4 gs = {0.f, 0.f, 0.f, 0.f};
    if (grp == 0 && i4 * 4 < sg.n) {
      gs = red[0][col];
#pragma unroll
      for (int t = 1; t < 16; ++t) gs += red[t][col];
    }
    if (grp == 0 && i4 * 4 < sg.n) {
      tail_update4<ADAM>(a, e0, gs, wv, lr, lr_t, b1, b2, eps, gsc);
    }
  } else {
    const long nb = gridDim.x - a.reduce_blocks;
    for (long i = a.dense_lo4 + (long)(blockIdx.x - a.reduce_blocks) * 256 + threadIdx.x; i < a.dense_hi4; i += nb * 256) {
      f32x4 wv = reinterpret_cast<const f32x4*>(a.w)[i];
      f32x4 gv = reinterpret_cast<const f32x4*>(a.g)[i];
      if (ADAM) {
        f32x4 mv = reinterpret_cast<const f32x4*>(a.m)[i], vv = reinterpret_cast<const f32x4*>(a.v)[i];
        adam4(wv, gv, mv, vv, lr_t, b1, b2, eps, gsc);
        reinterpret_cast<f32x4*>(a.m)[i] = mv;
        reinterpret_cast<f32x4*>(a.v)[i] = vv;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) wv[e] = fmaf(-lr, gv[e], wv[e]);
      }
      reinterpret_cast<f32x4*>(a.w)[i] = wv;
      if (a.shadow) store_bf16x4(a.shadow + 4 * i, wv[0], wv[1], wv[2], wv[3]);
    }
  }
  advance_state(a.adv);   // cursor / global_step / dropout offset: nothing in this kernel reads them
}

// ---------------------------------------------------------------------------------------------
// Data-parallel tail, coarse-grained: the already slab-reduced tail of the gradient arena (reduce_segments) makes
// the one-shot peer-to-peer round trip of p2p.hip in kTailXBlocks fat workgroups (64 system-scope fences instead of
// one per 64 elements) and the SAME launch applies the update to what it has just summed: the separate update
// launch of the data-parallel step (fused_reduce_sgd with G = 1) disappears.
// ---------------------------------------------------------------------------------------------
constexpr int kTailXBlocks = 64;

template <bool ADAM>
__global__ __launch_bounds__(256) void p2p_tail_allreduce_update(FusedSgdArgs a, TailP2P pp, int n) {
  const int tid = threadIdx.x, b = blockIdx.x, W = pp.world;
  const float lr = a.hyper[0] * a.hyper[4];
  float lr_t = 0.f, b1 = 0.f, b2 = 0.f, eps = 0.f, gsc = 0.f;
  if (ADAM) {
    b1 = a.hyper[1]; b2 = a.hyper[2]; eps = a.hyper[3]; gsc = a.hyper[4];
    const float t = (float)*a.adam_t;
    lr_t = a.hyper[0] * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
  }
  const int par = pp.seq & 1u;
  const int per = (n / 4 + kTailXBlocks - 1) / kTailXBlocks;
  const int lo4 = b * per, hi4 = min(n / 4, lo4 + per);
  const size_t slot = ((size_t)par * W + pp.rank) * pp.n_max;
  const f32x4* src = reinterpret_cast<const f32x4*>(a.g + pp.tail_off);
  for (int i = lo4 + tid; i < hi4; i += 256) {
    const f32x4 v = src[i];
#pragma unroll 1
    for (int k = 0; k < W; ++k) reinterpret_cast<f32x4*>(pp.stage[(pp.rank + k) % W] + slot)[i] = v;
  }
  __shared__ int p2p_bad;
  __threadfence_system();
  if (tid == 0) p2p_bad = 0;
  __syncthreads();
  if (tid < W) {
    __hip_atomic_store(pp.flags[tid] + ((size_t)par * W + pp.rank) * kTailFlagSlots + b, pp.seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned* f = pp.flags[pp.rank] + ((size_t)par * W + tid) * kTailFlagSlots + b;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != pp.seq) {
      if (wall_clock64() - t0 > pp.timeout_ticks) {
        *pp.status = 1;
        p2p_bad = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  // a peer never arrived: the sum would be incomplete.  Leave parameters, moments and step state untouched (the host
  // sees *status at its next health check and ends the run; a checkpoint taken meanwhile is clean)
  if (p2p_bad) return;
  const float* mine = pp.stage[pp.rank] + (size_t)par * W * pp.n_max;
  for (int i = lo4 + tid; i < hi4; i += 256) {
    const long e0 = pp.tail_off + 4L * i;
    const f32x4 wv = *reinterpret_cast<const f32x4*>(a.w + e0);
    f32x4 gs = reinterpret_cast<const f32x4*>(mine)[i];
    for (int q = 1; q < W; ++q) gs += reinterpret_cast<const f32x4*>(mine + (size_t)q * pp.n_max)[i];
    tail_update4<ADAM>(a, e0, gs, wv, lr, lr_t, b1, b2, eps, gsc);
  }
  advance_state(a.adv);
}

}  // namespace dtf

using namespace dtf;

extern "C" {

static int grid_for(long n4) {
  long b = (n4 + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

// Updates n floats at w/g (any 16-byte aligned sub-range of the arena).  state == NULL: no advance.
int dtf_sgd_update(void* w, const void* g, void* shadow, const void* hyper, long n, void* state,
                   long cursor_inc, long step_inc, long drop_inc, hipStream_t s) {
  if (n & 3) return -3;
  StepAdvance adv{(int64_t*)state, cursor_inc, step_inc, drop_inc, -1};
  hipLaunchKernelGGL(sgd_update<float>, dim3(grid_for(n / 4)), dim3(256), 0, s, (float*)w, (const float*)g,
                     (bf16*)shadow, (const float*)hyper, n / 4, adv);
  return (int)hipGetLastError();
}
// g: bf16 gradient (8-byte aligned)
int dtf_sgd_update_g16(void* w, const void* g, void* shadow, const void* hyper, long n, void* state,
                       long cursor_inc, long step_inc, long drop_inc, hipStream_t s) {
  if ((n & 3) || ((uintptr_t)g & 7)) return -3;
  StepAdvance adv{(int64_t*)state, cursor_inc, step_inc, drop_inc, -1};
  hipLaunchKernelGGL(sgd_update<bf16>, dim3(grid_for(n / 4)), dim3(256), 0, s, (float*)w, (const bf16*)g,
                     (bf16*)shadow, (const float*)hyper, n / 4, adv);
  return (int)hipGetLastError();
}

// tstate: the device step-state array; Adam's t for this range is tstate[adam_slot] (bumped after
// the update).  state == NULL: cursor/global_step/dropout are left alone.
int dtf_adam_update(void* w, const void* g, void* m, void* v, void* shadow, const void* hyper, long n,
                    void* tstate, int adam_slot, void* state, long cursor_inc, long step_inc, long drop_inc,
                    hipStream_t s) {
  if ((n & 3) || adam_slot < 3 || adam_slot > 7 || !tstate) return -3;
  hipLaunchKernelGGL(adam_update<float>, dim3(grid_for(n / 4)), dim3(256), 0, s, (float*)w, (const float*)g,
                     (float*)m, (float*)v, (bf16*)shadow, (const float*)hyper, n / 4,
                     (const int64_t*)tstate + adam_slot);
  StepAdvance adv{(int64_t*)tstate, state ? cursor_inc : 0, state ? step_inc : 0, state ? drop_inc : 0, adam_slot};
  hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(1), 0, s, adv);
  return (int)hipGetLastError();
}
int dtf_adam_update_g16(void* w, const void* g, void* m, void* v, void* shadow, const void* hyper, long n,
                        void* tstate, int adam_slot, void* state, long cursor_inc, long step_inc, long drop_inc,
                        hipStream_t s) {
  if ((n & 3) || adam_slot < 3 || adam_slot > 7 || !tstate || ((uintptr_t)g & 7)) return -3;
  hipLaunchKernelGGL(adam_update<bf16>, dim3(grid_for(n / 4)), dim3(256), 0, s, (float*)w, (const bf16*)g,
                     (float*)m, (float*)v, (bf16*)shadow, (const float*)hyper, n / 4,
                     (const int64_t*)tstate + adam_slot);
  StepAdvance adv{(int64_t*)tstate, state ? cursor_inc : 0, state ? step_inc : 0, state ? drop_inc : 0, adam_slot};
  hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(1), 0, s, adv);
  return (int)hipGetLastError();
}

int dtf_advance_state(void* state, long cursor_inc, long step_inc, long drop_inc, hipStream_t s) {
  StepAdvance adv{(int64_t*)state, cursor_inc, step_inc, drop_inc, -1};
  hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(1), 0, s, adv);
  return (int)hipGetLastError();
}

int dtf_cast_bf16(const void* w, void* shadow, long n, hipStream_t s) {
  if (n & 3) return -3;
  hipLaunchKernelGGL(cast_bf16, dim3(grid_for(n / 4)), dim3(256), 0, s, (const float*)w, (bf16*)shadow, n / 4);
  return (int)hipGetLastError();
}

// up to 4 segments: part[k] is [G[k]][stride[k]], out[k] has n[k] floats (n, stride multiples of 4,
// pointers 16-byte aligned)
int dtf_reduce_segments(int count, const void** part, void** out, const int* G, const int* n,
                        const long* stride, hipStream_t s) {
  if (count < 1 || count > 4) return -2;
  ReduceSegs segs;
  int blocks = 0;
  for (int k = 0; k < 4; ++k) {
    if (k < count) {
      if ((n[k] & 3) || (stride[k] & 3) || ((uintptr_t)part[k] & 15) || ((uintptr_t)out[k] & 15)) return -3;
      segs.s[k] = ReduceSeg{(const float*)part[k], (float*)out[k], G[k], n[k], stride[k]};
      blocks += (n[k] / 4 + 15) / 16;
    } else {
      segs.s[k] = ReduceSeg{nullptr, nullptr, 0, 0, 0};
    }
    segs.blocks_end[k] = blocks;
  }
  segs.count = count;
  hipLaunchKernelGGL(reduce_segments, dim3(blocks), dim3(256), 0, s, segs);
  return (int)hipGetLastError();
}

// Fused single-GPU SGD tail (see fused_reduce_sgd).  Segments: part[k] is [G[k]][stride[k]] and
// lands at arena float offset seg_off[k] (n[k] floats); [dense_lo, dense_hi) is the float range whose
// gradient is already final.  wc2T and out_wT (fp32 [10][H] transpose of the H x 10 tensor at out_w_off) may be NULL.
static int fused_tail_launch(bool adam, void* m, void* v, int adam_slot, int count, const void** part, const int* G,
                             const int* n, const long* stride, const long* seg_off, void* w, void* g, void* shadow,
                             const void* hyper, long dense_lo, long dense_hi, void* wc2T, long wc2_off, int c1, int c2,
                             void* out_wT, long out_w_off, int H, void* state, long cursor_inc, long step_inc,
                             long drop_inc, hipStream_t s) {
  if (count < 1 || count > 3 || (dense_lo & 3) || (dense_hi & 3) || (wc2_off & 3)) return -3;
  if (adam && (!m || !v || !state || adam_slot < 3 || adam_slot > 6)) return -3;
  if (wc2T && ((c1 % 32) || (c2 % 32))) return -3;
  FusedSgdArgs a;
  int blocks = 0;
  for (int k = 0; k < 3; ++k) {
    if (k < count) {
      if ((n[k] & 3) || (stride[k] & 3) || (seg_off[k] & 3) || ((uintptr_t)part[k] & 15)) return -3;
      a.s[k] = FusedSeg{(const float*)part[k], G[k], n[k], stride[k], seg_off[k]};
      blocks += (n[k] / 4 + 15) / 16;
    } else {
      a.s[k] = FusedSeg{nullptr, 0, 0, 0, 0};
    }
    a.blocks_end[k] = blocks;
  }
  a.count = count;
  a.reduce_blocks = blocks;
  a.w = (float*)w;
  a.g = (float*)g;
  a.shadow = (bf16*)shadow;
  a.hyper = (const float*)hyper;
  a.dense_lo4 = dense_lo / 4;
  a.dense_hi4 = dense_hi / 4;
  a.out_wT = (float*)out_wT;
  a.out_w_off = out_w_off;
  a.H = H;
  a.wc2T = (bf16*)wc2T;
  a.wc2_off = wc2_off;
  a.c1 = c1;
  a.c2 = c2;
  a.adv = StepAdvance{(int64_t*)state, cursor_inc, step_inc, drop_inc, -1};
  a.m = (float*)m;
  a.v = (float*)v;
  a.adam_t = adam ? (const int64_t*)state + adam_slot : nullptr;
  if (adam) hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(1), 0, s, StepAdvance{(int64_t*)state, 0, 0, 0, adam_slot});
  const int dense_blocks = dense_hi > dense_lo ? grid_for((dense_hi - dense_lo) / 4) : 0;
  if (adam) hipLaunchKernelGGL(fused_reduce_sgd<true>, dim3(blocks + dense_blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(fused_reduce_sgd<false>, dim3(blocks + dense_blocks), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

int dtf_fused_reduce_sgd(int count, const void** part, const int* G, const int* n, const long* stride,
                         const long* seg_off, void* w, void* g, void* shadow, const void* hyper, long dense_lo,
                         long dense_hi, void* wc2T, long wc2_off, int c1, int c2, void* out_wT, long out_w_off, int H,
                         void* state, long cursor_inc, long step_inc, long drop_inc, hipStream_t s) {
  return fused_tail_launch(false, nullptr, nullptr, -1, count, part, G, n, stride, seg_off, w, g, shadow, hyper, dense_lo,
                           dense_hi, wc2T, wc2_off, c1, c2, out_wT, out_w_off, H, state, cursor_inc, step_inc, drop_inc, s);
}

// The same tail with the Adam update (TF semantics, identical arithmetic to dtf_adam_update): two launches, a
// 1-thread kernel that advances the step counter state[adam_slot] and the fused kernel.
int dtf_fused_reduce_adam(int count, const void** part, const int* G, const int* n, const long* stride,
                          const long* seg_off, void* w, void* g, void* m, void* v, void* shadow, const void* hyper,
                          long dense_lo, long dense_hi, void* wc2T, long wc2_off, int c1, int c2, void* out_wT,
                          long out_w_off, int H, void* state, int adam_slot, long cursor_inc, long step_inc,
                          long drop_inc, hipStream_t s) {
  return fused_tail_launch(true, m, v, adam_slot, count, part, G, n, stride, seg_off, w, g, shadow, hyper, dense_lo,
                           dense_hi, wc2T, wc2_off, c1, c2, out_wT, out_w_off, H, state, cursor_inc, step_inc, drop_inc, s);
}

// Data-parallel tail, coarse-grained (p2p_tail_allreduce_update): g[tail_off, tail_off + n) holds this rank's reduced
// tail gradient; after the launch it holds the sum over ranks and the update has been applied to those elements.
int dtf_p2p_tail_update(int adam, void* w, void* g, void* m, void* v, void* shadow, const void* hyper, void* wc2T,
                        long wc2_off, int c1, int c2, void* out_wT, long out_w_off, int H, void* state, int adam_slot,
                        long cursor_inc, long step_inc, long drop_inc, void* const* bases, int world, int rank, long n,
                        long n_max, unsigned seq, int* status, double timeout_s, long tail_off, hipStream_t s) {
  if (world < 1 || world > 8 || rank < 0 || rank >= world || (n & 3) || (n_max & 3) || n > n_max || seq == 0 ||
      (tail_off & 3) || (wc2_off & 3))
    return -2;
  if (adam && (!m || !v || !state || adam_slot < 3 || adam_slot > 6)) return -3;
  if (wc2T && ((c1 % 32) || (c2 % 32))) return -3;
  FusedSgdArgs a;
  for (int k = 0; k < 3; ++k) {
    a.s[k] = FusedSeg{nullptr, 0, 0, 0, 0};
    a.blocks_end[k] = 0;
  }
  a.count = 0;
  a.reduce_blocks = 0;
  a.w = (float*)w;
  a.g = (float*)g;
  a.shadow = (bf16*)shadow;
  a.hyper = (const float*)hyper;
  a.dense_lo4 = a.dense_hi4 = 0;
  a.out_wT = (float*)out_wT;
  a.out_w_off = out_w_off;
  a.H = H;
  a.wc2T = (bf16*)wc2T;
  a.wc2_off = wc2_off;
  a.c1 = c1;
  a.c2 = c2;
  a.adv = StepAdvance{(int64_t*)state, cursor_inc, step_inc, drop_inc, -1};
  a.m = (float*)m;
  a.v = (float*)v;
  a.adam_t = adam ? (const int64_t*)state + adam_slot : nullptr;
  TailP2P pp;
  for (int r = 0; r < 8; ++r) {
    char* base = static_cast<char*>(bases[r < world ? r : rank]);
    pp.stage[r] = reinterpret_cast<float*>(base);
    pp.flags[r] = reinterpret_cast<unsigned*>(base + 2L * world * n_max * 4);
  }
  pp.status = status;
  pp.world = world;
  pp.rank = rank;
  pp.n_max = (int)n_max;
  pp.seq = seq;
  pp.timeout_ticks = (long long)(timeout_s * 1e8);
  pp.tail_off = tail_off;
  if (adam) hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(1), 0, s, StepAdvance{(int64_t*)state, 0, 0, 0, adam_slot});
  if (adam) hipLaunchKernelGGL(p2p_tail_allreduce_update<true>, dim3(kTailXBlocks), dim3(256), 0, s, a, pp, (int)n);
  else hipLaunchKernelGGL(p2p_tail_allreduce_update<false>, dim3(kTailXBlocks), dim3(256), 0, s, a, pp, (int)n);
  return (int)hipGetLastError();
}

}  // extern "C"
