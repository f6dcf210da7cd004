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
#include "optim_core.hpp"

namespace dtf {

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

}  // extern "C"
