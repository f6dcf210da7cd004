// Classifier head: logits GEMM (N = 10) fused with softmax cross-entropy
// forward AND backward, the accuracy metric, and the ReLU/dropout gradient of
// fc1 (reference .idea/MNISTDist.py:89 logits, :148 reduce_mean(softmax_xent),
// :152-153 argmax-equality accuracy; TF's SoftmaxCrossEntropyWithLogits kernel
// also emits softmax - y, which is what this kernel hands to backward).
//
// N = 10 is far too skinny for the matrix cores, so one workgroup owns one batch row:
// threads stride the hidden dimension with 16-byte loads, reduce the 10
// partial dot products across the workgroup, and every thread then holds the full
// logits vector.  The row's loss, correctness bit, dlogits and the gradient
// wrt the fc1 pre-activation (da = (h1 > 0) * (dlogits . W^T) / keep) are all
// produced from registers in the same launch; h1 is stored post-ReLU and
// post-dropout, so "h1 > 0" is exactly relu'(a) * dropout_mask.
//
// head_wgrad reduces over the batch: dW_out = h1^T . dlogits, db_out, and the
// fc1 bias gradient db_d1 = colsum(da), as deterministic split-K slabs.
#include "common.hpp"
#include "head_wgrad_core.hpp"

// Diagnostic (-DDTF_TIMING=1, tools/conv_timeline.py run .. head): thread 0 of every workgroup stamps the 100 MHz
// wall clock at the phase boundaries of its first row

namespace dtf {

#if DTF_TIMING
__device__ long long* g_head_timing = nullptr;
#define DTF_HSTAMP(slot)                                                                                     \
  do {                                                                                                       \
    if (threadIdx.x == 0 && g_head_timing && b == (int)blockIdx.x) g_head_timing[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define DTF_HSTAMP(slot) do {} while (0)
#endif

constexpr int kCls = 10;

// Optional fused fc1 epilogue: when `part` is given, the row of h1 is first produced here from the
// split-K partial products of the fc1 GEMM (sum of S slabs + bias -> ReLU -> dropout, exactly the
// arithmetic of gemm.hip's splitk_bias_relu_dropout) and written out for head_wgrad.
struct Fc1Epilogue {
  const float* part;      // [S][B][H] fp32 or nullptr (h1 is an input then)
  const float* bias1;     // [H]
  int S;
  DropoutArgs drop;
};

// Sum of S split-K slabs in slab order, ((p0 + p1) + p2) + ..., for the VQ float4s a thread owns, with four slabs
// in flight per wait: the plain "load, add, next" loop compiles to a chain of S dependent memory round trips (ISA: one
// `s_waitcnt vmcnt(0)` per slab).
template <int VQ>
__device__ __forceinline__ void sum_slabs4(const float* p, size_t stride, int S, f32x4 (&acc)[VQ]) {
  auto ld = [&](int i, int q) { return *reinterpret_cast<const f32x4*>(p + (size_t)i * stride + 4 * q); };
  int i;
  if (S >= 4) {
    f32x4 v[4][VQ];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < VQ; ++q) v[k][q] = ld(k, q);
#pragma unroll
    for (int q = 0; q < VQ; ++q) acc[q] = ((v[0][q] + v[1][q]) + v[2][q]) + v[3][q];
    i = 4;
  } else {
#pragma unroll
    for (int q = 0; q < VQ; ++q) acc[q] = ld(0, q);
    i = 1;
  }
  for (; i + 4 <= S; i += 4) {
    f32x4 v[4][VQ];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < VQ; ++q) v[k][q] = ld(i + k, q);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < VQ; ++q) acc[q] += v[k][q];
  }
  for (; i < S; ++i)
#pragma unroll
    for (int q = 0; q < VQ; ++q) acc[q] += ld(i, q);
}

// labels: int32 dataset labels; idx: optional gather (label = labels[idx[b]]).
// wT: the output weight TRANSPOSED, fp32 [10][H].  A lane owns 4 consecutive hidden units per
// 256-wide iteration, so every load below is one contiguous 1 KiB (fp32) or 512 B (bf16) wave
// access.  With the native [H][10] layout each lane walked 80 consecutive floats and a single wave
// load touched 64 cache lines; the texture path made that 9 us for 2 MB of activations.
// One workgroup of NW waves per batch row (grid-stride); a thread owns V = 4 VQ consecutive hidden units of every
// 1024-wide slice.  Two instantiations:
//  * <4 waves, V = 4>: any H.  4 waves per row instead of one, so a CU holds 16+ independent load streams (one wave
//    per row left each SIMD with a single latency-bound wave).
//  * <2 waves, V = 8, ONEPASS>: H == 1024, a thread sees exactly one slice.  The slice of h1 and the ten slices of wT
//    stay in registers for the backward half instead of being loaded again behind the row's own stores, and the
//    per-wave fixed work (ten 64-lane reductions, softmax, the cross-wave exchange) is paid by two waves per row, not
//    four: at 1024 rows the kernel is bound by VALU issue and by the rate at which 4096 waves can be launched
//    (in-kernel stamps, tools/conv_timeline.py run 1024 head: workgroup starts spread over 2.7 us of a 5.4 us kernel).
//    ONE wave per row (16 units per lane, no LDS, no barriers, four rows to a workgroup) was measured too: its row
//    takes 4.4 us against 2.8 us, step 0.1429 vs 0.1416 ms -- the latency chain of a lone wave per SIMD again.
//    TWO rows per 256-thread workgroup (half as many workgroups to dispatch, same waves): 0.1405 vs 0.1408 ms --
//    the start spread is a wave-launch rate, not a workgroup-dispatch rate; not kept.
template <typename AT, int NW, int V, bool ONEPASS>
__global__ __launch_bounds__(NW * 64) void head_fwd_bwd(
    AT* __restrict__ h1, Fc1Epilogue epi, const float* __restrict__ wT, const float* __restrict__ bias,
    const int* __restrict__ labels, const int* __restrict__ idx, float* __restrict__ logits_out,
    float* __restrict__ loss_rows, float* __restrict__ correct_rows, float* __restrict__ dlogits,
    AT* __restrict__ da, int B, int H, float grad_scale, float inv_keep, int backward) {
  static_assert(NW * 64 * V == 1024 && (NW == 2 || NW == 4) && V % 4 == 0, "NW waves cover a 1024-wide slice per pass");
  constexpr int VQ = V / 4;
  __shared__ __attribute__((aligned(16))) float zred[kCls][4];   // [class][wave]: one 16-byte read per class
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool do_drop = epi.part && epi.drop.keep_prob < 1.0f;
  const uint64_t drop_off = (do_drop && epi.drop.state) ? (uint64_t)epi.drop.state[2] : 0;
  const size_t BH = (size_t)B * H;
  const __amdgpu_buffer_rsrc_t lab_rs = make_rsrc(labels, 0x7ffffffcull);
  const __amdgpu_buffer_rsrc_t idx_rs = make_rsrc(idx ? idx : labels, 0x7ffffffcull);

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    DTF_HSTAMP(0);
    // two dependent loads (gather index, then label), needed only after the reductions.  As plain (scalar) loads the
    // compiler sinks the second one to its use -- a memory round trip between the two barriers of the reduction --
    // so both are raw buffer loads (never sunk), the first issued here with the row's other loads, the second once
    // those have arrived.  (Always a valid address: a guarded load is a branch with a full wait.)
    const int li_raw = __builtin_amdgcn_raw_buffer_load_b32(idx_rs, 0, idx ? b * 4 : 0, 0);
    AT* hrow = h1 + (size_t)b * H;
    float z[kCls];
#pragma unroll
    for (int c = 0; c < kCls; ++c) z[c] = 0.f;
    f32x4 hkeep[VQ], wkeep[kCls][VQ];
    for (int j0 = tid * V, once = 1; ONEPASS ? once : (j0 < H); j0 += 1024, once = 0) {
      f32x4 hf[VQ];
      if (epi.part) {
        f32x4 bv[VQ];
#pragma unroll
        for (int q = 0; q < VQ; ++q) bv[q] = *reinterpret_cast<const f32x4*>(epi.bias1 + j0 + 4 * q);
        sum_slabs4<VQ>(epi.part + (size_t)b * H + j0, BH, epi.S, hf);
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = fmaxf(hf[q][e] + bv[q][e], 0.f);
            if (do_drop) v *= dropout_scale(epi.drop, drop_off, (uint64_t)b * (uint64_t)H + j0 + 4 * q + e);
            hf[q][e] = (float)(AT)v;   // downstream math sees exactly the stored value (bf16-rounded on the mixed path)
          }
          store_act4(hrow + j0 + 4 * q, hf[q]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < VQ; ++q) hf[q] = load_act4(hrow + j0 + 4 * q);
      }
#pragma unroll
      for (int c = 0; c < kCls; ++c) {
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wT + (size_t)c * H + j0 + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) z[c] = fmaf(hf[q][e], wv[e], z[c]);
          if (ONEPASS) wkeep[c][q] = wv;
        }
      }
      if (ONEPASS) {
#pragma unroll
        for (int q = 0; q < VQ; ++q) hkeep[q] = hf[q];
      }
    }
    const int label = __builtin_amdgcn_raw_buffer_load_b32(lab_rs, (idx ? li_raw : b) * 4, 0, 0);
    DTF_HSTAMP(1);   // h1 row produced + partial logits
    // fixed-order reduction: lanes (DPP tree), then the waves in index order
#pragma unroll
    for (int c = 0; c < kCls; ++c) z[c] = wave_sum_lane63(z[c]);
    __syncthreads();   // zred free (previous row consumed)
    if (lane == 63) {
#pragma unroll
      for (int c = 0; c < kCls; ++c) zred[c][wave] = z[c];
    }
    __syncthreads();
    // Lane-parallel softmax: lane 16 r + c of every wave owns class c (c >= 10: padding), so the row costs one LDS
    // read, one exp and two 16-lane DPP reductions per lane instead of ten of each (the replicated softmax was a
    // third of this kernel's instructions).
    const int cls = lane & 15;
    const bool live = cls < kCls;
    const int cidx = live ? cls : 0;
    const f32x4 zw = *reinterpret_cast<const f32x4*>(&zred[cidx][0]);
    const float zc = (NW == 4 ? (zw[0] + zw[1]) + (zw[2] + zw[3]) : zw[0] + zw[1]) + bias[cidx];

    DTF_HSTAMP(2);   // logits reduced
    const float zmax = row16_max(live ? zc : -INFINITY);
    const float pc = live ? __expf(zc - zmax) : 0.f;
    const float se = row16_sum(pc);                       // fixed tree order; identical in all four rows
    const float inv = 1.f / se;
    const int label_u = __builtin_amdgcn_readfirstlane(label);
    // first class that attains the maximum (the tie-break of a sequential scan); rows replicate, row 0 has the low bits
    const unsigned eq = (unsigned)__ballot(live && zc == zmax) & 0x3ffu;
    const int amax = eq ? __builtin_ctz(eq) : 0;
    const float zl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zc), label_u & 15));
    const float dlc = (pc * inv - (cls == label_u ? 1.f : 0.f)) * grad_scale;
    float dl[kCls];
#pragma unroll
    for (int c = 0; c < kCls; ++c) dl[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlc), c));

    if (tid == 0) {
      loss_rows[b] = __logf(se) + zmax - zl;
      correct_rows[b] = (amax == label_u) ? 1.f : 0.f;
    }
    if (logits_out && tid < kCls) logits_out[(size_t)b * kCls + tid] = zc;
    DTF_HSTAMP(3);   // softmax / loss
    if (!backward) continue;
    if (tid < kCls) dlogits[(size_t)b * kCls + tid] = dlc;
    AT* drow = da + (size_t)b * H;
    for (int j0 = tid * V, once = 1; ONEPASS ? once : (j0 < H); j0 += 1024, once = 0) {
#pragma unroll
      for (int q = 0; q < VQ; ++q) {
        const f32x4 hv = ONEPASS ? hkeep[q] : load_act4(hrow + j0 + 4 * q);   // own earlier store when fused
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < kCls; ++c) {
          const f32x4 wv = ONEPASS ? wkeep[c][q] : *reinterpret_cast<const f32x4*>(wT + (size_t)c * H + j0 + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) sacc[e] = fmaf(dl[c], wv[e], sacc[e]);
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (hv[e] > 0.f) ? sacc[e] * inv_keep : 0.f;
        store_act4(drow + j0 + 4 * q, o);
      }
    }
    DTF_HSTAMP(4);   // da row stored
  }
}

// out[c][j] = in[j][c]  (fp32; keeps the transposed output weight of head_fwd_bwd current)
__global__ __launch_bounds__(256) void transpose_head_w(const float* __restrict__ in, float* __restrict__ out, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // index into out
  if (i < H * kCls) out[i] = in[(size_t)(i % H) * kCls + i / H];
}

// Debug export of the dropout keep mask the fused fc1 epilogue applies for (seed, offset): out[i] = 1 if element i
// (= row * H + column) is kept.  Same dropout_scale() call as head_fwd_bwd; tests compare it with the host mirror
// (ops/reference.py dropout_mask_reference) and feed it to the oracle.
__global__ __launch_bounds__(256) void dropout_mask_export(DropoutArgs drop, uint64_t offset, unsigned char* __restrict__ out,
                                                           long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = dropout_scale(drop, offset, (uint64_t)i) != 0.f ? 1 : 0;
}

// K9: batch reductions of the per-row loss / correctness the head kernel leaves behind (reference .idea/MNISTDist.py:148
// reduce_mean(cost), :153 reduce_mean(cast(correct_pred))).  One workgroup, fixed order (deterministic), fp64 partial
// sums; acc[0] = sum of losses, acc[1] = number of correct rows.  accumulate != 0 adds to acc (evaluation walks a
// dataset in chunks and reads acc once at the end: one host synchronisation per evaluation).
__global__ __launch_bounds__(256) void metrics_reduce(const float* __restrict__ loss_rows, const float* __restrict__ correct_rows,
                                                      int n, double* __restrict__ acc, int accumulate) {
  __shared__ double red[2][256];
  double l = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    l += (double)loss_rows[i];
    c += (double)correct_rows[i];
  }
  red[0][threadIdx.x] = l;
  red[1][threadIdx.x] = c;
  __syncthreads();
  for (int h = 128; h >= 1; h >>= 1) {
    if ((int)threadIdx.x < h) {
      red[0][threadIdx.x] += red[0][threadIdx.x + h];
      red[1][threadIdx.x] += red[1][threadIdx.x + h];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    acc[0] = (accumulate ? acc[0] : 0.0) + red[0][0];
    acc[1] = (accumulate ? acc[1] : 0.0) + red[1][0];
  }
}

// grid = (H/256, KS).  slab layout per k-split: [H*10 dW_out][12: db_out padded][H db_d1]
// (body in head_wgrad_core.hpp: the training step runs it inside the fc1 dgrad GEMM launch instead)
// NSUB waves per workgroup.  The fp32 set runs 8, like the guest workgroups of its fc1 dgrad launch (gemm_f32 HEADW):
// the two routes then sum in the same order, and the data-parallel step (guests) stays bit-identical to a replay
// through the stand-alone kernel -- fp32 has no bf16 rounding of the weights to absorb a last-bit difference.
template <typename AT, int NSUB>
__global__ __launch_bounds__(NSUB * 64) void head_wgrad(HeadWgradArgsT<AT> a) {
  __shared__ __attribute__((aligned(16))) float red[head_wgrad_lds_floats<NSUB>()];
  head_wgrad_block<NSUB, AT>(a, blockIdx.x, blockIdx.y, red);
}

}  // namespace dtf

using namespace dtf;

#if DTF_TIMING
extern "C" int dtf_head_set_timing_buf(void* p) {
  long long* q = (long long*)p;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dtf::g_head_timing), &q, sizeof(q));
}
#endif

// DTF_HEAD_ONEPASS=0 / dtf_head_set_onepass(0): the general (reloading) kernel also at H == 1024 (A/B measurements,
// the bitwise test of the two)
static int g_head_onepass = -1;
static bool head_onepass() {
  if (g_head_onepass < 0) {
    const char* e = getenv("DTF_HEAD_ONEPASS");
    g_head_onepass = !(e && e[0] == '0');
  }
  return g_head_onepass != 0;
}
extern "C" int dtf_head_set_onepass(int on) {   // < 0: back to the default
  g_head_onepass = on < 0 ? -1 : (on ? 1 : 0);
  return 0;
}

template <typename AT>
static int head_launch(void* h1, Fc1Epilogue epi, const void* wT, const void* bias, const void* labels,
                       const void* idx, void* logits_out, void* loss_rows, void* correct_rows, void* dlogits, void* da,
                       int B, int H, float grad_scale, float inv_keep, int backward, hipStream_t s) {
  if (H % 4) return -3;
  int blocks = B < 4096 ? B : 4096;
  const bool onepass = H == 1024 && head_onepass();
  auto kern = onepass ? head_fwd_bwd<AT, 2, 8, true> : head_fwd_bwd<AT, 4, 4, false>;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(onepass ? 128 : 256), 0, s, (AT*)h1, epi, (const float*)wT,
                     (const float*)bias, (const int*)labels, (const int*)idx, (float*)logits_out,
                     (float*)loss_rows, (float*)correct_rows, (float*)dlogits, (AT*)da, B, H, grad_scale,
                     inv_keep, backward);
  return (int)hipGetLastError();
}

extern "C" {

// h1 [B][H] bf16 is an input; wT = output weight transposed, fp32 [10][H]
int dtf_head_fwd_bwd(const void* h1, const void* wT, const void* bias, const void* labels, const void* idx,
                     void* logits_out, void* loss_rows, void* correct_rows, void* dlogits, void* da, int B,
                     int H, float grad_scale, float inv_keep, int backward, hipStream_t s) {
  return head_launch<bf16>(const_cast<void*>(h1), Fc1Epilogue{nullptr, nullptr, 0, DropoutArgs{1.0f, 0, nullptr}}, wT, bias,
                           labels, idx, logits_out, loss_rows, correct_rows, dlogits, da, B, H, grad_scale, inv_keep,
                           backward, s);
}

// the same with fp32 activations (h1, da): --dtype fp32
int dtf_head_f32_fwd_bwd(const void* h1, const void* wT, const void* bias, const void* labels, const void* idx,
                         void* logits_out, void* loss_rows, void* correct_rows, void* dlogits, void* da, int B,
                         int H, float grad_scale, float inv_keep, int backward, hipStream_t s) {
  return head_launch<float>(const_cast<void*>(h1), Fc1Epilogue{nullptr, nullptr, 0, DropoutArgs{1.0f, 0, nullptr}}, wT, bias,
                            labels, idx, logits_out, loss_rows, correct_rows, dlogits, da, B, H, grad_scale, inv_keep,
                            backward, s);
}

// fused fc1 epilogue + head: h1 = dropout(relu(sum_s part[s] + bias1)) is produced (and stored) here
int dtf_head_fused_fwd_bwd(const void* part, int S, const void* bias1, float keep_prob, uint64_t seed,
                           const void* state, void* h1, const void* wT, const void* bias, const void* labels,
                           const void* idx, void* logits_out, void* loss_rows, void* correct_rows, void* dlogits,
                           void* da, int B, int H, float grad_scale, float inv_keep, int backward, hipStream_t s) {
  if (!part || S < 1) return -3;
  Fc1Epilogue epi{(const float*)part, (const float*)bias1, S, DropoutArgs{keep_prob, seed, (const int64_t*)state}};
  return head_launch<bf16>(h1, epi, wT, bias, labels, idx, logits_out, loss_rows, correct_rows, dlogits, da, B, H,
                           grad_scale, inv_keep, backward, s);
}

int dtf_head_f32_fused_fwd_bwd(const void* part, int S, const void* bias1, float keep_prob, uint64_t seed,
                               const void* state, void* h1, const void* wT, const void* bias, const void* labels,
                               const void* idx, void* logits_out, void* loss_rows, void* correct_rows, void* dlogits,
                               void* da, int B, int H, float grad_scale, float inv_keep, int backward, hipStream_t s) {
  if (!part || S < 1) return -3;
  Fc1Epilogue epi{(const float*)part, (const float*)bias1, S, DropoutArgs{keep_prob, seed, (const int64_t*)state}};
  return head_launch<float>(h1, epi, wT, bias, labels, idx, logits_out, loss_rows, correct_rows, dlogits, da, B, H,
                            grad_scale, inv_keep, backward, s);
}

// acc (2 doubles) = [sum loss_rows[0..n), sum correct_rows[0..n)]; accumulate != 0: added to what acc holds
int dtf_metrics_reduce(const void* loss_rows, const void* correct_rows, int n, void* acc, int accumulate, hipStream_t s) {
  if (n < 0 || !acc) return -2;
  hipLaunchKernelGGL(metrics_reduce, dim3(1), dim3(256), 0, s, (const float*)loss_rows, (const float*)correct_rows, n,
                     (double*)acc, accumulate);
  return (int)hipGetLastError();
}

// wT[10][H] = w[H][10]^T (fp32)
int dtf_transpose_head_w(const void* w, void* wT, int H, hipStream_t s) {
  hipLaunchKernelGGL(transpose_head_w, dim3((H * kCls + 255) / 256), dim3(256), 0, s, (const float*)w, (float*)wT, H);
  return (int)hipGetLastError();
}

// part: [KS][H*10 + 12 + H] fp32
int dtf_head_wgrad(const void* h1, const void* dlogits, const void* da, void* part, int B, int H, int KS,
                   hipStream_t s) {
  if (KS <= 0) return -2;
  int rps = (B + KS - 1) / KS;
  if (H % 4) return -3;
  HeadWgradArgs a{(const bf16*)h1, (const float*)dlogits, (const bf16*)da, (float*)part, B, H, rps,
                  (H + kHeadCols - 1) / kHeadCols};
  hipLaunchKernelGGL((head_wgrad<bf16, 4>), dim3(a.nbx, KS), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

int dtf_head_f32_wgrad(const void* h1, const void* dlogits, const void* da, void* part, int B, int H, int KS,
                       hipStream_t s) {
  if (KS <= 0) return -2;
  int rps = (B + KS - 1) / KS;
  if (H % 4) return -3;
  HeadWgradArgsT<float> a{(const float*)h1, (const float*)dlogits, (const float*)da, (float*)part, B, H, rps,
                          (H + kHeadCols - 1) / kHeadCols};
  hipLaunchKernelGGL((head_wgrad<float, 8>), dim3(a.nbx, KS), dim3(512), 0, s, a);
  return (int)hipGetLastError();
}

int dtf_dropout_mask(uint64_t seed, uint64_t offset, float keep_prob, void* out, long n, hipStream_t s) {
  if (n <= 0 || !out) return -3;
  DropoutArgs drop{keep_prob, seed, nullptr};
  const long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(dropout_mask_export, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, s, drop, offset,
                     (unsigned char*)out, n);
  return (int)hipGetLastError();
}

}  // extern "C"
