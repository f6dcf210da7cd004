// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// Everything here assumes wave64, the unified VGPR/AGPR file and the
// 16x16x32 bf16 MFMA fragment maps of CDNA4:
//   A frag: lane l holds A[row = l&15][k = 8*(l>>4) + j], j = 0..7
//   B frag: lane l holds B[k = 8*(l>>4) + j][col = l&15]
//   C/D   : lane l holds D[row = 4*(l>>4) + r][col = l&15], r = 0..3
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "diag.hpp"

namespace dtf {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kWave = 64;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// Counter-based RNG: Philox4x32-10 (Salmon et al.), used for dropout masks and
// the truncated-normal initialiser.  Stateless: (key, counter) -> 4 x u32.
// ---------------------------------------------------------------------------
struct Philox {
  static __host__ __device__ __forceinline__ void mulhilo(uint32_t a, uint32_t b,
                                                         uint32_t& hi, uint32_t& lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
  }
  static __host__ __device__ __forceinline__ void run(uint64_t key, uint64_t ctr_lo,
                                                     uint64_t ctr_hi, uint32_t out[4]) {
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
    uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      uint32_t hi0, lo0, hi1, lo1;
      mulhilo(0xD2511F53u, c0, hi0, lo0);
      mulhilo(0xCD9E8D57u, c2, hi1, lo1);
      uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
};

__host__ __device__ __forceinline__ float u32_to_unit_float(uint32_t x) {
  // [0,1) with 24 bits of mantissa.
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------------------
// Stateless epoch shuffle: a 4-round Feistel network over 2^(2*hb) >= n with
// cycle walking gives a bijection on [0, n).  Mirrors
// distributed_tensorflow_amd/data/sampler.py::feistel_permute bit for bit.
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

__host__ __device__ __forceinline__ uint32_t feistel_permute(uint32_t i, uint32_t n,
                                                            uint32_t hb, uint32_t key) {
  const uint32_t mask = (1u << hb) - 1u;
  do {
    uint32_t L = i >> hb, R = i & mask;
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
      uint32_t t = L ^ (mix32(R ^ key ^ (r * 0x9E3779B9u)) & mask);
      L = R;
      R = t;
    }
    i = (L << hb) | R;
  } while (i >= n);
  return i;
}

// Counter-based dropout (TF: x * floor(keep + u) / keep): the mask of element `elem` at dropout
// offset `offset` is a pure function of (seed, elem, offset), so it is never stored.
struct DropoutArgs {
  float keep_prob;        // 1.0 -> disabled
  uint64_t seed;
  const int64_t* state;   // device step state; state[2] = dropout offset (graph-replay safe)
};

__device__ __forceinline__ float dropout_scale(const DropoutArgs& d, uint64_t offset, uint64_t elem) {
  uint32_t r[4];
  Philox::run(d.seed, elem, offset, r);
  return u32_to_unit_float(r[0]) < d.keep_prob ? 1.0f / d.keep_prob : 0.f;
}

// ---------------------------------------------------------------------------
// Max-pool + ReLU gradient routing.  The forward kernels store, per pooled element, the 2x2
// argmax q in 0..3 -- or kPoolDead when the pooled activation is not positive -- so backward
// needs the u8 code only (no re-read of the pooled tensor).  expand_pool_grad turns 8 pooled
// gradients (4 dwords of bf16 pairs) + 8 codes into the 4 window positions: o[q] = g where
// code == q, else 0.  Pure 32-bit integer ops on packed pairs.
// ---------------------------------------------------------------------------
constexpr unsigned kPoolDead = 4;

// Packed 16-bit formulation (v_pk_lshlrev_b16 / v_pk_ashrrev_i16): the two codes of a dword go to its
// 16-bit halves, onehot = 1 << code, and the mask of window position q is bit q of onehot smeared over
// the half (shift it to bit 15, arithmetic shift back).  3 independent VALU ops per (dword, q) and no
// VCC: the compare + select version was a serial v_cmp -> s_nop -> v_cndmask chain through one VCC
// (~1300 cycles per call in conv2 wgrad's staging, tools/conv_timeline.py).
__device__ __forceinline__ void expand_pool_grad(const u32x4& g, const u32x2& a, u32x4 (&o)[4]) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const unsigned ab = (d < 2 ? a[0] : a[1]) >> ((d & 1) * 16);
    const unsigned c = (ab & 0xffu) | ((ab & 0xff00u) << 8);          // codes 0..4 in the two halves
    const s16x2 onehot = (s16x2){1, 1} << __builtin_bit_cast(s16x2, c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const s16x2 sh = {(short)(15 - q), (short)(15 - q)};
      const s16x2 m = (onehot << sh) >> (s16x2){15, 15};              // 0xffff where code == q
      o[q][d] = g[d] & __builtin_bit_cast(unsigned, m);
    }
  }
}

// Wave-level sum across all 64 lanes, result in every lane (wave-uniform).  Six DPP adds on the VALU
// (quad swaps, half-row / row mirrors, row broadcasts into the next rows; the total lands in lane 63 and is
// read back as a scalar) instead of six __shfl_xor = ds_bpermute round trips through the LDS pipe: the ten
// reductions of the head kernel were 2.4 us of a 7 us row (tools/conv_timeline.py run 1024 head).
#define DTF_DPP_ADD(v, ctrl, row_mask) \
  ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (row_mask), 0xf, false)))
// the total is valid in lane 63 only (for a caller whose lane 63 consumes it itself: no scalar round trip)
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = DTF_DPP_ADD(v, 0xB1, 0xf);    // quad_perm [1,0,3,2]
  v = DTF_DPP_ADD(v, 0x4E, 0xf);    // quad_perm [2,3,0,1]
  v = DTF_DPP_ADD(v, 0x141, 0xf);   // row_half_mirror: sums of 8
  v = DTF_DPP_ADD(v, 0x140, 0xf);   // row_mirror: every lane holds its row's sum of 16
  v = DTF_DPP_ADD(v, 0x142, 0xa);   // row_bcast15 into rows 1 and 3
  v = DTF_DPP_ADD(v, 0x143, 0xc);   // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum_lane63(v)), 63));
}

// Sum / maximum over the 16 lanes of a DPP row, result in every lane of the row (four VALU steps)
__device__ __forceinline__ float row16_sum(float v) {
  v = DTF_DPP_ADD(v, 0xB1, 0xf);
  v = DTF_DPP_ADD(v, 0x4E, 0xf);
  v = DTF_DPP_ADD(v, 0x141, 0xf);
  v = DTF_DPP_ADD(v, 0x140, 0xf);
  return v;
}
#define DTF_DPP_MAX(v, ctrl) \
  fmaxf((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false)))
__device__ __forceinline__ float row16_max(float v) {
  v = DTF_DPP_MAX(v, 0xB1);
  v = DTF_DPP_MAX(v, 0x4E);
  v = DTF_DPP_MAX(v, 0x141);
  v = DTF_DPP_MAX(v, 0x140);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------
// Raw buffer loads: unconditional (out-of-range reads return 0), no per-load address VGPR pair, and --
// unlike loads through a const __restrict__ pointer -- never sunk to their first use by the scheduler,
// so a prefetch stays where it is written (pair with __builtin_amdgcn_sched_barrier(0)).
// ---------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
  const uint32_t n = bytes > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ u32x2 buf_load8(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}

// XCD-aware tile assignment.  Workgroups are dealt round-robin over the 8 XCDs (private 4 MiB
// L2 each), so linear ids L and L+8 share an L2.  These GEMMs are small enough that every tile is
// resident at once and the whole grid marches through K in step; what bounds them is fabric/HBM
// traffic (measured: ~11 B/clk/CU, i.e. operand panels re-fetched by every XCD).  Giving each XCD a
// compact BH x (chunk/BH) rectangle of tiles makes its L2 fetch each A/B panel once per rectangle
// instead of once per tile.  Bijective for any tile count (guide T1).
__device__ __forceinline__ void xcd_tile_linear(int L, int gx, int gy, int& tx, int& ty) {
  constexpr int NX = 8, BH = 4;
  const int total = gx * gy;
  const int xcd = L % NX, slot = L / NX;
  const int q = total / NX, r = total % NX;
  const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  // banded column-major order: bands of BH tile rows, walked column by column
  const int band = idx / (BH * gx);
  const int rem = idx - band * BH * gx;
  const int h = min(BH, gy - band * BH);
  tx = rem / h;
  ty = band * BH + rem % h;
}
__device__ __forceinline__ void xcd_tile(int gx, int gy, int& tx, int& ty) {
  xcd_tile_linear(blockIdx.x + gx * blockIdx.y, gx, gy, tx, ty);
}

// ---------------------------------------------------------------------------
// Producer / consumer hand-over inside a workgroup WITHOUT s_barrier (conv5x5_wgrad_pc, the ring GEMMs): one progress
// word PER WAVE in LDS (a shared counter cannot tell two steps of a fast wave from one step of two waves).  LDS
// operations of one wave execute in order, so the word's update (release) follows the wave's own LDS writes / reads;
// a waiting wave polls all words of the other side with one read (lane l reads word l & n_mask) + s_sleep.
// The spin is bounded: a wave that gives up computes garbage, but the grid always drains.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void pc_signal(unsigned* word, unsigned steps_done, int lane) {
  if (lane == 0) __hip_atomic_store(word, steps_done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void pc_wait_all_ge(unsigned* words, int n_mask, unsigned target, int lane) {
  for (int spin = 0; spin < (1 << 22); ++spin) {
    const unsigned v = __hip_atomic_load(words + (lane & n_mask), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (__builtin_amdgcn_ballot_w64(v < target) == 0ull) return;
    __builtin_amdgcn_s_sleep(1);
  }
}

}  // namespace dtf
