// Helpers of the exact-fp32 kernel set (--dtype fp32): the reference computes in float32 end to end
// (reference .idea/MNISTDist.py:113-115 float32 placeholders, :43,48 float32 variables), so this set keeps
// every activation, weight and gradient in fp32 and multiplies on the CDNA4 f32-input matrix cores:
//
//   v_mfma_f32_32x32x2_f32 : A lane l -> A[i = l&31][k = l>>5], B lane l -> B[k = l>>5][j = l&31]
//                            C/D 16 regs: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
//   v_mfma_f32_16x16x4_f32 : A lane l -> A[i = l&15][k = l>>4], B lane l -> B[k = l>>4][j = l&15]
//                            C/D 4 regs : col = l&15, row = 4*(l>>4) + r
//
// Both are bit-for-bit a k-ordered fmaf chain (one rounding per product, fp32 accumulate) and run at
// the f32 vector rate (64 FLOP/clk/SIMD = 157 TF/s per chip): 1/16 of the bf16 matrix rate, so in
// these kernels the matrix pipe is the bound by a wide margin and everything else (staging, LDS
// traffic, epilogues) only has to stay out of its way.
//
// Operand idiom used throughout: a lane fetches FOUR consecutive k of its row/column with one 16-byte
// read (element e <-> k = kblock + 4*(lane k-group) + e) and issues four MFMAs, one per e; A and B use
// the same k <-> (group, e) map, and any bijection of k is a valid GEMM.
#pragma once
#include "common.hpp"

namespace dtf {

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ f32x16 mfma32f(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16f(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4x4 outer products, block = lane >> 2; A lane l -> A_block[i = l&3],
// B lane l -> B_block[j = l&3], C/D 4 regs: col = l&3, rows r.  Two passes (8 cycles) for 512 FLOP: the same
// 64 FLOP/clk/SIMD with a quarter of the rows of a 16x16x4.
__device__ __forceinline__ f32x4 mfma4f(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 buf_load16f(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// Max-pool / ReLU gradient routing in fp32: 4 pooled gradients + their 4 argmax codes (one byte each, kPoolDead
// where the pooled activation is not positive) -> the four window positions q = 2*dy + dx.
__device__ __forceinline__ void expand_pool_grad_f32(const f32x4& g, unsigned codes, f32x4 (&o)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned c = (codes >> (8 * e)) & 0xffu;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q][e] = (c == (unsigned)q) ? g[e] : 0.f;
  }
}

}  // namespace dtf
