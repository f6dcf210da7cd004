// Core of the first conv block (5x5 SAME, one input channel, + bias + ReLU + 2x2 max-pool + argmax
// code) shared by the stand-alone forward kernel (conv1.hip) and the fused conv1 -> conv2 forward
// (conv2.hip).  See conv1.hip for the derivation of the tiling.
#pragma once
#include "common.hpp"

namespace dtf {

constexpr int kC1Pitch = 96;   // pixels per padded LDS row (48 dwords: rows of lane groups g, g+1 never share a bank)
constexpr int kC1ImgElems = 32 * kC1Pitch;   // zero-haloed 28x28 image: rows 2..29, columns 2..29

// Fused batch sampler (state != nullptr): the block derives the dataset row of image b from the
// device-resident sample cursor itself -- the stateless Feistel epoch shuffle of data.hip's
// sample_indices, bit for bit -- and publishes it in idx_out[b] for the later consumers of the
// batch (head labels, conv1 wgrad).  This removes the sampler launch from the head of every step.
struct SampleArgs {
  const int64_t* state;   // nullptr: rows come from idx (or are 0..B-1)
  int* idx_out;
  uint32_t n, hb, seed;
  int shuffle;
};

struct SampleCursor {
  uint32_t epoch, pos;
};

__device__ __forceinline__ SampleCursor sample_cursor(const SampleArgs& smp) {
  SampleCursor c = {0u, 0u};
  if (smp.state) {   // one 64-bit division per thread, once; everything after is 32-bit
    const uint64_t v = (uint64_t)smp.state[0];
    c.epoch = (uint32_t)(v / smp.n);
    c.pos = (uint32_t)(v % smp.n);
  }
  return c;
}

// dataset row of batch element b; `publish` = this thread writes idx_out[b]
__device__ __forceinline__ size_t sample_row(const SampleArgs& smp, const SampleCursor& cur, const int* idx, int b,
                                             bool publish) {
  if (smp.state) {
    const uint32_t at = cur.pos + (uint32_t)b;        // batches may span epochs (B can exceed n)
    const uint32_t epoch = cur.epoch + at / smp.n, pos = at % smp.n;
    const uint32_t r = smp.shuffle ? feistel_permute(pos, smp.n, smp.hb, mix32(smp.seed + epoch * 0x9E3779B9u)) : pos;
    if (publish) smp.idx_out[b] = (int)r;
    return r;
  }
  return idx ? (size_t)idx[b] : (size_t)b;
}

// weight fragments of one 32-channel chunk: lane (n = i16, g) holds k = 8g + j  <->  (ky, kx = j)
struct Conv1Frags {
  bf16x8 wf[2][2];
  float bv[2];
};

__device__ __forceinline__ Conv1Frags conv1_load_frags(const float* __restrict__ w, const float* __restrict__ bias,
                                                       int C, int cchunk, int lane) {
  const int g = lane >> 4, i16 = lane & 15;
  Conv1Frags f;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = cchunk * 32 + nt * 16 + i16;
    f.bv[nt] = bias[n];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f.wf[0][nt][j] = (j < 5) ? (bf16)w[(g * 5 + j) * C + n] : (bf16)0.f;
      f.wf[1][nt][j] = (j < 5 && g == 0) ? (bf16)w[(20 + j) * C + n] : (bf16)0.f;
    }
  }
  return f;
}

// One wave computes pooled row `rp` (14 pixels x 32 channels of the chunk) of the image staged in
// `img` and leaves it in its private LDS staging rows: orow [px][32 ch] bf16, arow [px][32 ch] argmax code.
__device__ __forceinline__ void conv1_pooled_row(const bf16* img, int rp, const Conv1Frags& f, bf16* orow,
                                                 uint8_t* arow, int lane) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  const int g = lane >> 4, i16 = lane & 15;
  f32x4 acc[2][2][2];   // [out row][E/O][n-tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[a][e][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int orow_i = 0; orow_i < 2; ++orow_i) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // input row (padded coords) = y + ky, ky = g (ks 0) or 4 (ks 1); pixels 2m .. 2m+9
      const int r = 2 * rp + orow_i + (ks == 0 ? g : 4);
      const unsigned* p = reinterpret_cast<const unsigned*>(img) + (r * kC1Pitch) / 2 + i16;
      unsigned d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
      u4 ev = {d0, d1, d2, d3};
      u4 od = {__builtin_amdgcn_alignbit(d1, d0, 16), __builtin_amdgcn_alignbit(d2, d1, 16),
               __builtin_amdgcn_alignbit(d3, d2, 16), __builtin_amdgcn_alignbit(d4, d3, 16)};
      bf16x8 ae = __builtin_bit_cast(bf16x8, ev), ao = __builtin_bit_cast(bf16x8, od);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc[orow_i][0][nt] = mfma16(ae, f.wf[ks][nt], acc[orow_i][0][nt]);
        acc[orow_i][1][nt] = mfma16(ao, f.wf[ks][nt], acc[orow_i][1][nt]);
      }
    }
  }
  // pool: lane (g, i16) owns pooled columns px = 4g + r of channels nt*16 + i16
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v0 = acc[0][0][nt][r], v1 = acc[0][1][nt][r], v2 = acc[1][0][nt][r], v3 = acc[1][1][nt][r];
      float best = v0;
      int bi = 0;
      if (v1 > best) { best = v1; bi = 1; }
      if (v2 > best) { best = v2; bi = 2; }
      if (v3 > best) { best = v3; bi = 3; }
      const int px = 4 * g + r;
      const float o = fmaxf(best + f.bv[nt], 0.f);
      orow[px * 32 + nt * 16 + i16] = (bf16)o;
      arow[px * 32 + nt * 16 + i16] = (uint8_t)(o > 0.f ? bi : (int)kPoolDead);
    }
}

}  // namespace dtf
