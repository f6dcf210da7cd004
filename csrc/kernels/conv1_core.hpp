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

// ---------------------------------------------------------------------------------------------
// Filter/bias gradient building blocks (see conv1.hip, conv1_wgrad, for the GEMM view), shared with
// the fused conv2-dgrad -> conv1-wgrad kernel in conv2.hip.
//   cp  : five column-shifted copies of the zero-haloed image, [5][32 rows][kCpPitch] (+8 pad)
//   dyt : the pool/ReLU-expanded output gradient, [rows][32 x][32 ch] bf16 with the 16-byte chunk
//         pair swizzled by (x >> 3) & 1; x = 28..31 and everything outside the image stay zero
// ---------------------------------------------------------------------------------------------
constexpr int kCpPitch = 40;                  // elements per copy row (80 B: conflict-free taps)
constexpr int kCpStride = 32 * kCpPitch + 8;  // elements per shifted copy
constexpr int kCpBytes = 5 * kCpStride * 2;   // 12880
constexpr int kDy1RowBytes = 32 * 32 * 2;     // one image row of the gradient tile

struct Conv1WgradLane {
  int a_off[2];        // element offset of (copy kx, row ky) for the two m-tiles, or -1 = constant fragment
  bf16x8 a_const[2];   // row 25 = all ones (bias gradient), rows 26.. = zero
};

__device__ __forceinline__ Conv1WgradLane conv1w_lane(int lane) {
  const int i16 = lane & 15;
  Conv1WgradLane L;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = mt * 16 + i16;
    L.a_off[mt] = (m < 25) ? (m % 5) * kCpStride + (m / 5) * kCpPitch : -1;
    const bf16 cv = (m == 25) ? (bf16)1.0f : (bf16)0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) L.a_const[mt][e] = cv;
  }
  return L;
}

// pixel pair i (0..391: pixels (y, 2*x2), (y, 2*x2+1), one dword) -> the five shifted copies.  An even
// shift keeps the pair dword-aligned, an odd one splits it into two 2-byte stores.
__device__ __forceinline__ void conv1w_write_pair(bf16* cp, int i, unsigned v) {
  const int y = i / 14, x = 2 * (i % 14);
  bf16* rowp = cp + (y + 2) * kCpPitch;
#pragma unroll
  for (int sft = 0; sft < 5; ++sft) {
    const int c = x - sft + 2;   // column of pixel x in copy sft (copy holds x = c + sft - 2)
    bf16* dst = rowp + sft * kCpStride + c;
    if ((sft & 1) == 0) {
      if (c >= 0 && c < 31) *reinterpret_cast<unsigned*>(dst) = v;
    } else {
      if (c >= 0) reinterpret_cast<unsigned short*>(dst)[0] = (unsigned short)(v & 0xffffu);
      if (c + 1 >= 0 && c + 1 < 32) reinterpret_cast<unsigned short*>(dst)[1] = (unsigned short)(v >> 16);
    }
  }
}

// item (pooled pixel (py, px), 8-channel chunk c8): 8 pooled gradients + 8 argmax codes -> the four
// window positions of the gradient tile; y_base = first image row held by dyt
__device__ __forceinline__ void conv1w_expand(bf16* dyt, int py, int px, int c8, const u32x4& gv, const u32x2& a2,
                                              int y_base) {
  u32x4 o[4];
  expand_pool_grad(gv, a2, o);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int y = 2 * py + (q >> 1) - y_base, x = 2 * px + (q & 1);
    const int slot = (((c8 >> 1) ^ ((x >> 3) & 1)) << 1) | (c8 & 1);
    *reinterpret_cast<u32x4*>(dyt + (y * 32 + x) * 32 + slot * 8) = o[q];
  }
}

// one K block (2 image rows x 16 columns; kb = 2 * row_pair + column_half) of the 32 x 32 output tile
__device__ __forceinline__ void conv1w_kblock(const bf16* cp, const bf16* dyt, int kb, int y_base,
                                              const Conv1WgradLane& L, int lane, f32x4 (&acc)[2][2]) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int g = lane >> 4, i16 = lane & 15, qr = i16 >> 2, pc = i16 & 3;
  const int yrow = 2 * (kb >> 1) + (g >> 1);
  const int xb = (kb & 1) * 16 + 8 * (g & 1);
  bf16x8 bfrag[2], afrag[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const bf16* p0 = dyt + ((yrow - y_base) * 32 + xb + qr) * 32 + ((j ^ ((xb >> 3) & 1)) << 4) + pc * 4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * 32));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    bfrag[j] = __builtin_bit_cast(bf16x8, v);
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    // every lane reads (constant rows read a valid dummy address) and then selects
    const int off = (L.a_off[mt] >= 0 ? L.a_off[mt] : 0) + yrow * kCpPitch + xb;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(cp + off);
    afrag[mt] = L.a_off[mt] >= 0 ? v : L.a_const[mt];
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[mt][j] = mfma16(afrag[mt], bfrag[j], acc[mt][j]);
}

// cross-wave (4 waves) reduction through `red` (>= 16 KB of LDS, free to overwrite), then rows
// m < 26 of the 32 x 32 tile go to the slab [26][C] at channel chunk cchunk.  Call after a barrier.
__device__ __forceinline__ void conv1w_reduce_store(const f32x4 (&acc)[2][2], float* red, float* slab, int C,
                                                    int cchunk, int wave, int lane) {
  const int g = lane >> 4, i16 = lane & 15;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 16 + (mt * 2 + j) * 4 + r) * 64 + lane] = acc[mt][j][r];
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = ((mt * 2 + j) * 4 + r) * 64 + lane;
        const float s = (red[e] + red[16 * 64 + e]) + (red[32 * 64 + e] + red[48 * 64 + e]);
        const int m = mt * 16 + 4 * g + r;
        if (m < 26) slab[m * C + cchunk * 32 + j * 16 + i16] = s;
      }
}

}  // namespace dtf
