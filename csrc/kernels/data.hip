// Device-resident input pipeline and initialisers (reference
// .idea/MNISTDist.py:167,178: input_data.read_data_sets + mnist.train.next_batch
// -> host numpy -> feed_dict copy; :42-49 truncated_normal / constant init).
//
// The whole dataset stays in HBM; a "batch" is only a vector of row indices.
// Epoch shuffling is a keyed Feistel bijection evaluated on the fly, so there
// is no permutation array, no host RNG and nothing that a HIP-graph replay
// would freeze: the cursor lives in device memory (state[0]).
#include "common.hpp"

namespace dtf {

// idx[i] = perm_epoch(pos) for the global sample number cursor + i.  The one 64-bit division
// (cursor / n) is done by a single lane per block; lanes then use 32-bit arithmetic only.
__global__ void sample_indices(const int64_t* __restrict__ state, int* __restrict__ idx, int B, uint32_t n,
                               uint32_t hb, uint32_t seed, int shuffle) {
  __shared__ uint32_t s_epoch, s_pos;
  if (threadIdx.x == 0) {
    uint64_t c = (uint64_t)state[0];
    s_epoch = (uint32_t)(c / n);
    s_pos = (uint32_t)(c % n);
  }
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  uint32_t epoch = s_epoch + (s_pos + (uint32_t)i) / n;   // batches may span epochs (B can exceed n)
  uint32_t pos = (s_pos + (uint32_t)i) % n;
  idx[i] = shuffle ? (int)feistel_permute(pos, n, hb, mix32(seed + epoch * 0x9E3779B9u)) : (int)pos;
}

// Synthetic MNIST-shaped data, integer-hash only so the host generator
// (distributed_tensorflow_amd/data/synthetic.py) reproduces it exactly:
//   label  = mix32(seed ^ (i * 0x85EBCA6B)) % 10
//   pixel  = T ? (64 + u6)/128 : u6/256, T = class template bit of the 4x4 cell
__global__ void gen_synthetic(bf16* __restrict__ images, int* __restrict__ labels, long n, uint32_t seed) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = n * 784;
  for (; t < total; t += (long)gridDim.x * blockDim.x) {
    uint32_t i = (uint32_t)(t / 784), p = (uint32_t)(t % 784);
    uint32_t label = mix32(seed ^ (i * 0x85EBCA6Bu)) % 10u;
    uint32_t y = p / 28, x = p % 28;
    uint32_t cell = (y >> 2) * 7 + (x >> 2);
    uint32_t tb = mix32(0xC0FFEEu + label * 49u + cell) & 1u;
    uint32_t u6 = mix32(seed * 0x9E3779B9u + (uint32_t)t * 0x27D4EB2Fu + 0x165667B1u) >> 26;
    float v = tb ? (float)(64 + u6) * (1.0f / 128.0f) : (float)u6 * (1.0f / 256.0f);
    images[t] = (bf16)v;
    if (p == 0) labels[i] = (int)label;
  }
}

// TF truncated_normal: N(0,1) redrawn until |z| <= 2, times stddev.
__global__ void init_truncated_normal(float* __restrict__ w, long n, float stddev, uint64_t seed,
                                      uint64_t stream) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) {
    float z = 0.f;
    for (uint32_t attempt = 0; attempt < 64; ++attempt) {
      uint32_t r[4];
      Philox::run(seed, (uint64_t)i, (stream << 8) | attempt, r);
      float u1 = ((float)(r[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0,1]
      float u2 = u32_to_unit_float(r[1]);
      float rad = sqrtf(-2.0f * logf(u1));
      float z0 = rad * cosf(6.283185307179586f * u2);
      float z1 = rad * sinf(6.283185307179586f * u2);
      if (fabsf(z0) <= 2.0f) { z = z0; break; }
      if (fabsf(z1) <= 2.0f) { z = z1; break; }
    }
    w[i] = z * stddev;
  }
}

__global__ void fill_f32(float* __restrict__ w, long n, float v) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) w[i] = v;
}

// uint8 [n] (raw idx pixels) -> bf16 in [0,1]  (TF input_data: float32(x) / 255)
__global__ void u8_to_bf16_unit(const uint8_t* __restrict__ src, bf16* __restrict__ dst, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = (bf16)((float)src[i] * (1.0f / 255.0f));
}

}  // namespace dtf

using namespace dtf;

extern "C" {

static int grid1d(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

int dtf_sample_indices(const void* state, void* idx, int B, unsigned n, unsigned hb, unsigned seed,
                       int shuffle, hipStream_t s) {
  if (n == 0 || (1ull << (2 * hb)) < n) return -3;
  hipLaunchKernelGGL(sample_indices, dim3((B + 255) / 256), dim3(256), 0, s, (const int64_t*)state,
                     (int*)idx, B, n, hb, seed, shuffle);
  return (int)hipGetLastError();
}

int dtf_gen_synthetic(void* images, void* labels, long n, unsigned seed, hipStream_t s) {
  hipLaunchKernelGGL(gen_synthetic, dim3(grid1d(n * 784)), dim3(256), 0, s, (bf16*)images, (int*)labels, n,
                     seed);
  return (int)hipGetLastError();
}

int dtf_init_truncated_normal(void* w, long n, float stddev, uint64_t seed, uint64_t stream, hipStream_t s) {
  hipLaunchKernelGGL(init_truncated_normal, dim3(grid1d(n)), dim3(256), 0, s, (float*)w, n, stddev, seed,
                     stream);
  return (int)hipGetLastError();
}

int dtf_fill_f32(void* w, long n, float v, hipStream_t s) {
  hipLaunchKernelGGL(fill_f32, dim3(grid1d(n)), dim3(256), 0, s, (float*)w, n, v);
  return (int)hipGetLastError();
}

int dtf_u8_to_bf16_unit(const void* src, void* dst, long n, hipStream_t s) {
  hipLaunchKernelGGL(u8_to_bf16_unit, dim3(grid1d(n)), dim3(256), 0, s, (const uint8_t*)src, (bf16*)dst, n);
  return (int)hipGetLastError();
}

}  // extern "C"
