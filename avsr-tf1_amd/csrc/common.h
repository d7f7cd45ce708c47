// Shared device/host helpers for the gfx950 (MI355X, wave64) AVSR hot-path kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AVSR_OK 0
#define AVSR_ERR_ARG (-1)
#define AVSR_ERR_HIP (-2)
#define AVSR_ERR_UNSUPPORTED (-3)

#define AVSR_CHECK_LAUNCH()                         \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return AVSR_ERR_HIP;     \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace avsr {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- stateless counter RNG, bit-identical to oracle/avsr_oracle.py::hash_u32 -------------
__host__ __device__ __forceinline__ uint32_t hash_u32(uint32_t seed, uint32_t stream, uint32_t idx) {
  uint32_t key = seed * 0x9E3779B9u + stream * 0x85EBCA6Bu;
  uint32_t x = idx ^ key;
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ float uniform01(uint32_t seed, uint32_t stream, uint32_t idx) {
  return (float)(hash_u32(seed, stream, idx) >> 8) * (1.0f / 16777216.0f);
}

// ---- wave64 reductions ------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over aligned groups of 16 lanes
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum over 256 threads (4 waves); red must hold >= 4 floats; all threads get the result
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---------------------------------------------------------------------------------------------
// Device fills / copies as KERNELS.  The engine never enqueues hipMemsetAsync / hipMemcpyAsync: captured into a hipGraph they become
// memset / memcpy nodes, and graphs holding such nodes misbehaved on ROCm 7.0 -- a kernel node keeps every
// dependency on the compute queue.  All sizes are multiples of 4 bytes (float / int32 buffers); 2-D forms take byte pitches.
__global__ static void k_fill_words(uint32_t* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0u;
}
__global__ static void k_copy_words(uint32_t* d, const uint32_t* s, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ static void k_fill_words_2d(uint32_t* p, long pitch_w, long width_w, long rows) {
  const long total = width_w * rows;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) p[(i / width_w) * pitch_w + i % width_w] = 0u;
}
__global__ static void k_copy_words_2d(uint32_t* d, long dpitch_w, const uint32_t* s, long spitch_w, long width_w, long rows) {
  const long total = width_w * rows;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    d[(i / width_w) * dpitch_w + i % width_w] = s[(i / width_w) * spitch_w + i % width_w];
}
static inline int k_blocks(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }
static inline hipError_t dev_zero(void* p, size_t bytes, hipStream_t s) {
  if (!bytes) return hipSuccess;
  if (bytes % 4 || !p) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_fill_words, dim3(k_blocks((long)(bytes / 4))), dim3(256), 0, s, (uint32_t*)p, (long)(bytes / 4));
  return hipGetLastError();
}
static inline hipError_t dev_copy(void* d, const void* src, size_t bytes, hipStream_t s) {
  if (!bytes) return hipSuccess;
  if (bytes % 4 || !d || !src) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_copy_words, dim3(k_blocks((long)(bytes / 4))), dim3(256), 0, s, (uint32_t*)d, (const uint32_t*)src, (long)(bytes / 4));
  return hipGetLastError();
}
static inline hipError_t dev_zero_2d(void* p, size_t pitch, size_t width, size_t rows, hipStream_t s) {
  if (!width || !rows) return hipSuccess;
  if (pitch % 4 || width % 4 || !p) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_fill_words_2d, dim3(k_blocks((long)(width / 4 * rows))), dim3(256), 0, s, (uint32_t*)p, (long)(pitch / 4), (long)(width / 4), (long)rows);
  return hipGetLastError();
}
static inline hipError_t dev_copy_2d(void* d, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, hipStream_t s) {
  if (!width || !rows) return hipSuccess;
  if (dpitch % 4 || spitch % 4 || width % 4 || !d || !src) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_copy_words_2d, dim3(k_blocks((long)(width / 4 * rows))), dim3(256), 0, s, (uint32_t*)d, (long)(dpitch / 4), (const uint32_t*)src,
                     (long)(spitch / 4), (long)(width / 4), (long)rows);
  return hipGetLastError();
}


// Several independent fills / copies in ONE launch (the set-up of a decoder block was six 4-5 us launches in a row on the step's critical
// chain).  The operations of a batch must not depend on each other; 2-D forms take byte pitches like dev_copy_2d.
#define DEV_BATCH_MAX 12
struct DevOp { uint32_t* d; const uint32_t* s; long dpitch, spitch, width, rows; };      // s == NULL: fill with zeros; all in words
struct DevBatchArgs { int n; int blk0[DEV_BATCH_MAX + 1]; DevOp op[DEV_BATCH_MAX]; };
__global__ static void k_dev_batch(const DevBatchArgs A) {
  int j = 0;
#pragma unroll 1
  for (int k = 1; k < A.n; ++k) if ((int)blockIdx.x >= A.blk0[k]) j = k;
  j = __builtin_amdgcn_readfirstlane(j);
  uint32_t* const d = A.op[j].d; const uint32_t* const s = A.op[j].s;
  const long dp = A.op[j].dpitch, sp = A.op[j].spitch, w = A.op[j].width, total = w * A.op[j].rows;
  const long nb = A.blk0[j + 1] - A.blk0[j];
  for (long i = (long)((int)blockIdx.x - A.blk0[j]) * blockDim.x + threadIdx.x; i < total; i += nb * blockDim.x) {
    const long r = i / w, c = i - r * w;
    d[r * dp + c] = s ? s[r * sp + c] : 0u;
  }
}
struct DevBatch {
  DevBatchArgs a;
  hipStream_t st;
  bool bad;
  explicit DevBatch(hipStream_t s) : st(s), bad(false) { a.n = 0; a.blk0[0] = 0; }
  hipError_t flush() {
    if (bad) return hipErrorInvalidValue;
    if (!a.n) return hipSuccess;
    hipLaunchKernelGGL(k_dev_batch, dim3(a.blk0[a.n]), dim3(256), 0, st, a);
    a.n = 0; a.blk0[0] = 0;
    return hipGetLastError();
  }
  void add(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t rows) {
    if (!width || !rows) return;
    if (dpitch % 4 || spitch % 4 || width % 4 || !d) { bad = true; return; }
    if (a.n == DEV_BATCH_MAX && flush() != hipSuccess) { bad = true; return; }
    DevOp& o = a.op[a.n];
    o.d = (uint32_t*)d; o.s = (const uint32_t*)s; o.dpitch = (long)(dpitch / 4); o.spitch = (long)(spitch / 4); o.width = (long)(width / 4); o.rows = (long)rows;
    int nb = k_blocks(o.width * o.rows);
    if (nb > 256) nb = 256;
    a.blk0[a.n + 1] = a.blk0[a.n] + nb;
    ++a.n;
  }
  void zero(void* p, size_t bytes) { add(p, bytes, nullptr, 0, bytes, 1); }
  void copy(void* d, const void* s, size_t bytes) { if (!s) bad = true; else add(d, bytes, s, bytes, bytes, 1); }
  void zero_2d(void* p, size_t pitch, size_t width, size_t rows) { add(p, pitch, nullptr, 0, width, rows); }
  void copy_2d(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t rows) { if (!s) bad = true; else add(d, dpitch, s, spitch, width, rows); }
};

}  // namespace avsr
