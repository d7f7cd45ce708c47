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

}  // namespace avsr
