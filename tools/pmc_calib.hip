// Known-bytes kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md "HBM": FETCH_SIZE =
// TCC_EA0_RDREQ x 64 B, so requests wider than 64 B are under-counted; "calibrate on a known byte count in your own access pattern").
// Every kernel moves exactly N bytes (printed by main) with ONE access form the engine's kernels use:
//   rd16   16 B per lane, lanes contiguous (1 KB per wave-instruction): global_load_dwordx4 -- gemm / conv loaders, records
//   rd16b  the same through raw buffer loads (buffer_load_dwordx4)              -- conv_gen / conv_q4 / persistent kernels
//   rd4    4 B per lane, lanes contiguous (256 B per wave-instruction)           -- colsum (old), scalar epilogues
//   rd8    8 B per lane
//   rd16s  16 B per lane at a 32-byte lane stride (every other 16-B piece of a line: half-line requests)
//   rdlds  16 B per lane straight into LDS (buffer_load_dwordx4 ... lds)        -- LDS-DMA form
//   rd16_cached  rd16 over a 64 MB window launched four times (fits the 256 MB Infinity Cache: are cache hits counted?)
//   wr16 / wr4   16 B / 4 B per lane stores
// tools/pmc_calibrate.py runs this under `rocprofv3 --pmc FETCH_SIZE`, `--pmc WRITE_SIZE` (separate passes) and, if the counters
// exist, `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum`, and writes profiles/r04_pmc_calibration.json.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void calib_rd16(const f32x4* p, long n, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
// the same loads over a window that fits the Infinity Cache, launched repeatedly: are cache hits counted by FETCH_SIZE?
__global__ void calib_rd16_cached(const f32x4* p, long n, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
__global__ void calib_rd16b(const float* p, long n16, float* sink) {
  // raw buffer loads over 1 GB windows (32-bit offsets)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long per = (1L << 30) / 16;
  for (long w0 = 0; w0 < n16; w0 += per) {
    const long cnt = n16 - w0 < per ? n16 - w0 : per;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p + w0 * 4), 0, 0x80000000u, 0x00020000);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (long)gridDim.x * blockDim.x) {
      typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
      acc += __builtin_bit_cast(f32x4, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(rs, (int)(i * 16), 0, 0));
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
__global__ void calib_rd4(const float* p, long n, float* sink) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ void calib_rd8(const f32x2* p, long n, float* sink) {
  f32x2 acc = {0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += p[i];
  if (acc[0] + acc[1] == 12345.678f) sink[0] = acc[0];
}
// every lane reads the FIRST 16 bytes of its own 32-byte slot: half of every 64-byte piece is touched, n slots -> 16 n bytes requested
__global__ void calib_rd16s(const f32x4* p, long nslots, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (long)gridDim.x * blockDim.x) acc += p[2 * i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
__global__ __launch_bounds__(256) void calib_rdlds(const float* p, long n16, float* sink) {
  __shared__ __attribute__((aligned(16))) float buf[4 * 256 * 4];      // 4 KB per wave
  const int wave = threadIdx.x >> 6;
  const long per = (1L << 30) / 16;
  float acc = 0.f;
  for (long w0 = 0; w0 < n16; w0 += per) {
    const long cnt = n16 - w0 < per ? n16 - w0 : per;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p + w0 * 4), 0, 0x80000000u, 0x00020000);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (long)gridDim.x * blockDim.x) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(buf + wave * 1024), 16, (int)(i * 16), 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  acc = buf[threadIdx.x];
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ void calib_wr16(f32x4* p, long n) {
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void calib_wr4(float* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 1.f;
}

int main() {
  const long BYTES = 1L << 31;                 // 2 GiB: 8x the Infinity Cache
  const long SMALL = 64L << 20;                // 64 MiB: fits it
  float *buf, *sink;
  CK(hipMalloc(&buf, BYTES));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 0, BYTES));
  CK(hipDeviceSynchronize());
  const dim3 g(256 * 8), b(256);
  printf("bytes_large %ld bytes_small %ld\n", BYTES, SMALL);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_rd16, g, b, 0, 0, (const f32x4*)buf, BYTES / 16, sink);
    hipLaunchKernelGGL(calib_rd16b, g, b, 0, 0, (const float*)buf, BYTES / 16, sink);
    hipLaunchKernelGGL(calib_rd4, g, b, 0, 0, (const float*)buf, BYTES / 4, sink);
    hipLaunchKernelGGL(calib_rd8, g, b, 0, 0, (const f32x2*)buf, BYTES / 8, sink);
    hipLaunchKernelGGL(calib_rd16s, g, b, 0, 0, (const f32x4*)buf, BYTES / 32, sink);
    hipLaunchKernelGGL(calib_rdlds, g, b, 0, 0, (const float*)buf, BYTES / 16, sink);
    hipLaunchKernelGGL(calib_wr16, g, b, 0, 0, (f32x4*)buf, BYTES / 16);
    hipLaunchKernelGGL(calib_wr4, g, b, 0, 0, buf, BYTES / 4);
    CK(hipDeviceSynchronize());
  }
  // cache-resident re-read: warm the 64 MiB window, then time-independent counters of further passes over it
  for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(calib_rd16_cached, g, b, 0, 0, (const f32x4*)buf, SMALL / 16, sink);
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  printf("ok\n");
  return 0;
}
