// What does the shader clock do under sustained fp32 MFMA load on MI355X?  The roofline peaks of MI355X_MICROARCH.md (157.3 TF fp32 matrix)
// assume 2.4 GHz; the per-wave cycle stamps of the convolution dissection (profiles/r04_conv_deep_dissection.txt) imply ~1.7 GHz.
//   every wave: s_memtime (shader-clock counter) and s_memrealtime (100 MHz constant) around a loop of independent MFMAs;
//   variants: v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 / v_mfma_f32_4x4x1_16b_f32 / plain v_fma_f32, 1 / 2 waves per SIMD, a short and a long run.
// prints: shader cycles per real microsecond (= MHz), achieved TFLOP/s over the whole chip.
// build: hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void spin(int iters, long* out, float* sink) {
  const long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  if (KIND == 0) {
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    for (int i = 0; i < iters; ++i) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
    }
    if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 12345.f) sink[0] = 1.f;
  } else if (KIND == 1) {
    f32x4 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    for (int i = 0; i < iters; ++i) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc3, 0, 0, 0);
    }
    if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 12345.f) sink[0] = 1.f;
  } else if (KIND == 3) {
    f32x4 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0}, acc4 = {0}, acc5 = {0}, acc6 = {0}, acc7 = {0};
    for (int i = 0; i < iters; ++i) {
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc3, 0, 0, 0);
      acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc4, 0, 0, 0);
      acc5 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc5, 0, 0, 0);
      acc6 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc6, 0, 0, 0);
      acc7 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc7, 0, 0, 0);
    }
    if (acc0[0] + acc1[1] + acc2[2] + acc3[3] + acc4[0] + acc5[1] + acc6[2] + acc7[3] == 12345.f) sink[0] = 1.f;
  } else {
    float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    for (int i = 0; i < iters * 16; ++i) { x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a); }
    if (x0 + x1 + x2 + x3 == 12345.f) sink[0] = 1.f;
  }
  const long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    long* o = out + ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    o[0] = c1 - c0; o[1] = r1 - r0;
  }
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("device %s, %d CUs, clockRate %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
  const int NB = 512;
  long* out; float* sink;
  CK(hipMalloc(&out, NB * 4 * 2 * sizeof(long))); CK(hipMalloc(&sink, 64));
  std::vector<long> h(NB * 4 * 2);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[4] = {"v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x4_f32", "v_fma_f32", "v_mfma_f32_4x4x1_16b_f32"};
  for (int kind = 0; kind < 4; ++kind)
    for (int blocks : {256, 512})
      for (int iters : {2000, 20000, 200000}) {
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipEventRecord(e0));
          if (kind == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, 0, iters, out, sink);
          else if (kind == 1) hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, iters, out, sink);
          else if (kind == 2) hipLaunchKernelGGL(spin<2>, dim3(blocks), dim3(256), 0, 0, iters, out, sink);
          else hipLaunchKernelGGL(spin<3>, dim3(blocks), dim3(256), 0, 0, iters, out, sink);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), out, blocks * 4 * 2 * sizeof(long), hipMemcpyDeviceToHost));
        double sc = 0, sr = 0;
        for (int i = 0; i < blocks * 4; ++i) { sc += h[2 * i]; sr += h[2 * i + 1]; }
        const double mhz = sc / (sr / 100.0);               // real-time counter: 100 MHz
        const double flop = kind == 0 ? 4.0 * 4096 : (kind == 1 ? 4.0 * 2048 : (kind == 3 ? 8.0 * 512 : 16.0 * 4 * 2 * 64));
        const double tf = (double)blocks * 4 * iters * flop / (ms * 1e-3) / 1e12;
        printf("%-24s %3d blocks x 256 thr, %6d iters: %8.3f ms  shader clock %6.0f MHz  %6.1f TFLOP/s  (cycles per wave %.0f)\n", names[kind], blocks,
               iters, ms, mhz, tf, sc / (blocks * 4));
      }
  printf("ok\n");
  return 0;
}
