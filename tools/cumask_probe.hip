// Do CU-masked streams (hipExtStreamCreateWithCUMask) partition MI355X by XCD, and do two masked streams run side by side?
// Background: the persistent recurrences hold every CU's registers while their pipes are 25-39 % busy, so the
// lip CNN cannot co-reside; a spatial split (recurrence on XCDs 0-3, CNN on XCDs 4-7) needs the dispatcher to honour CU masks.
//   part 1: census -- for a set of masks, which XCC_ID / CU the workgroups of a 2048-block launch land on
//   part 2: a 2 ms spin kernel on one half beside a streaming kernel on the other half: alone vs together (wall clock)
// build: hipcc --offload-arch=gfx950 -O3 tools/cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void census(int* xcc, int* hwid) {
  if (threadIdx.x == 0) {
    xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;          // HW_REG_XCC_ID[3:0]
    hwid[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);              // HW_REG_HW_ID: wave, simd, cu, sh, se ...
  }
  // stay resident for a moment so that the launch spreads over every eligible CU
  const long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
}
__global__ void spin(long ticks, int* out) {
  const long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) {}
  if (threadIdx.x == 0 && out) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void stream_copy(const f32x4* a, f32x4* b, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  printf("device %s, %d CUs\n", pr.name, pr.multiProcessorCount);
  const int NB = 2048;
  int *xcc, *hwid;
  CK(hipMalloc(&xcc, NB * 4)); CK(hipMalloc(&hwid, NB * 4));
  std::vector<int> hx(NB), hh(NB);
  struct M { const char* name; uint32_t w[8]; };
  std::vector<M> masks;
  { M m = {"all 256", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}}; masks.push_back(m); }
  { M m = {"bits 0-127", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}}; masks.push_back(m); }
  { M m = {"bits 128-255", {0, 0, 0, 0, ~0u, ~0u, ~0u, ~0u}}; masks.push_back(m); }
  { M m = {"bits 0-31", {~0u, 0, 0, 0, 0, 0, 0, 0}}; masks.push_back(m); }
  { M m = {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}}; masks.push_back(m); }
  { M m = {"bits i with (i%8)<4", {0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu}}; masks.push_back(m); }
  { M m = {"bits i with (i%8)>=4", {0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u}}; masks.push_back(m); }
  { M m = {"bit 0 only", {1u, 0, 0, 0, 0, 0, 0, 0}}; masks.push_back(m); }
  { M m = {"bits 0-7", {0xffu, 0, 0, 0, 0, 0, 0, 0}}; masks.push_back(m); }
  std::vector<hipStream_t> st(masks.size());
  for (size_t i = 0; i < masks.size(); ++i) {
    hipError_t e = hipExtStreamCreateWithCUMask(&st[i], 8, masks[i].w);
    if (e != hipSuccess) { printf("mask '%s': hipExtStreamCreateWithCUMask failed: %s\n", masks[i].name, hipGetErrorString(e)); st[i] = nullptr; continue; }
    CK(hipMemsetAsync(xcc, 0xff, NB * 4, st[i]));
    hipLaunchKernelGGL(census, dim3(NB), dim3(64), 0, st[i], xcc, hwid);
    CK(hipStreamSynchronize(st[i]));
    CK(hipMemcpy(hx.data(), xcc, NB * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hh.data(), hwid, NB * 4, hipMemcpyDeviceToHost));
    int hist[16] = {0};
    std::vector<int> cus;
    for (int b = 0; b < NB; ++b) {
      hist[hx[b] & 15]++;
      const int key = ((hx[b] & 15) << 16) | (hh[b] & 0xffff00);      // (xcc, se/sh/cu fields)
      bool seen = false;
      for (int c : cus) if (c == key) { seen = true; break; }
      if (!seen) cus.push_back(key);
    }
    printf("mask '%-22s': blocks per XCC:", masks[i].name);
    for (int x = 0; x < 8; ++x) printf(" %4d", hist[x]);
    printf("   distinct (xcc, hw-id cu fields): %zu\n", cus.size());
  }
  // ---- part 2: concurrency of two half-chip streams ----
  const long N = 1L << 28;                                 // 256 Mi floats4?  -> 1 GiB per buffer
  f32x4 *a, *b;
  CK(hipMalloc(&a, N)); CK(hipMalloc(&b, N));
  CK(hipMemset(a, 0, N));
  int* sx; CK(hipMalloc(&sx, 4096 * 4));
  auto run = [&](hipStream_t sa, hipStream_t sb, bool do_spin, bool do_copy, const char* what) {
    CK(hipDeviceSynchronize());
    const double t0 = now();
    if (do_spin) hipLaunchKernelGGL(spin, dim3(128 * 3), dim3(256), 0, sa, (long)200000 * 100 / 10, sx);   // 100 MHz timer: ~20 ms? (ticks)
    if (do_copy) for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(stream_copy, dim3(2048), dim3(256), 0, sb, a, b, N / 16);
    CK(hipDeviceSynchronize());
    printf("%-58s %.3f ms\n", what, 1e3 * (now() - t0));
  };
  for (int rep = 0; rep < 2; ++rep) {
    run(st[1], st[2], true, false, "spin (384 blocks x 256 thr) on mask bits 0-127 alone");
    run(st[1], st[2], false, true, "8 x 1 GiB copy on mask bits 128-255 alone");
    run(st[1], st[2], true, true, "both, disjoint masks");
    run(st[0], st[0], false, true, "8 x 1 GiB copy on the unmasked stream");
    run(st[5], st[6], true, false, "spin on mask (i%8)<4 alone");
    run(st[5], st[6], false, true, "copy on mask (i%8)>=4 alone");
    run(st[5], st[6], true, true, "both, interleaved masks");
  }
  printf("ok\n");
  return 0;
}
