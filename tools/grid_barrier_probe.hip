// Probe: cost of a grid-wide barrier among G co-resident workgroups (256 threads each) on gfx950.
//   flat : every workgroup adds to ONE agent-scope counter and polls it
//   tree : workgroups of an XCD add to that XCD's counter; the last arrival of each XCD adds to the global counter; all poll global
// Every wait is bounded.  hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o gbp && ./gbp
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }

template <int MODE>
__global__ __launch_bounds__(256) void bar(int* ctr, int* err, long* cycles, int iters, int G) {
  __shared__ int s_last;
  const int tid = threadIdx.x;
  const int x = xcc_id();
  int* gctr = ctr;            // [0] global
  int* xctr = ctr + 32 * (1 + x);
  int* xn = ctr + 32 * 16;    // per-XCD population, counted once
  if (tid == 0) atomicAdd(xn + 32 * x, 1);
  // initial flat barrier so populations are complete
  if (tid == 0) {
    __hip_atomic_fetch_add(gctr + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int s = 0; s < (1 << 22); ++s) if (__hip_atomic_load(gctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= G) break;
  }
  __syncthreads();
  const int pop = __hip_atomic_load(xn + 32 * x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int nx = 0;
  for (int i = 0; i < 8; ++i) nx += __hip_atomic_load(xn + 32 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
  const long t0 = wall_clock64();
  for (int it = 1; it <= iters; ++it) {
    __syncthreads();
    if (tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (MODE == 0) {
        __hip_atomic_fetch_add(gctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int s = 0;
        while (__hip_atomic_load(gctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < it * G) if (++s > (1 << 22)) { *err = 1; break; }
      } else {
        const int prev = __hip_atomic_fetch_add(xctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == it * pop - 1) __hip_atomic_fetch_add(gctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int s = 0;
        while (__hip_atomic_load(gctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < it * nx) if (++s > (1 << 22)) { *err = 1; break; }
      }
    }
    __syncthreads();
  }
  const long t1 = wall_clock64();
  if (tid == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

int main() {
  int *ctr, *err; long* cyc;
  hipMalloc(&ctr, 4096 * 4); hipMalloc(&err, 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int G : {64, 128, 256}) {
    for (int mode = 0; mode < 2; ++mode) {
      hipMemset(ctr, 0, 4096 * 4); hipMemset(err, 0, 4);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(bar<0>, dim3(G), dim3(256), 0, 0, ctr, err, cyc, iters, G);
      else hipLaunchKernelGGL(bar<1>, dim3(G), dim3(256), 0, 0, ctr, err, cyc, iters, G);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      int herr; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      printf("G=%3d %s: %.3f us per barrier (err %d)\n", G, mode ? "tree" : "flat", 1e3 * ms / iters, herr);
    }
  }
  return 0;
}
