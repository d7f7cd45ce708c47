// Probe: XCD placement of workgroups and the latency / staleness of two inter-workgroup hand-off protocols.
//   local : plain payload stores + s_waitcnt + sc0 flag store | sc1 flag poll + 16-B sc1 payload loads   (same-XCD pairs)
//   agent : sc1 payload stores + s_waitcnt + agent atomic add | agent poll + 16-B sc1 payload loads        (any placement)
// hipcc --offload-arch=gfx950 -O3 tools/xcd_probe.hip -o xcd_probe && ./xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }   // HW_REG_XCC_ID[3:0]

__global__ void where(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

__device__ __forceinline__ f32x4 ld16_sc1(const float* base, int elem_off) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, elem_off * 4, 0, 16);
  return __builtin_bit_cast(f32x4, v);
}

// pairs: block p (ping side) and partner; payload 256 floats per direction; iters round trips
template <int MODE>
__global__ __launch_bounds__(64) void pingpong(float* buf, int* flags, int* stats, int iters, int stride) {
  const int pair = blockIdx.x / (2 * stride) * stride + blockIdx.x % stride;
  const int side = (blockIdx.x / stride) & 1;
  float* mine = buf + (long)(pair * 2 + side) * 256;
  const float* theirs = buf + (long)(pair * 2 + (side ^ 1)) * 256;
  int* myflag = flags + (pair * 2 + side) * 32;
  const int* theirflag = flags + (pair * 2 + (side ^ 1)) * 32;
  const int lane = threadIdx.x;
  int stale = 0, timeouts = 0;
  const long t0 = wall_clock64();
  for (int it = 1; it <= iters; ++it) {
    if (side == 0 || it > 0) {
      // wait for partner's previous message (side 1 waits for `it`, side 0 waits for it-1)
      const int need = side == 0 ? it - 1 : it;
      if (need > 0) {
        int spins = 0;
        while (true) {
          int f = __hip_atomic_load(theirflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (f >= need) break;
          if (++spins > (1 << 20)) { ++timeouts; break; }
        }
        f32x4 v = ld16_sc1(theirs, lane * 4);
        if (v[0] != (float)need || v[3] != (float)need) ++stale;
      }
    }
    // publish
    f32x4 w = {(float)it, (float)it, (float)it, (float)it};
    if (MODE == 0) {
      *reinterpret_cast<f32x4*>(mine + lane * 4) = w;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(myflag, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      for (int e = 0; e < 4; ++e) __hip_atomic_store(mine + lane * 4 + e, w[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(myflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  const long t1 = wall_clock64();
  if (lane == 0) { stats[blockIdx.x * 4 + 0] = (int)(t1 - t0); stats[blockIdx.x * 4 + 1] = timeouts; stats[blockIdx.x * 4 + 3] = xcc_id(); }
  atomicAdd(&stats[blockIdx.x * 4 + 2], stale);
}

int main() {
  const int NB = 64;
  int* d; hipMalloc(&d, NB * 4);
  where<<<NB, 64>>>(d);
  std::vector<int> h(NB); hipMemcpy(h.data(), d, NB * 4, hipMemcpyDeviceToHost);
  printf("xcc of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %d", h[i]); printf("\n");
  int bad = 0; for (int i = 0; i < NB; ++i) bad += (h[i] != h[i % 8]); printf("blocks not on xcc[b%%8]: %d\n", bad);
  float* buf; int* flags; int* stats;
  hipMalloc(&buf, 64 * 2 * 256 * 4); hipMalloc(&flags, 64 * 2 * 32 * 4); hipMalloc(&stats, 256 * 16);
  const int iters = 2000;
  for (int mode = 0; mode < 2; ++mode)
    for (int stride : {8, 1}) {       // partner = block + stride: 8 -> same XCD (if round-robin), 1 -> neighbouring XCD
      if (mode == 0 && stride == 1) continue;   // the local protocol is only defined for same-XCD pairs
      hipMemset(buf, 0, 64 * 2 * 256 * 4); hipMemset(flags, 0, 64 * 2 * 32 * 4); hipMemset(stats, 0, 256 * 16);
      const int nb = 16;             // 8 pairs
      if (mode == 0) pingpong<0><<<nb, 64>>>(buf, flags, stats, iters, stride);
      else pingpong<1><<<nb, 64>>>(buf, flags, stats, iters, stride);
      hipDeviceSynchronize();
      std::vector<int> s(nb * 4); hipMemcpy(s.data(), stats, nb * 16, hipMemcpyDeviceToHost);
      double us = 0; int to = 0, st = 0;
      for (int b = 0; b < nb; ++b) { us += s[b * 4] / 100.0; to += s[b * 4 + 1]; st += s[b * 4 + 2]; }
      printf("mode %s partner+%d: %.3f us per one-way hop, timeouts %d, stale lanes %d (xcc of block0=%d, block%d=%d)\n",
             mode == 0 ? "local(plain+sc0 flag)" : "agent(sc1+atomic)", stride, us / nb / iters / 2.0, to, st, s[3], stride, s[stride * 4 + 3]);
    }
  return 0;
}
