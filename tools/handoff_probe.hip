// Probe: what a step-to-step all-to-all hand-off costs UNDER LOAD inside one XCD, for the two traffic patterns of the persistent encoder
// kernels, with the shipped protocol and with self-validating (tagged) operands.
//
//   pattern F (forward, rnn_persist.hip):  96 workgroups x 256 threads per XCD in 3 "layers" of 32; per step a workgroup needs the whole
//              h(t-1) of its layer for its 8 rows (8 KB, 16-byte L1-bypassing loads) and produces 8 rows x 8 units.
//   pattern B (BPTT, rnn_persist_bwd.hip K-split): 64 workgroups x 512 threads per XCD in 4 "cells" of 16; per step a workgroup publishes
//              a [16 x 256] partial slab (16 KB) and each of 256 threads sums the 16 producers' values of its (row, unit).
//
//   protocol 0 (shipped): plain stores -> s_waitcnt vmcnt(0) -> barrier -> progress word; consumer: wave 0 polls the progress words ->
//              barrier -> operand loads.       Chain per step: store ack + flag propagation + poll round trip + operand round trip.
//   protocol 1 (tagged):  every 16-byte (F: 3 values + tag) / 8-byte (B: value + tag) granule carries the step number; a granule is
//              written by ONE store instruction, so it is valid or stale as a whole.  The producer never drains or flags; the consumer
//              loads its operand granules and retries until every tag matches.     Chain per step: store -> operand round trip.
//   `work`: dependent FMAs per step standing in for MFMA phase + epilogue.
//
// hipcc --offload-arch=gfx950 -O3 tools/handoff_probe.hip -o /tmp/handoff_probe && /tmp/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define OOB ((int)0x80000000)
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x80000000u, 0x00020000); }
__device__ __forceinline__ f32x4 ld16(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16)); }
__device__ __forceinline__ f32x2 ld8(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_bit_cast(f32x2, (u32x2)__builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 16)); }
__device__ __forceinline__ float ld4s(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 16)); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Args { float* ring; float* rec; int* flags; int* claim; int* stat; int steps, proto, work; };

__device__ __forceinline__ float busy(float x, int n) {
  for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
  return x;
}
__device__ __forceinline__ float val(int t, int layer, int row, int unit) { return (float)((t * 7 + layer * 3 + row * 5 + unit) & 1023); }

// ------------------------------------------------------------------ pattern F
__global__ __launch_bounds__(256) void fwd_pattern(const Args A) {
  __shared__ int s_slot;
  __shared__ float red[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gx = __builtin_amdgcn_readfirstlane(xcc_id());
  if (tid == 0) s_slot = atomicAdd(A.claim + gx, 1);
  __syncthreads();
  const int slot = __builtin_amdgcn_readfirstlane(s_slot);
  if (slot >= 96) return;
  const int layer = slot >> 5, ct = slot & 31;
  // rings per (xcd, layer): proto 0 [2][8 rows][256]; proto 1 [2][8 rows][32 prod][3 chunks][4]
  const int ring_floats = A.proto ? 2 * 8 * 32 * 12 : 2 * 8 * 256;
  float* const ring = A.ring + (long)(gx * 3 + layer) * ring_floats;
  const __amdgpu_buffer_rsrc_t rr = rsrc(ring);
  int* const flags = A.flags + (gx * 3 + layer) * 32;
  float* const rec = A.rec + ((long)(gx * 96 + slot) * A.steps) * 64 * 8;        // per-step records: 8 floats per (row, unit)
  const int row = lane >> 3, unit = lane & 7;
  long bad = 0, retries = 0;
  float sink = 0.f;
  const long t0 = wall_clock64();
  for (int t = 1; t <= A.steps; ++t) {
    float acc = 0.f;
    if (A.proto == 0) {
      if (wave == 0 && t > 1) {
        for (int spins = 0; spins < (1 << 15); ++spins) {
          const int v = lane < 32 ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
          if (__all(v >= t - 1)) break;
          if (spins == (1 << 15) - 1) ++retries;
        }
      }
      lds_barrier();
      if (t > 1) {
        const int base = ((t - 1) & 1) * 8 * 256 * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = tid + 256 * j, r = c >> 6, k = (c & 63) * 4;          // chunk c: row r, floats k .. k+3
          const f32x4 v = ld16(rr, base + (r * 256 + k) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { bad += v[e] != val(t - 1, layer, r, k + e); acc += v[e]; }
        }
      }
    } else {
      lds_barrier();
      if (t > 1) {
        const int base = ((t - 1) & 1) * 8 * 32 * 12 * 4;
        const float want = (float)(t - 1);
        f32x4 v[3];
        for (int spins = 0; spins < (1 << 15); ++spins) {
          asm volatile("" ::: "memory");              // (the buffer-load builtin is not volatile: keep the poll inside the loop)
#pragma unroll
          for (int j = 0; j < 3; ++j) v[j] = ld16(rr, base + (tid + 256 * j) * 16);
          const bool ok = v[0][3] == want && v[1][3] == want && v[2][3] == want;
          if (__all(ok)) break;
          ++retries;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int c = tid + 256 * j, r = c / 96, p = (c % 96) / 3, cc = c % 3;
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            const int u = cc * 3 + e;
            if (u < 8) bad += v[j][e] != val(t - 1, layer, r, p * 8 + u);
            acc += v[j][e];
          }
        }
      }
    }
    acc = busy(acc, A.work);
    red[tid] = acc;
    lds_barrier();
    if (wave == 0) {
      sink += red[lane] + red[lane + 64] + red[lane + 128] + red[lane + 192];
      const float mine = val(t, layer, row, ct * 8 + unit);
      if (A.proto == 0) {
        ring[(t & 1) * 8 * 256 + row * 256 + ct * 8 + unit] = mine;
      } else {
        const float n1 = __shfl_down(mine, 1, 64), n2 = __shfl_down(mine, 2, 64);
        if (unit % 3 == 0) {
          f32x4 w = {mine, unit + 1 < 8 ? n1 : 0.f, unit + 2 < 8 ? n2 : 0.f, (float)t};
          *reinterpret_cast<f32x4*>(ring + (t & 1) * 8 * 32 * 12 + ((row * 32 + ct) * 3 + unit / 3) * 4) = w;
        }
      }
      // the step's records (gates 16 B, cell state, two more output copies): time-indexed, never read back here
      float* const rp = rec + ((long)(t - 1) * 64 + lane) * 8;
      *reinterpret_cast<f32x4*>(rp) = f32x4{mine, mine, mine, mine};
      rp[4] = mine; rp[5] = mine; rp[6] = mine;
    }
    if (A.proto == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + ct, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  const long t1 = wall_clock64();
  if (tid == 0) { A.stat[(gx * 96 + slot) * 4] = (int)(t1 - t0); A.stat[(gx * 96 + slot) * 4 + 3] = (int)sink; }
  atomicAdd(A.stat + (gx * 96 + slot) * 4 + 1, (int)bad);
  atomicAdd(A.stat + (gx * 96 + slot) * 4 + 2, (int)retries);
}

// ------------------------------------------------------------------ pattern B
__global__ __launch_bounds__(512) void bwd_pattern(const Args A) {
  __shared__ int s_slot;
  __shared__ float red[512];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gx = __builtin_amdgcn_readfirstlane(xcc_id());
  if (tid == 0) s_slot = atomicAdd(A.claim + gx, 1);
  __syncthreads();
  const int slot = __builtin_amdgcn_readfirstlane(s_slot);
  if (slot >= 64) return;
  const int cell = slot >> 4, ct = slot & 15;
  // slabs per (xcd, cell): [2 parities][16 producers][16 rows][256 units] x (1 float | value + tag)
  const int G = A.proto ? 2 : 1;
  const int slab = 16 * 256 * G;
  float* const ring = A.ring + (long)(gx * 4 + cell) * 2 * 16 * slab;
  const __amdgpu_buffer_rsrc_t rr = rsrc(ring);
  int* const flags = A.flags + (gx * 4 + cell) * 32;
  float* const rec = A.rec + ((long)(gx * 64 + slot) * A.steps) * 256 * 4;
  const int er = (tid & 255) >> 4, eu = tid & 15;
  long bad = 0, retries = 0;
  float sink = 0.f;
  const long t0 = wall_clock64();
  for (int t = 1; t <= A.steps; ++t) {
    float acc = 0.f;
    if (A.proto == 0) {
      if (wave == 0 && t > 1) {
        for (int spins = 0; spins < (1 << 15); ++spins) {
          const int v = lane < 16 ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
          if (__all(v >= t - 1)) break;
        }
      }
      lds_barrier();
      if (t > 1 && tid < 256) {
        const int o = (((t - 1) & 1) * 16 * slab + er * 256 + ct * 16 + eu) * 4;
        float ps[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) ps[p] = ld4s(rr, o + p * slab * 4);
#pragma unroll
        for (int p = 0; p < 16; ++p) { bad += ps[p] != val(t - 1, cell, er, p * 16 + ct * 16 + eu); acc += ps[p]; }
      }
    } else {
      lds_barrier();
      if (t > 1 && tid < 256) {
        const int o = (((t - 1) & 1) * 16 * slab + (er * 256 + ct * 16 + eu) * 2) * 4;
        const float want = (float)(t - 1);
        f32x2 ps[16];
        for (int spins = 0; spins < (1 << 15); ++spins) {
          bool ok = true;
          asm volatile("" ::: "memory");
#pragma unroll
          for (int p = 0; p < 16; ++p) ps[p] = ld8(rr, o + p * slab * 4);
#pragma unroll
          for (int p = 0; p < 16; ++p) ok = ok && ps[p][1] == want;
          if (__all(ok)) break;
          ++retries;
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) { bad += ps[p][0] != val(t - 1, cell, er, p * 16 + ct * 16 + eu); acc += ps[p][0]; }
      }
    }
    acc = busy(acc, A.work);
    red[tid] = acc;
    lds_barrier();
    sink += red[tid ^ 1];
    // my [16 x 256] partial slab: wave w owns units [32 w, 32 w + 32), lane -> (row, 4 units) x 2
    float* const dst = ring + (long)((t & 1) * 16 + ct) * slab;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = lane + 64 * j, rrow = idx >> 3, c4 = idx & 7, n = wave * 32 + c4 * 4;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = val(t, cell, rrow, ct * 16 + n + e);          // (consumer p = ct reads unit n + e of ITS tile: see check above)
      if (A.proto == 0) {
        *reinterpret_cast<f32x4*>(dst + rrow * 256 + n) = v;
      } else {
        const float tg = (float)t;
        *reinterpret_cast<f32x4*>(dst + (rrow * 256 + n) * 2) = f32x4{v[0], tg, v[1], tg};
        *reinterpret_cast<f32x4*>(dst + (rrow * 256 + n) * 2 + 4) = f32x4{v[2], tg, v[3], tg};
      }
    }
    if (tid < 256) *reinterpret_cast<f32x4*>(rec + ((long)(t - 1) * 256 + tid) * 4) = f32x4{acc, acc, acc, acc};     // d gates record
    if (A.proto == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + ct, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  const long t1 = wall_clock64();
  if (tid == 0) { A.stat[(gx * 64 + slot) * 4] = (int)(t1 - t0); A.stat[(gx * 64 + slot) * 4 + 3] = (int)sink; }
  atomicAdd(A.stat + (gx * 64 + slot) * 4 + 1, (int)bad);
  atomicAdd(A.stat + (gx * 64 + slot) * 4 + 2, (int)retries);
}

// ------------------------------------------------------------------ pattern B2: the same BPTT exchange with 1024-thread workgroups of 32 units
// (8 producers per cell instead of 16: half the slab bytes through the L2 and half the partial loads per consumer), protocol 0 only
__global__ __launch_bounds__(1024) void bwd_pattern_1024(const Args A) {
  __shared__ int s_slot;
  __shared__ float red[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gx = __builtin_amdgcn_readfirstlane(xcc_id());
  if (tid == 0) s_slot = atomicAdd(A.claim + gx, 1);
  __syncthreads();
  const int slot = __builtin_amdgcn_readfirstlane(s_slot);
  if (slot >= 32) return;
  const int cell = slot >> 3, ct = slot & 7;
  const int slab = 16 * 256;
  float* const ring = A.ring + (long)(gx * 4 + cell) * 2 * 8 * slab;
  const __amdgpu_buffer_rsrc_t rr = rsrc(ring);
  int* const flags = A.flags + (gx * 4 + cell) * 32;
  float* const rec = A.rec + ((long)(gx * 32 + slot) * A.steps) * 512 * 4;
  const int er = (tid & 511) >> 5, eu = tid & 31;
  long bad = 0;
  float sink = 0.f;
  const long t0 = wall_clock64();
  for (int t = 1; t <= A.steps; ++t) {
    float acc = 0.f;
    if (wave == 0 && t > 1) {
      for (int spins = 0; spins < (1 << 15); ++spins) {
        const int v = lane < 8 ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
        if (__all(v >= t - 1)) break;
      }
    }
    lds_barrier();
    if (t > 1 && tid < 512) {
      const int o = (((t - 1) & 1) * 8 * slab + er * 256 + ct * 32 + eu) * 4;
      float ps[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) ps[p] = ld4s(rr, o + p * slab * 4);
#pragma unroll
      for (int p = 0; p < 8; ++p) { bad += ps[p] != val(t - 1, cell, er, p * 32 + ct * 32 + eu); acc += ps[p]; }
    }
    acc = busy(acc, A.work);
    red[tid] = acc;
    lds_barrier();
    sink += red[tid ^ 1];
    float* const dst = ring + (long)((t & 1) * 8 + ct) * slab;
    {
      const int rrow = lane >> 2, n = wave * 16 + (lane & 3) * 4;          // wave w owns units [16 w, 16 w + 16): 16 rows x 4 lanes x 16 B
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = val(t, cell, rrow, ct * 32 + n + e);
      *reinterpret_cast<f32x4*>(dst + rrow * 256 + n) = v;
    }
    if (tid < 512) *reinterpret_cast<f32x4*>(rec + ((long)(t - 1) * 512 + tid) * 4) = f32x4{acc, acc, acc, acc};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + ct, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  const long t1 = wall_clock64();
  if (tid == 0) { A.stat[(gx * 32 + slot) * 4] = (int)(t1 - t0); A.stat[(gx * 32 + slot) * 4 + 3] = (int)sink; }
  atomicAdd(A.stat + (gx * 32 + slot) * 4 + 1, (int)bad);
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int steps = argc > 1 ? atoi(argv[1]) : 1000;
  Args A{};
  const size_t ring_bytes = 64ul << 20, rec_bytes = 3ul << 30;
  hipMalloc(&A.ring, ring_bytes); hipMalloc(&A.rec, rec_bytes); hipMalloc(&A.flags, 1 << 16); hipMalloc(&A.claim, 64); hipMalloc(&A.stat, 1 << 16);
  A.steps = steps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pattern = 0; pattern < 3; ++pattern)
    for (int work : {0, 100, 300})
      for (int proto = 0; proto < 2; ++proto) {
        if (pattern == 2 && proto == 1) continue;
        A.proto = proto; A.work = work;
        float best = 1e9f; int bad = 0; long retries = 0; double tick_us = 0;
        for (int rep = 0; rep < 3; ++rep) {
          hipMemset(A.ring, 0, ring_bytes); hipMemset(A.flags, 0, 1 << 16); hipMemset(A.claim, 0, 64); hipMemset(A.stat, 0, 1 << 16);
          hipDeviceSynchronize();
          hipEventRecord(e0);
          if (pattern == 0) fwd_pattern<<<8 * 96, 256>>>(A); else if (pattern == 1) bwd_pattern<<<8 * 64, 512>>>(A); else bwd_pattern_1024<<<8 * 32, 1024>>>(A);
          hipEventRecord(e1);
          if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
          float ms; hipEventElapsedTime(&ms, e0, e1);
          const int nw = pattern == 0 ? 8 * 96 : pattern == 1 ? 8 * 64 : 8 * 32;
          std::vector<int> s(nw * 4); hipMemcpy(s.data(), A.stat, nw * 16, hipMemcpyDeviceToHost);
          int mx = 0; bad = 0; retries = 0;
          for (int w = 0; w < nw; ++w) { mx = s[w * 4] > mx ? s[w * 4] : mx; bad += s[w * 4 + 1]; retries += s[w * 4 + 2]; }
          tick_us = mx / 100.0 / steps;
          best = ms < best ? ms : best;
        }
        printf("pattern %s  work %3d  protocol %s : %.3f us per step (launch %.3f ms / %d steps; slowest workgroup %.3f us), wrong values %d, retried polls per workgroup-step %.2f\n",
               pattern == 0 ? "F (96 wg x 8 KB reads)" : pattern == 1 ? "B (64 wg x 16 KB slabs)" : "B2 (32 wg of 1024 threads x 16 KB slabs)", work, proto == 0 ? "0 drain+flag" : "1 tagged     ", best * 1e3 / steps,
               best, steps, tick_us, bad, (double)retries / ((pattern == 0 ? 8 * 96 : pattern == 1 ? 8 * 64 : 8 * 32) * (double)steps));
      }
  return 0;
}
