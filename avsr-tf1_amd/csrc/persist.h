// Device helpers shared by the persistent RNN kernels (rnn_persist.hip forward, rnn_persist_bwd.hip backward).
#pragma once
#include "common.h"

#define P_HDR 256          // sync words reserved ahead of the counters: [0] sticky error flag, [16..] debug timing

namespace avsr {

extern int32_t* g_sync;
extern int64_t g_sync_ints;
extern int g_persist_mode;

__device__ __forceinline__ f32x4 ld4_sc1(const float* p) {
  // 16-byte load that bypasses this CU's L1 (the line may have been rewritten by another CU since we last read it)
  typedef unsigned long long u64;
  const u64 a = __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const u64 b = __hip_atomic_load(reinterpret_cast<const u64*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  f32x4 v;
  v[0] = __builtin_bit_cast(float, (unsigned)(a & 0xffffffffu)); v[1] = __builtin_bit_cast(float, (unsigned)(a >> 32));
  v[2] = __builtin_bit_cast(float, (unsigned)(b & 0xffffffffu)); v[3] = __builtin_bit_cast(float, (unsigned)(b >> 32));
  return v;
}
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// bounded wait for *ctr >= target; false (and *err = 1) on timeout or if another workgroup already failed
__device__ __forceinline__ bool wait_ge(const int* ctr, int target, int* err) {
  for (int spins = 0; spins < (1 << 21); ++spins) {
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    if ((spins & 1023) == 1023 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    __builtin_amdgcn_s_sleep(2);
  }
  __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

__device__ __forceinline__ float p_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float p_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }
// DropoutWrapper mask scale.  The seed VALUE is read once per kernel: a load inside the step loop would put an
// s_waitcnt vmcnt(0) -- and with it every operand prefetched for the next step -- on the critical path.
__device__ __forceinline__ float p_drop(bool on, uint32_t seedv, uint32_t stream, uint32_t idx, float keep) {
  if (!on || keep >= 1.0f) return 1.0f;
  return uniform01(seedv, stream, idx) < keep ? 1.0f / keep : 0.0f;
}

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }   // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ f32x4 ldx_sc1(__amdgpu_buffer_rsrc_t r, int elem_off) {
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(r, elem_off * 4, 0, 16);                      // aux 16 = sc1
  return __builtin_bit_cast(f32x4, v);
}


// Workgroup barrier that does NOT drain the vector-memory queue.  __syncthreads() carries a workgroup-scope fence, which
// the compiler lowers to s_waitcnt vmcnt(0) before s_barrier: every operand prefetched for the NEXT step would be waited
// for on the critical path of THIS step.  The persistent kernels only need LDS ordering (and plain arrival) at their
// in-step barriers; the publish barrier drains stores explicitly.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16-byte write-through store (one fabric write; four scalar sc1 stores cost ~6x per byte)
__device__ __forceinline__ void stx_sc1(__amdgpu_buffer_rsrc_t r, int elem_off, f32x4 v) {
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r, elem_off * 4, 0, 16);
}
// Raw buffer resource of 2 GiB: byte offsets < 2^31 are in range, P_OOB (2^31) is out of range, so an operand a lane
// must not fetch is "loaded" with offset P_OOB: the hardware returns 0 without touching memory and the load stays
// unconditional.  (Loads inside divergent branches make the compiler give up counting vmcnt and wait for everything.)
#define P_OOB ((int)0x80000000)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x80000000u, 0x00020000);
}
// byte-offset forms
__device__ __forceinline__ f32x4 ldb_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(f32x4, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ f32x4 ldb4(__amdgpu_buffer_rsrc_t r, int byte_off) {
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(f32x4, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
// A 16-byte buffer load the compiler's wait-count pass does not see (same 2 GiB resource as make_rsrc, as four SGPR words).  The
// pass flushes vmcnt(0) in the preheader of any loop that contains stores whenever loads issued before the loop are still in flight
// (SIInsertWaitcnts "flush in preheader"), which serialises a register prefetch with the very loop it is meant to overlap; a hidden
// load leaves the wait to the caller (vm_wait_all() before the first use of the destination).  The compiler's own waits stay safe:
// vmcnt counts in order, so extra older operations can only make one of its waits longer, never shorter.
// Limits (round 4, measured the hard way on a ring of prefetched operands in the weight-gradient kernel): the register allocator believes a
// hidden load's destination holds its value from the asm statement on.  If it ever COPIES that value before the caller's wait -- a phi
// copy at a loop's back edge, a move between register classes under pressure -- it copies the stale content, and when it then reuses the
// old register (e.g. for an address) the landing load overwrites it: wrong operands, or a memory fault.  Use hidden loads only where the
// destination is consumed by straight-line code behind one vm_wait_all() + vm_landed() (the convolution's frame prefetch: issued at the
// bottom of a pass, committed at the top of the next, the same registers in every pass), never for values rotated through a ring across
// iterations; tests/test_gpu_conv.py and the full-size property check in tests/test_gpu_bench.py run the users of this after every build.
typedef int i32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_ make_rsrc_words(const void* p) {
  const unsigned long a = reinterpret_cast<unsigned long>(p);
  i32x4_ r = {__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu)), (int)0x80000000, 0x00020000};
  return r;
}
__device__ __forceinline__ void ldb4_hidden(f32x4& d, i32x4_ rs, int byte_off) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(d) : "v"(byte_off), "s"(rs));
}
__device__ __forceinline__ void vm_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void vm_landed(f32x4& d) { asm volatile("" : "+v"(d)); }   // orders the uses of d behind vm_wait_all()
__device__ __forceinline__ float ldb1(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

}  // namespace avsr
