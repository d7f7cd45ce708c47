// Shared by the fused persistent decode kernels: dec_persist.hip (forward) and dec_persist_bwd.hip (BPTT).
#pragma once
#include "step.h"
#include "attn.h"
#include "avsr_hip.h"
#include "prof.h"
#include "persist.h"

#define DP_NW 32          // workgroups per row group (= CUs of one XCD)
#define DP_R 8            // rows per group
#define DP_WPR 4          // workgroups per row in the attention phase
#define DP_NT 512         // threads per workgroup: 8 waves, two per SIMD (<= 256 registers per lane each)
#define DP_WV 8
#define DP_CPW 7          // 16-wide K chunks per wave, cell product   (E + A + H <= 896)
#define DP_APW 4          // 16-wide K chunks per wave, attention layer (H + D <= 512)
#define DP_MISC 4032      // floats of LDS ahead of the resident value rows (8-row groups)
#define DP_MISC16 6272    // same for 16-row groups (the reduction buffer doubles)
#define DP_LDS_BYTES 163840

namespace avsr {

struct DPLaunch;

struct DPMech {
  const float* keys; const float* values; long values_sb, values_st; const int* len; const float* g;
  const float* watt_t;
  float* scores; float* ctx; float* pstat;      // records [B][L][T], [B][L][D], [L][2][nc_rec][B]
  float* ppm; float* ppl; float* ppctx;         // quarter partials [4][B], [4][B], [4][B][D]
  const float* v; const float* bq; const float* wq_t; float* pq;   // Bahdanau: v [H], bias [H] (normed only), query layer^T [H][H], processed-query record [B][L][H]
  int T, D, type, nc_rec, ch, lds_off;
};

struct DPLaunch {
  int B, L, H, E, V, n_mech, mode, oa;
  int l_begin, l_end, b0, ngroups;
  int go_id, eos_id, A, KW;
  int UW, AW, NWA, drop;
  int uwsh, awsh;
  int bah;                                      // the (one) mechanism is Bahdanau / normed Bahdanau: processed-query phase, tanh scores
  int R;                                        // rows per group (8, or 16 for the attentive layer when its memories fit)
  int* err; int* claim; int* flags;
  const float* wt; const float* bias;
  float* gates; float* cs; float* cell_out; float* att; float* attd; float* hs_seq; float* state;
  int* steplen;
  const float* embedding; const float* wout_t; const float* bout;
  float* logits; int* ids; int* tok; int* n_unfinished;
  float* xs; const int* labels; int* fed;
  const int* seed; float k_in, k_st, k_out, prob; uint32_t cid4;
  float* plog;
  DPMech m[2];
};

__device__ __forceinline__ float ld1_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 16));
}
__device__ __forceinline__ void stb4(__amdgpu_buffer_rsrc_t r, int byte_off, f32x4 v) {
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r, byte_off, 0, 0);
}
// DPP reductions: __shfl_xor lowers to ds_bpermute (an LDS round trip per step); inside a row of 16 lanes the data-parallel
// primitives rotate for one VALU issue.  row16_*: every lane of the row gets the row's result.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0x128>(v);      // row_ror:8
  v += dpp_f<0x124>(v);      // row_ror:4
  v += dpp_f<0x122>(v);      // row_ror:2
  v += dpp_f<0x121>(v);      // row_ror:1
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<0x128>(v));
  v = fmaxf(v, dpp_f<0x124>(v));
  v = fmaxf(v, dpp_f<0x122>(v));
  v = fmaxf(v, dpp_f<0x121>(v));
  return v;
}
__device__ __forceinline__ float rdlane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
// whole-wave results (uniform): the four row results combined in a fixed order
__device__ __forceinline__ float wave64_sum(float v) {
  v = row16_sum(v);
  return (rdlane_f(v, 0) + rdlane_f(v, 16)) + (rdlane_f(v, 32) + rdlane_f(v, 48));
}
__device__ __forceinline__ float wave64_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(rdlane_f(v, 0), rdlane_f(v, 16)), fmaxf(rdlane_f(v, 32), rdlane_f(v, 48)));
}
__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }


// Fills L for the descriptor; AVSR_ERR_UNSUPPORTED when the fused kernels do not cover it (dec_persist.hip).
int dp_plan(const avsr_attn_rnn& d, DPLaunch& L, int* variant, size_t* lds_bytes);
extern int g_dec_fused;

}  // namespace avsr
