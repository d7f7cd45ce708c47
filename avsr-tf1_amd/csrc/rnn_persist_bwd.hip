// Persistent BPTT of the masked multi-layer LSTM stacks (fast path of avsr_rnn_bwd; same results as the
// one-launch-per-step wavefront in rnn.hip, which stays as the fallback).
//
// Layout.  Batch rows are independent, so the batch is cut into groups of 16 rows (one full MFMA row tile) and
// every group owns a PAIR of XCDs (group r -> XCDs 2r, 2r+1).  The (stack, layer) cells of a group are divided
// between the two XCDs of its pair so that each cell's own recurrence -- the step-to-step critical path -- stays
// inside one XCD's L2, and only layer-to-layer edges may cross (the layer above runs ahead of the layer below in
// BPTT, so a crossing edge is off the critical path).  A workgroup = 512 threads = 16 rows x 16 units (32 for a
// top-of-stack cell) with its slice of Wh^T (and of the upper layer's Wx^T) resident in registers for the whole
// sequence; its dc / dh carries live in registers too.
//
// Hand-off forms (MI355X_MICROARCH.md "inter-workgroup visibility"; tools/xcd_probe.hip):
//   same XCD : plain stores (line stays in that L2) -> s_waitcnt vmcnt(0) -> barrier -> progress word;
//              reader: one coalesced L1-bypassing poll of the progress words -> 16-byte sc1 loads (L2-served).
//   crossing : 16-byte sc1 stores -> s_waitcnt vmcnt(0) -> barrier -> agent-scope arrival counter;
//              reader: agent-scope poll of the counter -> 16-byte sc1 loads.
// Placement is never assumed: a workgroup reads HW_REG_XCC_ID and claims a slot of the XCD it actually runs on.
// Only liveness depends on every XCD receiving its share of the grid; every wait is bounded and a miss raises
// the sticky error word (avsr_rnn_set_persistent documents it).
#include "step.h"
#include "avsr_hip.h"
#include "prof.h"
#include "persist.h"

#define B_MAX_TASKS 12
namespace avsr {
float* g_persist_scratch = nullptr;      // float scratch of the K-split kernel: partial d h slabs, the helpers' dx records (caller-owned)
int64_t g_persist_scratch_floats = 0;
}
extern "C" int avsr_rnn_set_persistent_scratch(float* scratch, int64_t floats) {
  avsr::g_persist_scratch = scratch; avsr::g_persist_scratch_floats = scratch ? floats : 0;
  return AVSR_OK;
}
#define B_CH 8            // 16-wide K chunks per wave per operand part (4H / 16 / 8 waves, H <= 256)

namespace avsr {

struct BTask {
  const float* w_own; const float* w_up; long ldw_own, ldw_up;
  const int* len;
  const float* gates; const float* cs;
  float* dgates; float* ring; const float* up_dgates;
  float* part;                              // K-split kernel: partial d h slabs [2 parities][nct producers][B][H]
  float* dx; int kind;                      // K-split kernel: kind 0 CELL (dx = input from its helper, or null), 1 HELP (dx = output [B,T,H])
  const float* dout; long dout_sb, dout_st;
  const float* dh_final; const float* dc_final;
  int* prog; const int* prog_up;
  int* ctr; const int* ctr_up;
  int B, T, H, H_up, reverse, nct, nct_up, slot_begin, half, ntile, up_remote;
  const int32_t* seed; float k_st, k_out, k_in; uint32_t r_st, r_out, r_in; int in_W, in_coff;
};
// b0: first batch row of this launch.  solo (K-split kernel, batches above 64 utterances whose tasks fit ONE XCD's 64 slots): a
// 16-row group owns one XCD instead of a pair -- eight groups per launch, so 128 utterances are one pass over the chip instead of
// two sequential 64-row slices (every edge is XCD-local then).
struct BLaunch { int ntask, ngroups, wpx0, wpx1, b0, solo; int* err; int* claim; BTask task[B_MAX_TASKS]; };

__global__ __launch_bounds__(512) void rnn_persist_bwd_kernel(const BLaunch L) {
  __shared__ __attribute__((aligned(16))) float red[8][2][16][16];
  __shared__ int s_slot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xcc = __builtin_amdgcn_readfirstlane(xcc_id());
  if (tid == 0) s_slot = __hip_atomic_fetch_add(L.claim + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int slot = __builtin_amdgcn_readfirstlane(s_slot);
  const int g = xcc >> 1, half = xcc & 1;
  if (g >= L.ngroups || slot >= (half ? L.wpx1 : L.wpx0)) return;
  int ti = -1;
#pragma unroll
  for (int i = 0; i < B_MAX_TASKS; ++i)
    if (i < L.ntask && L.task[i].half == half && slot >= L.task[i].slot_begin) ti = i;    // same-half tasks: ascending slot_begin
  ti = __builtin_amdgcn_readfirstlane(ti);
  const BTask& tk = L.task[ti];
  const int ct = slot - tk.slot_begin;
  const int H = tk.H, T = tk.T, reverse = tk.reverse;
  const int NT = tk.ntile;
  const bool top = tk.w_up == nullptr;           // no upper layer inside the stack
  const int unit0 = ct * 16 * NT, row0 = L.b0 + g * 16;
  const int i = lane & 15, q = lane >> 4;
  const int K_own = 4 * H, K_up = 4 * tk.H_up;
  // ---- K split.  Every wave runs two runs of <= 8 chunks (16 K each): operand part A with weights wA into acc0,
  // part B with wB into acc1.
  //   cell with a layer above : A = own d gates (ring), chunks split over the 8 waves; B = upper layer's d gates
  //                             (prefetched a step ahead), chunks split over the 8 waves; both for column tile 0
  //   top cell, 2 column tiles: wave w serves tile w>>2 with the K quarter w&3: A = first half of the quarter, B = second
  //   top cell, 1 column tile : A = own d gates split over the 8 waves; B unused
  const bool two = top && NT == 2;
  const int nco = K_own >> 4, ncu = top ? 0 : K_up >> 4;
  int a0, nA, b0, nB;                        // first chunk / chunk count of parts A and B (chunk = 16 K)
  if (two) {
    const int kq = wave & 3, q0 = (kq * nco) / 4, q1 = ((kq + 1) * nco) / 4, qm = q0 + (q1 - q0 + 1) / 2;
    a0 = q0; nA = qm - q0; b0 = qm; nB = q1 - qm;
  } else {
    a0 = (wave * nco) / 8; nA = ((wave + 1) * nco) / 8 - a0;
    b0 = (wave * ncu) / 8; nB = ((wave + 1) * ncu) / 8 - b0;
  }
  const int wtile = two ? wave >> 2 : 0;     // column tile this wave multiplies

  // ---- weight slices: registers for the whole sequence ----
  f32x4 wA[B_CH], wB[B_CH];
  {
    const int uu = unit0 + wtile * 16 + i;
#pragma unroll
    for (int c = 0; c < B_CH; ++c) {
      wA[c] = (c < nA && uu < H) ? ld4(tk.w_own + (long)uu * tk.ldw_own + (a0 + c) * 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < nB && uu < H) v = top ? ld4(tk.w_own + (long)uu * tk.ldw_own + (b0 + c) * 16 + 4 * q)
                                    : ld4(tk.w_up + (long)uu * tk.ldw_up + (b0 + c) * 16 + 4 * q);
      wB[c] = v;
    }
  }

  // ---- epilogue ownership: one thread per (row, unit) for all steps ----
  const int tile = tid >> 8, er = (tid & 255) >> 4, eu = tid & 15;
  const int b = row0 + er, u = unit0 + tile * 16 + eu;
  const bool eok = tile < NT && b < tk.B && u < H;
  const int len_b = eok ? (tk.len ? tk.len[b] : T) : 0;
  float dc_carry = (eok && tk.dc_final) ? tk.dc_final[(long)b * H + u] : 0.f;
  float dh_carry = (eok && tk.dh_final) ? tk.dh_final[(long)b * H + u] : 0.f;
  const int rec_b = b * T * H + u;
  const __amdgpu_buffer_rsrc_t gates_rs = make_rsrc(tk.gates), cs_rs = make_rsrc(tk.cs), dout_rs = make_rsrc(tk.dout);
  const bool has_dout = tk.dout != nullptr;
  float* const dgates_p = tk.dgates;
  const __amdgpu_buffer_rsrc_t dgates_rs = make_rsrc(tk.dgates);
  float* const ring_p = tk.ring;
  const int dout_b = (int)(b * tk.dout_sb) + u, dout_st = (int)tk.dout_st;
  const bool drop_on = tk.seed != nullptr;
  const uint32_t seedv = drop_on ? (uint32_t)tk.seed[0] : 0u;
  const float k_st = tk.k_st, k_out = tk.k_out, k_in = tk.k_in;
  const uint32_t r_st = tk.r_st, r_out = tk.r_out, r_in = tk.r_in;
  const int in_W = tk.in_W, in_coff = tk.in_coff;
  const bool publish_remote = tk.ctr != nullptr;

  // ---- A-operand row of this lane (byte offsets; P_OOB = "reads as zero") ----
  const int ab = row0 + i;
  const bool aok = ab < tk.B;
  const int len_a = aok ? (tk.len ? tk.len[ab] : T) : 0;
  const __amdgpu_buffer_rsrc_t ring_rs = make_rsrc(tk.ring);
  const __amdgpu_buffer_rsrc_t up_rs = make_rsrc(tk.up_dgates);
  const int ring_par = tk.B * K_own * 4;                                   // bytes between the two ring slots
  const int ringA = aok ? (ab * K_own + a0 * 16 + 4 * q) * 4 : P_OOB;
  const int ringB = aok ? (ab * K_own + b0 * 16 + 4 * q) * 4 : P_OOB;     // top cell with two tiles only
  const int up_row = (ab * T * K_up + b0 * 16 + 4 * q) * 4;

  // ---- progress polling (wave 0): lanes 0-31 own progress words, lanes 32-63 the upper layer ----
  int* const my_prog = tk.prog + g * 32 + ct;
  int* const my_ctr = publish_remote ? tk.ctr + (long)g * T : nullptr;
  const int up_remote = tk.up_remote, nct_up = tk.nct_up;
  const int* poll_own = nullptr; const int* poll_up = nullptr;
  if (wave == 0) {
    if (lane < 32) { if (lane < tk.nct) poll_own = tk.prog + g * 32 + lane; }
    else if (!top) {
      if (up_remote) { if (lane == 32) poll_up = tk.ctr_up + (long)g * T; }
      else if (lane - 32 < nct_up) poll_up = tk.prog_up + g * 32 + (lane - 32);
    }
  }
  // own: steps completed >= need_own.  upper: step index t_up completed (local: progress >= T - t_up; crossing: counter[t_up] >= nct_up)
  auto wait_progress = [&](int need_own, int t_up) {
    if (wave != 0) return;
    const int* p = lane < 32 ? poll_own : (poll_up ? (up_remote ? poll_up + t_up : poll_up) : nullptr);
    const int need = lane < 32 ? need_own : (up_remote ? nct_up : T - t_up);
    for (int spins = 0; spins < (1 << 21); ++spins) {
      const int v = p ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
      if (__all(v >= need)) return;
#ifdef POLL_SLEEP
      __builtin_amdgcn_s_sleep(POLL_SLEEP);
#endif
      if ((spins & 1023) == 1023 && __hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
    if (lane == 0) __hip_atomic_store(L.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // part-B operands of a cell with a layer above (d gates of that layer at the same step) run one step ahead
  f32x4 aB[B_CH];
#pragma unroll
  for (int c = 0; c < B_CH; ++c) aB[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_up = [&](int t) {
    const int o = (aok && t >= 0 && t < len_a) ? up_row + (reverse ? len_a - 1 - t : t) * (K_up * 4) : P_OOB;
#pragma unroll
    for (int c = 0; c < B_CH; ++c) aB[c] = ldb_sc1(up_rs, c < nB ? o + c * 64 : P_OOB);
  };
  // epilogue operands of the next step (records written by the forward pass: cold in HBM) are fetched a step ahead
  f32x4 n_g = {0.f, 0.f, 0.f, 0.f};
  float n_c = 0.f, n_cp = 0.f, n_do = 0.f;
  auto load_rec = [&](int t) {
    const bool v = eok && t >= 0 && t < len_b;
    const int tau = reverse ? len_b - 1 - t : t;
    const int o = v ? (rec_b + tau * H) * 4 : P_OOB;
    n_g = ldb4(gates_rs, v ? o * 4 : P_OOB);
    n_c = ldb1(cs_rs, o);
    n_cp = ldb1(cs_rs, (v && t > 0) ? o + (reverse ? H : -H) * 4 : P_OOB);
    n_do = ldb1(dout_rs, (v && has_dout) ? (dout_b + tau * dout_st) * 4 : P_OOB);
  };
  if (!top) {
    wait_progress(0, T - 1);
    __syncthreads();
    load_up(T - 1);
  }
  load_rec(T - 1);
#ifdef PERSIST_TIMING
  long tm[6] = {0, 0, 0, 0, 0, 0};
#define BTICK(k) { const long now_ = __builtin_amdgcn_s_memtime(); tm[k] += now_ - last_; last_ = now_; }
#else
#define BTICK(k)
#endif

  for (int t = T - 1; t >= 0; --t) {
#ifdef PERSIST_TIMING
    long last_ = __builtin_amdgcn_s_memtime();
#endif
    // hand the records prefetched during the previous step over HERE (pinned by the empty asm): left to the compiler the
    // copy sinks to just before the next prefetch, behind loads issued in between, and its wait serialises them
    f32x4 g4 = n_g;
    float c = n_c, cprev = n_cp, dout_ext = n_do;
    asm volatile("" : "+v"(g4), "+v"(c), "+v"(cprev), "+v"(dout_ext));
    // dependencies: step t+1 of this cell (every column tile of my rows); the cell above one step ahead (t-1)
    wait_progress(T - 1 - t, t > 0 ? t - 1 : 0);
    lds_barrier();
    BTICK(0)
    // ---- recurrent operand: d gates of step t+1 from the two-slot ring (zero-filled by the host for t = T-1) ----
    f32x4 aA[B_CH];
    {
      const int par = ((t + 1) & 1) * ring_par;
      const int oA = aok ? ringA + par : P_OOB, oB = aok ? ringB + par : P_OOB;
#pragma unroll
      for (int cc = 0; cc < B_CH; ++cc) aA[cc] = ldb_sc1(ring_rs, cc < nA ? oA + cc * 64 : P_OOB);
      if (two) {
#pragma unroll
        for (int cc = 0; cc < B_CH; ++cc) aB[cc] = ldb_sc1(ring_rs, cc < nB ? oB + cc * 64 : P_OOB);
      }
    }
#ifdef PERSIST_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // timing build only: isolate the recurrent-operand latency
    BTICK(1)
#endif
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (!two) {
      // part B (upper layer, or nothing): operands arrived a step ago, so these MFMAs run under the loads just issued
#pragma unroll
      for (int cc = 0; cc < B_CH; ++cc)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aB[cc][e], wB[cc][e], acc1, 0, 0, 0);
      load_up(t - 1);                          // refill in place: consumed a step from now
    }
    load_rec(t - 1);                           // issued last: vmcnt retires in order (see rnn_persist.hip)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int cc = 0; cc < B_CH; ++cc)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aA[cc][e], wA[cc][e], acc0, 0, 0, 0);
    if (two) {
#pragma unroll
      for (int cc = 0; cc < B_CH; ++cc)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aB[cc][e], wB[cc][e], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { red[wave][0][q * 4 + r][i] = acc0[r]; red[wave][1][q * 4 + r][i] = acc1[r]; }
    lds_barrier();
    BTICK(2)

    // ---- LSTM cell backward (same arithmetic as EP_LSTM_BWD in step.hip) ----
    if (eok) {
      f32x4 dg = {0.f, 0.f, 0.f, 0.f};
      const bool valid = t < len_b;
      int tau = t;
      if (valid) {
        tau = reverse ? len_b - 1 - t : t;
        float zA = 0.f, zB = 0.f;
        if (two) {                               // the four waves of my column tile, both halves of their K quarter
#pragma unroll
          for (int w = 0; w < 4; ++w) zA += red[tile * 4 + w][0][er][eu] + red[tile * 4 + w][1][er][eu];
        } else {
#pragma unroll
          for (int w = 0; w < 8; ++w) { zA += red[w][0][er][eu]; zB += red[w][1][er][eu]; }
        }
        const long bt = (long)b * T + tau;
        const uint32_t oidx = (uint32_t)(bt * H + u);
        const uint32_t iidx = (uint32_t)(bt * in_W + in_coff + u);
        const float dout = dout_ext + zB * p_drop(drop_on, seedv, r_in, iidx, k_in);
        const float dh = dout * p_drop(drop_on, seedv, r_out, oidx, k_out) + (zA + dh_carry) * p_drop(drop_on, seedv, r_st, oidx, k_st);
        const float tc = p_tanh(c);
        float dc = dh * g4[3] * (1.f - tc * tc) + dc_carry;
        if (!(fabsf(c) < 1.0f)) dc = 0.f;      // cell_clip = 1.0: no gradient through a clipped cell
        dg[3] = dh * tc * g4[3] * (1.f - g4[3]);
        dg[0] = dc * g4[1] * g4[0] * (1.f - g4[0]);
        dg[1] = dc * g4[0] * (1.f - g4[1] * g4[1]);
        dg[2] = dc * cprev * g4[2] * (1.f - g4[2]);
        dc_carry = dc * g4[2];
        dh_carry = 0.f;
      }
      // ring first (the recurrence reads it), then the record (zero at padding position t past the utterance)
      st4(ring_p + (long)(t & 1) * (ring_par >> 2) + (long)b * K_own + u * 4, dg);
      const int ro = (rec_b + tau * H) * 4;
      if (publish_remote) stx_sc1(dgates_rs, ro, dg);
      else st4(dgates_p + ro, dg);
    }
    BTICK(3)
    // ---- publish ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    BTICK(4)
    if (tid == 0) {
      __hip_atomic_store(my_prog, T - t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (my_ctr) __hip_atomic_fetch_add(my_ctr + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    BTICK(5)
  }
#ifdef PERSIST_TIMING
  if (tid == 0 && ct == 0 && g == 0)
    for (int k = 0; k < 6; ++k) L.err[80 + ti * 8 + k] = (int)(tm[k] / T);
#endif
}

// ======================================================================================================
// K-split variant (round 2; first choice of avsr_rnn_bwd_persistent, the kernel above takes what this one declines).
// The kernel above partitions the recurrent product dh = dgates(t+1) . Wh^T by output UNITS, so every workgroup of a layer pulls the
// whole [16 x 4H] dgates tile (64 KB at H = 256) through its CU's L2 port on the step-to-step chain and only then multiplies, and it
// carries the product for the layer below (another 64 KB operand, 32 more MFMAs per wave) in the same workgroup: taking parts away
//  showed that second product costing 2.3 us of the 5.4 us step although it is on no recurrence.  Here
//   CELL  the recurrent product is partitioned by K: a workgroup multiplies the [16 x 64] dgates slice it has JUST produced (its own 16
//         units x 4 gates, through LDS into the MFMA A layout) with its 64 rows of Wh^T into a partial dh for ALL units and publishes that
//         [16 x H] slab (whole 128-byte rows, 16 bytes per lane: narrow stores retire one by one ahead of the drain); a consumer sums the nct
//         partial values of its own (row, unit) - 16 scalar loads per thread instead of 64 KB per workgroup.  Chain per step: hand-off ->
//         16 partial loads -> cell backward -> 32 MFMAs per wave -> publish.
//   HELP  the product for the layer below is its own task (as in rnn_persist_bwd2.hip): dx(t) = input mask . dgates_above(t) . Wx^T by units,
//         K over the 8 waves, written to a [B, T, H] record the cell below reads one step ahead with its forward records.  No recurrence:
//         it follows the cell above as that cell publishes.
// Both are 512-thread workgroups held to 128 VGPRs (two per CU, 64 per XCD); every cell is 16 units per workgroup (no 32-unit top cells:
// their 64 MFMAs per wave would set the pace).  Crossing edges of the XCD pair are placed INTO helpers where possible (a cell polling
// an agent-scope counter on its chain cost 2.46 vs 2.22 ms).  c4: 2.70 -> 2.2 ms per launch.
__device__ __forceinline__ float ldb1_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 16));
}

// NW: waves per workgroup.  8 = 512 threads owning 16 units (two workgroups per CU).  16 (round 5) = 1024 threads owning 32 units, one
// workgroup per CU: a cell then has HALF as many producers -- half the partial-slab bytes through the XCD's L2 per step and half the
// partial loads per consumer -- with the same matrix work and registers per wave (one 16-column tile of the partial d h with K = 128
// instead of two with K = 64).  tools/handoff_probe.hip: the bare exchange of this kernel's pattern costs 1.73 us per step with 16
// producers of 512 threads and 1.22 us with 8 of 1024.
template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void rnn_persist_bwdk_kernel(const BLaunch L) {
  constexpr int UW = 2 * NW, UWSH = NW == 8 ? 4 : 5;       // units per workgroup
  constexpr int TPW = 16 / NW;                              // 16-column tiles of the partial d h per wave (H <= 256)
  constexpr int KC = UW / 4;                                // 16-deep chunks of the workgroup's own gate columns
  __shared__ __attribute__((aligned(16))) float red[NW][16][16];
  __shared__ __attribute__((aligned(16))) float dgs[16][UW * 4 + 4];
  __shared__ __attribute__((aligned(16))) float stg[NW][16][TPW * 16 + 4];
  __shared__ int s_slot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xcc = __builtin_amdgcn_readfirstlane(xcc_id());
  if (tid == 0) s_slot = __hip_atomic_fetch_add(L.claim + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int slot = __builtin_amdgcn_readfirstlane(s_slot);
  const int g = L.solo ? xcc : xcc >> 1, half = L.solo ? 0 : xcc & 1;
  if (g >= L.ngroups || slot >= (half ? L.wpx1 : L.wpx0)) return;
  int ti = -1, sb_best = -1;                       // the task of this half whose slot range holds my slot (ranges: cells first, then helpers)
#pragma unroll
  for (int i = 0; i < B_MAX_TASKS; ++i)
    if (i < L.ntask && L.task[i].half == half && slot >= L.task[i].slot_begin && L.task[i].slot_begin > sb_best) { ti = i; sb_best = L.task[i].slot_begin; }
  ti = __builtin_amdgcn_readfirstlane(ti);
  const BTask& tk = L.task[ti];
  const int ct = slot - tk.slot_begin;
  const int H = tk.H, T = tk.T, reverse = tk.reverse, nct = tk.nct;
  const bool helper = tk.kind == 1;
  const int unit0 = ct * UW, row0 = L.b0 + g * 16;
  const int i = lane & 15, q = lane >> 4;

  // ---- ownership of (row, unit) by threads 0 .. 16 * UW - 1 for all steps ----
  const int er = (tid & (16 * UW - 1)) >> UWSH, eu = tid & (UW - 1);
  const int b = row0 + er, u = unit0 + eu;
  const bool eok = tid < 16 * UW && b < tk.B && u < H;
  const int len_b = eok ? (tk.len ? tk.len[b] : T) : 0;
  const int rec_b = b * T * H + u;
  const bool drop_on = tk.seed != nullptr;
  const uint32_t seedv = drop_on ? (uint32_t)tk.seed[0] : 0u;
  const bool publish_remote = tk.ctr != nullptr;
  const int ab = row0 + i;
  const bool aok = ab < tk.B;
  const int len_a = aok ? (tk.len ? tk.len[ab] : T) : 0;

  // ---- progress polling (wave 0): lanes 0-31 own progress words (cells), lanes 32-63 the producer this task reads ----
  int* const my_prog = tk.prog + g * 32 + ct;
  int* const my_ctr = publish_remote ? tk.ctr + (long)g * T : nullptr;
  const bool has_up = tk.prog_up != nullptr || tk.ctr_up != nullptr;
  const int up_remote = tk.up_remote, nct_up = tk.nct_up;
  const int* poll_own = nullptr; const int* poll_up = nullptr;
  if (wave == 0) {
    if (lane < 32) { if (!helper && lane < nct) poll_own = tk.prog + g * 32 + lane; }
    else if (has_up) {
      if (up_remote) { if (lane == 32) poll_up = tk.ctr_up + (long)g * T; }
      else if (lane - 32 < nct_up) poll_up = tk.prog_up + g * 32 + (lane - 32);
    }
  }
  // own: steps completed >= need_own.  producer: step index t_up completed (local: progress >= T - t_up; crossing: counter[t_up] >= nct_up)
  auto wait_progress = [&](int need_own, int t_up) {
    if (wave != 0) return;
    const int* p = lane < 32 ? poll_own : (poll_up ? (up_remote ? poll_up + t_up : poll_up) : nullptr);
    const int need = lane < 32 ? need_own : (up_remote ? nct_up : T - t_up);
    for (int spins = 0; spins < (1 << 21); ++spins) {
      const int v = p ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
      if (__all(v >= need)) return;
      if ((spins & 1023) == 1023 && __hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
    if (lane == 0) __hip_atomic_store(L.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto publish = [&](int t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(my_prog, T - t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (my_ctr) __hip_atomic_fetch_add(my_ctr + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  if (helper) {
    // ================= HELP: dx(t) = input mask . (d gates of the layer above at step t) . Wx^T, by units, K over the 8 waves.
    // No recurrence: it follows the cell above as closely as that cell publishes, off every step-to-step chain. =================
    const int K_up = 4 * tk.H_up, ncu = K_up >> 4;
    const int kw = wave & 7, tl = wave >> 3;               // K eighth / 16-unit tile of this wave
    const int b0 = (kw * ncu) / 8, nB = ((kw + 1) * ncu) / 8 - b0;
    f32x4 wB[B_CH];
    {
      const int uu = unit0 + tl * 16 + i;
#pragma unroll
      for (int c = 0; c < B_CH; ++c)
        wB[c] = (c < nB && uu < H) ? ld4(tk.w_up + (long)uu * tk.ldw_up + (b0 + c) * 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const __amdgpu_buffer_rsrc_t up_rs = make_rsrc(tk.up_dgates);
    const int up_row = (ab * T * K_up + b0 * 16 + 4 * q) * 4;
    float* const dx_p = tk.dx;
    const float k_in = tk.k_in; const uint32_t r_in = tk.r_in; const int in_W = tk.in_W, in_coff = tk.in_coff;
    for (int t = T - 1; t >= 0; --t) {
      wait_progress(0, t);
      lds_barrier();
      f32x4 aB[B_CH];
      const int o = (aok && t < len_a) ? up_row + (reverse ? len_a - 1 - t : t) * (K_up * 4) : P_OOB;
#pragma unroll
      for (int c = 0; c < B_CH; ++c) aB[c] = ldb_sc1(up_rs, c < nB ? o + c * 64 : P_OOB);
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < B_CH; cc += 2)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aB[cc][e], wB[cc][e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aB[cc + 1][e], wB[cc + 1][e], acc1, 0, 0, 0);
        }
      acc0 += acc1;
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][q * 4 + r][i] = acc0[r];
      lds_barrier();
      if (eok && t < len_b) {
        const int tau = reverse ? len_b - 1 - t : t;
        const long bt = (long)b * T + tau;
        float z = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) z += red[(eu >> 4) * 8 + w][er][eu & 15];
        const float v = z * p_drop(drop_on, seedv, r_in, (uint32_t)(bt * in_W + in_coff + u), k_in);
        if (publish_remote) st_sc1(dx_p + bt * H + u, v);
        else dx_p[bt * H + u] = v;
      }
      publish(t);
    }
    return;
  }

  // ================= CELL =================
  // my 64 rows of Wh^T for two 16-unit column tiles per wave (product by K), resident for the whole sequence
  f32x4 wR[TPW][KC];
#pragma unroll
  for (int nt = 0; nt < TPW; ++nt) {
    const int n = (wave * TPW + nt) * 16 + i;
#pragma unroll
    for (int c = 0; c < KC; ++c)
      wR[nt][c] = (n < H && unit0 * 4 + c * 16 + 4 * q < 4 * H) ? ld4(tk.w_own + (long)n * tk.ldw_own + unit0 * 4 + c * 16 + 4 * q)
                                                              : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float dc_carry = (eok && tk.dc_final) ? tk.dc_final[(long)b * H + u] : 0.f;
  float dh_carry = (eok && tk.dh_final) ? tk.dh_final[(long)b * H + u] : 0.f;
  const __amdgpu_buffer_rsrc_t gates_rs = make_rsrc(tk.gates), cs_rs = make_rsrc(tk.cs), dout_rs = make_rsrc(tk.dout);
  const __amdgpu_buffer_rsrc_t dx_rs = make_rsrc(tk.dx);
  const bool has_dout = tk.dout != nullptr, has_dx = tk.dx != nullptr;
  float* const dgates_p = tk.dgates;
  const __amdgpu_buffer_rsrc_t dgates_rs = make_rsrc(tk.dgates);
  const int dout_b = (int)(b * tk.dout_sb) + u, dout_st = (int)tk.dout_st;
  const float k_st = tk.k_st, k_out = tk.k_out;
  const uint32_t r_st = tk.r_st, r_out = tk.r_out;

  // partial slabs [parity][producer][B][H]: consumer offsets of my (row, unit), producer rows of my accumulator tiles
  const __amdgpu_buffer_rsrc_t part_rs = make_rsrc(tk.part);
  const int slab = tk.B * H * 4;                                           // bytes per producer slab
  const int par_bytes = nct * slab;                                        // bytes per parity
  const int cons_off = (b * H + u) * 4;                                    // used by eok threads only
  float* const part_p = tk.part;

  // epilogue operands of the next step: forward records (cold in HBM) and the helper's dx, fetched a step ahead
  f32x4 n_g = {0.f, 0.f, 0.f, 0.f};
  float n_c = 0.f, n_cp = 0.f, n_do = 0.f, n_dx = 0.f;
  auto load_rec = [&](int t) {
    const bool v = eok && t >= 0 && t < len_b;
    const int tau = reverse ? len_b - 1 - t : t;
    const int o = v ? (rec_b + tau * H) * 4 : P_OOB;
    n_g = ldb4(gates_rs, v ? o * 4 : P_OOB);
    n_c = ldb1(cs_rs, o);
    n_cp = ldb1(cs_rs, (v && t > 0) ? o + (reverse ? H : -H) * 4 : P_OOB);
    n_do = ldb1(dout_rs, (v && has_dout) ? (dout_b + tau * dout_st) * 4 : P_OOB);
    n_dx = ldb1_sc1(dx_rs, (v && has_dx) ? o : P_OOB);                  // produced during this launch
  };
  if (has_up) wait_progress(0, T - 1);
  __syncthreads();
  load_rec(T - 1);
#ifdef PERSIST_TIMING
  long tmk[6] = {0, 0, 0, 0, 0, 0};                   // tools/persist_probe.py: wait | partial slabs | cell | product + stores | publish
#define KTICK(k_) { const long now_ = __builtin_amdgcn_s_memtime(); tmk[k_] += now_ - lastk_; lastk_ = now_; }
#else
#define KTICK(k_)
#endif
  for (int t = T - 1; t >= 0; --t) {
#ifdef PERSIST_TIMING
    long lastk_ = __builtin_amdgcn_s_memtime();
#endif
    f32x4 g4 = n_g;
    float c = n_c, cprev = n_cp, dout_ext = n_do + n_dx;
    asm volatile("" : "+v"(g4), "+v"(c), "+v"(cprev), "+v"(dout_ext));
    // dependencies: step t+1 of this cell (every producer of my rows); my helper one step ahead (t-1)
    wait_progress(T - 1 - t, t > 0 ? t - 1 : 0);
    lds_barrier();
    KTICK(0)
    // ---- recurrent operand: the nct partial d h values of my (row, unit) from step t+1 (nothing at t = T-1) ----
    float ps[16];
    {
      const bool pv = eok && t + 1 < T;
      const int o = cons_off + ((t + 1) & 1) * par_bytes;
#pragma unroll
      for (int p = 0; p < 16; ++p) ps[p] = ldb1_sc1(part_rs, (pv && p < nct) ? o + p * slab : P_OOB);
    }
    load_rec(t - 1);
    asm volatile("" ::: "memory");
#ifdef PERSIST_TIMING
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");    // timing build only: isolate the partial-slab latency (the 5 record loads stay in flight)
    KTICK(1)
#endif
    // ---- LSTM cell backward (same arithmetic as EP_LSTM_BWD in step.hip; dx already carries the layer above's input mask) ----
    if (tid < 16 * UW) {
      f32x4 dg = {0.f, 0.f, 0.f, 0.f};
      const bool valid = eok && t < len_b;
      int tau = t;
      if (valid) {
        tau = reverse ? len_b - 1 - t : t;
        float zA = ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
        zA += ((ps[8] + ps[9]) + (ps[10] + ps[11])) + ((ps[12] + ps[13]) + (ps[14] + ps[15]));
        const long bt = (long)b * T + tau;
        const uint32_t oidx = (uint32_t)(bt * H + u);
        const float dh = dout_ext * p_drop(drop_on, seedv, r_out, oidx, k_out) + (zA + dh_carry) * p_drop(drop_on, seedv, r_st, oidx, k_st);
        const float tc = p_tanh(c);
        float dc = dh * g4[3] * (1.f - tc * tc) + dc_carry;
        if (!(fabsf(c) < 1.0f)) dc = 0.f;
        dg[3] = dh * tc * g4[3] * (1.f - g4[3]);
        dg[0] = dc * g4[1] * g4[0] * (1.f - g4[0]);
        dg[1] = dc * g4[0] * (1.f - g4[1] * g4[1]);
        dg[2] = dc * cprev * g4[2] * (1.f - g4[2]);
        dc_carry = dc * g4[2];
        dh_carry = 0.f;
      }
      *reinterpret_cast<f32x4*>(&dgs[er][eu * 4]) = dg;               // the recurrence's operand first, then the record
      if (eok) {
        const int ro = (rec_b + tau * H) * 4;
        if (publish_remote) stx_sc1(dgates_rs, ro, dg);
        else st4(dgates_p + ro, dg);
      }
    }
    lds_barrier();
    KTICK(2)
    // ---- my K slice of the recurrent product: [16 x 64] d gates . 64 rows of Wh^T -> partial d h of ALL units, two tiles per wave ----
    {
      f32x4 A[KC];
#pragma unroll
      for (int cc = 0; cc < KC; ++cc) A[cc] = *reinterpret_cast<const f32x4*>(&dgs[i][cc * 16 + 4 * q]);
      float* const dst = part_p + ((long)((t & 1) * nct + ct) * tk.B) * H;
      {
        // the two column tiles' accumulator chains alternate (one after the other they are 2 x 16 DEPENDENT MFMAs: 40 cycles each
        // instead of 32)
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        if constexpr (TPW == 2) {
#pragma unroll
          for (int cc = 0; cc < KC; ++cc)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[cc][e], wR[0][cc][e], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[cc][e], wR[TPW - 1][cc][e], acc1, 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0x7F6);           // MFMAs keep this order (everything else may move across)
            }
#pragma unroll
          for (int r = 0; r < 4; ++r) { stg[wave][q * 4 + r][i] = acc0[r]; stg[wave][q * 4 + r][(TPW - 1) * 16 + i] = acc1[r]; }
        } else {
          // one column tile per wave, K = 128: the chunk pairs' chains alternate (same reason)
#pragma unroll
          for (int cc = 0; cc < KC; cc += 2)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[cc][e], wR[0][cc][e], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[cc + 1][e], wR[0][cc + 1][e], acc1, 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0x7F6);
            }
          acc0 += acc1;
#pragma unroll
          for (int r = 0; r < 4; ++r) stg[wave][q * 4 + r][i] = acc0[r];
        }
      }
      // the wave's [16 x 32] ([16 x 16]) tile leaves as whole 128-byte (64-byte) rows, 16 bytes per lane (narrow stores retire one by one
      // ahead of the drain)
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int idx = lane + 64 * j, rr = idx / (4 * TPW), c4 = idx % (4 * TPW);
        const f32x4 v = *reinterpret_cast<const f32x4*>(&stg[wave][rr][c4 * 4]);
        const int n = wave * (16 * TPW) + c4 * 4, rb = row0 + rr;
        if (n < H && rb < tk.B) st4(dst + (long)rb * H + n, v);
      }
    }
    KTICK(3)
    publish(t);
    KTICK(4)
  }
#ifdef PERSIST_TIMING
  if (tid == 0 && ct == 0 && g == 0)
    for (int k_ = 0; k_ < 6; ++k_) L.err[80 + ti * 8 + k_] = (int)(tmk[k_] / T);
#endif
}

// Choose the XCD half of every cell: all assignments are enumerated (<= 2^8); feasible ones keep each half within
// `cap` workgroups; the winner has the fewest crossing layer edges, then the smallest larger half.
// ecost[i]: price of a crossing edge into task i (K-split kernel: 1 when the CONSUMER is a helper - its agent-scope poll and remote
// operand loads are off every chain -, 3 when it is a cell, which would poll across the XCD pair on its step-to-step chain:
// measured 2.22 vs 2.46 ms on c4); null = 1 each.
static bool assign_halves(const int* cost, const int* upper, int n, int cap, int* half_out, const int* ecost = nullptr) {
  int best = -1, best_cross = 1 << 30, best_load = 1 << 30;
  for (int m = 0; m < (1 << n); ++m) {
    int load[2] = {0, 0}, cross = 0;
    for (int i = 0; i < n; ++i) {
      load[(m >> i) & 1] += cost[i];
      if (upper[i] >= 0 && (((m >> i) ^ (m >> upper[i])) & 1)) cross += ecost ? ecost[i] : 1;
    }
    if (load[0] > cap || load[1] > cap) continue;
    const int mx = load[0] > load[1] ? load[0] : load[1];
    if (cross < best_cross || (cross == best_cross && mx < best_load)) { best = m; best_cross = cross; best_load = mx; }
  }
  if (best < 0) return false;
  for (int i = 0; i < n; ++i) half_out[i] = (best >> i) & 1;
  return true;
}

static_assert(sizeof(BLaunch) <= 4000, "launch descriptor must fit the kernel-argument segment");

}  // namespace avsr

// Returns AVSR_ERR_UNSUPPORTED when the persistent path is disabled or the configuration does not fit it
// (avsr_rnn_bwd then uses one launch per wavefront step).  The caller has already zeroed every layer's dstate
// (ring slots) and nothing else of the launch path's setup is needed: the final-state gradients are read here.
#include <cstdio>
#include <cstdlib>
// AVSR_PERSIST_DEBUG=1 prints which precondition sent a call back to the per-step launches
#define UNSUP(code) do { if (getenv("AVSR_PERSIST_DEBUG")) fprintf(stderr, "[avsr] persistent BPTT not used: reason %d (rnn_persist_bwd.hip)\n", code); return AVSR_ERR_UNSUPPORTED; } while (0)
static int bwd_persistent(const avsr_rnn_stack* st, int32_t n, void* stream, int dry, bool ksplit, bool wide);

int avsr_rnn_bwd_persistent(const avsr_rnn_stack* st, int32_t n, void* stream, int dry) {
  // the K-split kernel first: 16-unit workgroups of 512 threads; configurations it declines take the kernel above.  AVSR_RNN_BWD_WIDE=1
  // selects the 32-unit / 1024-thread form where every layer allows and the tasks fit: built in round 5 on the strength of
  // tools/handoff_probe.hip (the bare exchange 1.73 -> 1.22 us per step), same results (232 parity tests), and measured SLOWER in the
  // real kernel -- c4 4.28 -> 4.77 us per sequential step, c2 4.52 -> 4.67, c5 4.62 -> 4.79 (profiles/r05_experiments.txt): with one
  // 16-wave workgroup per CU every in-step barrier waits for sixteen waves and nothing else runs on the CU meanwhile.  Off by default.
  static const int ksplit = getenv("AVSR_RNN_BWD_KSPLIT") ? atoi(getenv("AVSR_RNN_BWD_KSPLIT")) : 1;
  static const int wide = getenv("AVSR_RNN_BWD_WIDE") ? atoi(getenv("AVSR_RNN_BWD_WIDE")) : 0;
  if (ksplit && wide) {
    const int rc = bwd_persistent(st, n, stream, dry, true, true);
    if (rc != AVSR_ERR_UNSUPPORTED) return rc;
  }
  if (ksplit) {
    const int rc = bwd_persistent(st, n, stream, dry, true, false);
    if (rc != AVSR_ERR_UNSUPPORTED) return rc;
  }
  return bwd_persistent(st, n, stream, dry, false, false);
}

static int bwd_persistent(const avsr_rnn_stack* st, int32_t n, void* stream, int dry, bool ksplit, bool wide) {
  const int UWH = wide ? 32 : 16;                    // units per workgroup of the K-split kernel
  using namespace avsr;
  int32_t* sync = g_sync; const int64_t sync_ints = g_sync_ints;
  if (!sync || !(g_persist_mode & 2)) return AVSR_ERR_UNSUPPORTED;
  if (ksplit && !g_persist_scratch) return AVSR_ERR_UNSUPPORTED;
  long part_used = 0;
  static thread_local BLaunch L;
  L = BLaunch{};
  double flops = 0.0;
  int cost[B_MAX_TASKS], upper[B_MAX_TASKS], half[B_MAX_TASKS];
  const int B = st[0].B;
  int ngroups = B < 64 ? (B + 15) / 16 : 4;            // per launch: 4 XCD pairs x 16 rows; larger batches run as 64-row slices
  for (int i = 0; i < n; ++i) {
    const avsr_rnn_stack& S = st[i];
    if (S.cell != 0 || S.B != B) UNSUP(3);
    const int first = L.ntask;
    for (int l = 0; l < S.n_layers; ++l) {
      const avsr_rnn_layer& Ly = S.layer[l];
      if (L.ntask >= B_MAX_TASKS) UNSUP(4);
      const int H = Ly.units, in = Ly.in_dim;
      const bool top = l + 1 >= S.n_layers;
      if (H % 16 || H > 32 * B_CH || (long)S.B * S.T * H * 4 >= (1L << 29)) UNSUP(5);
      if (!top && (S.layer[l + 1].units % 16 || S.layer[l + 1].units > 32 * B_CH || S.layer[l + 1].in_dim != H)) UNSUP(6);
      if (Ly.dout && (long)S.B * (S.T + 2) * Ly.ld_dout >= (1L << 29)) UNSUP(7);
      const int t_i = L.ntask++;
      BTask& tk = L.task[t_i];
      tk.w_own = Ly.w + (long)in * 4 * H; tk.ldw_own = 4 * H;
      if (!top && !ksplit) { const avsr_rnn_layer& Up = S.layer[l + 1]; tk.w_up = Up.w; tk.ldw_up = 4 * Up.units; tk.H_up = Up.units; tk.up_dgates = Up.dgates; }
      tk.len = S.len; tk.gates = Ly.gates; tk.cs = Ly.cs; tk.dgates = Ly.dgates; tk.ring = Ly.dstate;
      if (Ly.dout) { tk.dout = Ly.dout + Ly.ld_dout + Ly.dout_col; tk.dout_sb = (long)(S.T + 2) * Ly.ld_dout; tk.dout_st = Ly.ld_dout; }
      if (top) { tk.dh_final = S.dh_final; tk.dc_final = S.dc_final; }
      tk.B = S.B; tk.T = S.T; tk.H = H; tk.reverse = S.reverse;
      flops += 2.0 * S.B * S.T * (4.0 * H + (top ? 0.0 : 4.0 * S.layer[l + 1].units)) * H;
      if (wide && H % 32) UNSUP(14);
      tk.ntile = (!ksplit && top && H % 32 == 0) ? 2 : 1;
      tk.nct = ksplit ? H / UWH : H / (16 * tk.ntile);
      if (tk.nct > 32) UNSUP(8);
      if (ksplit) {
        if (tk.nct > 16 || (long)2 * tk.nct * S.B * H * 4 >= (1L << 30)) UNSUP(11);
        const long need = (long)2 * tk.nct * S.B * H;
        if (part_used + need > g_persist_scratch_floats) UNSUP(12);
        tk.part = g_persist_scratch + part_used; part_used += need;
      }
      cost[t_i] = tk.nct; upper[t_i] = top ? -1 : t_i + 1;
      if (S.seed) {
        const uint32_t cid = (uint32_t)(S.cell_id_base + l);
        tk.seed = S.seed; tk.k_st = S.keep_state; tk.k_out = S.keep_out; tk.k_in = 1.0f;
        tk.r_st = cid * 4 + 1; tk.r_out = cid * 4 + 2;
        if (!top && !ksplit) { tk.k_in = S.keep_in; tk.r_in = (cid + 1) * 4; tk.in_W = H; tk.in_coff = 0; }
      }
      if (ksplit && !top) {
        // HELP task right behind its cell: the cell reads the helper (upper[cell] = cell + 1), the helper reads the cell above
        // (upper[helper] = helper + 1 = the next layer's cell)
        if (L.ntask >= B_MAX_TASKS) UNSUP(4);
        const avsr_rnn_layer& Up = S.layer[l + 1];
        const int h_i = L.ntask++;
        BTask& hk = L.task[h_i];
        hk.kind = 1;
        hk.w_up = Up.w; hk.ldw_up = 4 * Up.units; hk.H_up = Up.units; hk.up_dgates = Up.dgates;
        hk.len = S.len; hk.B = S.B; hk.T = S.T; hk.H = H; hk.reverse = S.reverse;
        hk.ntile = 1; hk.nct = H / UWH;
        const long need = (long)S.B * S.T * H;
        if (part_used + need > g_persist_scratch_floats) UNSUP(13);
        hk.dx = g_persist_scratch + part_used; part_used += need;
        tk.dx = hk.dx;
        cost[h_i] = hk.nct; upper[h_i] = h_i + 1;
        if (S.seed) {
          const uint32_t cid = (uint32_t)(S.cell_id_base + l);
          hk.seed = S.seed; hk.k_in = S.keep_in; hk.r_in = (cid + 1) * 4; hk.in_W = H; hk.in_coff = 0;
          hk.k_st = hk.k_out = 1.0f;
        }
      }
    }
    (void)first;
  }
  // one 512-thread workgroup per CU (its register budget admits no second one): <= 32 per XCD, keep a margin
  // (the K-split kernel fills an XCD: 32, like the forward kernel's 96 three-per-CU workgroups)
  // (K-split kernel: 512-thread workgroups held to 128 VGPRs, two per CU)
  int ecost[B_MAX_TASKS];
  for (int i = 0; i < L.ntask; ++i) ecost[i] = L.task[i].kind == 1 ? 1 : 3;
  static const int solo_on = getenv("AVSR_RNN_BWD_SOLO") ? atoi(getenv("AVSR_RNN_BWD_SOLO")) : 1;
  int total_cost = 0;
  for (int i = 0; i < L.ntask; ++i) total_cost += cost[i];
  const int cap = ksplit ? (wide ? 32 : 64) : 28;                         // workgroup slots of an XCD (1024 threads: one per CU)
  const bool solo = ksplit && solo_on && B > 64 && total_cost <= cap;     // one XCD per 16-row group, 128 rows per launch
  if (solo) { for (int i = 0; i < L.ntask; ++i) half[i] = 0; ngroups = 8; }
  else if (!assign_halves(cost, upper, L.ntask, cap, half, ksplit ? ecost : nullptr)) UNSUP(9);
  long words = P_HDR + 8;
  int slots[2] = {0, 0};
  // slots are claimed in arrival order and the dispatcher fills every CU once before it doubles up: cells take the first slots (a CU
  // to themselves as far as they go), helpers the rest
  for (int pass = 0; pass < 2; ++pass)
    for (int i = 0; i < L.ntask; ++i) {
      BTask& tk = L.task[i];
      if ((tk.kind == 1) != (pass == 1)) continue;
      tk.half = half[i]; tk.slot_begin = slots[half[i]]; slots[half[i]] += tk.nct;
      tk.prog = sync + words; words += 8 * 32;            // (solo: one row of progress words per XCD)
    }
  for (int i = 0; i < L.ntask; ++i) {
    BTask& tk = L.task[i];
    if (upper[i] < 0) continue;
    BTask& up = L.task[upper[i]];
    tk.prog_up = up.prog; tk.nct_up = up.nct;
    if (up.half != tk.half) {               // crossing edge: the upper cell publishes through memory
      tk.up_remote = 1;
      if (!up.ctr) { up.ctr = sync + words; words += (long)ngroups * up.T; }
      tk.ctr_up = up.ctr;
    }
  }
  if (words > sync_ints) UNSUP(10);
  if (dry) return AVSR_OK;
  L.err = sync; L.claim = sync + P_HDR; L.wpx0 = slots[0]; L.wpx1 = slots[1]; L.solo = solo ? 1 : 0;
  const int slice = solo ? 128 : 64;
  hipStream_t s = (hipStream_t)stream;
  const int wpx = slots[0] > slots[1] ? slots[0] : slots[1];
  for (int b0 = 0; b0 < B; b0 += slice) {              // rows are independent: consecutive launches over 64-row (solo: 128-row) slices
    const int rows = B - b0 < slice ? B - b0 : slice;
    L.b0 = b0; L.ngroups = (rows + 15) / 16;
    if (avsr::dev_zero(sync + P_HDR, sizeof(int32_t) * (words - P_HDR), s) != hipSuccess) return AVSR_ERR_HIP;
    {
      ProfScope ps(PROF_RNN_PERSIST_BWD, s, flops * rows / B);
      if (ksplit && wide) hipLaunchKernelGGL(rnn_persist_bwdk_kernel<16>, dim3(8 * wpx), dim3(1024), 0, s, L);
      else if (ksplit) hipLaunchKernelGGL(rnn_persist_bwdk_kernel<8>, dim3(8 * wpx), dim3(512), 0, s, L);
      else hipLaunchKernelGGL(rnn_persist_bwd_kernel, dim3(8 * wpx), dim3(512), 0, s, L);
    }
    AVSR_CHECK_LAUNCH();
  }
  return AVSR_OK;
}
