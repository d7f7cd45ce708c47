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

#define B_MAX_TASKS 8
#define B_CH 8            // 16-wide K chunks per wave per operand part (4H / 16 / 8 waves, H <= 256)

namespace avsr {

struct BTask {
  const float* w_own; const float* w_up; long ldw_own, ldw_up;
  const int* len;
  const float* gates; const float* cs;
  float* dgates; float* ring; const float* up_dgates;
  const float* dout; long dout_sb, dout_st;
  const float* dh_final; const float* dc_final;
  int* prog; const int* prog_up;
  int* ctr; const int* ctr_up;
  int B, T, H, H_up, reverse, nct, nct_up, slot_begin, half, ntile, up_remote;
  const int32_t* seed; float k_st, k_out, k_in; uint32_t r_st, r_out, r_in; int in_W, in_coff;
};
struct BLaunch { int ntask, ngroups, wpx0, wpx1, b0; int* err; int* claim; BTask task[B_MAX_TASKS]; };   // b0: first batch row of this launch

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

// Choose the XCD half of every cell: all assignments are enumerated (<= 2^8); feasible ones keep each half within
// `cap` workgroups; the winner has the fewest crossing layer edges, then the smallest larger half.
static bool assign_halves(const int* cost, const int* upper, int n, int cap, int* half_out) {
  int best = -1, best_cross = 1 << 30, best_load = 1 << 30;
  for (int m = 0; m < (1 << n); ++m) {
    int load[2] = {0, 0}, cross = 0;
    for (int i = 0; i < n; ++i) {
      load[(m >> i) & 1] += cost[i];
      if (upper[i] >= 0 && (((m >> i) ^ (m >> upper[i])) & 1)) ++cross;
    }
    if (load[0] > cap || load[1] > cap) continue;
    const int mx = load[0] > load[1] ? load[0] : load[1];
    if (cross < best_cross || (cross == best_cross && mx < best_load)) { best = m; best_cross = cross; best_load = mx; }
  }
  if (best < 0) return false;
  for (int i = 0; i < n; ++i) half_out[i] = (best >> i) & 1;
  return true;
}

}  // namespace avsr

// Returns AVSR_ERR_UNSUPPORTED when the persistent path is disabled or the configuration does not fit it
// (avsr_rnn_bwd then uses one launch per wavefront step).  The caller has already zeroed every layer's dstate
// (ring slots) and nothing else of the launch path's setup is needed: the final-state gradients are read here.
#include <cstdio>
#include <cstdlib>
// AVSR_PERSIST_DEBUG=1 prints which precondition sent a call back to the per-step launches
#define UNSUP(code) do { if (getenv("AVSR_PERSIST_DEBUG")) fprintf(stderr, "[avsr] persistent BPTT not used: reason %d (rnn_persist_bwd.hip)\n", code); return AVSR_ERR_UNSUPPORTED; } while (0)
int avsr_rnn_bwd_persistent(const avsr_rnn_stack* st, int32_t n, void* stream, int dry) {
  using namespace avsr;
  int32_t* sync = g_sync; const int64_t sync_ints = g_sync_ints;
  if (!sync || !(g_persist_mode & 2)) return AVSR_ERR_UNSUPPORTED;
  static thread_local BLaunch L;
  L = BLaunch{};
  double flops = 0.0;
  int cost[B_MAX_TASKS], upper[B_MAX_TASKS], half[B_MAX_TASKS];
  const int B = st[0].B;
  const int ngroups = B < 64 ? (B + 15) / 16 : 4;      // per launch: 4 XCD pairs x 16 rows; larger batches run as 64-row slices
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
      if (!top) { const avsr_rnn_layer& Up = S.layer[l + 1]; tk.w_up = Up.w; tk.ldw_up = 4 * Up.units; tk.H_up = Up.units; tk.up_dgates = Up.dgates; }
      tk.len = S.len; tk.gates = Ly.gates; tk.cs = Ly.cs; tk.dgates = Ly.dgates; tk.ring = Ly.dstate;
      if (Ly.dout) { tk.dout = Ly.dout + Ly.ld_dout + Ly.dout_col; tk.dout_sb = (long)(S.T + 2) * Ly.ld_dout; tk.dout_st = Ly.ld_dout; }
      if (top) { tk.dh_final = S.dh_final; tk.dc_final = S.dc_final; }
      tk.B = S.B; tk.T = S.T; tk.H = H; tk.reverse = S.reverse;
      flops += 2.0 * S.B * S.T * (4.0 * H + (top ? 0.0 : 4.0 * S.layer[l + 1].units)) * H;
      tk.ntile = (top && H % 32 == 0) ? 2 : 1;
      tk.nct = H / (16 * tk.ntile);
      if (tk.nct > 32) UNSUP(8);
      cost[t_i] = tk.nct; upper[t_i] = top ? -1 : t_i + 1;
      if (S.seed) {
        const uint32_t cid = (uint32_t)(S.cell_id_base + l);
        tk.seed = S.seed; tk.k_st = S.keep_state; tk.k_out = S.keep_out; tk.k_in = 1.0f;
        tk.r_st = cid * 4 + 1; tk.r_out = cid * 4 + 2;
        if (!top) { tk.k_in = S.keep_in; tk.r_in = (cid + 1) * 4; tk.in_W = H; tk.in_coff = 0; }
      }
    }
    (void)first;
  }
  // one 512-thread workgroup per CU (its register budget admits no second one): <= 32 per XCD, keep a margin
  if (!assign_halves(cost, upper, L.ntask, 28, half)) UNSUP(9);
  long words = P_HDR + 8;
  int slots[2] = {0, 0};
  for (int i = 0; i < L.ntask; ++i) {
    BTask& tk = L.task[i];
    tk.half = half[i]; tk.slot_begin = slots[half[i]]; slots[half[i]] += tk.nct;
    tk.prog = sync + words; words += 4 * 32;
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
  L.err = sync; L.claim = sync + P_HDR; L.wpx0 = slots[0]; L.wpx1 = slots[1];
  hipStream_t s = (hipStream_t)stream;
  const int wpx = slots[0] > slots[1] ? slots[0] : slots[1];
  for (int b0 = 0; b0 < B; b0 += 64) {                 // rows are independent: consecutive launches over 64-row slices
    const int rows = B - b0 < 64 ? B - b0 : 64;
    L.b0 = b0; L.ngroups = (rows + 15) / 16;
    if (avsr::dev_zero(sync + P_HDR, sizeof(int32_t) * (words - P_HDR), s) != hipSuccess) return AVSR_ERR_HIP;
    {
      ProfScope ps(PROF_RNN_PERSIST_BWD, s, flops * rows / B);
      hipLaunchKernelGGL(rnn_persist_bwd_kernel, dim3(8 * wpx), dim3(512), 0, s, L);
    }
    AVSR_CHECK_LAUNCH();
  }
  return AVSR_OK;
}
