// Persistent BPTT, split form (first choice of avsr_rnn_bwd; rnn_persist_bwd.hip is the fused form it falls back to).
//
// The fused form gives every cell workgroup BOTH products of a BPTT step -- dG(t+1)·Whᵀ (the recurrence) and
// dG_upper(t)·Wxᵀ (the gradient arriving from the layer above) -- in one 512-thread workgroup per CU.  All waves of a
// CU then move in lock step: the matrix pipe idles while they wait / load / publish, and the waits idle while it
// multiplies (tools/persist_probe.py: 4.1k cycles of MFMA + ~3k cycles of hand-off per step, serialised).
// Here the two products are separate tasks:
//   CELL(l)   recurrence only: A = d gates of step t+1 (2-slot ring), W = Whᵀ slice, epilogue = LSTM cell backward.
//             The layer-above term arrives as a plain [B,T,H] operand dx (prefetched a step ahead with the records).
//   HELP(l)   for every cell with a layer above: dx(t) = dropout_mask ⊙ (dG_upper(t)·Wxᵀ).  No recurrence, so it runs
//             as far ahead as the layer above allows and its MFMAs fill the pipe while the cell it shares a CU with
//             is in a hand-off phase.
// Every workgroup is 256 threads = 16 batch rows (one MFMA row tile) x 16 units with a 16 x K<=1024 weight slice in
// 64 VGPRs per thread; 16-row groups live on XCD pairs, tasks are placed on the two XCDs of the pair by exhaustive
// search over the dependency edges (fewest crossing edges, then balance).  Hand-off forms, placement by XCC_ID, bounded
// waits and the sticky error word are those of rnn_persist_bwd.hip / persist.h.
#include "step.h"
#include "avsr_hip.h"
#include "prof.h"
#include "persist.h"

#include <cstdio>
#include <cstdlib>

#define S_MAX_TASKS 8      // the launch descriptor travels by value: keep it well inside the 4 KB kernarg limit
#define S_CH 16           // 16-wide K chunks per wave: K <= 1024 over 4 waves

namespace avsr {

extern float* g_persist_scratch;         // float scratch for the dx operands (caller-owned, avsr_rnn_set_persistent_scratch; rnn_persist_bwd.hip)
extern int64_t g_persist_scratch_floats;

struct STask {
  int kind;                               // 0 CELL, 1 HELP
  const float* w; long ldw;               // weight rows (one per unit of THIS layer), K contiguous
  const float* a; int a_is_ring;          // A operand: CELL ring [2][B][K] | HELP the upper layer's d gates record [B,T,K]
  const int* len;
  const float* gates; const float* cs;    // CELL: forward records
  float* dgates; float* ring;             // CELL outputs
  float* dx;                              // CELL: input (null = no layer above); HELP: output   [B,T,H]
  const float* dout; long dout_sb, dout_st;
  const float* dh_final; const float* dc_final;
  int* prog;                              // my progress words [4 groups][32]
  const int* dep1; int dep1_n;            // CELL: own progress words
  const int* dep2; int dep2_n; int dep2_remote;   // CELL: helper; HELP: upper cell.  remote: agent counter [groups][T]
  int* ctr;                               // my agent-scope arrival counters [4][T] when a consumer sits on the other XCD
  int B, T, H, K, reverse, nct, slot_begin, half;
  const int32_t* seed; float k_st, k_out, k_in; uint32_t r_st, r_out, r_in; int in_W, in_coff;
};
struct SLaunch { int ntask, ngroups, wpx0, wpx1, b0; int* err; int* claim; STask task[S_MAX_TASKS]; };

__global__ __launch_bounds__(256) void rnn_persist_bwd_split_kernel(const SLaunch L) {
  __shared__ __attribute__((aligned(16))) float red[4][16][16];
  __shared__ int s_slot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xcc = __builtin_amdgcn_readfirstlane(xcc_id());
  if (tid == 0) s_slot = __hip_atomic_fetch_add(L.claim + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int slot = __builtin_amdgcn_readfirstlane(s_slot);
  const int g = xcc >> 1, half = xcc & 1;
  if (g >= L.ngroups || slot >= (half ? L.wpx1 : L.wpx0)) return;
  int ti = 0;
#pragma unroll
  for (int i = 0; i < S_MAX_TASKS; ++i)
    if (i < L.ntask && L.task[i].half == half && slot >= L.task[i].slot_begin) ti = i;      // same-half tasks: ascending slot_begin
  ti = __builtin_amdgcn_readfirstlane(ti);
  const STask& tk = L.task[ti];
  const int ct = slot - tk.slot_begin;
  const int H = tk.H, T = tk.T, K = tk.K, reverse = tk.reverse;
  const bool cell = tk.kind == 0;
  const int unit0 = ct * 16, row0 = L.b0 + g * 16;
  const int i = lane & 15, q = lane >> 4;
  const int nch = K >> 4;
  const int c0 = (wave * nch) / 4, nc = ((wave + 1) * nch) / 4 - c0;           // <= S_CH (host-checked)

  // ---- weight slice: registers for the whole sequence ----
  f32x4 wv[S_CH];
  {
    const int uu = unit0 + i;
#pragma unroll
    for (int c = 0; c < S_CH; ++c)
      wv[c] = (c < nc && uu < H) ? ld4(tk.w + (long)uu * tk.ldw + (c0 + c) * 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- epilogue ownership: one thread per (row, unit) for all steps ----
  const int er = tid >> 4, eu = tid & 15;
  const int b = row0 + er, u = unit0 + eu;
  const bool eok = b < tk.B && u < H;
  const int len_b = eok ? (tk.len ? tk.len[b] : T) : 0;
  float dc_carry = (cell && eok && tk.dc_final) ? tk.dc_final[(long)b * H + u] : 0.f;
  float dh_carry = (cell && eok && tk.dh_final) ? tk.dh_final[(long)b * H + u] : 0.f;
  const int rec_b = b * T * H + u;
  const __amdgpu_buffer_rsrc_t gates_rs = make_rsrc(tk.gates), cs_rs = make_rsrc(tk.cs), dout_rs = make_rsrc(tk.dout);
  const __amdgpu_buffer_rsrc_t dx_rs = make_rsrc(tk.dx), dgates_rs = make_rsrc(tk.dgates);
  const bool has_dout = tk.dout != nullptr, has_dx = cell && tk.dx != nullptr;
  float* const dgates_p = tk.dgates; float* const ring_p = tk.ring; float* const dx_p = tk.dx;
  const int dout_b = (int)(b * tk.dout_sb) + u, dout_st = (int)tk.dout_st;
  const bool drop_on = tk.seed != nullptr;
  const uint32_t seedv = drop_on ? (uint32_t)tk.seed[0] : 0u;
  const float k_st = tk.k_st, k_out = tk.k_out, k_in = tk.k_in;
  const uint32_t r_st = tk.r_st, r_out = tk.r_out, r_in = tk.r_in;
  const int in_W = tk.in_W, in_coff = tk.in_coff;
  const bool publish_remote = tk.ctr != nullptr;

  // ---- A-operand row of this lane (byte offsets; P_OOB = reads as zero) ----
  const int ab = row0 + i;
  const bool aok = ab < tk.B;
  const int len_a = aok ? (tk.len ? tk.len[ab] : T) : 0;
  const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(tk.a);
  const int ring_par = tk.B * K * 4;
  const int a_row = cell ? (ab * K + c0 * 16 + 4 * q) * 4 : (ab * T * K + c0 * 16 + 4 * q) * 4;

  // ---- progress polling (wave 0): lanes 0-31 dependency 1, lanes 32-63 dependency 2 ----
  int* const my_prog = tk.prog + g * 32 + ct;
  int* const my_ctr = publish_remote ? tk.ctr + (long)g * T : nullptr;
  const int dep2_remote = tk.dep2_remote, dep2_n = tk.dep2_n;
  const int* poll1 = nullptr; const int* poll2 = nullptr;
  if (wave == 0) {
    if (lane < 32) { if (tk.dep1 && lane < tk.dep1_n) poll1 = tk.dep1 + g * 32 + lane; }
    else if (tk.dep2) {
      if (dep2_remote) { if (lane == 32) poll2 = tk.dep2 + (long)g * T; }
      else if (lane - 32 < dep2_n) poll2 = tk.dep2 + g * 32 + (lane - 32);
    }
  }
  // dependency 1: steps completed >= need1.  dependency 2: step index t2 completed (local: progress >= T - t2; crossing: counter[t2] >= n)
  auto wait_progress = [&](int need1, int t2) {
    if (wave != 0) return;
    const int* p = lane < 32 ? poll1 : (poll2 ? (dep2_remote ? poll2 + t2 : poll2) : nullptr);
    const int need = lane < 32 ? need1 : (dep2_remote ? dep2_n : T - t2);
    for (int spins = 0; spins < (1 << 21); ++spins) {
      const int v = p ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
      if (__all(v >= need)) return;
      if ((spins & 1023) == 1023 && __hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
    if (lane == 0) __hip_atomic_store(L.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // CELL: epilogue operands of the next step (forward records: cold in HBM; dx: written by the helper) one step ahead
  f32x4 n_g = {0.f, 0.f, 0.f, 0.f};
  float n_c = 0.f, n_cp = 0.f, n_do = 0.f, n_dx = 0.f;
  auto load_rec = [&](int t) {
    const bool v = cell && eok && t >= 0 && t < len_b;
    const int tau = reverse ? len_b - 1 - t : t;
    const int o = v ? (rec_b + tau * H) * 4 : P_OOB;
    n_g = ldb4(gates_rs, v ? o * 4 : P_OOB);
    n_c = ldb1(cs_rs, o);
    n_cp = ldb1(cs_rs, (v && t > 0) ? o + (reverse ? H : -H) * 4 : P_OOB);
    n_do = ldb1(dout_rs, (v && has_dout) ? (dout_b + tau * dout_st) * 4 : P_OOB);
    n_dx = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b32(dx_rs, (v && has_dx) ? o : P_OOB, 0, 16));   // sc1: produced during this launch
  };
  if (cell) {
    if (has_dx) { wait_progress(0, T - 1); }
    __syncthreads();
    load_rec(T - 1);
  }

  for (int t = T - 1; t >= 0; --t) {
    // records prefetched during the previous step are handed over HERE (see rnn_persist_bwd.hip)
    f32x4 g4 = n_g;
    float c = n_c, cprev = n_cp, dout_ext = n_do + n_dx;
    asm volatile("" : "+v"(g4), "+v"(c), "+v"(cprev), "+v"(dout_ext));
    // CELL: step t+1 of this cell; its helper one step ahead (t-1).  HELP: step t of the cell above.
    if (cell) wait_progress(T - 1 - t, t > 0 ? t - 1 : 0);
    else wait_progress(0, t);
    lds_barrier();
    // ---- A operand: all chunks of this wave in flight ----
    f32x4 av[S_CH];
    {
      int o;
      if (cell) o = aok ? a_row + ((t + 1) & 1) * ring_par : P_OOB;
      else o = (aok && t < len_a) ? a_row + (reverse ? len_a - 1 - t : t) * (K * 4) : P_OOB;
#pragma unroll
      for (int cc = 0; cc < S_CH; ++cc) av[cc] = ldb_sc1(a_rs, (cc < nc && o != P_OOB) ? o + cc * 64 : P_OOB);
    }
    if (cell) load_rec(t - 1);                 // issued last: vmcnt retires in order
    asm volatile("" ::: "memory");
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < S_CH; ++cc) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cc][0], wv[cc][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cc][1], wv[cc][1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cc][2], wv[cc][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cc][3], wv[cc][3], acc1, 0, 0, 0);
    }
    acc0 += acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][q * 4 + r][i] = acc0[r];
    lds_barrier();

    if (eok) {
      const float z = (red[0][er][eu] + red[1][er][eu]) + (red[2][er][eu] + red[3][er][eu]);
      const bool valid = t < len_b;
      const int tau = valid ? (reverse ? len_b - 1 - t : t) : t;
      const long bt = (long)b * T + tau;
      if (cell) {
        // ---- LSTM cell backward (same arithmetic as EP_LSTM_BWD in step.hip; dx already carries the input mask) ----
        f32x4 dg = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
          const uint32_t oidx = (uint32_t)(bt * H + u);
          const float dh = dout_ext * p_drop(drop_on, seedv, r_out, oidx, k_out) + (z + dh_carry) * p_drop(drop_on, seedv, r_st, oidx, k_st);
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
        st4(ring_p + (long)(t & 1) * (ring_par >> 2) + (long)b * K + u * 4, dg);
        const int ro = (rec_b + tau * H) * 4;
        if (publish_remote) stx_sc1(dgates_rs, ro, dg);
        else st4(dgates_p + ro, dg);
      } else if (valid) {
        // ---- helper: gradient wrt this layer's emitted output coming from the layer above, through its input mask ----
        const float v = z * p_drop(drop_on, seedv, r_in, (uint32_t)(bt * in_W + in_coff + u), k_in);
        if (publish_remote) st_sc1(dx_p + bt * H + u, v);
        else dx_p[bt * H + u] = v;
      }
    }
    // ---- publish ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(my_prog, T - t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (my_ctr) __hip_atomic_fetch_add(my_ctr + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Place every task on one XCD of the pair: all assignments enumerated (<= 2^12); feasible ones keep each half within
// `cap` workgroups; fewest crossing edges wins, then the smaller larger half.  edge e: consumer ec[e] reads producer ep[e].
static bool place_tasks(const int* cost, int n, const int* ep, const int* ec, int ne, int cap, int* half_out) {
  int best = -1, best_cross = 1 << 30, best_load = 1 << 30;
  for (int m = 0; m < (1 << n); ++m) {
    int load[2] = {0, 0}, cross = 0;
    for (int i = 0; i < n; ++i) load[(m >> i) & 1] += cost[i];
    if (load[0] > cap || load[1] > cap) continue;
    for (int e = 0; e < ne; ++e) cross += ((m >> ep[e]) ^ (m >> ec[e])) & 1;
    const int mx = load[0] > load[1] ? load[0] : load[1];
    if (cross < best_cross || (cross == best_cross && mx < best_load)) { best = m; best_cross = cross; best_load = mx; }
  }
  if (best < 0) return false;
  for (int i = 0; i < n; ++i) half_out[i] = (best >> i) & 1;
  return true;
}

static_assert(sizeof(SLaunch) <= 3072, "launch descriptor must fit the kernel-argument segment");

}  // namespace avsr


#define UNSUP2(code) do { if (getenv("AVSR_PERSIST_DEBUG")) fprintf(stderr, "[avsr] split persistent BPTT not used: reason %d (rnn_persist_bwd2.hip)\n", code); return AVSR_ERR_UNSUPPORTED; } while (0)

// Returns AVSR_ERR_UNSUPPORTED when this form is disabled or does not fit (the caller then tries the fused form).
int avsr_rnn_bwd_persistent_split(const avsr_rnn_stack* st, int32_t n, void* stream, int dry) {
  using namespace avsr;
  int32_t* sync = g_sync; const int64_t sync_ints = g_sync_ints;
  if (!sync || !(g_persist_mode & 4) || !g_persist_scratch) return AVSR_ERR_UNSUPPORTED;
  static thread_local SLaunch L;
  L = SLaunch{};
  double flops = 0.0;
  int cost[S_MAX_TASKS], half[S_MAX_TASKS], ep[2 * S_MAX_TASKS], ec[2 * S_MAX_TASKS], ne = 0;
  int cell_of[AVSR_MAX_STACKS][AVSR_MAX_LAYERS], help_of[AVSR_MAX_STACKS][AVSR_MAX_LAYERS];
  const int B = st[0].B;
  const int ngroups = B < 64 ? (B + 15) / 16 : 4;
  long dx_used = 0;
  for (int i = 0; i < n; ++i) {
    const avsr_rnn_stack& S = st[i];
    if (S.cell != 0 || S.B != B) UNSUP2(3);
    for (int l = 0; l < S.n_layers; ++l) {
      const avsr_rnn_layer& Ly = S.layer[l];
      const int H = Ly.units, in = Ly.in_dim;
      const bool top = l + 1 >= S.n_layers;
      if (H % 16 || H > 256 || (long)S.B * S.T * H * 4 >= (1L << 29)) UNSUP2(5);
      if (!top && (S.layer[l + 1].units % 16 || S.layer[l + 1].units > 256 || S.layer[l + 1].in_dim != H)) UNSUP2(6);
      if (Ly.dout && (long)S.B * (S.T + 2) * Ly.ld_dout >= (1L << 29)) UNSUP2(7);
      if (L.ntask + (top ? 1 : 2) > S_MAX_TASKS) UNSUP2(4);
      const int ci = L.ntask++;
      cell_of[i][l] = ci; help_of[i][l] = -1;
      STask& tk = L.task[ci];
      tk.kind = 0;
      tk.w = Ly.w + (long)in * 4 * H; tk.ldw = 4 * H; tk.K = 4 * H;
      tk.a = Ly.dstate; tk.a_is_ring = 1;
      tk.len = S.len; tk.gates = Ly.gates; tk.cs = Ly.cs; tk.dgates = Ly.dgates; tk.ring = Ly.dstate;
      if (Ly.dout) { tk.dout = Ly.dout + Ly.ld_dout + Ly.dout_col; tk.dout_sb = (long)(S.T + 2) * Ly.ld_dout; tk.dout_st = Ly.ld_dout; }
      if (top) { tk.dh_final = S.dh_final; tk.dc_final = S.dc_final; }
      tk.B = S.B; tk.T = S.T; tk.H = H; tk.reverse = S.reverse;
      tk.nct = H / 16; cost[ci] = tk.nct;
      flops += 2.0 * S.B * S.T * 4.0 * H * H;
      if (S.seed) {
        const uint32_t cid = (uint32_t)(S.cell_id_base + l);
        tk.seed = S.seed; tk.k_st = S.keep_state; tk.k_out = S.keep_out; tk.k_in = 1.0f;
        tk.r_st = cid * 4 + 1; tk.r_out = cid * 4 + 2;
      }
      if (!top) {
        const avsr_rnn_layer& Up = S.layer[l + 1];
        const int hi = L.ntask++;
        help_of[i][l] = hi;
        STask& hk = L.task[hi];
        hk.kind = 1;
        hk.w = Up.w; hk.ldw = 4 * Up.units; hk.K = 4 * Up.units;
        hk.a = Up.dgates; hk.a_is_ring = 0;
        hk.len = S.len;
        hk.B = S.B; hk.T = S.T; hk.H = H; hk.reverse = S.reverse;
        hk.nct = H / 16; cost[hi] = hk.nct;
        if ((dx_used + (long)S.B * S.T * H) > g_persist_scratch_floats) UNSUP2(11);
        hk.dx = g_persist_scratch + dx_used; dx_used += (long)S.B * S.T * H;
        L.task[ci].dx = hk.dx;
        flops += 2.0 * S.B * S.T * 4.0 * Up.units * H;
        if (S.seed) {
          const uint32_t cid = (uint32_t)(S.cell_id_base + l);
          hk.seed = S.seed; hk.k_in = S.keep_in; hk.r_in = (cid + 1) * 4; hk.in_W = H; hk.in_coff = 0;
          hk.k_st = hk.k_out = 1.0f;
        }
      }
    }
    // edges: cell(l+1) -> help(l) -> cell(l)
    for (int l = 0; l + 1 < S.n_layers; ++l) {
      ep[ne] = cell_of[i][l + 1]; ec[ne] = help_of[i][l]; ++ne;
      ep[ne] = help_of[i][l]; ec[ne] = cell_of[i][l]; ++ne;
    }
  }
  // 256-thread workgroups at <= 168 VGPRs: three per CU fit; plan for two (64 per XCD) and keep the third as margin
  if (!place_tasks(cost, L.ntask, ep, ec, ne, 64, half)) UNSUP2(9);
  long words = P_HDR + 8;
  int slots[2] = {0, 0};
  for (int i = 0; i < L.ntask; ++i) {
    STask& tk = L.task[i];
    tk.half = half[i]; tk.slot_begin = slots[half[i]]; slots[half[i]] += tk.nct;
    tk.prog = sync + words; words += 4 * 32;
    if (tk.kind == 0) { tk.dep1 = tk.prog; tk.dep1_n = tk.nct; }
  }
  for (int e = 0; e < ne; ++e) {
    STask& p = L.task[ep[e]];
    STask& c = L.task[ec[e]];
    c.dep2 = p.prog; c.dep2_n = p.nct; c.dep2_remote = 0;
    if (p.half != c.half) {                 // crossing edge: the producer publishes through memory
      if (!p.ctr) { p.ctr = sync + words; words += (long)4 * p.T; }
      c.dep2 = p.ctr; c.dep2_remote = 1;
    }
  }
  if (words > sync_ints) UNSUP2(10);
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
      hipLaunchKernelGGL(rnn_persist_bwd_split_kernel, dim3(8 * wpx), dim3(256), 0, s, L);
    }
    AVSR_CHECK_LAUNCH();
  }
  return AVSR_OK;
}
