// Persistent encoder forward, pair layout (first choice of avsr_rnn_fwd when the batch fits: <= 64 rows per launch slice).
//
// Same work per workgroup as the XCD-local kernel of rnn_persist.hip, re-laid out like the BPTT kernel: batch groups of 16 rows
// (a full MFMA row tile -- the 8-row groups pad half of every 16x16x4 tile, and rocprof shows that kernel's matrix pipe 43 %
// busy, half of it on padding) on PAIRS of XCDs; the cells of a group are placed on the two XCDs by exhaustive search (fewest
// crossing layer edges, then balance).  A cell's own recurrence never leaves its XCD: h(t) goes through a two-slot ring
// [2][B][H] (plain stores, progress words).  The sequence the layer above reads (xt_seq under dropout, else out) is written
// with write-through stores and announced through an agent-scope counter only when that layer sits on the other XCD.
#include "step.h"
#include "avsr_hip.h"
#include "prof.h"
#include "persist.h"

#define P_MAX_TASKS 8
#define P_XC 4
#define P_HC 4
#ifdef PERSIST_TIMING
#define TICK(k) { const long now_ = __builtin_amdgcn_s_memtime(); tm[k] += now_ - last_; last_ = now_; }
#else
#define TICK(k)
#endif

namespace avsr {

struct FTask {
  const float* wt; const float* bias; const int* len;
  float* gates; float* cs;
  float* out; long out_sb, out_st;
  float* hs_w; long hs_sb, hs_st;
  float* ring;                              // [2][B][H] state h of the previous step (layer.state of the launch path)
  const float* x_r; long x_sb, x_st;
  float* xt_w; long xt_sb, xt_st;
  float* h_final; float* c_final;
  int* done; const int* done_lower;         // progress words [4 groups][32]; done_lower = counters [4][T] when low_remote
  int* ctr;                                 // my arrival counters [4][T] when the layer above is on the other XCD
  int B, T, H, in, hoisted, reverse, nct, nct_lower, wg_begin, uw, half, low_remote;
  const int32_t* seed; float k_st, k_out, k_in; uint32_t r_st, r_out, r_in; int in_W, in_coff;
};
struct FLaunch { int ntask, ngroups, wpx0, wpx1, b0; int* err; int* claim; FTask task[P_MAX_TASKS]; };
static_assert(sizeof(FLaunch) <= 3072, "launch descriptor must fit the kernel-argument segment");

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void rnn_persist_fwd_pair_kernel(const FLaunch L) {
  __shared__ __attribute__((aligned(16))) float red[4][4][16][16];
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
  for (int i = 0; i < P_MAX_TASKS; ++i)
    if (i < L.ntask && L.task[i].half == half && slot >= L.task[i].wg_begin) ti = i;    // same-half tasks: ascending wg_begin
  ti = __builtin_amdgcn_readfirstlane(ti);
  const FTask& tk = L.task[ti];
  const int ct = slot - tk.wg_begin;
  const int UW = tk.uw, uw_shift = UW == 16 ? 4 : 3;
  const int row0 = L.b0 + g * 16, col0 = ct * UW * 4, unit0 = ct * UW;
  const int i = lane & 15, q = lane >> 4;
  const int H = tk.H, T = tk.T;
  const int hoisted = tk.hoisted, reverse = tk.reverse;
  const bool wide = hoisted && UW == 16;          // recurrent-only layer: the input-part registers hold two more column tiles
  const int Kx = hoisted ? 0 : tk.in;
  const int ncx = (Kx + 15) >> 4, nchh = (H + 15) >> 4;
  const int xg0 = (wave * ncx) / 4, nxw = ((wave + 1) * ncx) / 4 - xg0;
  const int hg0 = (wave * nchh) / 4, nhw = ((wave + 1) * nchh) / 4 - hg0;
  const long ldw = tk.in + H;

  f32x4 wa[P_XC][2], wb[P_HC][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = col0 + nt * 16 + i;
#pragma unroll
    for (int c = 0; c < P_HC; ++c) {
      const int k = (hg0 + c) * 16 + 4 * q;
      wb[c][nt] = (c < nhw && col < 4 * H && k < H) ? ld4(tk.wt + (long)col * ldw + tk.in + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < P_XC; ++c) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (wide) {
        const int k = (hg0 + c) * 16 + 4 * q, col2 = col + 32;
        if (c < nhw && col2 < 4 * H && k < H) v = ld4(tk.wt + (long)col2 * ldw + tk.in + k);
      } else {
        const int k = (xg0 + c) * 16 + 4 * q;
        if (c < nxw && col < 4 * H && k < Kx) v = ld4(tk.wt + (long)col * ldw + k);
      }
      wa[c][nt] = v;
    }
  }

  // epilogue ownership: thread e (< 16 * UW) owns (row er, unit eu) for all steps
  const int er = tid >> uw_shift, eu = tid & (UW - 1);
  const int b = row0 + er, u = unit0 + eu;
  const bool eok = tid < 16 * UW && b < tk.B && u < H;
  const int len_b = eok ? (tk.len ? tk.len[b] : T) : 0;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (eok && tk.bias) bias4 = ld4(tk.bias + u * 4);
  float c_state = 0.f, h_state = 0.f;

  const int ab = row0 + i;
  const bool aok = ab < tk.B;
  const int len_a = aok ? (tk.len ? tk.len[ab] : T) : 0;
  const int xrow = (int)(ab * tk.x_sb) + 4 * q, hrow = ab * H + 4 * q;      // h(t-1): 2-slot ring [2][B][H], plain stores, this XCD only
  const int x_st = (int)tk.x_st, ring_par = tk.B * H;
  // unconditional raw buffer loads, out-of-range offset = reads zero: exact vmcnt counting keeps the prefetches in flight
  const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(tk.x_r), h_rs = make_rsrc(tk.ring), z_rs = make_rsrc(tk.gates);
  const bool has_low = !hoisted;
  // progress words: 32 per (task, group); wave 0 polls own (lanes 0-31) and lower (lanes 32-63) in one load
  int* const my_flag = tk.done + g * 32 + ct;
  int* const my_ctr = tk.ctr ? tk.ctr + (long)g * T : nullptr;       // agent-scope arrival counters when the layer above sits on the other XCD
  const int low_remote = tk.low_remote, nct_lower = tk.nct_lower;
  const int* poll_ptr = nullptr;
  if (wave == 0) {
    if (lane < 32) { if (lane < tk.nct) poll_ptr = tk.done + g * 32 + lane; }
    else if (has_low) {
      if (low_remote) { if (lane == 32) poll_ptr = tk.done_lower + (long)g * T; }
      else if (lane - 32 < nct_lower) poll_ptr = tk.done_lower + g * 32 + (lane - 32);
    }
  }
  const int rec_b = b * T * H + u;
  const int out_b = (int)(b * tk.out_sb) + u, hsw_b = (int)(b * tk.hs_sb) + u, xtw_b = (int)(b * tk.xt_sb) + u;
  const int out_st = (int)tk.out_st, hs_st = (int)tk.hs_st, xt_st = (int)tk.xt_st;
  float* const gates_p = tk.gates; float* const cs_p = tk.cs; float* const out_p = tk.out;
  float* const hsw_p = tk.hs_w; float* const xtw_p = tk.xt_w; float* const ring_p = tk.ring;
  const bool x_remote = tk.ctr != nullptr;
  const bool drop_on = tk.seed != nullptr;
  const uint32_t seedv = drop_on ? (uint32_t)tk.seed[0] : 0u;
  const float k_st = tk.k_st, k_out = tk.k_out, k_in = tk.k_in;
  const uint32_t r_st = tk.r_st, r_out = tk.r_out, r_in = tk.r_in;
  const int in_W = tk.in_W, in_coff = tk.in_coff;

  // wave 0: wait until own progress >= need_own and lower progress >= need_low (bounded)
  // own: steps completed >= need_own.  lower: steps completed >= need_low (local: progress word; crossing: counter[need_low-1] >= nct)
  auto wait_progress = [&](int need_own, int need_low) {
    if (wave != 0) return;
    const bool rem = lane >= 32 && low_remote;
    const int* pp = rem && poll_ptr ? poll_ptr + (need_low - 1) : poll_ptr;
    const int need = lane < 32 ? need_own : (rem ? nct_lower : need_low);
    for (int spins = 0; spins < (1 << 21); ++spins) {
      const int v = pp ? __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
      if (__all(v >= need)) return;
#ifdef POLL_SLEEP
      __builtin_amdgcn_s_sleep(POLL_SLEEP);
#endif
      if ((spins & 1023) == 1023 && __hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
    if (lane == 0) __hip_atomic_store(L.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  f32x4 xcur[P_XC];
#pragma unroll
  for (int c = 0; c < P_XC; ++c) xcur[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_x = [&](int t, f32x4* dst) {
    const bool v = aok && t < len_a;
    const int xo = xrow + (reverse ? len_a - 1 - t : t) * x_st;
#pragma unroll
    for (int c = 0; c < P_XC; ++c) {
      const int k = (xg0 + c) * 16;
      dst[c] = ldb_sc1(x_rs, (c < nxw && v && k + 4 * q < Kx) ? (xo + k) * 4 : P_OOB);
    }
  };
  if (has_low) {
    wait_progress(0, 1);
    __syncthreads();
    load_x(0, xcur);
  }
  f32x4 znext = {0.f, 0.f, 0.f, 0.f};
  znext = ldb4(z_rs, (eok && hoisted && 0 < len_b) ? (rec_b + (reverse ? len_b - 1 : 0) * H) * 16 : P_OOB);
#ifdef PERSIST_TIMING
  long tm[6] = {0, 0, 0, 0, 0, 0};
#endif
  for (int t = 0; t < T; ++t) {
#ifdef PERSIST_TIMING
    long last_ = __builtin_amdgcn_s_memtime();
#endif
    f32x4 zpre = znext;                          // prefetched during the previous step; handed over here (see rnn_persist_bwd.hip)
    asm volatile("" : "+v"(zpre));
    // dependencies: step t-1 of this layer (all column tiles of my rows); the layer below one step ahead
    wait_progress(t, t + 2 < T ? t + 2 : T);
    lds_barrier();
    TICK(0)
    const bool avalid = aok && t < len_a;
    const int ho_ = hrow + (t & 1) * ring_par;                    // slot t&1 holds the state after step t-1
    f32x4 hv[P_HC];
#pragma unroll
    for (int c = 0; c < P_HC; ++c) {
      const int k = (hg0 + c) * 16;
      hv[c] = ldb_sc1(h_rs, (c < nhw && avalid && t > 0 && k + 4 * q < H) ? (ho_ + k) * 4 : P_OOB);
    }
#ifdef PERSIST_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // timing build only: isolate the recurrent-operand latency
    TICK(1)
#endif
    f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!hoisted) {
      // input part first: its operands arrived a step ago, so these MFMAs run under the loads just issued
#pragma unroll
      for (int c = 0; c < P_XC; ++c)
        if (c < nxw) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xcur[c][e], wa[c][nt][e], acc[nt], 0, 0, 0);
        }
      load_x(t + 1 < T ? t + 1 : T, xcur);     // refill in place: consumed a step from now (past the end: nothing is fetched)
    }
    // hoisted x.Wx of the NEXT step (cold in HBM).  Issued last: vmcnt retires in order, so a slow load must be
    // younger than the recurrent operands or it would stall their wait.
    znext = ldb4(z_rs, (eok && hoisted && t + 1 < len_b) ? (rec_b + (reverse ? len_b - 2 - t : t + 1) * H) * 16 : P_OOB);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < P_HC; ++c)
      if (c < nhw) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[c][e], wb[c][nt][e], acc[nt], 0, 0, 0);
        if (wide) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[2 + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[c][e], wa[c][nt][e], acc[2 + nt], 0, 0, 0);
        }
      }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][nt][q * 4 + r][i] = acc[nt][r];
    lds_barrier();
    TICK(2)
    if (eok) {
      const bool valid = t < len_b;
      if (valid) {
        const int tau = reverse ? len_b - 1 - t : t;
        const long bt = (long)b * T + tau;
        f32x4 z = bias4 + zpre;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
          const int cc = eu * 4 + gi;
          z[gi] += (red[0][cc >> 4][er][cc & 15] + red[1][cc >> 4][er][cc & 15]) + (red[2][cc >> 4][er][cc & 15] + red[3][cc >> 4][er][cc & 15]);
        }
        f32x4 g4;
        g4[0] = p_sigmoid(z[0]); g4[1] = p_tanh(z[1]); g4[2] = p_sigmoid(z[2] + 1.0f); g4[3] = p_sigmoid(z[3]);
        float c = g4[2] * c_state + g4[0] * g4[1];
        c = fminf(1.0f, fmaxf(-1.0f, c));
        const float h = g4[3] * p_tanh(c);
        const uint32_t oidx = (uint32_t)(bt * H + u);
        const float ho = h * p_drop(drop_on, seedv, r_out, oidx, k_out);
        const float hs = h * p_drop(drop_on, seedv, r_st, oidx, k_st);
        ring_p[((t + 1) & 1) * ring_par + b * H + u] = hs;             // recurrence: this XCD only
        const float xn = ho * p_drop(drop_on, seedv, r_in, (uint32_t)(bt * in_W + in_coff + u), k_in);
        // the sequence the layer above consumes (xt_seq under dropout, else out) goes through memory when that layer is remote
        if (x_remote && !xtw_p) st_sc1(out_p + (out_b + tau * out_st), ho); else out_p[out_b + tau * out_st] = ho;
        if (hsw_p) hsw_p[hsw_b + tau * hs_st] = hs;
        if (xtw_p) { if (x_remote) st_sc1(xtw_p + (xtw_b + tau * xt_st), xn); else xtw_p[xtw_b + tau * xt_st] = xn; }
        st4(gates_p + (long)(rec_b + tau * H) * 4, g4);
        cs_p[rec_b + tau * H] = c;
        c_state = c;
        h_state = hs;
      } else {                     // past the utterance: zero output at padding position t, state carried in registers
        ring_p[((t + 1) & 1) * ring_par + b * H + u] = h_state;
        if (x_remote && !xtw_p) st_sc1(out_p + (out_b + t * out_st), 0.f); else out_p[out_b + t * out_st] = 0.f;
        if (hsw_p) hsw_p[hsw_b + t * hs_st] = 0.f;
        if (xtw_p) { if (x_remote) st_sc1(xtw_p + (xtw_b + t * xt_st), 0.f); else xtw_p[xtw_b + t * xt_st] = 0.f; }
      }
    }
    TICK(3)
    // publish: stores drained into the XCD's L2, then this workgroup's progress word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TICK(4)
    if (tid == 0) {
      __hip_atomic_store(my_flag, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (my_ctr) __hip_atomic_fetch_add(my_ctr + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    TICK(5)
  }
#ifdef PERSIST_TIMING
  if (tid == 0 && ct == 0 && g == 0)
    for (int k = 0; k < 6; ++k) L.err[16 + ti * 8 + k] = (int)(tm[k] / T);
#endif
  if (eok) {
    if (tk.h_final) tk.h_final[(long)b * H + u] = h_state;
    if (tk.c_final) tk.c_final[(long)b * H + u] = c_state;
  }
}


// same search as the BPTT kernel (rnn_persist_bwd2.hip): edge e = consumer ec[e] reads producer ep[e]
static bool place_cells(const int* cost, int n, const int* ep, const int* ec, int ne, int cap, int* half_out) {
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

}  // namespace avsr

// Returns AVSR_ERR_UNSUPPORTED when this layout is disabled (mode bit 3) or does not fit; the caller then tries the other forms.
int avsr_rnn_fwd_persistent_pair(const avsr_rnn_stack* st, int32_t n, void* stream, int dry) {
  using namespace avsr;
  int32_t* sync = g_sync; const int64_t sync_ints = g_sync_ints;
  if (!sync || !(g_persist_mode & 8)) return AVSR_ERR_UNSUPPORTED;
  static thread_local FLaunch L;
  L = FLaunch{};
  double flops = 0.0;
  int cost[P_MAX_TASKS], half[P_MAX_TASKS], ep[P_MAX_TASKS], ec[P_MAX_TASKS], ne = 0;
  const int B = st[0].B;
  for (int i = 0; i < n; ++i) {
    const avsr_rnn_stack& S = st[i];
    if (S.cell != 0 || S.B != B) return AVSR_ERR_UNSUPPORTED;
    for (int l = 0; l < S.n_layers; ++l) {
      const avsr_rnn_layer& Ly = S.layer[l];
      if (L.ntask >= P_MAX_TASKS) return AVSR_ERR_UNSUPPORTED;
      const int H = Ly.units, in = Ly.in_dim;
      if ((long)S.B * (S.T + 2) * (Ly.ld_out > 4 * H ? Ly.ld_out : 4 * H) >= (1L << 29)) return AVSR_ERR_UNSUPPORTED;
      if (H % 8 || !Ly.out || !Ly.state || H > 64 * P_HC || (!Ly.hoisted && in > 64 * P_XC) || in % 4) return AVSR_ERR_UNSUPPORTED;
      if (!Ly.hoisted && l == 0) return AVSR_ERR_UNSUPPORTED;
      if (S.seed && !Ly.hs_seq) return AVSR_ERR_UNSUPPORTED;
      const int ti = L.ntask++;
      FTask& tk = L.task[ti];
      tk.wt = Ly.wt; tk.bias = Ly.bias; tk.len = S.len;
      tk.gates = Ly.gates; tk.cs = Ly.cs; tk.ring = Ly.state;
      tk.out = Ly.out + Ly.ld_out + Ly.out_col; tk.out_sb = (long)(S.T + 2) * Ly.ld_out; tk.out_st = Ly.ld_out;
      if (S.seed) { tk.hs_w = Ly.hs_seq + H; tk.hs_sb = (long)(S.T + 2) * H; tk.hs_st = H; }
      if (!Ly.hoisted) {
        const avsr_rnn_layer& Lo = S.layer[l - 1];
        if (S.seed) { if (!Lo.xt_seq) return AVSR_ERR_UNSUPPORTED; tk.x_r = Lo.xt_seq + Lo.units; tk.x_sb = (long)(S.T + 2) * Lo.units; tk.x_st = Lo.units; }
        else { tk.x_r = Lo.out + Lo.ld_out + Lo.out_col; tk.x_sb = (long)(S.T + 2) * Lo.ld_out; tk.x_st = Lo.ld_out; }
        ep[ne] = ti - 1; ec[ne] = ti; ++ne;
      }
      if (S.seed && Ly.xt_seq) { tk.xt_w = Ly.xt_seq + H; tk.xt_sb = (long)(S.T + 2) * H; tk.xt_st = H; }
      tk.h_final = Ly.h_final; tk.c_final = Ly.c_final;
      tk.B = S.B; tk.T = S.T; tk.H = H; tk.in = in; tk.hoisted = Ly.hoisted; tk.reverse = S.reverse;
      flops += 2.0 * S.B * S.T * ((Ly.hoisted ? 0 : in) + H) * 4.0 * H;
      tk.uw = (Ly.hoisted && H % 16 == 0) ? 16 : 8;
      tk.nct = H / tk.uw;
      if (tk.nct > 32) return AVSR_ERR_UNSUPPORTED;
      cost[ti] = tk.nct;
      if (S.seed) {
        const uint32_t cid = (uint32_t)(S.cell_id_base + l);
        tk.seed = S.seed; tk.k_st = S.keep_state; tk.k_out = S.keep_out; tk.k_in = 1.0f;
        tk.r_st = cid * 4 + 1; tk.r_out = cid * 4 + 2;
        if (l + 1 < S.n_layers) { tk.k_in = S.keep_in; tk.r_in = (cid + 1) * 4; tk.in_W = H; tk.in_coff = 0; }
        else if (S.consumer_width > 0) { tk.k_in = S.consumer_keep; tk.r_in = (uint32_t)S.consumer_stream; tk.in_W = S.consumer_width; tk.in_coff = 0; }
      }
    }
  }
  // 256-thread workgroups at <= 168 VGPRs: three per CU fit (96 per XCD); plan for 80 and keep the rest as margin
  if (!place_cells(cost, L.ntask, ep, ec, ne, 80, half)) return AVSR_ERR_UNSUPPORTED;
  long words = P_HDR + 8;
  int slots[2] = {0, 0};
  for (int i = 0; i < L.ntask; ++i) {
    FTask& tk = L.task[i];
    tk.half = half[i]; tk.wg_begin = slots[half[i]]; slots[half[i]] += tk.nct;
    tk.done = sync + words; words += 4 * 32;
  }
  for (int e = 0; e < ne; ++e) {
    FTask& p = L.task[ep[e]];
    FTask& c = L.task[ec[e]];
    c.done_lower = p.done; c.nct_lower = p.nct; c.low_remote = 0;
    if (p.half != c.half) {
      if (!p.ctr) { p.ctr = sync + words; words += (long)4 * p.T; }
      c.done_lower = p.ctr; c.low_remote = 1;
    }
  }
  if (words > sync_ints) return AVSR_ERR_UNSUPPORTED;
  if (dry) return AVSR_OK;
  L.err = sync; L.claim = sync + P_HDR; L.wpx0 = slots[0]; L.wpx1 = slots[1];
  hipStream_t s = (hipStream_t)stream;
  const int wpx = slots[0] > slots[1] ? slots[0] : slots[1];
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int rows = B - b0 < 64 ? B - b0 : 64;
    L.b0 = b0; L.ngroups = (rows + 15) / 16;
    if (avsr::dev_zero(sync + P_HDR, sizeof(int32_t) * (words - P_HDR), s) != hipSuccess) return AVSR_ERR_HIP;
    {
      ProfScope ps(PROF_RNN_PERSIST_FWD, s, flops * rows / B);
      hipLaunchKernelGGL(rnn_persist_fwd_pair_kernel, dim3(8 * wpx), dim3(256), 0, s, L);
    }
    AVSR_CHECK_LAUNCH();
  }
  return AVSR_OK;
}
