// Persistent wavefront kernel for the masked multi-layer LSTM forward pass (opt-in fast path of avsr_rnn_fwd).
//
// Why: with one launch per wavefront step every step pays a dependent-launch boundary (~1.6 us), a cold-L2
// re-read of the step's weight slice (~10 MB of fabric traffic per launch), kernel-argument setup and an
// epilogue round trip -- ~8-10 us per step for <2 us of MFMA work (tools/step_probe.hip).  Here ONE launch
// covers the whole sequence: every workgroup owns 16 batch rows x 8 units of one (stack, layer) cell for all
// T steps, keeps its weight fragments in registers and its (c, h) state in registers, and the per-step
// all-to-all (each workgroup needs the full previous h of its 16 rows, and the layer below's output) goes
// through the sequence buffers themselves: producers store h with write-through (sc1) stores and bump a
// per-(row-tile, time) arrival counter; consumers poll that ONE counter (relaxed, agent scope), then read the
// rows with sc1 loads (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility":
// {sc1 stores + drained vmcnt + relaxed agent flag} / {relaxed poll + sc1 loads}; no dispatch-order or
// placement assumption).  Because the exchange buffers are the time-indexed records (never a ring), a fast
// layer can run ahead of a slow one without any overwrite hazard.
//
// Every spin is bounded: on timeout the workgroup raises *err and stops waiting, so a scheduling surprise
// yields a flagged wrong result instead of a hang.  Requirements (else the caller uses the launch path):
// LSTM, units % 8 == 0, (in + H) <= 512 per layer, total workgroups <= 512 (2 per CU: co-residency with margin).
#include "step.h"
#include "avsr_hip.h"
#include "prof.h"
#include "persist.h"
#include <cstring>
#include <cstdlib>

#define P_MAX_TASKS 8
#define P_MAXC 8

namespace avsr {

struct PTask {
  const float* wt; const float* bias; const int* len;
  float* gates; float* cs;
  float* out; long out_sb, out_st;          // emitted output of position tau: out + b*out_sb + tau*out_st
  float* hs_w; long hs_sb, hs_st;           // state-dropped h of position tau (dropout only; else the output doubles as state)
  const float* hs_r; long hsr_sb, hsr_st;   // index tau -> h consumed by the step at position tau (previous step's state)
  const float* x_r; long x_sb, x_st;        // lower layer's output as this layer consumes it, position tau (null: hoisted)
  float* xt_w; long xt_sb, xt_st;           // this layer's output as ITS consumer sees it (dropout only)
  float* h_final; float* c_final;
  int* done; const int* done_lower;         // arrival counters [nrt][T]
  int B, T, H, in, hoisted, reverse, nct, nct_lower, wg_begin, nrt, uw, part, prio;
  const int32_t* seed; float k_st, k_out, k_in; uint32_t r_st, r_out, r_in; int in_W, in_coff;
};
// b0: first batch row of this launch.  npart = 2 (XCD-local kernel only): the launch holds TWO independent stacks (the directions of
// a bidirectional encoder) that do not fit one XCD together: stack 0 lives on XCDs 0-3, stack 1 on XCDs 4-7, each in 16-row groups
// (<= 64 utterances), wg_begin counted per stack, pwg[p] = workgroups of stack p per XCD.
struct PLaunch { int ntask, wpx, ngroups, b0, npart, pwg[2]; int* err; int* claim; PTask task[P_MAX_TASKS]; };

// bounded wait for (*c0 >= n0 && *c1 >= n1): both counters are fetched in the same round trip
__device__ __forceinline__ bool wait_ge2(const int* c0, int n0, const int* c1, int n1, int* err) {
  for (int spins = 0; spins < (1 << 21); ++spins) {
    const int a = c0 ? __hip_atomic_load(c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : n0;
    const int b = c1 ? __hip_atomic_load(c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : n1;
    if (a >= n0 && b >= n1) return true;
    if ((spins & 1023) == 1023 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

#ifdef PERSIST_TIMING
#define TICK(k) { const long now_ = __builtin_amdgcn_s_memtime(); tm[k] += now_ - last_; last_ = now_; }
#else
#define TICK(k)
#endif
#define P_XC 4      // 16-wide K chunks per wave of the input part   (in  <= 256)
#define P_HC 4      // 16-wide K chunks per wave of the recurrent part (H <= 256)

__global__ __launch_bounds__(256) void rnn_persist_fwd_kernel(const PLaunch L) {
  __shared__ __attribute__((aligned(16))) float red[4][2][16][16];
  int ti = 0;
#pragma unroll
  for (int i = 1; i < P_MAX_TASKS; ++i)
    if (i < L.ntask && (int)blockIdx.x >= L.task[i].wg_begin) ti = i;
  ti = __builtin_amdgcn_readfirstlane(ti);       // uniform: task fields come through scalar loads
  const PTask& tk = L.task[ti];
  const int local = blockIdx.x - tk.wg_begin;
  const int rt = local / tk.nct, ct = local % tk.nct;
  const int row0 = rt * 16, col0 = ct * 32, unit0 = ct * 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int H = tk.H, T = tk.T;
  const int hoisted = tk.hoisted, reverse = tk.reverse;
  const int Kx = hoisted ? 0 : tk.in;
  // K split: every wave takes a quarter of the input chunks AND a quarter of the recurrent chunks, so the
  // recurrent part -- the only one on the step-to-step critical path -- is spread over all four SIMDs
  const int ncx = (Kx + 15) >> 4, nchh = (H + 15) >> 4;
  const int xg0 = (wave * ncx) / 4, nxw = ((wave + 1) * ncx) / 4 - xg0;      // <= P_XC (host-checked)
  const int hg0 = (wave * nchh) / 4, nhw = ((wave + 1) * nchh) / 4 - hg0;    // <= P_HC
  const long ldw = tk.in + H;

  // ---- weights of this workgroup's 32 gate columns: registers for the whole sequence ----
  f32x4 wx[P_XC][2], wh[P_HC][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = col0 + nt * 16 + i;
#pragma unroll
    for (int c = 0; c < P_XC; ++c) {
      const int k = (xg0 + c) * 16 + 4 * q;
      wx[c][nt] = (c < nxw && col < 4 * H && k < Kx) ? ld4(tk.wt + (long)col * ldw + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < P_HC; ++c) {
      const int k = (hg0 + c) * 16 + 4 * q;
      wh[c][nt] = (c < nhw && col < 4 * H && k < H) ? ld4(tk.wt + (long)col * ldw + tk.in + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  // ---- per-thread epilogue ownership: thread e (< 128) owns (row er, unit eu) for all steps ----
  const int er = tid >> 3, eu = tid & 7;
  const int b = row0 + er, u = unit0 + eu;
  const bool eok = tid < 128 && b < tk.B && u < H;
  const int len_b = eok ? (tk.len ? tk.len[b] : T) : 0;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (eok && tk.bias) bias4 = ld4(tk.bias + u * 4);
  float c_state = 0.f, h_state = 0.f;

  // A-operand row of this lane; 32-bit element offsets from uniform bases (host checks spans < 2^31 elements)
  const int ab = row0 + i;
  const bool aok = ab < tk.B;
  const int len_a = aok ? (tk.len ? tk.len[ab] : T) : 0;
  const int xrow = (int)(ab * tk.x_sb) + 4 * q, hrow = (int)(ab * tk.hsr_sb) + 4 * q;
  const int x_st = (int)tk.x_st, h_st = (int)tk.hsr_st;
  const float* const x_r = tk.x_r; const float* const hs_r = tk.hs_r;
  const int nct = tk.nct, nct_lower = tk.nct_lower;
  int* const done_mine = tk.done + (long)rt * T;
  const int* const done_low = hoisted ? nullptr : tk.done_lower + (long)rt * T;
  const int rec_b = b * T * H + u;                       // gates record: *4, cs record: *1
  const int out_b = (int)(b * tk.out_sb) + u, hsw_b = (int)(b * tk.hs_sb) + u, xtw_b = (int)(b * tk.xt_sb) + u;
  const int out_st = (int)tk.out_st, hs_st = (int)tk.hs_st, xt_st = (int)tk.xt_st;
  float* const gates_p = tk.gates; float* const cs_p = tk.cs; float* const out_p = tk.out;
  float* const hsw_p = tk.hs_w; float* const xtw_p = tk.xt_w;
  const bool drop_on = tk.seed != nullptr;
  const uint32_t seedv = drop_on ? (uint32_t)tk.seed[0] : 0u;
  const float k_st = tk.k_st, k_out = tk.k_out, k_in = tk.k_in;
  const uint32_t r_st = tk.r_st, r_out = tk.r_out, r_in = tk.r_in;
  const int in_W = tk.in_W, in_coff = tk.in_coff;

  // input-part operands run one step ahead of the recurrence: x(t+1) is fetched during step t
  f32x4 xcur[P_XC];
#pragma unroll
  for (int c = 0; c < P_XC; ++c) xcur[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_x = [&](int t, f32x4* dst) {
    const bool v = aok && t < len_a;
    const int xo = xrow + (reverse ? len_a - 1 - t : t) * x_st;
#pragma unroll
    for (int c = 0; c < P_XC; ++c) {
      const int k = (xg0 + c) * 16;
      dst[c] = (c < nxw && v && k + 4 * q < Kx) ? ld4_sc1(x_r + (xo + k)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if (done_low) {
    if (tid == 0) wait_ge2(done_low, nct_lower, nullptr, 0, L.err);
    __syncthreads();
    load_x(0, xcur);
  }

  f32x4 znext = {0.f, 0.f, 0.f, 0.f};
  if (eok && hoisted && 0 < len_b) znext = ld4(gates_p + (long)(rec_b + (reverse ? len_b - 1 : 0) * H) * 4);
#ifdef PERSIST_TIMING
  long tm[6] = {0, 0, 0, 0, 0, 0};
#endif
  for (int t = 0; t < T; ++t) {
#ifdef PERSIST_TIMING
    long last_ = __builtin_amdgcn_s_memtime();
#endif
    f32x4 zpre = znext;                          // prefetched during the previous step; handed over here (see rnn_persist_bwd.hip)
    asm volatile("" : "+v"(zpre));
    // ---- dependencies: previous step of this layer (all column tiles of my row tile); layer below one step ahead ----
    if (tid == 0) {
      const int tl = t + 1 < T ? t + 1 : T - 1;
      wait_ge2(t > 0 ? done_mine + (t - 1) : nullptr, nct, done_low ? done_low + tl : nullptr, nct_lower, L.err);
    }
    lds_barrier();
    TICK(0)

    // ---- recurrent operand rows (sc1: produced by other CUs during this launch), next step's input rows ----
    const bool avalid = aok && t < len_a;
    const int ho_ = hrow + (reverse ? len_a - 1 - t : t) * h_st;
    f32x4 hv[P_HC];
#pragma unroll
    for (int c = 0; c < P_HC; ++c) {
      const int k = (hg0 + c) * 16;
      hv[c] = (c < nhw && avalid && t > 0 && k + 4 * q < H) ? ld4_sc1(hs_r + (ho_ + k)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 2; ++e) acc[nt][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    // input part first: its operands arrived a step ago, so these MFMAs run under the loads just issued
#pragma unroll
    for (int c = 0; c < P_XC; ++c)
      if (c < nxw) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[nt][e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xcur[c][e], wx[c][nt][e], acc[nt][e & 1], 0, 0, 0);
      }
    if (done_low && t + 1 < T) load_x(t + 1, xcur);      // refill in place: consumed a step from now
    // hoisted x.Wx of the NEXT step (written before this launch, cold in HBM).  Issued last: vmcnt retires in
    // order, so a slow load must be younger than the recurrent operands or it would stall their wait.
    if (eok && hoisted && t + 1 < len_b) znext = ld4(gates_p + (long)(rec_b + (reverse ? len_b - 2 - t : t + 1) * H) * 4);
    asm volatile("" ::: "memory");
#ifdef PERSIST_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TICK(1)
#endif
#pragma unroll
    for (int c = 0; c < P_HC; ++c)
      if (c < nhw) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[nt][e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[c][e], wh[c][nt][e], acc[nt][e & 1], 0, 0, 0);
      }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 s4 = acc[nt][0] + acc[nt][1];
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][nt][(lane >> 4) * 4 + r][lane & 15] = s4[r];
    }
    lds_barrier();
    TICK(2)

    // ---- gates, cell clip, length masking, dropout; records + exchanged outputs ----
    if (eok) {
      const bool valid = t < len_b;
      if (valid) {
        const int tau = reverse ? len_b - 1 - t : t;
        const long bt = (long)b * T + tau;
        f32x4 z = bias4 + zpre;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cc = eu * 4 + g;
          z[g] += red[0][cc >> 4][er][cc & 15] + red[1][cc >> 4][er][cc & 15] + red[2][cc >> 4][er][cc & 15] + red[3][cc >> 4][er][cc & 15];
        }
        f32x4 g4;
        g4[0] = p_sigmoid(z[0]); g4[1] = p_tanh(z[1]); g4[2] = p_sigmoid(z[2] + 1.0f); g4[3] = p_sigmoid(z[3]);
        float c = g4[2] * c_state + g4[0] * g4[1];
        c = fminf(1.0f, fmaxf(-1.0f, c));
        const float h = g4[3] * p_tanh(c);
        const uint32_t oidx = (uint32_t)(bt * H + u);
        const float ho = h * p_drop(drop_on, seedv, r_out, oidx, k_out);
        const float hs = h * p_drop(drop_on, seedv, r_st, oidx, k_st);
        st_sc1(out_p + (out_b + tau * out_st), ho);                 // exchanged values first, records after
        if (hsw_p) st_sc1(hsw_p + (hsw_b + tau * hs_st), hs);
        if (xtw_p) st_sc1(xtw_p + (xtw_b + tau * xt_st), ho * p_drop(drop_on, seedv, r_in, (uint32_t)(bt * in_W + in_coff + u), k_in));
        st4(gates_p + (long)(rec_b + tau * H) * 4, g4);
        cs_p[rec_b + tau * H] = c;
        c_state = c;
        h_state = hs;
      } else {                     // past the utterance: zero output at padding position t, state carried in registers
        st_sc1(out_p + (out_b + t * out_st), 0.f);
        if (hsw_p) st_sc1(hsw_p + (hsw_b + t * hs_st), 0.f);
        if (xtw_p) st_sc1(xtw_p + (xtw_b + t * xt_st), 0.f);
      }
    }
    TICK(3)
    // ---- publish: every storing wave drains its write-through stores, then ONE arrival per workgroup ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TICK(4)
    if (tid == 0) __hip_atomic_fetch_add(done_mine + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    TICK(5)
  }
#ifdef PERSIST_TIMING
  if (tid == 0 && local == 0)
    for (int k = 0; k < 6; ++k) L.err[16 + ti * 8 + k] = (int)(tm[k] / T);
#endif
  if (eok) {
    if (tk.h_final) tk.h_final[(long)b * H + u] = h_state;
    if (tk.c_final) tk.c_final[(long)b * H + u] = c_state;
  }
}


// ======================================================================================================
// XCD-local variant.  Batch rows are independent, so the batch is cut into groups of 8 rows and every group
// is bound to ONE XCD: all (stack, layer, column-tile) workgroups of a group share that XCD's L2, the only
// coherence point they need.  Hand-off inside an XCD (tools/xcd_probe.hip: 0.44 us per hop against 1.0-1.1 us
// for the agent-scope forms): plain stores (the line stays in the L2) -> s_waitcnt -> per-workgroup progress
// word; consumers poll the progress words of their row group with ONE coalesced L1-bypassing load and read the
// rows with 16-byte sc1 (L1-bypass, L2-served) loads.
// Placement is never assumed: a workgroup READS its XCC_ID and claims a slot of that XCD's group from a
// per-XCD counter, so every member of a group is physically on the group's XCD whatever the dispatcher did.
// Only liveness depends on the dispatcher handing each XCD its share of the grid; every wait is bounded and a
// miss raises the sticky error word (the host then switches the path off).
// ======================================================================================================
// R: rows per group.  8 for batches up to 64 utterances (all eight XCDs busy); 16 -- a full MFMA row tile, no padding rows -- for
// larger batches: the same instruction count per step then carries twice the rows, so 128 utterances are ONE pass over the chip
// instead of two sequential 64-row slices.
// Q4 (8-row groups, H = 256): the RECURRENT product runs on v_mfma_f32_4x4x1_16B_f32 with A-block broadcast instead of 16x16x4 tiles whose
// second half is padding.  One instruction = 16 blocks of a 4x4x1 outer product; with cbsz = 3 the A operand (4 rows of h at one k) of
// block `abid` of each 8-block half is broadcast to the half, so the two halves multiply rows 0-3 at TWO different k with the 8 x 4 = 32
// gate columns of the workgroup (B: one weight per lane, no duplication); a second instruction takes rows 4-7 with the same weights.
// 8 rows x 32 columns x 64 k of a wave = 64 instructions x 8 cycles = 512 matrix-pipe cycles per step instead of 32 x 32 = 1024, the
// same 32 weight registers, and the h operand is 2 x 16 bytes per lane instead of 4.  (Layer 0 -- 64 columns, no input part -- uses
// cbsz = 4: all 16 blocks share the rows, one k per instruction.)  Layout probed on gfx950: tools/mfma4_probe.hip.
template <int R, bool Q4 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void rnn_persist_fwd_xcd_kernel(const PLaunch L) {
  static_assert(!Q4 || R == 8, "4x4 recurrent product: 8-row groups");
  __shared__ __attribute__((aligned(16))) float red[4][4][R][16];
  __shared__ __attribute__((aligned(16))) float red4[Q4 ? 2048 : 4];   // [4 waves][2 k-halves][8 rows][32 cols] / wide: [4][8][64]
  __shared__ int s_slot;
  const int gx = __builtin_amdgcn_readfirstlane(xcc_id());
  if (threadIdx.x == 0) s_slot = __hip_atomic_fetch_add(L.claim + gx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int slot = __builtin_amdgcn_readfirstlane(s_slot);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int part = L.npart == 2 ? gx >> 2 : 0;      // which stack this XCD works for
  const int g = L.npart == 2 ? gx & 3 : gx;         // row group of that stack
  if (g >= L.ngroups || slot >= (L.npart == 2 ? L.pwg[part] : L.wpx)) return;
  int ti = -1;
#pragma unroll
  for (int i = 0; i < P_MAX_TASKS; ++i)
    if (i < L.ntask && L.task[i].part == part && slot >= L.task[i].wg_begin) ti = i;
  ti = __builtin_amdgcn_readfirstlane(ti);
  if (ti < 0) return;
  const PTask& tk = L.task[ti];
  // The cells with an input product (layers > 0) set the pace of the wavefront; the recurrent-only layer-0 cells have slack (they run
  // ~20 % ahead).  Three workgroups share each SIMD's matrix pipe: the pace-setters take it first.  Round 3: 2.42 -> 2.37 ms at c4; a
  // distinct priority per cell (0..3) did nothing.  (`v_mfma_f32_4x4x1_16B_f32` for the 8-row products -- no padding rows -- was
  // sized and dropped: K = 1 per instruction needs one weight register per k, 128 per wave against 64 with 16x16x4.)
  if (tk.prio == 3) __builtin_amdgcn_s_setprio(3);
  else if (tk.prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (tk.prio == 1) __builtin_amdgcn_s_setprio(1);
  const int ct = slot - tk.wg_begin;
  const int UW = tk.uw, uw_shift = UW == 16 ? 4 : 3;
  const int row0 = L.b0 + g * R, col0 = ct * UW * 4, unit0 = ct * UW;
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
  // Q4: wq[ab] = W[column of this lane][k0 + 4*ab .. + 3] for the recurrent product (wide: 16 A-blocks; else 8, and wq[8 + 2c + nt]
  // holds the input part's fragments, i.e. wa)
  f32x4 wq[Q4 ? 16 : 1];
  const int q4_col = (hoisted && UW == 16) ? lane : (lane & 31);                 // column of the workgroup this lane multiplies (B operand)
  const int q4_kb = hg0 * 16 + ((hoisted && UW == 16) ? 0 : 32 * (lane >> 5));   // first k of this lane's half
  if constexpr (Q4) {
    const int col = col0 + q4_col;
#pragma unroll
    for (int ab = 0; ab < 16; ++ab) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ab < ((hoisted && UW == 16) ? 16 : 8) && col < 4 * H) v = ld4(tk.wt + (long)col * ldw + tk.in + q4_kb + 4 * ab);
      wq[ab] = v;
    }
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = col0 + nt * 16 + i;
#pragma unroll
    for (int c = 0; c < P_HC; ++c) {
      const int k = (hg0 + c) * 16 + 4 * q;
      wb[c][nt] = (!Q4 && c < nhw && col < 4 * H && k < H) ? ld4(tk.wt + (long)col * ldw + tk.in + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < P_XC; ++c) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (wide) {
        const int k = (hg0 + c) * 16 + 4 * q, col2 = col + 32;
        if (!Q4 && c < nhw && col2 < 4 * H && k < H) v = ld4(tk.wt + (long)col2 * ldw + tk.in + k);
      } else {
        const int k = (xg0 + c) * 16 + 4 * q;
        if (c < nxw && col < 4 * H && k < Kx) v = ld4(tk.wt + (long)col * ldw + k);
      }
      wa[c][nt] = v;
    }
  }

  if constexpr (Q4) {
    if (!wide) {
#pragma unroll
      for (int c = 0; c < P_XC; ++c)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) wq[8 + 2 * c + nt] = wa[c][nt];
    }
  }
  // epilogue ownership: thread e (< R * UW <= 256) owns (row er, unit eu) for all steps
  const int er = tid >> uw_shift, eu = tid & (UW - 1);
  const int b = row0 + er, u = unit0 + eu;
  const bool eok = tid < R * UW && b < tk.B && u < H;
  const int len_b = eok ? (tk.len ? tk.len[b] : T) : 0;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (eok && tk.bias) bias4 = ld4(tk.bias + u * 4);
  float c_state = 0.f, h_state = 0.f;

  // PT (8-row groups): the 16 rows of an MFMA tile are the group's 8 rows at TWO time steps - tile rows 0-7 serve even steps, rows 8-15
  // odd steps.  The recurrent operand fills only the rows of the current step's parity (the other half is the padding an 8-row group
  // cannot avoid there), but the input-part product of steps (t, t+1) is ONE full tile every second step: half its MFMAs.
  constexpr bool PT = (R == 8);
  const int sub = PT ? (i >> 3) : 0;               // which step of the pair this lane's tile row belongs to
  const int ab = row0 + (PT ? (i & 7) : i);
  const bool aok = (PT || i < R) && ab < tk.B;
  const int len_a = aok ? (tk.len ? tk.len[ab] : T) : 0;
  const int xrow = (int)(ab * tk.x_sb) + 4 * q, hrow = (int)(ab * tk.hsr_sb) + 4 * q;
  // Q4: this lane's A rows (block lane>>2 holds rows i4, i4 + 4 of the group at k = q4a_k .. + 3)
  const int q4_r0 = row0 + (lane & 3), q4_r1 = q4_r0 + 4;
  const int q4a_k = hg0 * 16 + 4 * (lane >> 2);
  const int q4_len0 = (Q4 && q4_r0 < tk.B) ? (tk.len ? tk.len[q4_r0] : T) : 0, q4_len1 = (Q4 && q4_r1 < tk.B) ? (tk.len ? tk.len[q4_r1] : T) : 0;
  const int q4_h0 = (int)(q4_r0 * tk.hsr_sb) + q4a_k, q4_h1 = (int)(q4_r1 * tk.hsr_sb) + q4a_k;
  const int x_st = (int)tk.x_st, h_st = (int)tk.hsr_st;
  // unconditional raw buffer loads, out-of-range offset = reads zero: exact vmcnt counting keeps the prefetches in flight
  const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(tk.x_r), h_rs = make_rsrc(tk.hs_r), z_rs = make_rsrc(tk.gates);
  const bool has_low = !hoisted;
  // progress words: 32 per (task, group); wave 0 polls own (lanes 0-31) and lower (lanes 32-63) in one load
  int* const my_flag = tk.done + gx * 32 + ct;    // progress words are per (task, XCD)
  const int* poll_ptr = nullptr;
  if (wave == 0) {
    if (lane < 32) { if (lane < tk.nct) poll_ptr = tk.done + gx * 32 + lane; }
    else if (has_low && lane - 32 < tk.nct_lower) poll_ptr = tk.done_lower + gx * 32 + (lane - 32);
  }
  const int rec_b = b * T * H + u;
  const int out_b = (int)(b * tk.out_sb) + u, hsw_b = (int)(b * tk.hs_sb) + u, xtw_b = (int)(b * tk.xt_sb) + u;
  const int out_st = (int)tk.out_st, hs_st = (int)tk.hs_st, xt_st = (int)tk.xt_st;
  float* const gates_p = tk.gates; float* const cs_p = tk.cs; float* const out_p = tk.out;
  float* const hsw_p = tk.hs_w; float* const xtw_p = tk.xt_w;
  const bool drop_on = tk.seed != nullptr;
  const uint32_t seedv = drop_on ? (uint32_t)tk.seed[0] : 0u;
  const float k_st = tk.k_st, k_out = tk.k_out, k_in = tk.k_in;
  const uint32_t r_st = tk.r_st, r_out = tk.r_out, r_in = tk.r_in;
  const int in_W = tk.in_W, in_coff = tk.in_coff;

  // wave 0: wait until own progress >= need_own and lower progress >= need_low (bounded)
  auto wait_progress = [&](int need_own, int need_low) {
    if (wave != 0) return;
    const int need = lane < 32 ? need_own : need_low;
    for (int spins = 0; spins < (1 << 21); ++spins) {
      const int v = poll_ptr ? __hip_atomic_load(poll_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
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
  auto load_x = [&](int t0, f32x4* dst) {          // PT: rows of steps (t0, t0 + 1) side by side
    const int t = t0 + sub;
    const bool v = aok && t < len_a;
    const int xo = xrow + (reverse ? len_a - 1 - t : t) * x_st;
#pragma unroll
    for (int c = 0; c < P_XC; ++c) {
      const int k = (xg0 + c) * 16;
      dst[c] = ldb_sc1(x_rs, (c < nxw && v && k + 4 * q < Kx) ? (xo + k) * 4 : P_OOB);
    }
  };
  if (has_low) {
    wait_progress(0, PT ? (2 < T ? 2 : T) : 1);
    __syncthreads();
    load_x(0, xcur);
  }
  f32x4 accx[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};      // PT: input part of the current pair of steps
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
    // (PT: the pair (t+1, t+2) is requested during the odd step t, so the layer below must then be three steps on)
    wait_progress(t, PT ? ((t & 1) ? (t + 3 < T ? t + 3 : T) : 0) : (t + 2 < T ? t + 2 : T));
    lds_barrier();
    TICK(0)
    const bool avalid = aok && t < len_a && (!PT || sub == (t & 1));
    const int ho_ = hrow + (reverse ? len_a - 1 - t : t) * h_st;
    f32x4 hv[P_HC];
    f32x4 a_lo = {0.f, 0.f, 0.f, 0.f}, a_hi = {0.f, 0.f, 0.f, 0.f};
    if constexpr (Q4) {
      a_lo = ldb_sc1(h_rs, (t > 0 && t < q4_len0) ? (q4_h0 + (reverse ? q4_len0 - 1 - t : t) * h_st) * 4 : P_OOB);
      a_hi = ldb_sc1(h_rs, (t > 0 && t < q4_len1) ? (q4_h1 + (reverse ? q4_len1 - 1 - t : t) * h_st) * 4 : P_OOB);
    } else {
#pragma unroll
      for (int c = 0; c < P_HC; ++c) {
        const int k = (hg0 + c) * 16;
        hv[c] = ldb_sc1(h_rs, (c < nhw && avalid && t > 0 && k + 4 * q < H) ? (ho_ + k) * 4 : P_OOB);
      }
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
      if (!PT || !(t & 1)) {
#pragma unroll
        for (int c = 0; c < P_XC; ++c)
          if (c < nxw) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xcur[c][e], Q4 ? wq[Q4 ? 8 + 2 * c + nt : 0][e] : wa[c][nt][e], acc[nt], 0, 0, 0);
          }
        if (PT) { accx[0] = acc[0]; accx[1] = acc[1]; }
      } else {
        acc[0] = accx[0]; acc[1] = accx[1];      // rows 8-15: the input part of this (odd) step, computed a step ago
      }
      if (!PT) load_x(t + 1 < T ? t + 1 : T, xcur);     // refill in place: consumed a step from now (past the end: nothing is fetched)
      else if (t & 1) load_x(t + 1 < T ? t + 1 : T, xcur);
    }
    // hoisted x.Wx of the NEXT step (cold in HBM).  Issued last: vmcnt retires in order, so a slow load must be
    // younger than the recurrent operands or it would stall their wait.
    znext = ldb4(z_rs, (eok && hoisted && t + 1 < len_b) ? (rec_b + (reverse ? len_b - 2 - t : t + 1) * H) * 16 : P_OOB);
    asm volatile("" ::: "memory");
    if constexpr (Q4) {
      // two accumulator chains (rows 0-3 / 4-7) alternate: an instruction never waits for its predecessor
      f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
#define Q4_STEP(CB, AB)                                                                                   \
      _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                     \
        r0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a_lo[v], wq[AB][v], r0, CB, AB, 0);                       \
        r1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a_hi[v], wq[AB][v], r1, CB, AB, 0);                       \
      }
      if (wide) {
        Q4_STEP(4, 0) Q4_STEP(4, 1) Q4_STEP(4, 2) Q4_STEP(4, 3) Q4_STEP(4, 4) Q4_STEP(4, 5) Q4_STEP(4, 6) Q4_STEP(4, 7)
        Q4_STEP(4, 8) Q4_STEP(4, 9) Q4_STEP(4, 10) Q4_STEP(4, 11) Q4_STEP(4, 12) Q4_STEP(4, 13) Q4_STEP(4, 14) Q4_STEP(4, 15)
      } else {
        Q4_STEP(3, 0) Q4_STEP(3, 1) Q4_STEP(3, 2) Q4_STEP(3, 3) Q4_STEP(3, 4) Q4_STEP(3, 5) Q4_STEP(3, 6) Q4_STEP(3, 7)
      }
#undef Q4_STEP
      // D: lane (block, j) register r = (row r, this lane's column): partial over this lane-half's k
      float* const dst = wide ? red4 + (wave * 8) * 64 + q4_col : red4 + ((wave * 2 + (lane >> 5)) * 8) * 32 + q4_col;
      const int rs = wide ? 64 : 32;
#pragma unroll
      for (int r = 0; r < 4; ++r) { dst[r * rs] = r0[r]; dst[(4 + r) * rs] = r1[r]; }
    } else {
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
    }
    if (PT ? (q >> 1) == (t & 1) : q < R / 4) {  // C rows (lane>>4)*4 + r carry batch rows (PT: the half of the tile of this step's parity)
      const int rq = PT ? (q & 1) : q;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][nt][rq * 4 + r][i] = acc[nt][r];
    }
    lds_barrier();
    TICK(2)
    // The gate epilogue is a dependent chain (LDS partials -> 5 exp / rcp pairs -> stores) on ONE or two waves, and it is on every
    // step's critical path, while the SIMD's other waves (two more workgroups per CU) poll or multiply: top priority for its duration
    // (round 6: 1.97 -> 1.94 ms per launch, profiles/r06_experiments.txt call B; storing the gate / cell RECORDS behind the publish
    // instead -- they have no reader inside the launch -- made it 2.08 ms: the late stores sit in front of the next step's operand loads)
    __builtin_amdgcn_s_setprio(3);
    if (eok) {
      const bool valid = t < len_b;
      if (valid) {
        const int tau = reverse ? len_b - 1 - t : t;
        const long bt = (long)b * T + tau;
        f32x4 z = bias4 + zpre;
        // the four gates of a unit are four consecutive floats of every partial: 16-byte LDS reads (one per partial instead of four;
        // the scalar form also put the 8 rows of a wave on the same banks), each element summed in the same order as before
        {
          const int c0 = eu * 4;
          if constexpr (Q4) {
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
            if (UW == 16) {
#pragma unroll
              for (int w = 0; w < 4; ++w) s4 += *reinterpret_cast<const f32x4*>(&red4[(w * 8 + er) * 64 + c0]);
            } else {
#pragma unroll
              for (int w = 0; w < 8; ++w) s4 += *reinterpret_cast<const f32x4*>(&red4[(w * 8 + er) * 32 + c0]);          // (wave, k-half) pairs in order
              const int ct4 = c0 >> 4, cl = c0 & 15;
              s4 += (*reinterpret_cast<const f32x4*>(&red[0][ct4][er][cl]) + *reinterpret_cast<const f32x4*>(&red[1][ct4][er][cl])) +
                    (*reinterpret_cast<const f32x4*>(&red[2][ct4][er][cl]) + *reinterpret_cast<const f32x4*>(&red[3][ct4][er][cl]));
            }
            z += s4;
          } else {
            const int ct4 = c0 >> 4, cl = c0 & 15;
            z += (*reinterpret_cast<const f32x4*>(&red[0][ct4][er][cl]) + *reinterpret_cast<const f32x4*>(&red[1][ct4][er][cl])) +
                 (*reinterpret_cast<const f32x4*>(&red[2][ct4][er][cl]) + *reinterpret_cast<const f32x4*>(&red[3][ct4][er][cl]));
          }
        }
        f32x4 g4;
        g4[0] = p_sigmoid(z[0]); g4[1] = p_tanh(z[1]); g4[2] = p_sigmoid(z[2] + 1.0f); g4[3] = p_sigmoid(z[3]);
        float c = g4[2] * c_state + g4[0] * g4[1];
        c = fminf(1.0f, fmaxf(-1.0f, c));
        const float h = g4[3] * p_tanh(c);
        const uint32_t oidx = (uint32_t)(bt * H + u);
        const float ho = h * p_drop(drop_on, seedv, r_out, oidx, k_out);
        const float hs = h * p_drop(drop_on, seedv, r_st, oidx, k_st);
        out_p[out_b + tau * out_st] = ho;                              // exchanged values first, records after
        if (hsw_p) hsw_p[hsw_b + tau * hs_st] = hs;
        if (xtw_p) xtw_p[xtw_b + tau * xt_st] = ho * p_drop(drop_on, seedv, r_in, (uint32_t)(bt * in_W + in_coff + u), k_in);
        st4(gates_p + (long)(rec_b + tau * H) * 4, g4);
        cs_p[rec_b + tau * H] = c;
        c_state = c;
        h_state = hs;
      } else {                     // past the utterance: zero output at padding position t, state carried in registers
        out_p[out_b + t * out_st] = 0.f;
        if (hsw_p) hsw_p[hsw_b + t * hs_st] = 0.f;
        if (xtw_p) xtw_p[xtw_b + t * xt_st] = 0.f;
      }
    }
    TICK(3)
    if (tk.prio == 3) __builtin_amdgcn_s_setprio(3);      // back to the task's static priority
    else if (tk.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (tk.prio == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    // publish: stores drained into the XCD's L2, then this workgroup's progress word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TICK(4)
    if (tid == 0) __hip_atomic_store(my_flag, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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

int32_t* g_sync = nullptr;
int g_persist_mode = 3;      // bit 0: agent-scope forward, bit 1: XCD-local forward + fused BPTT, bit 2: split BPTT (opt-in)

int64_t g_sync_ints = 0;

}  // namespace avsr

// sync: int32 device scratch [ints] owned by the caller: word 0 = sticky error flag (zero it when installing),
// words 1.. = arrival counters (zeroed by every persistent call).  NULL disables the persistent path.
extern "C" int avsr_rnn_set_persistent(int32_t* sync, int64_t ints) {
  avsr::g_sync = sync; avsr::g_sync_ints = sync ? ints : 0;
  return AVSR_OK;
}

namespace avsr {

// Fill the task table for one persistent launch.  local = XCD-local variant (8-row groups, progress words),
// else the agent-scope variant (16-row tiles, arrival counters).  Returns AVSR_ERR_UNSUPPORTED if it does not fit.
static double g_fwd_flops = 0.0;      // algorithmic FLOPs of the launch being built (event profiler)
static int build_tasks(const avsr_rnn_stack* st, int n, bool local, int32_t* sync, int64_t sync_ints, PLaunch& L, int* wg_out, long* words_out,
                       bool parts = false) {
  L = PLaunch{};
  g_fwd_flops = 0.0;
  if (parts && (!local || n != 2 || st[0].B > 64)) return AVSR_ERR_UNSUPPORTED;
  int wg = 0;
  long ctr = P_HDR + 8;                 // [P_HDR, P_HDR+8): per-XCD slot claim counters
  for (int i = 0; i < n; ++i) {
    const avsr_rnn_stack& S = st[i];
    if (S.cell != 0 || S.B != st[0].B) return AVSR_ERR_UNSUPPORTED;
    const int nrt = local ? (S.B + 7) / 8 : (S.B + 15) / 16;
    if (parts) { if (i == 1) L.pwg[0] = wg; wg = 0; }       // workgroup slots are counted per stack
    for (int l = 0; l < S.n_layers; ++l) {
      const avsr_rnn_layer& Ly = S.layer[l];
      if (L.ntask >= P_MAX_TASKS) return AVSR_ERR_UNSUPPORTED;
      const int H = Ly.units, in = Ly.in_dim;
      if ((long)S.B * (S.T + 2) * (Ly.ld_out > 4 * H ? Ly.ld_out : 4 * H) >= (1L << 29)) return AVSR_ERR_UNSUPPORTED;   // byte offsets < 2^31
      if (H % 8 || !Ly.out || H > 64 * P_HC || (!Ly.hoisted && in > 64 * P_XC) || in % 4) return AVSR_ERR_UNSUPPORTED;
      if (!Ly.hoisted && l == 0) return AVSR_ERR_UNSUPPORTED;
      if (S.seed && !Ly.hs_seq) return AVSR_ERR_UNSUPPORTED;
      PTask& tk = L.task[L.ntask++];
      tk.wt = Ly.wt; tk.bias = Ly.bias; tk.len = S.len;
      tk.gates = Ly.gates; tk.cs = Ly.cs;
      tk.out = Ly.out + Ly.ld_out + Ly.out_col; tk.out_sb = (long)(S.T + 2) * Ly.ld_out; tk.out_st = Ly.ld_out;
      const float* hseq; long hld;
      if (S.seed) { tk.hs_w = Ly.hs_seq + H; tk.hs_sb = (long)(S.T + 2) * H; tk.hs_st = H; hseq = Ly.hs_seq; hld = H; }
      else { hseq = Ly.out + Ly.out_col; hld = Ly.ld_out; }
      tk.hs_r = hseq + (S.reverse ? 2 * hld : 0); tk.hsr_sb = (long)(S.T + 2) * hld; tk.hsr_st = hld;
      if (!Ly.hoisted) {
        const avsr_rnn_layer& Lo = S.layer[l - 1];
        if (S.seed) { if (!Lo.xt_seq) return AVSR_ERR_UNSUPPORTED; tk.x_r = Lo.xt_seq + Lo.units; tk.x_sb = (long)(S.T + 2) * Lo.units; tk.x_st = Lo.units; }
        else { tk.x_r = Lo.out + Lo.ld_out + Lo.out_col; tk.x_sb = (long)(S.T + 2) * Lo.ld_out; tk.x_st = Lo.ld_out; }
      }
      if (S.seed && Ly.xt_seq) { tk.xt_w = Ly.xt_seq + H; tk.xt_sb = (long)(S.T + 2) * H; tk.xt_st = H; }
      tk.h_final = Ly.h_final; tk.c_final = Ly.c_final;
      tk.B = S.B; tk.T = S.T; tk.H = H; tk.in = in; tk.hoisted = Ly.hoisted; tk.reverse = S.reverse;
      g_fwd_flops += 2.0 * S.B * S.T * ((Ly.hoisted ? 0 : in) + H) * 4.0 * H;
      tk.uw = (local && Ly.hoisted && H % 16 == 0) ? 16 : 8;
      {
        // wave priority of the task's workgroups: the cells with an input product set the pace (default 2, recurrent-only layer 0: 0);
        // AVSR_RNN_PRIO="p0,p1,..." (by task index) overrides for experiments
        tk.prio = Ly.hoisted ? 0 : 2;
        static const char* pe = getenv("AVSR_RNN_PRIO");
        if (pe) { const char* q = pe; for (int k = 0; k < L.ntask - 1 && q; ++k) { q = strchr(q, ','); if (q) ++q; } if (q && *q >= '0' && *q <= '3') tk.prio = *q - '0'; }
      }
      tk.nct = H / tk.uw; tk.nrt = nrt; tk.wg_begin = wg; tk.part = parts ? i : 0;
      if (tk.nct > 32) return AVSR_ERR_UNSUPPORTED;
      wg += local ? tk.nct : nrt * tk.nct;
      tk.done = sync + ctr;
      ctr += local ? 8 * 32 : (long)nrt * S.T;
      if (!Ly.hoisted) { tk.done_lower = L.task[L.ntask - 2].done; tk.nct_lower = L.task[L.ntask - 2].nct; }
      if (S.seed) {
        const uint32_t cid = (uint32_t)(S.cell_id_base + l);
        tk.seed = S.seed; tk.k_st = S.keep_state; tk.k_out = S.keep_out; tk.k_in = 1.0f;
        tk.r_st = cid * 4 + 1; tk.r_out = cid * 4 + 2;
        if (l + 1 < S.n_layers) { tk.k_in = S.keep_in; tk.r_in = (cid + 1) * 4; tk.in_W = H; tk.in_coff = 0; }
        else if (S.consumer_width > 0) { tk.k_in = S.consumer_keep; tk.r_in = (uint32_t)S.consumer_stream; tk.in_W = S.consumer_width; tk.in_coff = 0; }
      }
    }
  }
  // co-residency: every workgroup of the launch must be resident at once (they wait on each other).
  // agent-scope kernel: <= 2 per CU chip-wide; XCD-local kernel: <= 3 per CU of one XCD (its VGPR budget admits 3).
  if (ctr > sync_ints) return AVSR_ERR_UNSUPPORTED;
  if (parts) {
    L.pwg[1] = wg; L.npart = 2;
    if (L.pwg[0] > 96 || L.pwg[1] > 96) return AVSR_ERR_UNSUPPORTED;
    wg = L.pwg[0] > L.pwg[1] ? L.pwg[0] : L.pwg[1];
  }
  if (local ? wg > 96 : wg > 512) return AVSR_ERR_UNSUPPORTED;
  L.err = sync; L.claim = sync + P_HDR; L.wpx = wg; L.ngroups = 0; L.b0 = 0;
  *wg_out = wg; *words_out = ctr;
  return AVSR_OK;
}

}  // namespace avsr

extern "C" int avsr_rnn_set_persistent_mode(int mode) { avsr::g_persist_mode = mode; return AVSR_OK; }

// Returns AVSR_ERR_UNSUPPORTED when the persistent path is disabled or the configuration does not fit it
// (avsr_rnn_fwd then uses one launch per wavefront step).
int avsr_rnn_fwd_persistent(const avsr_rnn_stack* st, int32_t n, void* stream, int dry) {
  using namespace avsr;
  int32_t* sync = g_sync; const int64_t sync_ints = g_sync_ints;
  if (!sync) return AVSR_ERR_UNSUPPORTED;
  static thread_local PLaunch L;
  hipStream_t s = (hipStream_t)stream;
  int wg = 0; long words = 0;
  static const int parts_first = getenv("AVSR_RNN_PARTS") ? atoi(getenv("AVSR_RNN_PARTS")) == 2 : 0;   // experiment: two stacks side by side even when they fit together
  if ((g_persist_mode & 2) && !(parts_first && n == 2 && st[0].B <= 64) && build_tasks(st, n, true, sync, sync_ints, L, &wg, &words) == AVSR_OK) {
    if (dry) return AVSR_OK;
    // 8 XCDs x R rows per launch (R = 8 up to 64 utterances, 16 above); a larger batch runs as consecutive launches over slices
    // (rows are independent)
    const int B = st[0].B;
    static const int rows16 = getenv("AVSR_RNN_ROWS16") ? atoi(getenv("AVSR_RNN_ROWS16")) : 1;
    const int R = (B > 64 && rows16) ? 16 : 8;
    for (int b0 = 0; b0 < B; b0 += 8 * R) {
      const int rows = B - b0 < 8 * R ? B - b0 : 8 * R;
      L.b0 = b0; L.ngroups = (rows + R - 1) / R;
      if (avsr::dev_zero(sync + P_HDR, sizeof(int32_t) * (words - P_HDR), s) != hipSuccess) return AVSR_ERR_HIP;
      {
        ProfScope ps(PROF_RNN_PERSIST_FWD, s, g_fwd_flops * rows / B);
        // 8-row groups at H = 256 everywhere: recurrent product on the 4x4x1 MFMA with A-block broadcast (no padding rows)
        static const int q4_on = getenv("AVSR_RNN_Q4") ? atoi(getenv("AVSR_RNN_Q4")) : 1;
        bool q4 = q4_on && R == 8;
        for (int i = 0; i < L.ntask && q4; ++i) q4 = L.task[i].H == 256 && (L.task[i].hoisted ? L.task[i].uw == 16 : (L.task[i].uw == 8 && L.task[i].in == 256));
        if (R == 16) hipLaunchKernelGGL(rnn_persist_fwd_xcd_kernel<16>, dim3(8 * wg), dim3(256), 0, s, L);
        else if (q4) hipLaunchKernelGGL((rnn_persist_fwd_xcd_kernel<8, true>), dim3(8 * wg), dim3(256), 0, s, L);
        else hipLaunchKernelGGL(rnn_persist_fwd_xcd_kernel<8>, dim3(8 * wg), dim3(256), 0, s, L);
      }
      AVSR_CHECK_LAUNCH();
    }
    return AVSR_OK;
  }
  // two stacks that do not fit one XCD together (the directions of a bidirectional encoder), <= 64 utterances: one launch, each
  // stack on four XCDs in 16-row groups -- the directions run side by side instead of one after the other
  static const int parts_on = getenv("AVSR_RNN_PARTS") ? atoi(getenv("AVSR_RNN_PARTS")) : 1;
  if ((g_persist_mode & 2) && parts_on && build_tasks(st, n, true, sync, sync_ints, L, &wg, &words, true) == AVSR_OK) {
    if (dry) return AVSR_OK;
    const int B = st[0].B;
    L.b0 = 0; L.ngroups = (B + 15) / 16;
    if (avsr::dev_zero(sync + P_HDR, sizeof(int32_t) * (words - P_HDR), s) != hipSuccess) return AVSR_ERR_HIP;
    {
      ProfScope ps(PROF_RNN_PERSIST_FWD, s, g_fwd_flops);
      hipLaunchKernelGGL(rnn_persist_fwd_xcd_kernel<16>, dim3(8 * wg), dim3(256), 0, s, L);
    }
    AVSR_CHECK_LAUNCH();
    return AVSR_OK;
  }
  if ((g_persist_mode & 1) && build_tasks(st, n, false, sync, sync_ints, L, &wg, &words) == AVSR_OK) {
    if (dry) return AVSR_OK;
    if (avsr::dev_zero(sync + P_HDR, sizeof(int32_t) * (words - P_HDR), s) != hipSuccess) return AVSR_ERR_HIP;
    {
      ProfScope ps(PROF_RNN_PERSIST_FWD, s, g_fwd_flops);
      hipLaunchKernelGGL(rnn_persist_fwd_kernel, dim3(wg), dim3(256), 0, s, L);
    }
    AVSR_CHECK_LAUNCH();
    return AVSR_OK;
  }
  return AVSR_ERR_UNSUPPORTED;
}
