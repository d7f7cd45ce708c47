// Fused persistent BPTT of the attention decoder / the AV-Align attentive layer: the backward loop of avsr_attn_rnn_bwd
// (gradient of avsr/decoder_bimodal.py:241-275, decoder_unimodal.py:320-350, encoder.py:265-290 -- AttentionWrapper step = LSTM cell ->
// Luong attention per memory -> attention layer) as ONE launch instead of four dependent launches per decode step.
//
// Same MI355X mapping as the forward kernel (dec_persist.hip): groups of 8 utterances bound to one XCD, 32 workgroups of 512
// threads, the quarter of one utterance's keys resident in VGPRs and its values in LDS, XCD-local hand-offs through the L2 +
// one progress word per workgroup.  Per step l (descending) three phases, i.e. three hand-offs:
//   A  d attention(l) columns [j*AW, (j+1)*AW) and the recurrent part of d h(l) for units [j*UW, (j+1)*UW), all 8 rows:
//      dG(l+1) . [W_att-rows | W_h-rows]^T on v_mfma_f32_16x16x4_f32 (K = 4H split over the 8 waves, the kernel rows resident in
//      registers); + the external gradient, x the input-dropout mask; attention record.  Then, without a hand-off, the
//      workgroup's K-slice of the attention layer's transpose: datt[:, slice] . W_att[:, slice]^T -> partial d cell_out (H) and
//      partial d context (D) of its mechanism, published as split-K partials (16 KB per workgroup).
//   B  attention backward of (row j/4, quarter j%4) of every memory: d context = sum of the 16 partials; d alpha_t = V_t . dctx from
//      the LDS-resident values; d s_t = alpha_t (d alpha_t - ctx . dctx) (alpha from the forward's score / softmax-statistic records);
//      partial d query = g sum_t d s_t K_t from the register-resident keys; score-gradient and context-gradient records.
//   C  LSTM cell backward for (row, unit) of the workgroup's unit slice: d out = external + sum of the 32 attention-layer
//      partials + the 4 x n_mech query partials; DropoutWrapper masks; gate gradients -> record + the rolling dG(l) of phase A.
// Records (dgates, datt, dctx, dscores) keep the layouts the post-loop weight-gradient GEMMs consume.  The rolling buffers
// (dstate) are left exactly as the per-step loop leaves them after step 0, so the caller's d h0 / d c0 tail is unchanged.
// Every wait is bounded (sticky error word of the persistent scratch, as in dec_persist.hip).
#include "dec_persist.h"

#define DB_KPW 8          // 16-deep K chunks of dG per wave (4H <= 1024)
#define DB_PART 512       // floats per (workgroup, row) partial: [d cell_out (H <= 256) | d context (D <= 256)]

namespace avsr {

struct DBMech {
  const float* keys; const float* values; long values_sb, values_st; const int* len; const float* g; const float* watt_t;
  const float* scores; const float* ctx; const float* pstat;       // forward records [B][L][T], [B][L][D], [L][2][nc_rec][B]
  float* dscores; float* dctx; float* pdq;                         // records [B][L][T], [B][L][D]; quarter partials [4][B][H]
  const float* v; const float* bq; const float* wq; const float* pq; float* dpq;   // Bahdanau: v, bias (normed), query layer [H][H], pq / d pq records [B][L][H]
  int T, D, type, nc_rec, ch, lds_off;
};

struct DBLaunch {
  int B, L, H, E, A, KW, n_mech, b0, ngroups;
  int UW, AW, NWA, drop, uwsh, awsh;
  int* err; int* claim; int* flags;
  const float* w;                                                  // cell kernel [E + A + H][4H]
  const float* gates; const float* cs; const float* c0; const int* steplen;
  const float* datt_ext; const float* dcell_ext;
  float* dgates; float* dstate; float* datt;
  float* part;                                                     // [groups][32][8][DB_PART]
  const int* seed; float k_in, k_st, k_out; uint32_t cid4;
  DBMech m[2];
};

// v[l] + v[l ^ 16] and v[l] + v[l ^ 32] with the gfx950 row / half swaps (one VALU issue each; __shfl_xor is an LDS-pipe round trip).
// Inline asm: the builtin form with both operands equal was folded to a + a by the compiler.
__device__ __forceinline__ float xor16_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float xor32_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}

// R: rows per group (8, or 16 = a full MFMA row tile for the attentive layer; see dec_persist.hip)
// BAH: the block's one mechanism is (normed) Bahdanau: the quarter partial of phase B is the gradient of the PROCESSED query
// (d pq[h] = sum_t ds_t v[h] (1 - tanh^2(keys_t[h] + pq[h] + b[h]))), and phase C pulls it through the query layer (d q = d pq . W_q^T,
// an MFMA product over this workgroup's units) before the cell backward; the d pq record feeds the post-loop d W_q GEMM.
template <int KR0, int KR1, int R, bool BAH = false>
__global__ __launch_bounds__(DP_NT) void dec_persist_bwd_kernel(const DBLaunch L) {
  static_assert(!BAH || (R == 8 && KR1 == 0), "Bahdanau: one mechanism, 8-row groups");
  constexpr int WPR = DP_NW / R;                // workgroups per row in the attention phase
  constexpr int RQ = R / 4;                     // lane groups q < RQ hold real rows of a C tile
  constexpr int RED_F = 256 * R;                // [8 waves][2 tiles][R][16]
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const red = lds;                       // [8][2][R][16] phase A partial sums / [8][256] query-gradient partials
  float* const s_datt = lds + RED_F;            // [R][16] this workgroup's d attention columns
  float* const s_dctx = lds + RED_F + 16 * R;   // [2][256] d context of the attention row
  float* const s_ds = s_dctx + 512;             // [2][128] d score of this share's frames
  float* const s_cd = s_ds + 256;               // [2] ctx . dctx, [2] softmax maximum, [2] 1 / softmax denominator
  int* const s_int = reinterpret_cast<int*>(s_cd + 16);   // [0..R) step lengths, [R] slot
  constexpr int MISC = R == 8 ? DP_MISC : DP_MISC16;
  float* const vals = lds + MISC;               // resident value rows of this workgroup's share
  static_assert(RED_F + 16 * R + 512 + 256 + 16 + R + 16 <= MISC, "LDS layout");

  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int g = __builtin_amdgcn_readfirstlane(xcc_id());
  if (tid0 == 0) s_int[R] = __hip_atomic_fetch_add(L.claim + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int j = __builtin_amdgcn_readfirstlane(s_int[R]);
  if (g >= L.ngroups || j >= DP_NW) return;

  const int B = L.B, Ls = L.L, H = L.H, E = L.E, A = L.A;
  const int H4 = 4 * H;
  const int rowbase = L.b0 + g * R;
  const int gg = rowbase / R;                 // group index into the partial workspace
  const bool drop = L.drop != 0;
  const uint32_t seedv = L.seed ? (uint32_t)L.seed[0] : 0u;
  const uint32_t cid4 = L.cid4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const long BH = (long)B * H;
  float* const dgroll = L.dstate;                // [2][B][4H]
  float* const dcbuf = L.dstate + 8 * BH;        // [2][B][H]
  float* const dhcarry = L.dstate + 10 * BH;     // [2][B][H]

  const int UW = L.UW, uwsh = L.uwsh, unit0 = j * UW;
  const int AW = L.AW, awsh = L.awsh, an0 = j * AW;
  const bool has_att = an0 < A;
  const int ma = has_att ? an0 / H : 0;
  const int nl0 = an0 - ma * H;
  const DBMech& Ma = L.m[ma];
  const int Da = Ma.D, ncols = H + Da;
  const int nk = H >> 2;                          // 16-deep chunks of K = 4H
  const int kg0 = (wave * nk) / DP_WV, nkw = ((wave + 1) * nk) / DP_WV - kg0;
  unsigned ukA[DB_KPW];

  // ---------------------------------------------------------------------------------------------------------
  // resident operands
  // ---------------------------------------------------------------------------------------------------------
  f32x4 wp[DB_KPW][2], wb[4], wqb[2];
  f32x4 k0[KR0 > 0 ? KR0 : 1][4], k1[KR1 > 0 ? KR1 : 1][4];
  int n_m[2] = {0, 0}, t0_m[2] = {0, 0};
  const int r_att = j / WPR, cq = j % WPR;
  const int b_att = rowbase + r_att;
  const bool att_row = b_att < B;
  float dc_state = 0.f, dh_carry = 0.f;
  {
    const int tid = tid0, lane = tid & 63, i = lane & 15, q = lane >> 4, s16 = tid & 15, rg = tid >> 4;
    for (int idx = tid; idx < MISC - RED_F; idx += DP_NT) lds[RED_F + idx] = 0.f;
    // (1) rows of the cell kernel behind this workgroup's d attention columns (tile 0) and d h units (tile 1); K split over the waves
#pragma unroll
    for (int cc = 0; cc < DB_KPW; ++cc) {
      const bool in = cc < nkw;
      const int kk = (kg0 + cc) << 4;
      ukA[cc] = in ? (unsigned)(kk * 4) : 0u;
      wp[cc][0] = (in && has_att && i < AW) ? ld4(L.w + (long)(E + an0 + i) * H4 + kk + 4 * q) : zero4;
      wp[cc][1] = (in && i < UW && unit0 + i < H) ? ld4(L.w + (long)(E + A + unit0 + i) * H4 + kk + 4 * q) : zero4;
    }
    // (2) this workgroup's K-slice of the attention layer's transpose: rows [nl0, nl0 + AW) of W_att^T, column tiles wave*4 .. wave*4+3
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = (wave * 4 + t) * 16 + i;
      f32x4 v = zero4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (has_att && n < ncols && 4 * q + e < AW) v[e] = Ma.watt_t[(long)(nl0 + 4 * q + e) * ncols + n];
      wb[t] = v;
    }
    // (2b) Bahdanau: rows [unit0, unit0 + UW) of the query layer (d q[n] = sum_k d pq[k] W_q[n][k]); K = H split over the waves
    wqb[0] = zero4; wqb[1] = zero4;
    if constexpr (BAH) {
      const int nqc = H >> 4, qg0 = (wave * nqc) / DP_WV, nqw = ((wave + 1) * nqc) / DP_WV - qg0;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
        wqb[cc] = (cc < nqw && i < UW && unit0 + i < H) ? ld4(L.m[0].wq + (long)(unit0 + i) * H + ((qg0 + cc) << 4) + 4 * q) : zero4;
    }
    // (3) attention memories of row r_att, quarter cq: keys -> registers (16 lanes per frame), values -> LDS
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (m >= L.n_mech) continue;
      const DBMech& M = L.m[m];
      const int len = att_row ? min(M.len ? M.len[b_att] : M.T, M.T) : 0;
      const int t0 = cq * M.ch;
      const int n = max(0, min(M.ch, len - t0));
      n_m[m] = n; t0_m[m] = t0;
      const float* kb = M.keys + ((long)b_att * M.T + t0) * H;
      if (m == 0) {
#pragma unroll
        for (int u = 0; u < KR0; ++u)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int fr = rg + 32 * u, k = 4 * s16 + 64 * jj;
            k0[u][jj] = (fr < n && k < H) ? ld4(kb + (long)fr * H + k) : zero4;
          }
      } else {
#pragma unroll
        for (int u = 0; u < KR1; ++u)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int fr = rg + 32 * u, k = 4 * s16 + 64 * jj;
            k1[u][jj] = (fr < n && k < H) ? ld4(kb + (long)fr * H + k) : zero4;
          }
      }
      const int d4 = M.D >> 2;
      const float* vb = M.values + (long)b_att * M.values_sb + (long)t0 * M.values_st;
      for (int idx = tid; idx < n * d4; idx += DP_NT) {
        const int fr = idx / d4, c4 = idx - fr * d4;
        st4(vals + M.lds_off + fr * M.D + 4 * c4, ld4(vb + (long)fr * M.values_st + 4 * c4));
      }
    }
    // (4) gradient state of (row er, unit eu); step lengths of the group's rows
    const int er = tid >> uwsh, eu = tid & (UW - 1), eb = rowbase + er, eun = unit0 + eu;
    if (tid < R * UW && eb < B && eun < H) {
      dc_state = dcbuf[(long)(Ls & 1) * BH + (long)eb * H + eun];
      dh_carry = dhcarry[(long)(Ls & 1) * BH + (long)eb * H + eun];
    }
    __syncthreads();
    if (tid < R) s_int[tid] = (rowbase + tid < B) ? L.steplen[rowbase + tid] : 0;
    __syncthreads();
  }
  const __amdgpu_buffer_rsrc_t dg_rs = make_rsrc(dgroll), part_rs = make_rsrc(L.part), gates_rs = make_rsrc(L.gates);
  const __amdgpu_buffer_rsrc_t ctx_rs = make_rsrc(L.m[wave < 2 ? (wave < L.n_mech ? wave : 0) : 0].ctx);
  const __amdgpu_buffer_rsrc_t pdq0_rs = make_rsrc(L.m[0].pdq), pdq1_rs = make_rsrc(L.m[L.n_mech > 1 ? 1 : 0].pdq);
  const __amdgpu_buffer_rsrc_t sc0_rs = make_rsrc(L.m[0].scores), sc1_rs = make_rsrc(L.m[L.n_mech > 1 ? 1 : 0].scores);
  const __amdgpu_buffer_rsrc_t ps0_rs = make_rsrc(L.m[0].pstat), ps1_rs = make_rsrc(L.m[L.n_mech > 1 ? 1 : 0].pstat);
  const __amdgpu_buffer_rsrc_t cs_rs = make_rsrc(L.cs), c0_rs = make_rsrc(L.c0), dce_rs = make_rsrc(L.dcell_ext), dae_rs = make_rsrc(L.datt_ext);

  // softmax temperature of scaled Luong (attention.py:43-54): read once, not inside the step loop
  float gscv[2] = {1.f, 1.f};
#pragma unroll
  for (int m = 0; m < 2; ++m)
    if (m < L.n_mech && L.m[m].type == ATT_SCALED_LUONG) gscv[m] = L.m[m].g[0];
  int* const flag_base = L.flags + g * 3 * 32;
  auto wait_all = [&](int phase, int need) {
    if (wave == 0) {
      const int* fp = flag_base + phase * 32 + (threadIdx.x & 31);
      bool ok = false;
      for (int spins = 0; spins < (1 << 21); ++spins) {
        const int v = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v >= need)) { ok = true; break; }
        if ((spins & 1023) == 1023 && __hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = true; break; }
      }
      if (!ok && threadIdx.x == 0) __hip_atomic_store(L.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    lds_barrier();
  };
  auto publish = [&](int phase, int value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag_base + phase * 32 + j, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
#ifdef DP_TIMING
  long tm[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long last_ = __builtin_amdgcn_s_memtime();
#define BTICK(k) { const long now_ = __builtin_amdgcn_s_memtime(); tm[k] += now_ - last_; last_ = now_; }
#else
#define BTICK(k)
#endif

  for (int l = Ls - 1; l >= 0; --l) {
    const int epoch = Ls - l;
    int tid = tid0;
    asm volatile("" : "+v"(tid));                 // opaque: per-thread indices below are recomputed, not kept across steps
    const int lane = tid & 63, i = lane & 15, q = lane >> 4;
    const int ab = rowbase + i;
    const bool aok = i < R && ab < B;
    const int er = tid >> uwsh, eu = tid & (UW - 1), eb = rowbase + er, eun = unit0 + eu;
    const bool eok = tid < R * UW && eb < B && eun < H;
    float zA = 0.f;
    const int s16 = tid & 15, rg = tid >> 4;
    const bool c_valid = eok && l < s_int[er & (R - 1)];
    const long c_bt = (long)eb * Ls + l;
    f32x4 c_g4 = zero4;
    float c_c = 0.f, c_prev = 0.f, c_dext = 0.f;
    float raw0[KR0 > 0 ? KR0 : 1], raw1[KR1 > 0 ? KR1 : 1], pm_pf = 0.f, pl_pf = 0.f;
    f32x4 ctx_pf = zero4;
    float a_ext = 0.f;                            // external gradient of this thread's attention column: requested before the wait
    {
      const int ar = tid >> awsh, ac = tid & (AW - 1), arb = rowbase + ar;
      a_ext = ldb1(dae_rs, (tid < R * AW && has_att && L.datt_ext && arb < B) ? (int)((((long)arb * Ls + l) * A + an0 + ac) * 4) : P_OOB);
    }
    // =====================================================================================================
    // A: d attention(l) columns and the recurrent d h(l) of this workgroup from dG(l+1); attention-layer transpose, split-K
    // =====================================================================================================
    if (epoch > 1) wait_all(2, epoch - 1);
    BTICK(0)
    {
      const unsigned dg_o = aok ? (unsigned)(((long)((l + 1) & 1) * B + ab) * H4) * 4u + (unsigned)(q * 16) : (unsigned)P_OOB;
      f32x4 dv[DB_KPW];
#pragma unroll
      for (int cc = 0; cc < DB_KPW; ++cc) dv[cc] = ldb_sc1(dg_rs, (int)(dg_o + ukA[cc]));
      f32x4 acc[2] = {zero4, zero4}, accb[2] = {zero4, zero4};
#pragma unroll
      for (int cc = 0; cc < DB_KPW; cc += 2)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[cc][e], wp[cc][nt][e], acc[nt], 0, 0, 0);
            accb[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[cc + 1][e], wp[cc + 1][nt][e], accb[nt], 0, 0, 0);
          }
      acc[0] += accb[0]; acc[1] += accb[1];
      if (q < RQ) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((wave * 2 + nt) * R + q * 4 + r) * 16 + i] = acc[nt][r];
      }
      // forward records phase B consumes (raw scores of this lane's frames, softmax statistics, the context): requested here, so
      // their latency runs under the rest of this phase and the hand-off
      {
        const bool lead = s16 == 0 && att_row;
        const unsigned so0 = (unsigned)((((long)b_att * Ls + l) * L.m[0].T + t0_m[0]) * 4), so1 = (unsigned)((((long)b_att * Ls + l) * L.m[1].T + t0_m[1]) * 4);
#pragma unroll
        for (int u = 0; u < KR0; ++u) raw0[u] = ldb1(sc0_rs, (lead && rg + 32 * u < n_m[0]) ? (int)(so0 + (unsigned)((rg + 32 * u) * 4)) : P_OOB);
#pragma unroll
        for (int u = 0; u < KR1; ++u) raw1[u] = ldb1(sc1_rs, (lead && rg + 32 * u < n_m[1]) ? (int)(so1 + (unsigned)((rg + 32 * u) * 4)) : P_OOB);
        if (wave < 2 && wave < L.n_mech) {
          const int D = L.m[wave].D;
          ctx_pf = ldb4(ctx_rs, (att_row && 4 * lane < D) ? (int)((((long)b_att * Ls + l) * D + 4 * lane) * 4) : P_OOB);
          // softmax statistics of this row and step: lane c takes recorded chunk c (the fused forward records them merged in chunk 0,
          // the per-step path one partial per chunk)
          const int nc = L.m[wave].nc_rec;
          const bool in = lane < nc && att_row;
          const unsigned o = (unsigned)((((long)(2 * l) * nc + lane) * B + b_att) * 4);
          pm_pf = ldb1(wave ? ps1_rs : ps0_rs, in ? (int)o : P_OOB);
          pl_pf = ldb1(wave ? ps1_rs : ps0_rs, in ? (int)(o + (unsigned)((long)nc * B * 4)) : P_OOB);
        }
      }
      lds_barrier();
      BTICK(1)
      if (tid < R * AW && has_att) {
        const int ar = tid >> awsh, ac = tid & (AW - 1), arb = rowbase + ar;
        float a = 0.f;
        if (arb < B && l < s_int[ar]) {
          const int o = ar * 16 + ac;
          float z = 0.f;
#pragma unroll
          for (int w = 0; w < DP_WV; ++w) z += red[(w * 2) * 16 * R + o];
          // the product is the gradient of step l+1's DROPPED attention input (AttentionWrapper feeds [x | attention] through the
          // DropoutWrapper's input mask, cells.py:46-54)
          z *= p_drop(drop, seedv, cid4, (uint32_t)(((long)arb * Ls + l + 1) * (E + A) + E + an0 + ac), L.k_in);
          a = z + a_ext;
        }
        if (arb < B) L.datt[((long)arb * Ls + l) * A + an0 + ac] = a;
        s_datt[ar * 16 + ac] = a;
      }
      if (tid < R * UW) {
        const int o = er * 16 + eu;
#pragma unroll
        for (int w = 0; w < DP_WV; ++w) zA += red[(w * 2 + 1) * 16 * R + o];
      }
      lds_barrier();
      if (has_att) {
        const f32x4 a4 = ld4(s_datt + (i & (R - 1)) * 16 + 4 * q);
        float* const prow = L.part + (((long)gg * DP_NW + j) * R + q * 4) * DB_PART + i;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int n0 = (wave * 4 + t) * 16;
          if (n0 < ncols) {                          // wave-uniform
            f32x4 c = zero4;
#pragma unroll
            for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], wb[t][e], c, 0, 0, 0);
            if (q < RQ) {
#pragma unroll
              for (int r = 0; r < 4; ++r) prow[(long)r * DB_PART + n0] = c[r];
            }
          }
        }
      }
      BTICK(2)
      publish(0, epoch);
      BTICK(3)
    }
    // =====================================================================================================
    // B: attention backward of (row r_att, quarter cq)  (gradient of attention.py:25-72 Luong / scaled Luong scores, the masked
    //    softmax and the context of contrib.seq2seq _compute_attention)
    // =====================================================================================================
    wait_all(0, epoch);
    BTICK(4)
    {
      const bool vrow = att_row && l < s_int[r_att];
      if (wave < 2 && wave < L.n_mech) {
        const DBMech& M = L.m[wave];
        const int D = M.D, c4 = lane;
        const bool cok = att_row && 4 * c4 < D;
        const int nwm = H >> awsh, w0 = wave * nwm;
        const unsigned base = cok ? (unsigned)((((long)gg * DP_NW + w0) * R + r_att) * DB_PART + H + 4 * c4) * 4u : (unsigned)P_OOB;
        f32x4 acc = zero4;
        for (int w = 0; w < nwm; w += 8) {
          f32x4 x[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) x[u] = ldb_sc1(part_rs, (w + u < nwm) ? (int)(base + (unsigned)((w + u) * R * DB_PART * 4)) : P_OOB);
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += x[u];
        }
        st4(s_dctx + wave * 256 + 4 * c4, acc);
        if (cq == 0 && cok) st4(M.dctx + ((long)b_att * Ls + l) * D + 4 * c4, acc);
        const float cd = wave64_sum(dot4(ctx_pf, acc));
        const bool in = lane < M.nc_rec && att_row;
        const float Mx = wave64_max(in ? pm_pf : -INFINITY);
        const float Lsum = wave64_sum((in && pm_pf != -INFINITY) ? __expf(pm_pf - Mx) * pl_pf : 0.f);
        if (lane == 0) { s_cd[wave] = cd; s_cd[2 + wave] = Mx; s_cd[4 + wave] = Lsum > 0.f ? 1.f / Lsum : 0.f; }
      }
      lds_barrier();
      BTICK(5)
      // records of phase C (gates, cell states): requested now, consumed after the next hand-off
      c_g4 = ldb4(gates_rs, c_valid ? (int)((c_bt * H + eun) * 16) : P_OOB);
      c_c = ldb1(cs_rs, c_valid ? (int)((c_bt * H + eun) * 4) : P_OOB);
      c_prev = (l == 0) ? ldb1(c0_rs, (c_valid && L.c0) ? (int)(((long)eb * H + eun) * 4) : P_OOB)
                        : ldb1(cs_rs, c_valid ? (int)(((c_bt - 1) * H + eun) * 4) : P_OOB);
      c_dext = ldb1(dce_rs, (c_valid && L.dcell_ext) ? (int)((c_bt * H + eun) * 4) : P_OOB);
      // d alpha_t = V_t . dctx, d s_t = alpha_t (d alpha_t - ctx . dctx) for the frames of this quarter, both memories
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (m >= L.n_mech) continue;
        const DBMech& M = L.m[m];
        const int D = M.D, n = n_m[m];
        const float gsc = gscv[m];
        // softmax statistics of this row and step from the forward's records (merged over the recorded chunks)
        const float Mx = s_cd[2 + m], invL = s_cd[4 + m];
        const float cd = s_cd[m];
        f32x4 d4[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) d4[jj] = ld4(s_dctx + m * 256 + 4 * s16 + 64 * jj);
        float* dsr = M.dscores + ((long)b_att * Ls + l) * M.T + t0_m[m];
#pragma unroll
        for (int u = 0; u < (m == 0 ? KR0 : KR1); ++u) {
          const int fr = rg + 32 * u;
          float a = 0.f;
          if (fr < n) {
            const float* vp = vals + M.lds_off + fr * D + 4 * s16;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              if (4 * s16 + 64 * jj < D) a += dot4(ld4(vp + 64 * jj), d4[jj]);
          }
          a = row16_sum(a);
          if (s16 == 0) {
            const float raw = m == 0 ? raw0[m == 0 ? u : 0] : raw1[m == 0 ? 0 : u];
            float dsv = 0.f;
            if (fr < n && vrow) dsv = __expf(raw * gsc - Mx) * invL * (a - cd);
            s_ds[m * 128 + fr] = dsv;
            if (att_row && fr < M.ch && t0_m[m] + fr < M.T) dsr[fr] = dsv;
          }
        }
      }
      lds_barrier();
      BTICK(11)
      // partial d query over this quarter from the register-resident keys, one memory at a time through the reduction buffer
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (m >= L.n_mech) continue;
        const DBMech& M = L.m[m];
        f32x4 pa[4] = {zero4, zero4, zero4, zero4};
        if constexpr (BAH) if (m == 0) {
          f32x4 v4[4], pb4[4];
          const float* pqr = M.pq + ((long)b_att * Ls + l) * H;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const bool in = att_row && 4 * s16 + 64 * jj < H;
            v4[jj] = in ? ld4(M.v + 4 * s16 + 64 * jj) : zero4;
            pb4[jj] = in ? ld4(pqr + 4 * s16 + 64 * jj) : zero4;
            if (in && M.bq) pb4[jj] += ld4(M.bq + 4 * s16 + 64 * jj);
          }
#pragma unroll
          for (int u = 0; u < KR0; ++u) {
            const float dsf = s_ds[rg + 32 * u];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float th = p_tanh(k0[u][jj][e] + pb4[jj][e]);
                pa[jj][e] += dsf * v4[jj][e] * (1.f - th * th);
              }
          }
        }
        if (!BAH && m == 0) {
#pragma unroll
          for (int u = 0; u < KR0; ++u) {
            const float dsf = s_ds[rg + 32 * u];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) pa[jj] += dsf * k0[u][jj];
          }
        } else if (m != 0) {
#pragma unroll
          for (int u = 0; u < KR1; ++u) {
            const float dsf = s_ds[128 + rg + 32 * u];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) pa[jj] += dsf * k1[u][jj];
          }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int e = 0; e < 4; ++e) pa[jj][e] = xor32_sum(xor16_sum(pa[jj][e]));
        if (m > 0) lds_barrier();                       // the previous memory's partials have been summed
        if (q == 0) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) st4(red + wave * 256 + 4 * s16 + 64 * jj, pa[jj]);
        }
        lds_barrier();
        if (tid < 64) {
          f32x4 sv = ld4(red + 4 * tid);
#pragma unroll
          for (int w = 1; w < DP_WV; ++w) sv += ld4(red + w * 256 + 4 * tid);
          if (att_row && 4 * tid < H) st4(M.pdq + (((long)cq * B + b_att) * H + 4 * tid), gscv[m] * sv);
        }
      }
      lds_barrier();
      BTICK(6)
      publish(1, epoch);
      BTICK(7)
    }
    // =====================================================================================================
    // C: LSTM cell backward of (row er, unit eun)  (cells.py:14-18 LSTMCell clip 1.0, forget bias 1.0; DropoutWrapper cells.py:46-54)
    // =====================================================================================================
    wait_all(1, epoch);
    BTICK(8)
    {
      const bool valid = c_valid;
      const long bt = c_bt;
      float pq[DP_NW / 2], pd[8];
      const unsigned po = eok ? (unsigned)((((long)gg * DP_NW) * R + er) * DB_PART + eun) * 4u : (unsigned)P_OOB;
#pragma unroll
      for (int w = 0; w < DP_NW / 2; ++w) pq[w] = ld1_sc1(part_rs, (w < L.NWA) ? (int)(po + (unsigned)(w * R * DB_PART * 4)) : P_OOB);
      const unsigned qo = eok ? (unsigned)((long)eb * H + eun) * 4u : (unsigned)P_OOB;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        pd[c] = ld1_sc1(pdq0_rs, (c < WPR) ? (int)(qo + (unsigned)((long)c * BH * 4)) : P_OOB);
        pd[4 + c] = ld1_sc1(pdq1_rs, (L.n_mech > 1 && c < WPR) ? (int)(qo + (unsigned)((long)c * BH * 4)) : P_OOB);
      }
      const float c = c_c, cprev = c_prev, dext = c_dext;
      const f32x4 g4 = c_g4;
      float dq_bah = 0.f;
      if constexpr (BAH) {
        // d pq of the 8 rows (sum of the four quarter partials) through the query layer: d q[r][unit] = sum_k d pq[r][k] W_q[unit][k]
        const int nqc = H >> 4, qg0 = (wave * nqc) / DP_WV, nqw = ((wave + 1) * nqc) / DP_WV - qg0;
        f32x4 a4[2] = {zero4, zero4};
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const unsigned ko = aok ? (unsigned)(((long)ab * H + ((qg0 + cc) << 4) + 4 * q) * 4) : (unsigned)P_OOB;
#pragma unroll
          for (int cqq = 0; cqq < 4; ++cqq)
            a4[cc] += ldb_sc1(pdq0_rs, (cc < nqw && cqq < WPR) ? (int)(ko + (unsigned)((long)cqq * BH * 4)) : P_OOB);
        }
        f32x4 accq = zero4;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
          for (int e = 0; e < 4; ++e) accq = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[cc][e], wqb[cc][e], accq, 0, 0, 0);
        if (q < RQ) {
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(wave * R + q * 4 + r) * 16 + i] = accq[r];
        }
        lds_barrier();
        if (tid < R * UW) {
          const int o = er * 16 + eu;
#pragma unroll
          for (int w = 0; w < DP_WV; ++w) dq_bah += red[w * 16 * R + o];
        }
        lds_barrier();
        // the d pq record of this step (post-loop d W_q GEMM, attention.py query_layer)
        if (eok) L.m[0].dpq[bt * H + eun] = valid ? ((pd[0] + pd[1]) + (pd[2] + pd[3])) : 0.f;
      }
      float dout = 0.f;
#pragma unroll
      for (int w = 0; w < DP_NW / 2; ++w) dout += pq[w];
      if (L.NWA > DP_NW / 2) {                       // wave-uniform: second half of the attention-layer partials
#pragma unroll
        for (int w = 0; w < DP_NW / 2; ++w) pq[w] = ld1_sc1(part_rs, (w + DP_NW / 2 < L.NWA) ? (int)(po + (unsigned)((w + DP_NW / 2) * R * DB_PART * 4)) : P_OOB);
#pragma unroll
        for (int w = 0; w < DP_NW / 2; ++w) dout += pq[w];
      }
      if (BAH) dout += dq_bah;                          // the query gradient arrives through the query layer
      else dout += ((pd[0] + pd[1]) + (pd[2] + pd[3])) + ((pd[4] + pd[5]) + (pd[6] + pd[7]));
      dout += dext;
      f32x4 dg = zero4;
      if (valid) {
        const uint32_t oidx = (uint32_t)(bt * H + eun);
        const float dh = dout * p_drop(drop, seedv, cid4 + 2, oidx, L.k_out) + (zA + dh_carry) * p_drop(drop, seedv, cid4 + 1, oidx, L.k_st);
        const float tc = p_tanh(c);
        float dc = dh * g4[3] * (1.f - tc * tc) + dc_state;
        if (!(fabsf(c) < 1.0f)) dc = 0.f;          // cell_clip = 1.0: no gradient through a clipped cell
        dg[3] = dh * tc * g4[3] * (1.f - g4[3]);
        dg[0] = dc * g4[1] * g4[0] * (1.f - g4[0]);
        dg[1] = dc * g4[0] * (1.f - g4[1] * g4[1]);
        dg[2] = dc * cprev * g4[2] * (1.f - g4[2]);
        dc_state = dc * g4[2];
        dh_carry = 0.f;
      }
      if (eok) {
        st4(L.dgates + (bt * H + eun) * 4, dg);
        st4(dgroll + ((long)(l & 1) * B + eb) * H4 + eun * 4, dg);
      }
      BTICK(9)
      publish(2, epoch);
      BTICK(10)
    }
  }
#ifdef DP_TIMING
  if (tid0 == 0 && j == 0 && g == 0)
    for (int k = 0; k < 12; ++k) L.err[(R == 16 ? 200 : 16) + k] = (int)(tm[k] / Ls);      // 16-row groups = the AV-Align attentive layer: its own words
#endif
  {
    const int tid = tid0;
    const int er = tid >> uwsh, eu = tid & (UW - 1), eb = rowbase + er, eun = unit0 + eu;
    if (tid < R * UW && eb < B && eun < H) {
      dcbuf[(long)eb * H + eun] = dc_state;
      dhcarry[(long)eb * H + eun] = dh_carry;
    }
  }
}

static const void* db_kernel(int variant) {
  if (variant == 3) return (const void*)dec_persist_bwd_kernel<2, 0, 16>;
  if (variant == 4) return (const void*)dec_persist_bwd_kernel<4, 0, 8, true>;
  return variant == 0 ? (const void*)dec_persist_bwd_kernel<4, 0, 8> : variant == 1 ? (const void*)dec_persist_bwd_kernel<1, 4, 8>
                                                                                    : (const void*)dec_persist_bwd_kernel<4, 1, 8>;
}

}  // namespace avsr

int64_t avsr_dec_persist_fwd_ws_floats(int32_t B, int32_t n_mech, int32_t Dmax);

// floats the backward kernel needs behind the forward region of the fused workspace
int64_t avsr_dec_persist_bwd_ws_floats(int32_t B, int32_t n_mech) {
  const int64_t rows = ((B + 15) / 16) * 16;       // whole groups of 8 or 16 rows
  return rows * DP_NW * DB_PART + (int64_t)n_mech * 4 * B * 256;
}

// The whole backward loop of avsr_attn_rnn_bwd as one persistent launch per 64-row slice.  The caller has zeroed dstate and copied
// the final-state gradients into it exactly as for the per-step path, and runs its d h0 / d c0 tail afterwards.
int avsr_dec_persist_bwd(const avsr_attn_rnn* dp, void* stream) {
  using namespace avsr;
  static thread_local DPLaunch F;
  static thread_local DBLaunch L;
  int variant = 0; size_t lds = 0;
  const avsr_attn_rnn& d = *dp;
  if (g_dec_fused == 2) return AVSR_ERR_UNSUPPORTED;
  int rc = dp_plan(d, F, &variant, &lds);
  if (rc) return rc;
  {
    // this kernel's OWN LDS size (MISC + the resident values): the forward's figure differs for 33..64-symbol vocabularies, whose layout
    // ahead of the values is smaller since round 5 -- launched with the forward's size, the last 256 bytes of the values lay outside the
    // allocation (caught by tests/test_gpu_beam.py: the memory-kernel gradient of a phoneme model)
    long vals = 0;
    for (int m = 0; m < d.n_mech; ++m) vals += (long)F.m[m].ch * d.mech[m].D;
    lds = sizeof(float) * (size_t)((F.R == 16 ? DP_MISC16 : DP_MISC) + vals);
    if (lds > DP_LDS_BYTES) return AVSR_ERR_UNSUPPORTED;
  }
  if (F.bah && (!d.mech[0].wq || !d.mech[0].dpq || !d.mech[0].pq || !d.mech[0].v)) return AVSR_ERR_ARG;
  if (!d.w || !d.dgates || !d.dstate || !d.datt) return AVSR_ERR_ARG;
  if (avsr_dec_persist_fwd_ws_floats(d.B, d.n_mech, 256) + avsr_dec_persist_bwd_ws_floats(d.B, d.n_mech) > d.fused_ws_floats) return AVSR_ERR_UNSUPPORTED;
  if ((long)d.B * d.L * d.H * 16 >= (1L << 31) || (long)((d.B + 15) / 16) * 16 * DP_NW * DB_PART * 4 >= (1L << 31) ||
      (long)4 * d.B * d.H * 4 >= (1L << 31))
    return AVSR_ERR_UNSUPPORTED;
  L = DBLaunch{};
  L.B = d.B; L.L = d.L; L.H = d.H; L.E = d.E; L.A = F.A; L.KW = F.KW; L.n_mech = d.n_mech;
  L.UW = F.UW; L.AW = F.AW; L.NWA = F.NWA; L.uwsh = F.uwsh; L.awsh = F.awsh;
  L.drop = (d.seed && (d.keep_in < 1.f || d.keep_state < 1.f || d.keep_out < 1.f)) ? 1 : 0;
  L.w = d.w; L.gates = d.gates; L.cs = d.cs; L.c0 = d.c0; L.steplen = d.steplen;
  L.datt_ext = d.datt_ext; L.dcell_ext = d.dcell_ext; L.dgates = d.dgates; L.dstate = d.dstate; L.datt = d.datt;
  L.seed = d.seed; L.k_in = d.keep_in; L.k_st = d.keep_state; L.k_out = d.keep_out; L.cid4 = (uint32_t)d.cell_id * 4;
  float* ws = d.fused_ws + avsr_dec_persist_fwd_ws_floats(d.B, d.n_mech, 256);
  L.part = ws; ws += (long)((d.B + 15) / 16) * 16 * DP_NW * DB_PART;
  for (int m = 0; m < d.n_mech; ++m) {
    const avsr_attn_mech& M = d.mech[m];
    if (!M.dscores || !M.dctx || !M.scores || !M.ctx || !M.pstat || !M.watt_t) return AVSR_ERR_ARG;
    DBMech& X = L.m[m];
    X.keys = M.keys; X.values = M.values; X.values_sb = M.values_sb; X.values_st = M.values_st; X.len = M.len; X.g = M.g; X.watt_t = M.watt_t;
    X.scores = M.scores; X.ctx = M.ctx; X.pstat = M.pstat; X.dscores = M.dscores; X.dctx = M.dctx;
    X.v = M.v; X.bq = (M.type == ATT_NORMED_BAHDANAU) ? M.bq : nullptr; X.wq = M.wq; X.pq = M.pq; X.dpq = M.dpq;
    X.pdq = ws; ws += 4L * d.B * 256;
    X.T = M.T; X.D = M.D; X.type = M.type; X.nc_rec = F.m[m].nc_rec; X.ch = F.m[m].ch; X.lds_off = F.m[m].lds_off;
    if ((long)d.B * d.L * M.D * 4 >= (1L << 31) || (long)d.B * d.L * M.T * 4 >= (1L << 31) || X.nc_rec > 64) return AVSR_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  int32_t* sync = g_sync;
  const long words = P_HDR + 8 + 8 * 3 * 32;
  if (words > g_sync_ints) return AVSR_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    for (int v = 0; v < 5; ++v)
      if (hipFuncSetAttribute(db_kernel(v), hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS_BYTES) != hipSuccess) return AVSR_ERR_HIP;
    attr_set = true;
  }
  L.err = sync; L.claim = sync + P_HDR; L.flags = sync + P_HDR + 8;
  const int slice = 8 * F.R;
  for (int b0 = 0; b0 < d.B; b0 += slice) {
    L.b0 = b0; L.ngroups = ((d.B - b0 < slice ? d.B - b0 : slice) + F.R - 1) / F.R;
    if (avsr::dev_zero(sync + P_HDR, sizeof(int32_t) * (words - P_HDR), s) != hipSuccess) return AVSR_ERR_HIP;
    {
      ProfScope ps(dp->prof_tag == 1 ? PROF_ALIGN_PERSIST_BWD : PROF_DEC_PERSIST_BWD, s);
      void* args[] = {(void*)&L};
      if (hipLaunchKernel(db_kernel(variant), dim3(8 * DP_NW), dim3(DP_NT), args, lds, s) != hipSuccess) return AVSR_ERR_HIP;
    }
    AVSR_CHECK_LAUNCH();
  }
  return AVSR_OK;
}
