// Fused persistent decode kernel: the whole attention-decoder loop (avsr/decoder_bimodal.py:241-275,
// avsr/decoder_unimodal.py:320-350: AttentionWrapper step = LSTM cell -> Luong score / masked softmax / context per
// memory -> attention layer -> output layer -> sample / arg-max -> next input) as ONE launch per call of
// avsr_attn_rnn_fwd, instead of 4 dependent launches per decode step.
//
// MI355X mapping.  Utterances are independent, so the batch is cut into groups of 8 rows and a group is bound to ONE
// XCD (32 CUs, one 256-thread workgroup per CU, claimed by XCC_ID exactly as rnn_persist.hip does).  Everything a
// group touches between two steps stays on that XCD:
//   * RESIDENT operands, loaded once per launch:  the workgroup's slice of the cell kernel (8 units x 4 gates x
//     (E+A+H) rows = 112 VGPRs/lane at c4), of the attention layers (32 VGPRs) and of the output layer; and its
//     QUARTER of one utterance's attention memories -- the keys in VGPRs (16 lanes per frame), the values in LDS.
//     At c4 (B=64, T_a=500, T_v=75, H=D=256) that is 288 KB of keys+values per CU = the whole 75.4 MB the per-step
//     attention kernel used to stream from HBM every step; here a decode step reads them from registers / LDS.
//   * per step three all-to-all hand-offs inside the XCD (h -> scores/context partials -> attention vector), each a
//     plain store into the XCD's L2 + one progress word per workgroup, polled by one coalesced L1-bypassing load
//     and read back with 16-byte sc1 loads (MI355X_MICROARCH.md "handoff"; measured 0.44 us per hop idle).
// Phases of step l in workgroup j of a group (NW = 32 workgroups, R = 8 rows, 4 workgroups per row):
//   P1 cell      gate columns of units [j*UW, (j+1)*UW) for all 8 rows: z = [x | att(l-1) | h(l-1)] . W on
//                v_mfma_f32_16x16x4_f32, K split over the 4 waves; gates, cell clip, masks, records; publishes h.
//   P2 attention row j/4, quarter j%4 of every memory: scores from the resident keys, quarter max / exp-sum,
//                un-normalised partial context from the resident values; publishes the partials.
//   P3 att layer attention columns [j*AW, (j+1)*AW) for all 8 rows: merges the four quarters' softmax partials while
//                loading its A operand, [h | ctx] . W_att; writes the attention record (+ its input-dropped copy),
//                and this workgroup's split-K share of the logits; publishes.
//   P4 sample    every workgroup sums the 32 logit shares (same order everywhere -> same tokens everywhere), draws the
//                scheduled sample / takes the arg-max, and builds the next step's input rows in LDS.
// Records (gates, cell states, outputs, raw scores, contexts, softmax statistics, logits, fed tokens) are written in
// the layouts avsr_attn_rnn_bwd and the post-loop GEMMs consume, so BPTT is unchanged.
//
// Every wait is bounded; a miss raises the sticky error word of the persistent-RNN scratch (same protocol as
// rnn_persist.hip) and the host redoes the call with one launch per phase.  Declines (AVSR_ERR_UNSUPPORTED -> per-step
// launches): GRU, multi-layer decoder cells, beam search, H or D > 256, V > 64, memories that
// do not fit the resident budget.
#include "dec_persist.h"

namespace avsr {

// KR0 / KR1: register capacity of the resident keys of mechanism 0 / 1, in frames per 16-lane group (32 groups per workgroup)
//
// Register discipline (two waves per SIMD -> 256 registers per lane): the resident operands take 152 of them (keys 80, cell
// kernel 56, attention layers 16).  Everything else a thread needs per step is RE-DERIVED from an opaque copy of its thread id
// inside the step loop -- otherwise the compiler hoists ~100 loop-invariant indices / 64-bit addresses out of the loop and
// spills them, and every reload sits on the critical path behind a full vmcnt wait.  Chunk tables are wave-uniform (SGPRs).
// R: rows (utterances) per group.  8 everywhere except the AV-Align attentive layer (MODE 0) when its memories fit: there a group
// is a FULL 16-row MFMA tile (no padding rows) and two workgroups instead of four share a row's memories -- half as many groups,
// so a 128-utterance batch is one pass over the chip instead of two sequential 64-row slices.
// BAH: the block's one mechanism is (normed) Bahdanau (attention.py:25-42): an extra phase computes the processed query
// pq = cell_out . W_q for the group (one more hand-off per step), the scores are v . tanh(keys + pq + b), and -- output_attention
// being False for this family -- the logits come from the cell output (split-K shares published with the cell phase).
// V64: vocabularies of 33..64 symbols (`phoneme`: V = 41, io_utils.py:354-370): the (row, symbol) phases of the output layer use all 512
// threads as 8 rows x 64 symbols (one wave per row) instead of 256 threads as 8 x 32, the output-kernel rows in LDS are round4(V) wide
// and the quarter's scores share the input-row buffer (round 5: no more LDS than the 32-symbol layout), the logit shares are [.][64].
template <int KR0, int KR1, int MODE, int R, bool BAH = false, bool V64 = false>
__global__ __launch_bounds__(DP_NT) void dec_persist_kernel(const DPLaunch L) {
  static_assert(R == 8 || (R == 16 && MODE == 0), "16-row groups: attentive layer only");
  static_assert(!BAH || (R == 8 && KR1 == 0), "Bahdanau: one mechanism, 8-row groups");
  static_assert(!V64 || (R == 8 && MODE >= 1), "64-symbol rows: decoder modes only");
  constexpr int VW = V64 ? 64 : 32, VSH = V64 ? 6 : 5;      // symbols per row of the (row, symbol) thread layout
  constexpr int NPH = BAH ? 4 : 3;              // hand-offs per step
  constexpr int PH_PQ = 1, PH_ATT = BAH ? 2 : 1, PH_LAYER = BAH ? 3 : 2;
  constexpr int WPR = DP_NW / R;                // workgroups per row in the attention phase (4 quarters / 2 halves)
  constexpr int RQ = R / 4;                     // lane groups q < RQ hold real rows of a C tile (row = 4q + r)
  constexpr int RED_F = 256 * R;                // floats of the reduction buffer [8 waves][2 tiles][R][16]
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const red = lds;                       // [8][2][R][16] cell / [2][2][256] context partials / [8][R][16] attention layer / [8][32] exp
  // V64: the quarter's scaled scores (written and read in P2 only) share the input-row buffer (read in P1, rewritten in P4: dead in
  // between, every phase boundary is a workgroup barrier), and the output-kernel rows are as wide as the vocabulary needs (round4(V),
  // not 64): 64 floats LESS ahead of the values than the 32-symbol layout at V = 41, so a phoneme model takes this kernel wherever a
  // character model does -- also at the benchmark shape, whose memories fill LDS to the last 2 KB (round 4 declined it there).
  constexpr int SPO = V64 ? 0 : 256;            // floats of the scores' own region
  const int WOS = V64 ? ((L.V + 3) & ~3) : 32;  // row stride of the output-kernel rows in LDS
  float* const s_p = lds + RED_F;               // [2][128] scaled scores of this quarter (P2) ...
  float* const s_x = lds + RED_F + SPO;         // [8][128] input rows of the next step (MODE >= 1: R = 8)
  float* const s_att = lds + RED_F + SPO + 1024;      // [R][16]  this workgroup's attention columns
  int* const s_int = reinterpret_cast<int*>(lds + RED_F + SPO + 1024 + 16 * R + 32);   // [0..R) tokens, [R..2R) step lengths, [2R] slot, [2R+1] unfinished, [2R+8..+12) zero pad
  constexpr int ZPAD = RED_F + SPO + 1024 + 16 * R + 32 + 2 * R + 8;                     // 4 zero floats: where operand slots of another source "read" LDS
  float* const s_wo = lds + RED_F + SPO + 1024 + 16 * R + 32 + 2 * R + 16;             // [16][WOS] this workgroup's rows of the output kernel
  float* const vals = lds + (R == 8 ? DP_MISC + (V64 ? 16 * WOS - 512 - 256 : 0) : DP_MISC16);      // resident value rows of this workgroup's share
  static_assert(2048 + 1280 + 128 + 32 + 16 + 16 + 512 == DP_MISC, "8-row layout");
  static_assert(4096 + 1280 + 256 + 32 + 32 + 16 + 512 <= DP_MISC16, "16-row layout");

  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int g = __builtin_amdgcn_readfirstlane(xcc_id());
  if (tid0 == 0) s_int[2 * R] = __hip_atomic_fetch_add(L.claim + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int j = __builtin_amdgcn_readfirstlane(s_int[2 * R]);
  if (g >= L.ngroups || j >= DP_NW) return;

  const int B = L.B, Ls = L.L, H = L.H, E = L.E, A = L.A, KW = L.KW, V = L.V;
  constexpr int mode = MODE;
  const int rowbase = L.b0 + g * R;
  const bool drop = L.drop != 0;
  const uint32_t seedv = L.seed ? (uint32_t)L.seed[0] : 0u;
  const uint32_t cid4 = L.cid4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const long BH = (long)B * H;
  float* const hbuf = L.state;                  // [2][B][H] ping-pong, then c [2][B][H]
  float* const cbuf = L.state + 2 * BH;

  // ---------------------------------------------------------------------------------------------------------
  // wave-uniform chunk tables (E, A, H, D are multiples of 16: a 16-wide chunk never straddles a source)
  // ---------------------------------------------------------------------------------------------------------
  const int UW = L.UW, uwsh = L.uwsh, unit0 = j * UW, col0 = unit0 * 4;
  // cell slots of a wave: 0 = input rows (LDS), 1..4 = attention record, 5..6 = h state; every wave takes an eighth of each source
  const int nx = (mode == 0) ? 0 : E >> 4, na = A >> 4, nh = H >> 4;
  const int xg0 = (wave * nx) / DP_WV, nxw = ((wave + 1) * nx) / DP_WV - xg0;
  const int ag0 = (wave * na) / DP_WV, naw = ((wave + 1) * na) / DP_WV - ag0;
  const int hg0 = (wave * nh) / DP_WV, nhw = ((wave + 1) * nh) / DP_WV - hg0;
  unsigned uk1[DP_CPW];                         // byte offset of the slot's chunk inside its global source row
  const int uxa = nxw > 0 ? RED_F + SPO + (xg0 << 4) : -1;   // LDS float index of the input-row chunk (s_x; MODE >= 1: R = 8)
  const int AW = L.AW, awsh = L.awsh, an0 = j * AW;
  const bool has_att = an0 < A;
  const int ma = has_att ? an0 / H : 0;
  const int nl0 = an0 - ma * H;
  const DPMech& Ma = L.m[ma];
  const int Da = Ma.D;
  // attention-layer slots: 0..1 = cell output, 2..3 = context partials
  const int nq = H >> 4, nd = Da >> 4;
  const int qg0 = (wave * nq) / DP_WV, nqw = ((wave + 1) * nq) / DP_WV - qg0;
  const int dg0 = (wave * nd) / DP_WV, ndw = ((wave + 1) * nd) / DP_WV - dg0;
  unsigned uk3[DP_APW];
  int cin[2];                                    // context slot in use (its merged rows go to the context record)

  // ---------------------------------------------------------------------------------------------------------
  // resident operands
  // ---------------------------------------------------------------------------------------------------------
  f32x4 wc[DP_CPW][2], wa[DP_APW], wq[2];
  f32x4 k0[KR0 > 0 ? KR0 : 1][4], k1[KR1 > 0 ? KR1 : 1][4];
  int n_m[2] = {0, 0}, t0_m[2] = {0, 0};
  const int r_att = j / WPR, cq = j % WPR;
  const int b_att = rowbase + r_att;
  const bool att_row = b_att < B;
  float c_state = 0.f, h_state = 0.f;
  {
    const int tid = tid0, lane = tid & 63, i = lane & 15, q = lane >> 4, s16 = tid & 15, rg = tid >> 4;
    // (1) cell kernel slice: gate columns of units [unit0, unit0 + UW); K split over the 8 waves
#pragma unroll
    for (int cc = 0; cc < DP_CPW; ++cc) {
      int kk, wbase;
      bool in;
      if (cc == 0) { kk = xg0 << 4; wbase = 0; in = nxw > 0; }
      else if (cc < 5) { kk = (ag0 + cc - 1) << 4; wbase = E; in = cc - 1 < naw; }
      else { kk = (hg0 + cc - 5) << 4; wbase = E + A; in = cc - 5 < nhw; }
      uk1[cc] = (in && cc > 0) ? (unsigned)(kk * 4) : 0u;      // unused slot: chunk 0 of the row against zero weights
      const int k = kk + 4 * q;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int cl = nt * 16 + i, col = col0 + cl;
        wc[cc][nt] = (in && cl < 4 * UW && col < 4 * H) ? ld4(L.wt + (long)col * KW + wbase + k) : zero4;
      }
    }
    // (2) attention-layer slice: attention columns [an0, an0 + AW) of mechanism ma
#pragma unroll
    for (int cc = 0; cc < DP_APW; ++cc) {
      const bool qs = cc < 2;
      const int kk = qs ? (qg0 + cc) << 4 : (dg0 + cc - 2) << 4, wbase = qs ? 0 : H;
      const bool in = has_att && (qs ? cc < nqw : cc - 2 < ndw);
      uk3[cc] = in ? (unsigned)(kk * 4) : 0u;
      if (cc >= 2) cin[cc - 2] = in ? 1 : 0;
      wa[cc] = (in && i < AW) ? ld4(Ma.watt_t + (long)(nl0 + i) * (H + Da) + wbase + kk + 4 * q) : zero4;
    }
    // (3) output-layer rows -> LDS [k][symbol]: the attention columns [an0, an0 + AW) (Luong family: logits from the attention
    //     vector) or the cell-output units [unit0, unit0 + UW) (Bahdanau family: logits from the cell output)
    for (int e = tid; e < 16 * VW; e += DP_NT) {
      const int k = e >> VSH, v = e & (VW - 1);
      float wv = 0.f;
      if (mode >= 1 && v < V) {
        if (!BAH) { if (L.oa && has_att && k < AW) wv = L.wout_t[(long)v * A + an0 + k]; }
        else if (k < UW && unit0 + k < H) wv = L.wout_t[(long)v * H + unit0 + k];
      }
      if (v < WOS) s_wo[k * WOS + v] = wv;
    }
    // (3b) Bahdanau: this workgroup's columns [unit0, unit0 + UW) of the query layer, K split over the waves like the attention layer's
    //      cell-output slots
    if (BAH) {
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int kk = (qg0 + cc) << 4;
        wq[cc] = (cc < nqw && i < UW && unit0 + i < H) ? ld4(L.m[0].wq_t + (long)(unit0 + i) * H + kk + 4 * q) : zero4;
      }
    }
    // (4) attention memories of row r_att, quarter cq: keys -> registers (16 lanes per frame), values -> LDS
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (m >= L.n_mech) continue;
      const DPMech& M = L.m[m];
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
    // (5) recurrent state of (row er, unit eu), tokens / step lengths of the group's rows, first input rows
    const int er = tid >> uwsh, eu = tid & (UW - 1), eb = rowbase + er, eun = unit0 + eu;
    if (tid < R * UW && eb < B && eun < H) {
      c_state = cbuf[(long)(L.l_begin & 1) * BH + (long)eb * H + eun];
      h_state = hbuf[(long)(L.l_begin & 1) * BH + (long)eb * H + eun];
    }
    if (tid < R) {
      const int b = rowbase + tid;
      s_int[tid] = (b < B && mode == 1) ? L.tok[b] : 0;
      s_int[R + tid] = (b < B) ? L.steplen[b] : 0;
    }
    if (tid == 0) s_int[2 * R + 1] = 0;
    if (tid < 4) lds[ZPAD + tid] = 0.f;
    __syncthreads();
    if (mode >= 1) {
      const int e4n = E >> 2;
      for (int idx = tid; idx < R * e4n; idx += DP_NT) {
        const int r = idx / e4n, e4 = idx - r * e4n, b = rowbase + r;
        f32x4 v = zero4;
        if (b < B) v = (mode == 2) ? ld4(L.xs + ((long)b * Ls + L.l_begin) * E + 4 * e4) : ld4(L.embedding + (long)s_int[r] * E + 4 * e4);
        st4(s_x + r * 128 + 4 * e4, v);
      }
    }
    __syncthreads();
  }
  const __amdgpu_buffer_rsrc_t ctx_rs = make_rsrc(Ma.ctx);
  const __amdgpu_buffer_rsrc_t att_rs = make_rsrc(drop ? L.attd : L.att), h_rs = make_rsrc(hbuf), co_rs = make_rsrc(L.cell_out);
  const __amdgpu_buffer_rsrc_t pc_rs = make_rsrc(Ma.ppctx), pm_rs = make_rsrc(Ma.ppm), pl_rs = make_rsrc(Ma.ppl);
  const __amdgpu_buffer_rsrc_t plog_rs = make_rsrc(L.plog), gates_rs = make_rsrc(L.gates), bias_rs = make_rsrc(L.bias);
  const __amdgpu_buffer_rsrc_t pq_rs = make_rsrc(L.m[0].pq);
  const int NWL = BAH ? DP_NW : L.NWA;           // workgroups holding a split-K share of the logits

  int* const flag_base = L.flags + g * NPH * 32;
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
  long tm[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long last_ = __builtin_amdgcn_s_memtime();
#define DTICK(k) { const long now_ = __builtin_amdgcn_s_memtime(); tm[k] += now_ - last_; last_ = now_; }
#else
#define DTICK(k)
#endif
  const float bout_v = (mode >= 1 && L.bout && (tid0 & (VW - 1)) < V) ? L.bout[tid0 & (VW - 1)] : 0.f;
  // recurrent operand rows of the NEXT cell step: h(l) is complete once every workgroup has published P1 of step l, so the
  // rows are fetched right after that wait (during P2) and consumed a step later, off the critical path
  f32x4 hv[2];
  {
    const int lane = tid0 & 63, i = lane & 15, q = lane >> 4, ab = rowbase + i;
    const unsigned h_o = (i < R && ab < B) ? (unsigned)((long)(L.l_begin & 1) * BH + (long)ab * H) * 4u + (unsigned)(q * 16) : (unsigned)P_OOB;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) hv[cc] = ldb_sc1(h_rs, (int)(h_o + uk1[5 + cc]));
  }
  for (int l = L.l_begin; l < L.l_end; ++l) {
    const int epoch = l - L.l_begin + 1;
    int tid = tid0;
    asm volatile("" : "+v"(tid));                 // opaque: per-thread indices below are recomputed, not kept across steps
    const int lane = tid & 63, i = lane & 15, q = lane >> 4;
    const int ab = rowbase + i;
    const bool aok = i < R && ab < B;             // MFMA A-operand row of this lane (R = 8: rows 8..15 of the tile are padding)
    DTICK(11)
    // =====================================================================================================
    // P1: LSTM cell (cells.py:14-18 LSTMCell clip 1.0, forget bias 1.0; DropoutWrapper cells.py:46-54)
    // =====================================================================================================
    {
      const int er = tid >> uwsh, eu = tid & (UW - 1), eb = rowbase + er, eun = unit0 + eu;
      const bool eok = tid < R * UW && eb < B && eun < H;
      const int e_steplen = s_int[R + (er & (R - 1))];
      const bool valid = eok && l < e_steplen;
      const long bt = (long)eb * Ls + l;
      // padding rows of the tile / rows beyond the batch: out-of-range offset, the load returns 0 without touching memory
      const unsigned att_o = aok ? (unsigned)(((long)ab * (Ls + 1) + l) * A) * 4u + (unsigned)(q * 16) : (unsigned)P_OOB;
      const int x_o = (i & 7) * 128 + 4 * q;        // (MODE >= 1 only: R = 8)
      f32x4 av[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) av[cc] = ldb_sc1(att_rs, (int)(att_o + uk1[1 + cc]));
      const f32x4 zpre = (MODE == 0) ? ldb4(gates_rs, valid ? (int)((bt * H + eun) * 16) : P_OOB) : zero4;
      const f32x4 bias4 = ldb4(bias_rs, (eok && L.bias) ? eun * 16 : P_OOB);
      f32x4 acc[2] = {zero4, zero4}, accb[2] = {zero4, zero4};       // two chains per column tile (40-cycle dependent latency)
      // input rows (LDS) and the recurrent h (fetched during the previous step) first: they run under the attention loads
      if (MODE != 0) {
        const f32x4 a4 = ld4(lds + (uxa >= 0 ? uxa + x_o : ZPAD));
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], wc[0][nt][e], acc[nt], 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          accb[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[0][e], wc[5][nt][e], accb[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[1][e], wc[6][nt][e], acc[nt], 0, 0, 0);
        }
#pragma unroll
      for (int cc = 0; cc < 4; cc += 2)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cc][e], wc[1 + cc][nt][e], acc[nt], 0, 0, 0);
            accb[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cc + 1][e], wc[2 + cc][nt][e], accb[nt], 0, 0, 0);
          }
      acc[0] += accb[0]; acc[1] += accb[1];
      if (q < RQ) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((wave * 2 + nt) * R + q * 4 + r) * 16 + i] = acc[nt][r];
      }
      lds_barrier();
      DTICK(0)
      float ho_keep = 0.f;
      if (eok) {
        const long so = ((long)eb * (Ls + 1) + l + 1) * H + eun;
        float hnext = h_state;
        if (valid) {
          f32x4 z = bias4 + zpre;
          {
            // the four gates of a unit are four consecutive columns of one 16-column tile: one 16-byte LDS read per wave partial instead
            // of four 4-byte ones (same per-element summation order)
            const int c0 = eu * 4;
            const float* ro = red + ((c0 >> 4) * R + er) * 16 + (c0 & 15);
            constexpr int WS = 32 * R;               // floats per wave
            z += ((ld4(ro) + ld4(ro + WS)) + (ld4(ro + 2 * WS) + ld4(ro + 3 * WS))) + ((ld4(ro + 4 * WS) + ld4(ro + 5 * WS)) + (ld4(ro + 6 * WS) + ld4(ro + 7 * WS)));
          }
          f32x4 g4;
          g4[0] = p_sigmoid(z[0]); g4[1] = p_tanh(z[1]); g4[2] = p_sigmoid(z[2] + 1.0f); g4[3] = p_sigmoid(z[3]);
          float c = g4[2] * c_state + g4[0] * g4[1];
          c = fminf(1.0f, fmaxf(-1.0f, c));
          const float h = g4[3] * p_tanh(c);
          const uint32_t oidx = (uint32_t)(bt * H + eun);
          const float ho = h * p_drop(drop, seedv, cid4 + 2, oidx, L.k_out);
          const float hs = h * p_drop(drop, seedv, cid4 + 1, oidx, L.k_st);
          L.cell_out[so] = ho;
          ho_keep = ho;
          hnext = hs;
          if (drop) L.hs_seq[so] = hs;
          st4(L.gates + (bt * H + eun) * 4, g4);
          L.cs[bt * H + eun] = c;
          c_state = c;
          h_state = hs;
        } else {
          L.cell_out[so] = 0.f;
          if (drop) L.hs_seq[so] = 0.f;
        }
        hbuf[(long)((l + 1) & 1) * BH + (long)eb * H + eun] = hnext;
      }
      if (BAH && mode >= 1) {
        // Bahdanau family (output_attention False): logits = cell_out . W_out; this workgroup's split-K share from its UW units (double-buffered by step
        // parity: the shares of step l are read in P4 of step l while a faster workgroup already writes those of step l + 1)
        if (tid < R * UW) s_att[er * 16 + eu] = ho_keep;
        lds_barrier();
        const int pr = (tid >> VSH) & 7, pv = tid & (VW - 1);
        if (tid < 8 * VW && pv < V) {
          float sacc = 0.f;
          for (int k = 0; k < UW; ++k) sacc += s_att[pr * 16 + k] * s_wo[k * WOS + pv];
          L.plog[((((long)(l & 1) * 8 + g) * DP_NW + j) * DP_R + pr) * VW + pv] = sacc;
        }
      }
      DTICK(1)
      publish(0, epoch);
      DTICK(2)
    }
    // =====================================================================================================
    // P2: scores, masked softmax partials, partial contexts of (row r_att, quarter cq)
    //     (attention.py:25-72 Luong / scaled Luong; contrib.seq2seq _compute_attention)
    // =====================================================================================================
    wait_all(0, epoch);
    DTICK(3)
    if (BAH) {
      // P1b: processed query pq = cell_out . W_q (attention.py:25-42 query_layer), columns [unit0, unit0 + UW) for the 8 rows
      const unsigned qo_ = aok ? (unsigned)(((long)ab * (Ls + 1) + l + 1) * H) * 4u + (unsigned)(q * 16) : (unsigned)P_OOB;
      f32x4 aq[2];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) aq[cc] = ldb_sc1(co_rs, (cc < nqw) ? (int)(qo_ + (unsigned)(((qg0 + cc) << 4) * 4)) : P_OOB);
      f32x4 acc = zero4;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[cc][e], wq[cc][e], acc, 0, 0, 0);
      if (q < RQ) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * R + q * 4 + r) * 16 + i] = acc[r];
      }
      lds_barrier();
      if (tid < R * UW) {
        const int er = tid >> uwsh, eu = tid & (UW - 1), eb = rowbase + er, eun = unit0 + eu;
        if (eb < B && eun < H) {
          const int o = er * 16 + eu;
          constexpr int WS = 16 * R;
          L.m[0].pq[((long)eb * Ls + l) * H + eun] = ((red[o] + red[WS + o]) + (red[2 * WS + o] + red[3 * WS + o])) + ((red[4 * WS + o] + red[5 * WS + o]) + (red[6 * WS + o] + red[7 * WS + o]));
        }
      }
      publish(PH_PQ, epoch);
      wait_all(PH_PQ, epoch);
    }
    {
      {
        const unsigned h_o = aok ? (unsigned)((long)((l + 1) & 1) * BH + (long)ab * H) * 4u + (unsigned)(q * 16) : (unsigned)P_OOB;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) hv[cc] = ldb_sc1(h_rs, (int)(h_o + uk1[5 + cc]));
      }
      const int s16 = tid & 15, rg = tid >> 4;
      f32x4 q4[4];
      const unsigned qo = att_row ? (unsigned)(((long)b_att * (Ls + 1) + l + 1) * H) * 4u + (unsigned)(s16 * 16) : (unsigned)P_OOB;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) q4[jj] = ldb_sc1(co_rs, (4 * s16 + 64 * jj < H) ? (int)(qo + 256 * jj) : P_OOB);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (m >= L.n_mech) continue;
        const DPMech& M = L.m[m];
        const float gsc = (M.type == ATT_SCALED_LUONG) ? M.g[0] : 1.f;
        float* const srow = M.scores + ((long)b_att * Ls + l) * M.T + t0_m[m];
        float* const sc = s_p + m * 128;
        const int n = n_m[m];
        if (m == 0) {
          f32x4 v4[4], pb4[4];
          if (BAH) {                                   // v . tanh(keys + pq + b): pq of this row from the phase above
            const unsigned po_ = att_row ? (unsigned)(((long)b_att * Ls + l) * H) * 4u + (unsigned)(s16 * 16) : (unsigned)P_OOB;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const bool in = 4 * s16 + 64 * jj < H;
              pb4[jj] = ldb_sc1(pq_rs, in ? (int)(po_ + 256 * jj) : P_OOB);
              v4[jj] = in ? ld4(M.v + 4 * s16 + 64 * jj) : zero4;
              if (in && M.bq) pb4[jj] += ld4(M.bq + 4 * s16 + 64 * jj);
            }
          }
#pragma unroll
          for (int u = 0; u < KR0; ++u) {
            float a;
            if (BAH) {
              a = 0.f;
#pragma unroll
              for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int e = 0; e < 4; ++e) a += v4[jj][e] * p_tanh(k0[u][jj][e] + pb4[jj][e]);
            } else {
              a = (dot4(k0[u][0], q4[0]) + dot4(k0[u][1], q4[1])) + (dot4(k0[u][2], q4[2]) + dot4(k0[u][3], q4[3]));
            }
            a = row16_sum(a);
            const int fr = rg + 32 * u;
            if (s16 == 0 && fr < n) { srow[fr] = a; sc[fr] = a * gsc; }
          }
        } else {
#pragma unroll
          for (int u = 0; u < KR1; ++u) {
            float a = (dot4(k1[u][0], q4[0]) + dot4(k1[u][1], q4[1])) + (dot4(k1[u][2], q4[2]) + dot4(k1[u][3], q4[3]));
            a = row16_sum(a);
            const int fr = rg + 32 * u;
            if (s16 == 0 && fr < n) { srow[fr] = a; sc[fr] = a * gsc; }
          }
        }
      }
      lds_barrier();
      DTICK(12)
      // every wave derives the quarter's max / exp-sum itself (two frames per lane): no block reduction, no barrier;
      // a lane keeps the numerators of frames (lane, lane + 64) and the context loop broadcasts them with readlane.
      // Partial contexts: the 8 waves are shared out between the mechanisms in proportion to their frame counts; a wave
      // walks every nw-th frame of its mechanism with one float4 column per lane, 4 frames in flight.
      const int nw0 = (L.n_mech < 2) ? DP_WV : max(1, min(DP_WV - 1, (DP_WV * n_m[0] + (n_m[0] + n_m[1]) / 2) / max(1, n_m[0] + n_m[1])));
      const int mw = (wave < nw0) ? 0 : 1, wl = mw ? wave - nw0 : wave, nwm = mw ? DP_WV - nw0 : nw0;
      {
        const DPMech& M = L.m[mw];
        const int D = M.D, n = n_m[mw];
        const float v0 = (lane < n) ? s_p[mw * 128 + lane] : -INFINITY, v1 = (lane + 64 < n) ? s_p[mw * 128 + 64 + lane] : -INFINITY;
        const float mx = wave64_max(fmaxf(v0, v1));
        const float p0 = (lane < n) ? __expf(v0 - mx) : 0.f, p1 = (lane + 64 < n) ? __expf(v1 - mx) : 0.f;
        const float lsum = wave64_sum(p0) + wave64_sum(p1);
        if (wl == 0 && lane == 0 && att_row) {
          M.ppm[(long)cq * B + b_att] = (n > 0) ? mx : -INFINITY;
          M.ppl[(long)cq * B + b_att] = lsum;
        }
        DTICK(13)
        f32x4 a4 = zero4;
        {
          // no divergence around the readlane broadcasts (a VALU select executed under a partial exec mask would leave the
          // source lanes of later frames stale): lanes beyond the value width walk column 0 and their sums are never read
          const float* vp = vals + M.lds_off + (4 * lane < D ? 4 * lane : 0);
          int fr = wl;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const float ph = half ? p1 : p0;                 // numerators of frames [64*half, 64*half + 64)
            const int nend = half ? n : min(n, 64);
            for (; fr + 3 * nwm < nend; fr += 4 * nwm) {
              f32x4 x[4];
              float pf[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int f = fr + u * nwm;
                x[u] = ld4(vp + f * D);
                pf[u] = rdlane_f(ph, f & 63);
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) a4 += pf[u] * x[u];
            }
            for (; fr < nend; fr += nwm) a4 += rdlane_f(ph, fr & 63) * ld4(vp + fr * D);
          }
        }
        st4(red + wave * 256 + 4 * lane, a4);
      }
      lds_barrier();
      DTICK(14)
      if (tid < 128 && att_row) {
        const int m = tid >> 6, c4 = tid & 63;
        if (m < L.n_mech && 4 * c4 < L.m[m].D) {
          const int w0 = m ? nw0 : 0, w1 = (m || L.n_mech < 2) ? DP_WV : nw0;
          // all eight waves' partial rows in flight at once, summed in the same order over [w0, w1) (a loop with run-time bounds made
          // this eight dependent LDS round trips: ~1000 ticks of the phase, profiles/r05_phase_ticks_before.txt "P2 rest")
          f32x4 pv8[DP_WV];
#pragma unroll
          for (int w = 0; w < DP_WV; ++w) pv8[w] = ld4(red + w * 256 + 4 * c4);
          f32x4 sacc = zero4;
          bool first = true;
#pragma unroll
          for (int w = 0; w < DP_WV; ++w)
            if (w >= w0 && w < w1) { sacc = first ? pv8[w] : sacc + pv8[w]; first = false; }
          st4(L.m[m].ppctx + ((long)cq * B + b_att) * L.m[m].D + 4 * c4, sacc);
        }
      }
      DTICK(4)
      publish(PH_ATT, epoch);
      DTICK(5)
    }
    // =====================================================================================================
    // P3: attention layer att_m = [cell_out, ctx_m] . W_att,m (attention.py:173-181), split-K share of the logits
    // =====================================================================================================
    wait_all(PH_ATT, epoch);
    DTICK(6)
    int label_pf = 0;                               // label of this step for the sampler (row tid < 8), fetched a phase early
    if (MODE == 2 && tid < DP_R && rowbase + tid < B && l + 1 < Ls) label_pf = L.labels[(long)(rowbase + tid) * Ls + l];
    {
      if (has_att) {
        // every operand of the phase is requested in one batch: statistics, 2 cell-output slots, 2 x 4 context partials
        float wgt[4], Mx = -INFINITY, Lsum = 0.f;
        f32x4 aq[2], sv[2][4];
        {
          float pmv[4], plv[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int o = (aok && c < WPR) ? (c * B + ab) * 4 : P_OOB;
            pmv[c] = ld1_sc1(pm_rs, o);
            plv[c] = ld1_sc1(pl_rs, o);
          }
          {
            const unsigned qo_ = aok ? (unsigned)(((long)ab * (Ls + 1) + l + 1) * H) * 4u + (unsigned)(q * 16) : (unsigned)P_OOB;
            const unsigned pc0_ = aok ? (unsigned)((long)ab * Da) * 4u + (unsigned)(q * 16) : (unsigned)P_OOB;
            const unsigned pcs_ = aok ? (unsigned)((long)B * Da) * 4u : 0u;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
              aq[cc] = ldb_sc1(co_rs, (int)(qo_ + uk3[cc]));
#pragma unroll
              for (int c = 0; c < 4; ++c) sv[cc][c] = ldb_sc1(pc_rs, (c < WPR) ? (int)(pc0_ + c * pcs_ + uk3[2 + cc]) : P_OOB);
            }
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) { wgt[c] = (aok && c < WPR) ? pmv[c] : -INFINITY; Mx = fmaxf(Mx, wgt[c]); }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float e = (wgt[c] == -INFINITY) ? 0.f : __expf(wgt[c] - Mx);
            Lsum += e * plv[c];
            wgt[c] = e;
          }
          const float inv = Lsum > 0.f ? 1.0f / Lsum : 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) wgt[c] *= inv;
        }
        DTICK(15)
        const bool saver = nl0 == 0;                                   // first workgroup of a mechanism keeps the records
        if (saver && aok && q == 0 && wave == 0) {                     // merged statistics in chunk 0, neutral elsewhere
          float* pmr = Ma.pstat + (long)(2 * l) * Ma.nc_rec * B;
          float* plr = Ma.pstat + (long)(2 * l + 1) * Ma.nc_rec * B;
          pmr[ab] = Mx; plr[ab] = Lsum;
          for (int c = 1; c < Ma.nc_rec; ++c) { pmr[(long)c * B + ab] = -INFINITY; plr[(long)c * B + ab] = 0.f; }
        }
        const unsigned cso = (aok && saver) ? (unsigned)(((long)ab * Ls + l) * Da) * 4u + (unsigned)(q * 16) : (unsigned)P_OOB;
        f32x4 acc = zero4, acc2 = zero4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[0][e], wa[0][e], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[1][e], wa[1][e], acc2, 0, 0, 0);
        }
        f32x4 c4[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          c4[cc] = (wgt[0] * sv[cc][0] + wgt[1] * sv[cc][1]) + (wgt[2] * sv[cc][2] + wgt[3] * sv[cc][3]);
          stb4(ctx_rs, cin[cc] ? (int)(cso + uk3[2 + cc]) : P_OOB, c4[cc]);   // context record
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(c4[0][e], wa[2][e], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(c4[1][e], wa[3][e], acc2, 0, 0, 0);
        }
        acc += acc2;
        if (q < RQ) {
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(wave * R + q * 4 + r) * 16 + i] = acc[r];
        }
      }
      lds_barrier();
      DTICK(16)
      const int pr = (tid >> VSH) & 7, pv = tid & (VW - 1);
      if (tid < R * AW && has_att) {
        const int ar = tid >> awsh, ac = tid & (AW - 1), arb = rowbase + ar;
        float a = 0.f;
        if (arb < B) {
          const int o = ar * 16 + ac;
          constexpr int WS = 16 * R;                 // floats per wave
          a = ((red[o] + red[WS + o]) + (red[2 * WS + o] + red[3 * WS + o])) + ((red[4 * WS + o] + red[5 * WS + o]) + (red[6 * WS + o] + red[7 * WS + o]));
          if (!(l < s_int[R + ar])) a = 0.f;
          const long ao = ((long)arb * (Ls + 1) + l + 1) * A + an0 + ac;
          L.att[ao] = a;
          if (drop) L.attd[ao] = a * p_drop(true, seedv, cid4, (uint32_t)(((long)arb * Ls + l + 1) * (E + A) + E + an0 + ac), L.k_in);
        }
        s_att[ar * 16 + ac] = a;
      }
      lds_barrier();
      if (!BAH && mode >= 1 && L.oa && has_att && tid < 8 * VW && pv < V) {
        float s = 0.f;
        for (int k = 0; k < AW; ++k) s += s_att[pr * 16 + k] * s_wo[k * WOS + pv];
        L.plog[((((long)(l & 1) * 8 + g) * DP_NW + j) * DP_R + pr) * VW + pv] = s;
      }
      DTICK(7)
      publish(PH_LAYER, epoch);
      DTICK(8)
    }
    // =====================================================================================================
    // P4: logits, sample / arg-max, next input rows (decoder_unimodal.py:304-309 ScheduledEmbeddingTrainingHelper,
    //     :176-217 GreedyEmbeddingHelper + dynamic_decode(impute_finished=True))
    // =====================================================================================================
    wait_all(PH_LAYER, epoch);
    DTICK(9)
    if (mode >= 1) {
      const int pr = (tid >> VSH) & 7, pv = tid & (VW - 1), pb = rowbase + pr;
      if (tid < 8 * VW) {
        float z = 0.f;
        const unsigned po = (unsigned)(((((long)(l & 1) * 8 + g) * DP_NW) * DP_R + pr) * VW + pv) * 4u;
        float part[DP_NW];
#pragma unroll
        for (int w = 0; w < DP_NW; ++w) part[w] = ld1_sc1(plog_rs, (w < NWL && pv < V) ? (int)(po + (unsigned)(w * DP_R * VW * 4)) : P_OOB);
#pragma unroll
        for (int w = 0; w < DP_NW; ++w) z += part[w];
        const bool valid = l < s_int[R + pr];
        z = valid ? z + bout_v : 0.f;
        if (pv < V && j == 0 && pb < B) L.logits[((long)pb * Ls + l) * V + pv] = z;
        // row maximum over the 32 (64) lanes of this row (exact, order-free), then the softmax numerators for the sampler
        float m0 = row16_max(pv < V ? z : -INFINITY);          // the row's lanes are two (four) DPP rows of this wave
        {
          const float mlo = fmaxf(rdlane_f(m0, 0), rdlane_f(m0, 16)), mhi = fmaxf(rdlane_f(m0, 32), rdlane_f(m0, 48));
          m0 = V64 ? fmaxf(mlo, mhi) : ((lane & 32) ? mhi : mlo);
        }
        if (mode == 2) red[pr * VW + pv] = pv < V ? expf(z - m0) : 0.f;
        if (mode == 1) {
          // first maximum (tf.argmax): lowest symbol whose logit equals the row maximum
          const unsigned long long bal = __ballot(pv < V && z == m0);
          const unsigned half = (unsigned)(bal >> (32 * ((tid >> 5) & 1)));
          const int first = V64 ? __builtin_ffsll((long long)bal) - 1 : __builtin_ffs((int)half) - 1;
          if (pv == 0) red[pr] = __builtin_bit_cast(float, first);
        }
      }
      lds_barrier();
      DTICK(17)
      if (tid < DP_R) {
        const int r = tid, b = rowbase + r;
        if (mode == 1) {
          int id = 0;
          bool unfin = false;
          if (b < B && l < s_int[R + r]) {
            id = __builtin_bit_cast(int, red[r]);
            s_int[r] = id;
            if (id == L.eos_id) s_int[R + r] = l + 1; else unfin = true;
          }
          if (j == 0 && b < B) L.ids[(long)b * Ls + l] = id;
          {
            const int cnt = __popcll(__ballot(unfin));      // rows of this group still decoding after step l
            if (tid == 0) s_int[2 * R + 1] = cnt;
          }
        } else if (l + 1 < Ls && b < B) {
          const uint32_t idx = (uint32_t)(b * Ls + l);
          int tk = label_pf;
          if (L.prob > 0.f && uniform01(seedv, 1000u, idx) < L.prob) {
            // expf(logit - max) per symbol, summed in index order exactly as the per-step sampler does; the row is pulled into
            // registers first (one dependent LDS read per symbol made this the longest stretch of the step)
            f32x4 ex4[VW / 4];
#pragma unroll
            for (int v4 = 0; v4 < VW / 4; ++v4) ex4[v4] = ld4(red + r * VW + 4 * v4);
            float tot = 0.f;
#pragma unroll
            for (int v = 0; v < VW; ++v) tot += (v < V) ? ex4[v >> 2][v & 3] : 0.f;
            const float target = uniform01(seedv, 1001u, idx) * tot;
            float run = 0.f;
            tk = V - 1;
            bool found = false;
#pragma unroll
            for (int v = 0; v < VW; ++v) {
              run += (v < V) ? ex4[v >> 2][v & 3] : 0.f;
              if (!found && v < V && run > target) { tk = v; found = true; }
            }
          }
          s_int[r] = tk;
          if (j == 0) L.fed[(long)b * Ls + l + 1] = tk;
        }
      }
      lds_barrier();
      DTICK(18)
      if (l + 1 < Ls) {
        const int e4n = E >> 2;
        for (int idx = tid; idx < DP_R * e4n; idx += DP_NT) {
          const int r = idx / e4n, e4 = idx - r * e4n, b = rowbase + r;
          f32x4 v = zero4;
          if (b < B) {
            v = ld4(L.embedding + (long)s_int[r] * E + 4 * e4);
            if (mode == 2) {
              if (drop && L.k_in < 1.0f) {
#pragma unroll
                for (int e = 0; e < 4; ++e)      // same expression as sched_sample_kernel (v / keep, not v * (1 / keep))
                  v[e] = uniform01(seedv, cid4, (uint32_t)(((long)b * Ls + l + 1) * (E + A) + 4 * e4 + e)) < L.k_in ? v[e] / L.k_in : 0.f;
              }
              if (j == (r << 2)) st4(L.xs + ((long)b * Ls + l + 1) * E + 4 * e4, v);
            }
          }
          st4(s_x + r * 128 + 4 * e4, v);
        }
      }
      lds_barrier();
      // greedy decode: every utterance of this group has emitted EOS -- dynamic_decode(impute_finished=True) emits zeros from here on and
      // the ids buffer is zeroed by the caller, so the group stops.  Every workgroup of the group holds the same tokens and step
      // lengths, hence takes this exit at the same step: nobody is left waiting for a hand-off.  (The host can therefore queue all
      // maximum_iterations steps as ONE launch instead of chunks with a host check in between.)
      if (mode == 1 && s_int[2 * R + 1] == 0) break;
    }
    DTICK(10)
  }
#ifdef DP_TIMING
  if (tid0 == 0 && j == 0 && g == 0)
    for (int k = 0; k < 24; ++k) L.err[(MODE == 0 ? 176 : 16) + k] = (int)(tm[k] / (L.l_end - L.l_begin));   // attentive layer: its own words
#endif
  // ---- state back to the ping-pong buffers (the next call / the caller's h_final, c_final copies read them) ----
  {
    const int tid = tid0;
    const int er = tid >> uwsh, eu = tid & (UW - 1), eb = rowbase + er, eun = unit0 + eu;
    if (tid < R * UW && eb < B && eun < H) cbuf[(long)(L.l_end & 1) * BH + (long)eb * H + eun] = c_state;
    if (mode == 1 && j == 0) {
      if (tid < DP_R && rowbase + tid < B) {
        L.steplen[rowbase + tid] = s_int[R + tid];
        L.tok[rowbase + tid] = s_int[tid];
      }
      if (tid == 0 && s_int[2 * R + 1] > 0) atomicAdd(L.n_unfinished, s_int[2 * R + 1]);
    }
  }
}

int g_dec_fused = 1;

// variant: 0 = one mechanism (<= 128 frames per quarter); 1 = (<= 32, <= 128); 2 = (<= 128, <= 32) frames per quarter;
// 3 = attentive layer (mode 0) in 16-row groups, one mechanism, <= 64 frames per half
static const void* dp_kernel(int variant, int mode, bool v64 = false) {
#define DPK(a, b) (mode == 0 ? (const void*)dec_persist_kernel<a, b, 0, 8> : mode == 1 ? (const void*)dec_persist_kernel<a, b, 1, 8> : (const void*)dec_persist_kernel<a, b, 2, 8>)
#define DPK64(a, b) (mode == 1 ? (const void*)dec_persist_kernel<a, b, 1, 8, false, true> : (const void*)dec_persist_kernel<a, b, 2, 8, false, true>)
  if (variant == 3) return (const void*)dec_persist_kernel<2, 0, 0, 16>;
  if (v64 && mode >= 1) {                                                                          // 33..64 symbols
    if (variant == 4) return mode == 1 ? (const void*)dec_persist_kernel<4, 0, 1, 8, true, true> : (const void*)dec_persist_kernel<4, 0, 2, 8, true, true>;
    return variant == 0 ? DPK64(4, 0) : variant == 1 ? DPK64(1, 4) : DPK64(4, 1);
  }
  if (variant == 4) return mode == 1 ? (const void*)dec_persist_kernel<4, 0, 1, 8, true> : (const void*)dec_persist_kernel<4, 0, 2, 8, true>;   // Bahdanau decoders
  return variant == 0 ? DPK(4, 0) : variant == 1 ? DPK(1, 4) : DPK(4, 1);
#undef DPK
#undef DPK64
}

static int g_dec_rows16 = -1;                   // -1: read AVSR_DEC_ROWS16 once (default on)

// Fills L for the descriptor; returns AVSR_ERR_UNSUPPORTED when the fused kernel does not cover it.
int dp_plan(const avsr_attn_rnn& d, DPLaunch& L, int* variant, size_t* lds_bytes) {
  if (!g_dec_fused || !g_sync || !d.fused_ws) return AVSR_ERR_UNSUPPORTED;
  if (d.cell != 0 || d.n_extra != 0 || d.n_mech < 1 || d.n_mech > 2 || d.mode < 0 || d.mode > 2) return AVSR_ERR_UNSUPPORTED;
  const int B = d.B, H = d.H, E = d.E, A = d.n_mech * H, KW = E + A + H;
  if (H > 256 || H % 4 || E % 4 || (d.mode != 0 && (E > 128 || d.V > 64))) return AVSR_ERR_UNSUPPORTED;   // mode 0: inputs hoisted, no logits
  const bool bah = d.n_mech == 1 && d.mech[0].type >= ATT_BAHDANAU;       // (normed) Bahdanau: one mechanism, decoder modes only
  if (bah && (d.mode < 1 || d.output_attention || !d.mech[0].v || !d.mech[0].wq_t || !d.mech[0].pq)) return AVSR_ERR_UNSUPPORTED;
  if (d.mode >= 1 && !d.output_attention && !bah) return AVSR_ERR_UNSUPPORTED;
  const int nx = d.mode == 0 ? 0 : (E + 15) / 16, NC = nx + (A + 15) / 16 + (H + 15) / 16;
  if ((NC + DP_WV - 1) / DP_WV > DP_CPW) return AVSR_ERR_UNSUPPORTED;
  if (H % 16 || E % 16) return AVSR_ERR_UNSUPPORTED;           // 16-wide operand chunks never straddle a source
  int UW = 1, AW = 1, uwsh = 0, awsh = 0;
  while (UW * DP_NW < H) { UW *= 2; ++uwsh; }
  while (AW * DP_NW < A) { AW *= 2; ++awsh; }
  if (UW > 8 || AW > 16 || H % AW) return AVSR_ERR_UNSUPPORTED;
  if ((long)B * (d.L + 1) * (A > H ? A : H) >= (1L << 29) || (long)4 * B * 256 >= (1L << 29)) return AVSR_ERR_UNSUPPORTED;
  if (avsr_attn_rnn_fused_ws_floats(B, d.n_mech, 256) > d.fused_ws_floats) return AVSR_ERR_UNSUPPORTED;
  L = DPLaunch{};
  L.B = B; L.L = d.L; L.H = H; L.E = E; L.V = d.V; L.n_mech = d.n_mech; L.mode = d.mode; L.oa = d.output_attention;
  L.go_id = d.go_id; L.eos_id = d.eos_id; L.A = A; L.KW = KW; L.UW = UW; L.AW = AW; L.NWA = (A + AW - 1) / AW; L.uwsh = uwsh; L.awsh = awsh;
  // 16-row groups (full MFMA row tiles, half as many groups): the attentive layer with ONE memory whose halves fit a workgroup
  if (g_dec_rows16 < 0) { const char* e = getenv("AVSR_DEC_ROWS16"); g_dec_rows16 = (e && e[0] == '0') ? 0 : 1; }
  L.R = DP_R;
  if (g_dec_rows16 && d.mode == 0 && d.n_mech == 1 && (d.mech[0].T + 1) / 2 <= 64 && UW * 16 <= DP_NT && AW * 16 <= DP_NT) L.R = 16;
  const int wpr = DP_NW / L.R;
  L.drop = (d.seed && d.mode != 1 && (d.keep_in < 1.f || d.keep_state < 1.f || d.keep_out < 1.f)) ? 1 : 0;
  L.wt = d.wt; L.bias = d.bias; L.gates = d.gates; L.cs = d.cs; L.cell_out = d.cell_out; L.att = d.att; L.attd = d.attd;
  L.hs_seq = d.hs_seq; L.state = d.state; L.steplen = d.steplen;
  L.embedding = d.embedding; L.wout_t = d.wout_t; L.bout = d.bout; L.logits = d.logits; L.ids = d.ids; L.tok = d.tok;
  L.n_unfinished = d.n_unfinished; L.xs = d.xs; L.labels = d.labels; L.fed = d.fed;
  L.seed = d.seed; L.k_in = d.keep_in; L.k_st = d.keep_state; L.k_out = d.keep_out; L.prob = d.sampling_prob;
  L.cid4 = (uint32_t)d.cell_id * 4;
  if (L.drop && (!d.hs_seq || !d.attd)) return AVSR_ERR_UNSUPPORTED;
  float* ws = d.fused_ws;
  L.plog = ws; ws += (long)((B + DP_R - 1) / DP_R + 16) * DP_NW * DP_R * 64;      // logit shares: [2 parities][8 groups of a slice][32][8][32 or 64]
  int lds_off = 0;
  for (int m = 0; m < d.n_mech; ++m) {
    const avsr_attn_mech& M = d.mech[m];
    if ((M.type > ATT_SCALED_LUONG && !bah) || M.D > 256 || M.D % 16 || M.T <= 0) return AVSR_ERR_UNSUPPORTED;
    if ((H + 15) / 16 + (M.D + 15) / 16 > DP_WV * DP_APW) return AVSR_ERR_UNSUPPORTED;
    if ((long)B * M.T * H >= (1L << 29)) return AVSR_ERR_UNSUPPORTED;
    DPMech& X = L.m[m];
    X.keys = M.keys; X.values = M.values; X.values_sb = M.values_sb; X.values_st = M.values_st; X.len = M.len; X.g = M.g;
    X.watt_t = M.watt_t; X.scores = M.scores; X.ctx = M.ctx; X.pstat = M.pstat;
    X.v = M.v; X.bq = (M.type == ATT_NORMED_BAHDANAU) ? M.bq : nullptr; X.wq_t = M.wq_t; X.pq = M.pq;
    X.T = M.T; X.D = M.D; X.type = M.type; X.nc_rec = (M.T + M.chunk - 1) / M.chunk; X.ch = (M.T + wpr - 1) / wpr;
    if (X.ch > 128) return AVSR_ERR_UNSUPPORTED;
    X.lds_off = lds_off; lds_off += X.ch * M.D;
    X.ppm = ws; ws += 4L * B; X.ppl = ws; ws += 4L * B; X.ppctx = ws; ws += 4L * B * M.D;
  }
  const bool v64 = d.mode >= 1 && d.V > 32;                                        // output-kernel rows round4(V) wide, scores in the input-row buffer
  const size_t bytes = sizeof(float) * ((size_t)(L.R == 16 ? DP_MISC16 : DP_MISC + (v64 ? 16 * ((d.V + 3) & ~3) - 512 - 256 : 0)) + lds_off);
  if (bytes > DP_LDS_BYTES) return AVSR_ERR_UNSUPPORTED;
  *lds_bytes = bytes;
  // register-resident key capacity (32 frames per pass): variant 0 = one mechanism up to 128 frames per quarter;
  // 1 = (<= 32, <= 128); 2 = (<= 128, <= 32)
  L.bah = bah ? 1 : 0;
  if (bah && (L.R != 8 || L.m[0].ch > 128)) return AVSR_ERR_UNSUPPORTED;
  if (bah) *variant = 4;
  else if (L.R == 16) *variant = 3;
  else if (d.n_mech == 1) *variant = 0;
  else if (L.m[0].ch <= 32 && L.m[1].ch <= 128) *variant = 1;
  else if (L.m[0].ch <= 128 && L.m[1].ch <= 32) *variant = 2;
  else return AVSR_ERR_UNSUPPORTED;
  return AVSR_OK;
}

}  // namespace avsr

int64_t avsr_dec_persist_bwd_ws_floats(int32_t B, int32_t n_mech);

// forward region of the fused workspace (the backward kernel's partials follow it)
int64_t avsr_dec_persist_fwd_ws_floats(int32_t B, int32_t n_mech, int32_t Dmax) {
  const int64_t groups = (B + DP_R - 1) / DP_R + 16;
  return groups * DP_NW * DP_R * 64 + (int64_t)n_mech * (8L * B + 4L * B * Dmax) + 64;
}

extern "C" int64_t avsr_attn_rnn_fused_ws_floats(int32_t B, int32_t n_mech, int32_t Dmax) {
  return avsr_dec_persist_fwd_ws_floats(B, n_mech, Dmax) + avsr_dec_persist_bwd_ws_floats(B, n_mech);
}

// 0: per-step launches; 1: fused forward and backward; 2: fused forward only; 3: fused backward only (A/B timing, tests)
extern "C" int avsr_attn_rnn_set_fused(int32_t on) { avsr::g_dec_fused = (on >= 0 && on <= 3) ? on : 1; return AVSR_OK; }

extern "C" int avsr_attn_rnn_fused_eligible(const avsr_attn_rnn* d) {
  using namespace avsr;
  if (!d) return 0;
  static thread_local DPLaunch L;
  int variant = 0; size_t lds = 0;
  return dp_plan(*d, L, &variant, &lds) == AVSR_OK ? 1 : 0;
}

// the SAME run-time conditions avsr_dec_persist_fwd declines on, on top of the plan: a caller that sizes its host checks by the answer
// (greedy decode runs all steps as ONE call when it is 1) must not be told "fused" and then get one launch per step and phase
extern "C" int avsr_attn_rnn_fused_fwd_active(const avsr_attn_rnn* d) {
  using namespace avsr;
  if (g_dec_fused == 3 || g_dec_fused == 0 || !g_sync) return 0;
  if ((long)(P_HDR + 8 + 8 * 4 * 32) > g_sync_ints) return 0;
  return avsr_attn_rnn_fused_eligible(d);
}

// Steps [l_begin, l_end) of the decoder as one persistent launch per 64-row slice.  The caller (avsr_attn_rnn_fwd) has
// initialised the state / slot-0 records exactly as for the per-step path and copies h_final / c_final afterwards.
int avsr_dec_persist_fwd(const avsr_attn_rnn* dp, int32_t l_begin, int32_t l_end, void* stream) {
  using namespace avsr;
  static thread_local DPLaunch L;
  int variant = 0; size_t lds = 0;
  if (g_dec_fused == 3) return AVSR_ERR_UNSUPPORTED;
  const int rc = dp_plan(*dp, L, &variant, &lds);
  if (rc) return rc;
  if (l_begin >= l_end) return AVSR_OK;
  hipStream_t s = (hipStream_t)stream;
  int32_t* sync = g_sync;
  const long words = P_HDR + 8 + 8 * 4 * 32;
  if (words > g_sync_ints) return AVSR_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    for (int v = 0; v < 5; ++v)
      for (int md = 0; md < 3; ++md)
        for (int w = 0; w < 2; ++w)
          if (hipFuncSetAttribute(dp_kernel(v, md, w != 0), hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS_BYTES) != hipSuccess) return AVSR_ERR_HIP;
    attr_set = true;
  }
  L.l_begin = l_begin; L.l_end = l_end;
  L.err = sync; L.claim = sync + P_HDR; L.flags = sync + P_HDR + 8;
  if (L.mode == 1 && avsr::dev_zero(L.n_unfinished, sizeof(int32_t), s) != hipSuccess) return AVSR_ERR_HIP;
  const int B = L.B;
  const int slice = 8 * L.R;                       // utterances per launch: one group per XCD
  for (int b0 = 0; b0 < B; b0 += slice) {
    L.b0 = b0; L.ngroups = ((B - b0 < slice ? B - b0 : slice) + L.R - 1) / L.R;
    if (avsr::dev_zero(sync + P_HDR, sizeof(int32_t) * (words - P_HDR), s) != hipSuccess) return AVSR_ERR_HIP;
    {
      ProfScope ps(dp->prof_tag == 1 ? PROF_ALIGN_PERSIST_FWD : PROF_DEC_PERSIST_FWD, s);
      void* args[] = {(void*)&L};
      if (hipLaunchKernel(dp_kernel(variant, L.mode, L.V > 32), dim3(8 * DP_NW), dim3(DP_NT), args, lds, s) != hipSuccess) return AVSR_ERR_HIP;
    }
    AVSR_CHECK_LAUNCH();
  }
  return AVSR_OK;
}
