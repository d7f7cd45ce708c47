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
// launches): GRU, multi-layer decoder cells, Bahdanau mechanisms, beam search, H or D > 256, V > 32, memories that
// do not fit the resident budget.
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
#define DP_MISC 4032      // floats of LDS ahead of the resident value rows
#define DP_LDS_BYTES 163840

namespace avsr {

struct DPMech {
  const float* keys; const float* values; long values_sb, values_st; const int* len; const float* g;
  const float* watt_t;
  float* scores; float* ctx; float* pstat;      // records [B][L][T], [B][L][D], [L][2][nc_rec][B]
  float* ppm; float* ppl; float* ppctx;         // quarter partials [4][B], [4][B], [4][B][D]
  int T, D, type, nc_rec, ch, lds_off;
};

struct DPLaunch {
  int B, L, H, E, V, n_mech, mode, oa;
  int l_begin, l_end, b0, ngroups;
  int go_id, eos_id, A, KW;
  int UW, AW, NWA, drop;
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
__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

// KR0 / KR1: register capacity of the resident keys of mechanism 0 / 1, in frames per 16-lane group (32 groups per workgroup)
template <int KR0, int KR1, int MODE>
__global__ __launch_bounds__(DP_NT) void dec_persist_kernel(const DPLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const red = lds;                       // [8][2][8][16] cell / [8][256] context partials / [8][8][16] attention layer
  float* const s_p = lds + 2048;                // [2][128] scores -> softmax numerators of this quarter (P2) ...
  float* const s_logit = lds + 2048;            // ... [8][32] logits of the group's rows (P4)
  float* const s_x = lds + 2304;                // [8][128] input rows of the next step
  float* const s_att = lds + 3328;              // [8][16]  this workgroup's attention columns
  float* const s_redw = lds + 3456;             // [32] wave partials of the block reductions
  int* const s_int = reinterpret_cast<int*>(lds + 3488);   // [0..7] tokens, [8..15] step lengths, [16] slot, [17] unfinished
  float* const s_wo = lds + 3520;               // [16][32] this workgroup's rows of the output kernel
  constexpr int ZPAD = 3512;                    // 4 zero floats (tail of s_int): where operand chunks of another source "read" LDS
  float* const vals = lds + DP_MISC;            // resident value rows of this workgroup's quarter

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = __builtin_amdgcn_readfirstlane(xcc_id());
  if (tid == 0) s_int[16] = __hip_atomic_fetch_add(L.claim + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int j = __builtin_amdgcn_readfirstlane(s_int[16]);
  if (g >= L.ngroups || j >= DP_NW) return;

  const int B = L.B, Ls = L.L, H = L.H, E = L.E, A = L.A, KW = L.KW, V = L.V;
  constexpr int mode = MODE;
  const int i = lane & 15, q = lane >> 4;
  const int s16 = tid & 15, rg = tid >> 4;      // attention: 16 lanes per frame, 32 frames per pass
  const int rowbase = L.b0 + g * DP_R;
  const bool drop = L.drop != 0;
  const uint32_t seedv = L.seed ? (uint32_t)L.seed[0] : 0u;
  const uint32_t cid4 = L.cid4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---------------------------------------------------------------------------------------------------------
  // resident operands
  // ---------------------------------------------------------------------------------------------------------
  // (1) cell kernel slice: gate columns of units [unit0, unit0 + UW); K split over the 8 waves
  const int UW = L.UW, unit0 = j * UW, col0 = unit0 * 4;
  const int nx = (mode == 0) ? 0 : (E + 15) >> 4, na = (A + 15) >> 4, nh = (H + 15) >> 4;
  const int NC = nx + na + nh;
  const int g0 = (wave * NC) / DP_WV, ncw = ((wave + 1) * NC) / DP_WV - g0;
  // MFMA A-operand row of this lane (rows 8..15 of the tile are padding)
  const int ab = rowbase + i;
  const bool aok = i < DP_R && ab < B;
  f32x4 wc[DP_CPW][2];
  // Branch-free operand fetch: every chunk slot issues one load per possible source (attention record, h state, LDS input
  // rows) each step; a slot's byte offset is P_OOB (reads 0, no memory access) / the LDS zero pad for the sources it is not of.
  unsigned ko[DP_CPW];
  int xa[DP_CPW], seg_att[DP_CPW];              // seg_att: wave-uniform "this slot reads the attention record" (else the h state)
#pragma unroll
  for (int cc = 0; cc < DP_CPW; ++cc) {
    const int gch = g0 + cc;
    int kk, kseg, wbase, seg;
    if (gch < nx) { kk = gch << 4; kseg = E; wbase = 0; seg = 0; }
    else if (gch < nx + na) { kk = (gch - nx) << 4; kseg = A; wbase = E; seg = 1; }
    else { kk = (gch - nx - na) << 4; kseg = H; wbase = E + A; seg = 2; }
    const int k = kk + 4 * q;
    const bool in = cc < ncw && k < kseg;
    ko[cc] = (in && seg != 0 && aok) ? (unsigned)(k * 4) : (unsigned)P_OOB;
    seg_att[cc] = __builtin_amdgcn_readfirstlane(seg == 1 ? 1 : 0);
    xa[cc] = (in && seg == 0 && i < DP_R) ? 2304 + i * 128 + k : ZPAD;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int cl = nt * 16 + i, col = col0 + cl;
      wc[cc][nt] = (in && cl < 4 * UW && col < 4 * H) ? ld4(L.wt + (long)col * KW + wbase + k) : zero4;
    }
  }
  // (2) attention-layer slice: attention columns [an0, an0 + AW) of mechanism ma
  const int AW = L.AW, an0 = j * AW;
  const bool has_att = an0 < A;
  const int ma = has_att ? an0 / H : 0;
  const int nl0 = an0 - ma * H;
  const DPMech& Ma = L.m[ma];
  const int Da = Ma.D;
  const int nq = (H + 15) >> 4, nd = (Da + 15) >> 4, NCa = nq + nd;
  const int ga0 = (wave * NCa) / DP_WV, ncaw = ((wave + 1) * NCa) / DP_WV - ga0;
  f32x4 wa[DP_APW];
  unsigned ko3[DP_APW];
  int seg_q[DP_APW];                             // wave-uniform "this slot reads the cell output" (else the context partials)
#pragma unroll
  for (int cc = 0; cc < DP_APW; ++cc) {
    const int gch = ga0 + cc;
    const bool qs = gch < nq;
    const int kk = qs ? gch << 4 : (gch - nq) << 4, kseg = qs ? H : Da, wbase = qs ? 0 : H;
    const int k = kk + 4 * q;
    const bool in = has_att && cc < ncaw && k < kseg;
    ko3[cc] = (in && aok) ? (unsigned)(k * 4) : (unsigned)P_OOB;
    seg_q[cc] = __builtin_amdgcn_readfirstlane(qs ? 1 : 0);
    wa[cc] = (in && i < AW) ? ld4(Ma.watt_t + (long)(nl0 + i) * (H + Da) + wbase + k) : zero4;
  }
  // (3) output-layer rows [an0, an0 + AW) -> LDS [k][symbol]
  {
    const int k = tid >> 5, v = tid & 31;
    s_wo[tid] = (mode >= 1 && L.oa && has_att && v < V && k < AW) ? L.wout_t[(long)v * A + an0 + k] : 0.f;
  }
  // (4) attention memories of row r_att, quarter cq: keys -> registers (16 lanes per frame), values -> LDS
  const int r_att = j >> 2, cq = j & 3;
  const int b_att = rowbase + r_att;
  const bool att_row = b_att < B;
  f32x4 k0[KR0 > 0 ? KR0 : 1][4], k1[KR1 > 0 ? KR1 : 1][4];
  int n_m[2] = {0, 0}, t0_m[2] = {0, 0};
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

  // ---------------------------------------------------------------------------------------------------------
  // per-thread roles and state
  // ---------------------------------------------------------------------------------------------------------
  // cell epilogue: thread e < 8*UW owns (row er, unit eu)
  const int er = tid / UW, eu = tid - er * UW;
  const int eb = rowbase + er, eun = unit0 + eu;
  const bool eok = tid < DP_R * UW && eb < B && eun < H;
  const long BH = (long)B * H;
  float* const hbuf = L.state;                  // [2][B][H] ping-pong, then c [2][B][H]
  float* const cbuf = L.state + 2 * BH;
  float c_state = 0.f, h_state = 0.f;
  f32x4 bias4 = zero4;
  int e_steplen = 0;
  if (eok) {
    c_state = cbuf[(long)(L.l_begin & 1) * BH + (long)eb * H + eun];
    h_state = hbuf[(long)(L.l_begin & 1) * BH + (long)eb * H + eun];
    if (L.bias) bias4 = ld4(L.bias + eun * 4);
    e_steplen = L.steplen[eb];
  }
  const __amdgpu_buffer_rsrc_t ctx_rs = make_rsrc(Ma.ctx);
  const __amdgpu_buffer_rsrc_t att_rs = make_rsrc(drop ? L.attd : L.att), h_rs = make_rsrc(hbuf), co_rs = make_rsrc(L.cell_out);
  const __amdgpu_buffer_rsrc_t pc_rs = make_rsrc(Ma.ppctx), pm_rs = make_rsrc(Ma.ppm), pl_rs = make_rsrc(Ma.ppl);
  const __amdgpu_buffer_rsrc_t plog_rs = make_rsrc(L.plog);
  // attention-layer epilogue: thread < 8*AW owns (row ar, column ac)
  const int ar = tid / AW, ac = tid - ar * AW;
  const int arb = rowbase + ar;
  const bool a_ok = has_att && tid < DP_R * AW && arb < B;
  int a_steplen = a_ok ? L.steplen[arb] : 0;
  // logits / sampling: thread (pr, pv), tid < 256; per-row tokens and step lengths of the group in s_int
  const int pr = (tid >> 5) & 7, pv = tid & 31;
  const bool p_on = tid < 256;
  const int pb = rowbase + pr;
  const float bout_v = (mode >= 1 && L.bout && pv < V) ? L.bout[pv] : 0.f;
  if (tid < DP_R) {
    const int b = rowbase + tid;
    s_int[tid] = (b < B && mode == 1) ? L.tok[b] : 0;
    s_int[8 + tid] = (b < B) ? L.steplen[b] : 0;
  }
  if (tid == 0) s_int[17] = 0;
  if (tid < 4) lds[ZPAD + tid] = 0.f;
  __syncthreads();
  // input rows of the first step of this call
  if (mode >= 1) {
    const int e4n = E >> 2;
    for (int idx = tid; idx < DP_R * e4n; idx += DP_NT) {
      const int r = idx / e4n, e4 = idx - r * e4n, b = rowbase + r;
      f32x4 v = zero4;
      if (b < B) v = (mode == 2) ? ld4(L.xs + ((long)b * Ls + L.l_begin) * E + 4 * e4) : ld4(L.embedding + (long)s_int[r] * E + 4 * e4);
      st4(s_x + r * 128 + 4 * e4, v);
    }
  }
  __syncthreads();

  int* const flag_base = L.flags + g * 3 * 32;
  auto wait_all = [&](int phase, int need) {
    if (wave == 0) {
      const int* fp = flag_base + phase * 32 + (lane & 31);
      bool ok = false;
      for (int spins = 0; spins < (1 << 21); ++spins) {
        const int v = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v >= need)) { ok = true; break; }
        if ((spins & 1023) == 1023 && __hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = true; break; }
      }
      if (!ok && lane == 0) __hip_atomic_store(L.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    lds_barrier();
  };
  auto publish = [&](int phase, int value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag_base + phase * 32 + j, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  for (int l = L.l_begin; l < L.l_end; ++l) {
    const int epoch = l - L.l_begin + 1;
    // =====================================================================================================
    // P1: LSTM cell (cells.py:14-18 LSTMCell clip 1.0, forget bias 1.0; DropoutWrapper cells.py:46-54)
    // =====================================================================================================
    {
      f32x4 zpre = zero4;
      if (mode == 0 && eok && l < e_steplen) zpre = ld4(L.gates + (((long)eb * Ls + l) * H + eun) * 4);
      f32x4 av[DP_CPW];
      const unsigned att_o = (unsigned)(((long)ab * (Ls + 1) + l) * A) * 4u, h_o = (unsigned)((long)(l & 1) * BH + (long)ab * H) * 4u;
#pragma unroll
      for (int cc = 0; cc < DP_CPW; ++cc)
        av[cc] = ldb_sc1(seg_att[cc] ? att_rs : h_rs, (int)((seg_att[cc] ? att_o : h_o) + ko[cc]));
      f32x4 acc[2] = {zero4, zero4};
#pragma unroll
      for (int cc = 0; cc < DP_CPW; ++cc) {
        f32x4 a4 = av[cc];                                         // an input-row slot read 0 from memory: its operand is in LDS
        if (MODE != 0) a4 += ld4(lds + xa[cc]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], wc[cc][nt][e], acc[nt], 0, 0, 0);
      }
      if (q < 2) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((wave * 2 + nt) * 8 + q * 4 + r) * 16 + i] = acc[nt][r];
      }
      lds_barrier();
      if (eok) {
        if (mode == 1) e_steplen = s_int[8 + er];
        const bool valid = l < e_steplen;
        const long bt = (long)eb * Ls + l;
        const long so = ((long)eb * (Ls + 1) + l + 1) * H + eun;
        float hnext = h_state;
        if (valid) {
          f32x4 z = bias4 + zpre;
#pragma unroll
          for (int gi = 0; gi < 4; ++gi) {
            const int cc = eu * 4 + gi;
            const int o = ((cc >> 4) * 8 + er) * 16 + (cc & 15);
            z[gi] += ((red[o] + red[256 + o]) + (red[512 + o] + red[768 + o])) + ((red[1024 + o] + red[1280 + o]) + (red[1536 + o] + red[1792 + o]));
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
      publish(0, epoch);
    }
    // =====================================================================================================
    // P2: scores, masked softmax partials, partial contexts of (row r_att, quarter cq)
    //     (attention.py:25-72 Luong / scaled Luong; contrib.seq2seq _compute_attention)
    // =====================================================================================================
    wait_all(0, epoch);
    {
      f32x4 q4[4];
      const int qo = (int)(((long)b_att * (Ls + 1) + l + 1) * H);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int k = 4 * s16 + 64 * jj;
        q4[jj] = ldb_sc1(co_rs, (att_row && k < H) ? (qo + k) * 4 : P_OOB);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (m >= L.n_mech) continue;
        const DPMech& M = L.m[m];
        const float gsc = (M.type == ATT_SCALED_LUONG) ? M.g[0] : 1.f;
        float* const srow = M.scores + ((long)b_att * Ls + l) * M.T + t0_m[m];
        float* const sc = s_p + m * 128;
        const int n = n_m[m];
        if (m == 0) {
#pragma unroll
          for (int u = 0; u < KR0; ++u) {
            float a = (dot4(k0[u][0], q4[0]) + dot4(k0[u][1], q4[1])) + (dot4(k0[u][2], q4[2]) + dot4(k0[u][3], q4[3]));
            a = group16_sum(a);
            const int fr = rg + 32 * u;
            if (s16 == 0 && fr < n) { srow[fr] = a; sc[fr] = a * gsc; }
          }
        } else {
#pragma unroll
          for (int u = 0; u < KR1; ++u) {
            float a = (dot4(k1[u][0], q4[0]) + dot4(k1[u][1], q4[1])) + (dot4(k1[u][2], q4[2]) + dot4(k1[u][3], q4[3]));
            a = group16_sum(a);
            const int fr = rg + 32 * u;
            if (s16 == 0 && fr < n) { srow[fr] = a; sc[fr] = a * gsc; }
          }
        }
      }
      lds_barrier();
      // quarter max / exp / sum of both mechanisms together (one element per thread: ch <= 128)
      float mx[2], lsum[2], pnum[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const float v = (tid < n_m[m]) ? s_p[m * 128 + tid] : -INFINITY;
        pnum[m] = v;
        const float wm = wave_max(v);
        if (lane == 0) s_redw[m * 8 + wave] = wm;
      }
      lds_barrier();
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float t = s_redw[m * 8];
#pragma unroll
        for (int w = 1; w < DP_WV; ++w) t = fmaxf(t, s_redw[m * 8 + w]);
        mx[m] = t;
        const float p = (tid < n_m[m]) ? expf(pnum[m] - t) : 0.f;
        if (tid < n_m[m]) s_p[m * 128 + tid] = p;
        const float ws = wave_sum(p);
        if (lane == 0) s_redw[16 + m * 8 + wave] = ws;
      }
      lds_barrier();
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < DP_WV; ++w) t += s_redw[16 + m * 8 + w];
        lsum[m] = t;
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (m >= L.n_mech) continue;
        const DPMech& M = L.m[m];
        const int D = M.D, d4 = D >> 2, n = n_m[m];
        if (tid == 0 && att_row) {
          M.ppm[(long)cq * B + b_att] = (n > 0) ? mx[m] : -INFINITY;
          M.ppl[(long)cq * B + b_att] = lsum[m];
        }
        if (lane < d4) {
          f32x4 a4 = zero4;
          const float* vp = vals + M.lds_off + 4 * lane;
          const float* pp = s_p + m * 128;
          for (int fr = wave; fr < n; fr += DP_WV) a4 += pp[fr] * ld4(vp + fr * D);
          st4(red + wave * 256 + 4 * lane, a4);
        }
        lds_barrier();
        if (tid < d4 && att_row) {
          f32x4 s = ld4(red + 4 * tid);
#pragma unroll
          for (int w = 1; w < DP_WV; ++w) s += ld4(red + w * 256 + 4 * tid);
          st4(M.ppctx + ((long)cq * B + b_att) * D + 4 * tid, s);
        }
        lds_barrier();
      }
      publish(1, epoch);
    }
    // =====================================================================================================
    // P3: attention layer att_m = [cell_out, ctx_m] . W_att,m (attention.py:173-181), split-K share of the logits
    // =====================================================================================================
    wait_all(1, epoch);
    {
      if (has_att) {
        // softmax merge weights of this lane's row over the four quarters
        float wgt[4], Mx = -INFINITY, Lsum = 0.f;
        {
          float pmv[4], plv[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int o = aok ? (c * B + ab) * 4 : P_OOB;
            pmv[c] = ld1_sc1(pm_rs, o);
            plv[c] = ld1_sc1(pl_rs, o);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) { wgt[c] = aok ? pmv[c] : -INFINITY; Mx = fmaxf(Mx, wgt[c]); }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float e = (wgt[c] == -INFINITY) ? 0.f : expf(wgt[c] - Mx);
            Lsum += e * plv[c];
            wgt[c] = e;
          }
          const float inv = Lsum > 0.f ? 1.0f / Lsum : 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) wgt[c] *= inv;
        }
        const bool saver = nl0 == 0;                                   // first workgroup of a mechanism keeps the records
        if (saver && aok && q == 0 && wave == 0) {                     // merged statistics in chunk 0, neutral elsewhere
          float* pmr = Ma.pstat + (long)(2 * l) * Ma.nc_rec * B;
          float* plr = Ma.pstat + (long)(2 * l + 1) * Ma.nc_rec * B;
          pmr[ab] = Mx; plr[ab] = Lsum;
          for (int c = 1; c < Ma.nc_rec; ++c) { pmr[(long)c * B + ab] = -INFINITY; plr[(long)c * B + ab] = 0.f; }
        }
        const unsigned qo = (unsigned)(((long)ab * (Ls + 1) + l + 1) * H) * 4u;
        const unsigned cso = (unsigned)(((long)ab * Ls + l) * Da) * 4u;
        const unsigned pcs = (unsigned)((long)B * Da) * 4u, pc0 = (unsigned)((long)ab * Da) * 4u;
        f32x4 acc = zero4;
        // two slots at a time: slot = 1 load of the cell output, or the 4 quarter partials of the context (merged here)
#pragma unroll
        for (int c0 = 0; c0 < DP_APW; c0 += 2) {
          f32x4 sv[2][4];
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const int cc = c0 + d;
            sv[d][0] = ldb_sc1(seg_q[cc] ? co_rs : pc_rs, (int)((seg_q[cc] ? qo : pc0) + ko3[cc]));
#pragma unroll
            for (int c = 1; c < 4; ++c) sv[d][c] = ldb_sc1(pc_rs, seg_q[cc] ? P_OOB : (int)(pc0 + c * pcs + ko3[cc]));
          }
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const int cc = c0 + d;
            const f32x4 a4 = ((seg_q[cc] ? 1.0f : wgt[0]) * sv[d][0] + wgt[1] * sv[d][1]) + (wgt[2] * sv[d][2] + wgt[3] * sv[d][3]);
            stb4(ctx_rs, (saver && !seg_q[cc]) ? (int)(cso + ko3[cc]) : P_OOB, a4);   // context record (out of range: dropped)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], wa[cc][e], acc, 0, 0, 0);
          }
        }
        if (q < 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(wave * 8 + q * 4 + r) * 16 + i] = acc[r];
        }
      }
      lds_barrier();
      if (tid < DP_R * AW && has_att) {
        float a = 0.f;
        if (a_ok) {
          if (mode == 1) a_steplen = s_int[8 + ar];
          const int o = ar * 16 + ac;
          a = ((red[o] + red[128 + o]) + (red[256 + o] + red[384 + o])) + ((red[512 + o] + red[640 + o]) + (red[768 + o] + red[896 + o]));
          if (!(l < a_steplen)) a = 0.f;
          const long ao = ((long)arb * (Ls + 1) + l + 1) * A + an0 + ac;
          L.att[ao] = a;
          if (drop) L.attd[ao] = a * p_drop(true, seedv, cid4, (uint32_t)(((long)arb * Ls + l + 1) * (E + A) + E + an0 + ac), L.k_in);
        }
        s_att[ar * 16 + ac] = a;
      }
      lds_barrier();
      if (mode >= 1 && L.oa && has_att && p_on && pv < V) {
        float s = 0.f;
        for (int k = 0; k < AW; ++k) s += s_att[pr * 16 + k] * s_wo[k * 32 + pv];
        L.plog[(((long)g * DP_NW + j) * DP_R + pr) * 32 + pv] = s;
      }
      publish(2, epoch);
    }
    // =====================================================================================================
    // P4: logits, sample / arg-max, next input rows (decoder_unimodal.py:304-309 ScheduledEmbeddingTrainingHelper,
    //     :176-217 GreedyEmbeddingHelper + dynamic_decode(impute_finished=True))
    // =====================================================================================================
    wait_all(2, epoch);
    if (mode >= 1) {
      if (p_on) {
        float z = 0.f;
#pragma unroll
        for (int w0 = 0; w0 < DP_NW; w0 += 8) {
          float part[8];
#pragma unroll
          for (int w = 0; w < 8; ++w) {
            const int o = (w0 + w < L.NWA && pv < V) ? (int)(((((long)g * DP_NW + w0 + w) * DP_R + pr) * 32 + pv) * 4) : P_OOB;
            part[w] = ld1_sc1(plog_rs, o);
          }
#pragma unroll
          for (int w = 0; w < 8; ++w) z += part[w];
        }
        const bool valid = l < s_int[8 + pr];
        z = valid ? z + bout_v : 0.f;
        if (pv < V) {
          s_logit[pr * 32 + pv] = z;
          if (j == 0 && pb < B) L.logits[((long)pb * Ls + l) * V + pv] = z;
        }
      }
      lds_barrier();
      if (tid < DP_R) {
        const int r = tid, b = rowbase + r;
        const float* lg = s_logit + r * 32;
        if (mode == 1) {
          int id = 0;
          bool unfin = false;
          if (b < B && l < s_int[8 + r]) {
            float best = lg[0];
            for (int v = 1; v < V; ++v)
              if (lg[v] > best) { best = lg[v]; id = v; }       // first maximum (tf.argmax)
            s_int[r] = id;
            if (id == L.eos_id) s_int[8 + r] = l + 1; else unfin = true;
          }
          if (j == 0 && b < B) L.ids[(long)b * Ls + l] = id;
          if (l == L.l_end - 1) {
            const int cnt = __popcll(__ballot(unfin));
            if (tid == 0) s_int[17] = cnt;
          }
        } else if (l + 1 < Ls && b < B) {
          const uint32_t idx = (uint32_t)(b * Ls + l);
          int tk = L.labels[(long)b * Ls + l];
          if (L.prob > 0.f && uniform01(seedv, 1000u, idx) < L.prob) {
            float m0 = lg[0];
            for (int v = 1; v < V; ++v) m0 = fmaxf(m0, lg[v]);
            float tot = 0.f;
            for (int v = 0; v < V; ++v) tot += expf(lg[v] - m0);
            const float target = uniform01(seedv, 1001u, idx) * tot;
            float run = 0.f;
            tk = V - 1;
            for (int v = 0; v < V; ++v) {
              run += expf(lg[v] - m0);
              if (run > target) { tk = v; break; }
            }
          }
          s_int[r] = tk;
          if (j == 0) L.fed[(long)b * Ls + l + 1] = tk;
        }
      }
      lds_barrier();
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
    }
  }
  // ---- state back to the ping-pong buffers (the next call / the caller's h_final, c_final copies read them) ----
  if (eok) cbuf[(long)(L.l_end & 1) * BH + (long)eb * H + eun] = c_state;
  if (mode == 1 && j == 0) {
    if (tid < DP_R && rowbase + tid < B) {
      L.steplen[rowbase + tid] = s_int[8 + tid];
      L.tok[rowbase + tid] = s_int[tid];
    }
    if (tid == 0 && s_int[17] > 0) atomicAdd(L.n_unfinished, s_int[17]);
  }
}

static int g_dec_fused = 1;

// variant: 0 = one mechanism (<= 128 frames per quarter); 1 = (<= 32, <= 128); 2 = (<= 128, <= 32) frames per quarter
static const void* dp_kernel(int variant, int mode) {
#define DPK(a, b) (mode == 0 ? (const void*)dec_persist_kernel<a, b, 0> : mode == 1 ? (const void*)dec_persist_kernel<a, b, 1> : (const void*)dec_persist_kernel<a, b, 2>)
  return variant == 0 ? DPK(4, 0) : variant == 1 ? DPK(1, 4) : DPK(4, 1);
#undef DPK
}

static inline int dp_quarter(int T) { return (T + DP_WPR - 1) / DP_WPR; }

// Fills L for the descriptor; returns AVSR_ERR_UNSUPPORTED when the fused kernel does not cover it.
static int dp_plan(const avsr_attn_rnn& d, DPLaunch& L, int* variant, size_t* lds_bytes) {
  if (!g_dec_fused || !g_sync || !d.fused_ws) return AVSR_ERR_UNSUPPORTED;
  if (d.cell != 0 || d.n_extra != 0 || d.n_mech < 1 || d.n_mech > 2 || d.mode < 0 || d.mode > 2) return AVSR_ERR_UNSUPPORTED;
  const int B = d.B, H = d.H, E = d.E, A = d.n_mech * H, KW = E + A + H;
  if (H > 256 || H % 4 || E % 4 || E > 128 || d.V > 32) return AVSR_ERR_UNSUPPORTED;
  if (d.mode >= 1 && !d.output_attention) return AVSR_ERR_UNSUPPORTED;
  const int nx = d.mode == 0 ? 0 : (E + 15) / 16, NC = nx + (A + 15) / 16 + (H + 15) / 16;
  if ((NC + DP_WV - 1) / DP_WV > DP_CPW) return AVSR_ERR_UNSUPPORTED;
  const int UW = (H + DP_NW - 1) / DP_NW, AW = (A + DP_NW - 1) / DP_NW;
  if (UW > 8 || AW > 16 || H % AW) return AVSR_ERR_UNSUPPORTED;
  if ((long)B * (d.L + 1) * (A > H ? A : H) >= (1L << 29) || (long)4 * B * 256 >= (1L << 29)) return AVSR_ERR_UNSUPPORTED;
  if (avsr_attn_rnn_fused_ws_floats(B, d.n_mech, 256) > d.fused_ws_floats) return AVSR_ERR_UNSUPPORTED;
  L = DPLaunch{};
  L.B = B; L.L = d.L; L.H = H; L.E = E; L.V = d.V; L.n_mech = d.n_mech; L.mode = d.mode; L.oa = d.output_attention;
  L.go_id = d.go_id; L.eos_id = d.eos_id; L.A = A; L.KW = KW; L.UW = UW; L.AW = AW; L.NWA = (A + AW - 1) / AW;
  L.drop = (d.seed && d.mode != 1 && (d.keep_in < 1.f || d.keep_state < 1.f || d.keep_out < 1.f)) ? 1 : 0;
  L.wt = d.wt; L.bias = d.bias; L.gates = d.gates; L.cs = d.cs; L.cell_out = d.cell_out; L.att = d.att; L.attd = d.attd;
  L.hs_seq = d.hs_seq; L.state = d.state; L.steplen = d.steplen;
  L.embedding = d.embedding; L.wout_t = d.wout_t; L.bout = d.bout; L.logits = d.logits; L.ids = d.ids; L.tok = d.tok;
  L.n_unfinished = d.n_unfinished; L.xs = d.xs; L.labels = d.labels; L.fed = d.fed;
  L.seed = d.seed; L.k_in = d.keep_in; L.k_st = d.keep_state; L.k_out = d.keep_out; L.prob = d.sampling_prob;
  L.cid4 = (uint32_t)d.cell_id * 4;
  if (L.drop && (!d.hs_seq || !d.attd)) return AVSR_ERR_UNSUPPORTED;
  float* ws = d.fused_ws;
  L.plog = ws; ws += (long)((B + DP_R - 1) / DP_R) * DP_NW * DP_R * 32;
  int lds_off = 0;
  for (int m = 0; m < d.n_mech; ++m) {
    const avsr_attn_mech& M = d.mech[m];
    if (M.type > ATT_SCALED_LUONG || M.D > 256 || M.D % 4 || M.T <= 0) return AVSR_ERR_UNSUPPORTED;
    if ((H + 15) / 16 + (M.D + 15) / 16 > DP_WV * DP_APW) return AVSR_ERR_UNSUPPORTED;
    if ((long)B * M.T * H >= (1L << 29)) return AVSR_ERR_UNSUPPORTED;
    DPMech& X = L.m[m];
    X.keys = M.keys; X.values = M.values; X.values_sb = M.values_sb; X.values_st = M.values_st; X.len = M.len; X.g = M.g;
    X.watt_t = M.watt_t; X.scores = M.scores; X.ctx = M.ctx; X.pstat = M.pstat;
    X.T = M.T; X.D = M.D; X.type = M.type; X.nc_rec = (M.T + M.chunk - 1) / M.chunk; X.ch = dp_quarter(M.T);
    if (X.ch > 128) return AVSR_ERR_UNSUPPORTED;
    X.lds_off = lds_off; lds_off += X.ch * M.D;
    X.ppm = ws; ws += 4L * B; X.ppl = ws; ws += 4L * B; X.ppctx = ws; ws += 4L * B * M.D;
  }
  const size_t bytes = sizeof(float) * ((size_t)DP_MISC + lds_off);
  if (bytes > DP_LDS_BYTES) return AVSR_ERR_UNSUPPORTED;
  *lds_bytes = bytes;
  // register-resident key capacity (32 frames per pass): variant 0 = one mechanism up to 128 frames per quarter;
  // 1 = (<= 32, <= 128); 2 = (<= 128, <= 32)
  if (d.n_mech == 1) *variant = 0;
  else if (L.m[0].ch <= 32 && L.m[1].ch <= 128) *variant = 1;
  else if (L.m[0].ch <= 128 && L.m[1].ch <= 32) *variant = 2;
  else return AVSR_ERR_UNSUPPORTED;
  return AVSR_OK;
}

}  // namespace avsr

extern "C" int64_t avsr_attn_rnn_fused_ws_floats(int32_t B, int32_t n_mech, int32_t Dmax) {
  const int64_t groups = (B + DP_R - 1) / DP_R;
  return groups * DP_NW * DP_R * 32 + (int64_t)n_mech * (8L * B + 4L * B * Dmax) + 64;
}

extern "C" int avsr_attn_rnn_set_fused(int32_t on) { avsr::g_dec_fused = on ? 1 : 0; return AVSR_OK; }

extern "C" int avsr_attn_rnn_fused_eligible(const avsr_attn_rnn* d) {
  using namespace avsr;
  if (!d) return 0;
  static thread_local DPLaunch L;
  int variant = 0; size_t lds = 0;
  return dp_plan(*d, L, &variant, &lds) == AVSR_OK ? 1 : 0;
}

// Steps [l_begin, l_end) of the decoder as one persistent launch per 64-row slice.  The caller (avsr_attn_rnn_fwd) has
// initialised the state / slot-0 records exactly as for the per-step path and copies h_final / c_final afterwards.
int avsr_dec_persist_fwd(const avsr_attn_rnn* dp, int32_t l_begin, int32_t l_end, void* stream) {
  using namespace avsr;
  static thread_local DPLaunch L;
  int variant = 0; size_t lds = 0;
  const int rc = dp_plan(*dp, L, &variant, &lds);
  if (rc) return rc;
  if (l_begin >= l_end) return AVSR_OK;
  hipStream_t s = (hipStream_t)stream;
  int32_t* sync = g_sync;
  const long words = P_HDR + 8 + 8 * 3 * 32;
  if (words > g_sync_ints) return AVSR_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    for (int v = 0; v < 3; ++v)
      for (int md = 0; md < 3; ++md)
        if (hipFuncSetAttribute(dp_kernel(v, md), hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS_BYTES) != hipSuccess) return AVSR_ERR_HIP;
    attr_set = true;
  }
  L.l_begin = l_begin; L.l_end = l_end;
  L.err = sync; L.claim = sync + P_HDR; L.flags = sync + P_HDR + 8;
  if (L.mode == 1 && avsr::dev_zero(L.n_unfinished, sizeof(int32_t), s) != hipSuccess) return AVSR_ERR_HIP;
  const int B = L.B;
  for (int b0 = 0; b0 < B; b0 += 64) {
    L.b0 = b0; L.ngroups = ((B - b0 < 64 ? B - b0 : 64) + DP_R - 1) / DP_R;
    if (avsr::dev_zero(sync + P_HDR, sizeof(int32_t) * (words - P_HDR), s) != hipSuccess) return AVSR_ERR_HIP;
    {
      ProfScope ps(PROF_DEC_PERSIST_FWD, s);
      void* args[] = {(void*)&L};
      if (hipLaunchKernel(dp_kernel(variant, L.mode), dim3(8 * DP_NW), dim3(DP_NT), args, lds, s) != hipSuccess) return AVSR_ERR_HIP;
    }
    AVSR_CHECK_LAUNCH();
  }
  return AVSR_OK;
}
