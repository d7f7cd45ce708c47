// Per-timestep "step" kernel: the sequential core of the AVSR hot path.
//
// One launch advances every RNN cell that is ready at this point of the layer/time wavefront
// (cells.py:61-102 MultiRNNCell inside encoder.py:80 / :110 tf.nn.dynamic_rnn, and the decoder
// cell inside decoder_*.py dynamic_decode): task i is one (stack, layer) cell at its own time step.
// A task is a small-M GEMM  z[B, N] = sum_s A_s[B, K_s] * W_s^T  on the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32) with a fused epilogue: LSTM gates + cell clip + sequence-length
// masking (forward), the LSTM gate backward (BPTT), or a plain dense layer.
//
// Geometry: grid = (ceil(N/16), ceil(B/16), ntask); 256 threads = 4 waves; the 4 waves split the
// concatenated K range in 16-wide chunks (chunk c -> wave c&3) and reduce through LDS.  Both
// operands are K-contiguous ("NT"), so every lane issues 16-byte loads: lane (i = l&15, q = l>>4)
// loads A[row0+i][16c+4q .. +3] and W[col0+i][16c+4q .. +3]; MFMA e consumes element e, i.e. the
// k-order inside a chunk is permuted identically on both operands (a dot product does not care).
// LSTM weights are stored gate-interleaved (column = 4*unit + {i,j,f,o}), so a 16-column tile is 4
// complete units and the epilogue needs no cross-workgroup exchange.
#include "step.h"
#include "prof.h"
#include "persist.h"

namespace avsr {

static int g_step_geo = -1;   // development override of the workgroup geometry (tools/step_probe.hip)

// ---- the shared MFMA core: returns this wave's partial 16x16 tile -------------------------------
// The concatenated K range (all sources) is cut into 16-wide chunks; wave w of NW owns the CONTIGUOUS
// chunk range [w*NC/NW, (w+1)*NC/NW) so every 128-byte line of an operand row is fetched by one wave.
// Loads are issued in batches of UN chunks (2*UN 16-byte loads per lane in flight) before the MFMAs
// that consume them; four independent accumulators keep the matrix pipe at its 32-cycle issue rate.

template <int KP, int UN, int RM, bool GROUPS>
__device__ __forceinline__ void mm16_partial(const StepTask& tk, int row0, int col0, int kp, bool save_ctx, f32x4 (&out)[RM]) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, q = lane >> 4;
  const int wcol = col0 + i;
  const bool wcol_ok = wcol < tk.N;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[RM][4];
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[r][e] = zero4;
  int NC = 0;
  for (int s = 0; s < tk.nsrc; ++s) NC += (tk.src[s].K + 15) >> 4;
  int g0 = (kp * NC) / KP, g1 = ((kp + 1) * NC) / KP;             // this wave's global chunk range
  if (GROUPS && tk.nsrc > 1) {
    // LSTM backward: source 0 (recurrent path, masked by the STATE dropout) is reduced by the lower half of the
    // waves, every other source (gradient of the cell OUTPUT) by the upper half, so the epilogue can mask them apart
    const int n0 = (tk.src[0].K + 15) >> 4, h = KP / 2;
    if (kp < h) { g0 = (kp * n0) / h; g1 = ((kp + 1) * n0) / h; }
    else { g0 = n0 + ((kp - h) * (NC - n0)) / h; g1 = n0 + ((kp - h + 1) * (NC - n0)) / h; }
  }
  int cbase = 0;
  for (int s = 0; s < tk.nsrc; ++s) {
    const StepSrc& S = tk.src[s];
    const int nch = (S.K + 15) >> 4;
    const int c0 = max(g0 - cbase, 0), c1 = min(g1 - cbase, nch);  // local chunk range of this source
    cbase += nch;
    if (c0 >= c1) continue;
    const float* wp = S.w + (long)wcol * S.ldw + (q << 2);
    if (S.kind == SRC_PLAIN || S.kind == SRC_OWNROW) {
      const float* ap[RM];
      bool aok[RM];
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        const int arow = row0 + 16 * r + i;
        aok[r] = arow < tk.B;
        long rb = arow;
        if (aok[r]) {
          if (S.kind == SRC_OWNROW) {}                    // current-step data of this very row (beam search: not the parent's)
          else if (s == 0 && tk.gather) rb = tk.gather[arow];
          else if (tk.gather2) rb = tk.gather2[arow];
        }
        ap[r] = S.a + rb * S.sb + (q << 2);
      }
      for (int c = c0; c < c1; c += UN) {
        f32x4 av[RM][UN], wv[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
          const int k = (c + j) << 4;
          const bool in = (c + j < c1) && (k + (q << 2) < S.K);
#ifdef PROBE_NO_LOAD
          wv[j] = zero4 + (float)lane;
#pragma unroll
          for (int r = 0; r < RM; ++r) av[r][j] = zero4 + (float)k;
#else
          wv[j] = (in && wcol_ok) ? ld4(wp + k) : zero4;
#pragma unroll
          for (int r = 0; r < RM; ++r) av[r][j] = (in && aok[r]) ? ld4(ap[r] + k) : zero4;
#endif
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
          if (c + j < c1) {
#pragma unroll
            for (int r = 0; r < RM; ++r) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                acc[r][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][j][e], wv[j][e], acc[r][e], 0, 0, 0);
            }
          }
        }
      }
    } else {
      // A row = sum_j wgt_j * slab_j[row]  (attention context from per-chunk softmax partials, or a plain sum).
      // Every load is an unconditional raw buffer load (out-of-range offset = reads 0) so that the statistics and all
      // slabs of a chunk are in flight together; guarded loads compiled to one branch + full vmcnt wait per load.
      const int ns = tk.nslab;
      const __amdgpu_buffer_rsrc_t pm_rs = make_rsrc(tk.pm), pl_rs = make_rsrc(tk.pl), a_rs = make_rsrc(S.a), w_rs = make_rsrc(S.w);
      const int w_off = wcol_ok ? (int)(((long)wcol * S.ldw + (q << 2)) * 4) : P_OOB;
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        const int arow = row0 + 16 * r + i;
        const bool arow_ok = arow < tk.B;
        float wgt[STEP_MAX_SLAB];
        if (S.kind == SRC_SOFTMAX) {
          float pmv[STEP_MAX_SLAB], plv[STEP_MAX_SLAB];
#pragma unroll
          for (int j = 0; j < STEP_MAX_SLAB; ++j) {
            const int o = (j < ns && arow_ok) ? (j * tk.B + arow) * 4 : P_OOB;
            pmv[j] = ldb1(pm_rs, o);
            plv[j] = ldb1(pl_rs, o);
          }
          float M = -INFINITY;
#pragma unroll
          for (int j = 0; j < STEP_MAX_SLAB; ++j) {
            wgt[j] = (j < ns && arow_ok) ? pmv[j] : -INFINITY;
            M = fmaxf(M, wgt[j]);
          }
          float Ls = 0.f;
#pragma unroll
          for (int j = 0; j < STEP_MAX_SLAB; ++j) {
            const float e = (wgt[j] == -INFINITY) ? 0.f : expf(wgt[j] - M);
            Ls += e * plv[j];                  // plv is 0 for slabs that do not exist
            wgt[j] = e;
          }
          const float inv = Ls > 0.f ? 1.0f / Ls : 0.f;
#pragma unroll
          for (int j = 0; j < STEP_MAX_SLAB; ++j) wgt[j] *= inv;
        } else {
#pragma unroll
          for (int j = 0; j < STEP_MAX_SLAB; ++j) wgt[j] = 1.0f;
        }
        const int a_off = (int)(((long)arow * S.sb + (q << 2)) * 4);
        const int slab_b = (int)(tk.slab_stride * 4);
        for (int c = c0; c < c1; ++c) {
          const int k = c << 4;
          const bool kin = k + (q << 2) < S.K;
          f32x4 sv[STEP_MAX_SLAB];
#pragma unroll
          for (int j = 0; j < STEP_MAX_SLAB; ++j)
            sv[j] = ldb4(a_rs, (kin && arow_ok && j < ns) ? a_off + j * slab_b + k * 4 : P_OOB);
          const f32x4 wv = ldb4(w_rs, kin ? w_off + k * 4 : P_OOB);
          f32x4 av = zero4;
#pragma unroll
          for (int j = 0; j < STEP_MAX_SLAB; ++j) av += wgt[j] * sv[j];
          if (kin && arow_ok && save_ctx && tk.ctx_save) st4(tk.ctx_save + (long)arow * tk.ctx_sb + k + (q << 2), av);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[r][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], wv[e], acc[r][e], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RM; ++r) out[r] = (acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]);
}

// fast activations: v_exp_f32 / v_rcp_f32 based (|err| ~1e-7 absolute, far inside the 1e-4 parity budget)
__device__ __forceinline__ float fsigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

// inverted-dropout factor of element idx: 1/keep if kept, 0 if dropped, 1 if dropout is off
__device__ __forceinline__ float drop_scale(const int32_t* seed, uint32_t stream, uint32_t idx, float keep) {
  if (!seed || keep >= 1.0f) return 1.0f;
  return uniform01((uint32_t)seed[0], stream, idx) < keep ? 1.0f / keep : 0.0f;
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if ((act & 3) == 1) return ftanh(v);
  if ((act & 3) == 2) return fsigmoid(v);
  return v;
}

// Workgroup = TM x TN output tiles of 16x16.  A wave owns the TM row tiles of ONE column tile for its
// K-part, so each weight fragment it loads feeds TM MFMAs (weights are the larger operand); the K range
// is split over KP waves and reduced through LDS.  NW = TN * KP waves.
template <int MODE, int TM, int TN, int KP>
__global__ __launch_bounds__(TN * KP * 64) void step_kernel(const StepLaunch L) {
  constexpr int NTILE = TM * TN;
  __shared__ __attribute__((aligned(16))) float red[KP][NTILE][16][16];
  const StepTask& tk = L.task[blockIdx.z];
  if ((int)blockIdx.x * TN * 16 >= tk.N || (int)blockIdx.y * TM * 16 >= tk.B) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = wave % TN, kp = wave / TN;
#ifdef PROBE_EMPTY
  if (tk.B < 0) tk.p5[tid] = 1.f;
  return;
#endif
  const int t = tk.t;

  // ---- epilogue operands are fetched BEFORE the matmul so their latency hides behind it ------------
  // LSTM fwd: one thread per (row, unit): tile e_tile, 64 threads each.  LINEAR / LSTM bwd: one per (row, col).
  constexpr int EPT = (MODE == EP_LSTM_FWD) ? 64 : 256;
  constexpr int NTHR = TN * KP * 64;
  static_assert(NTILE * EPT <= NTHR || true, "");
  // when the workgroup has fewer threads than epilogue items, threads loop (EPI_IT passes)
  constexpr int EPI_IT = (NTILE * EPT + NTHR - 1) / NTHR;
  static_assert(EPI_IT == 1, "geometry must give every epilogue item its own thread");
  const bool epi = tid < NTILE * EPT;
  const int e_tile = tid / EPT, e_l = tid % EPT;
  const int e_mt = e_tile / TN, e_nt = e_tile % TN;
  const int e_row0 = ((int)blockIdx.y * TM + e_mt) * 16, e_col0 = ((int)blockIdx.x * TN + e_nt) * 16;
  bool e_ok = false, valid = false;
  int b = 0, n = 0, len = 0, tau = 0;
  float pre0 = 0.f, pre1 = 0.f, pre2 = 0.f, pre3 = 0.f, pre2b = 0.f;
  f32x4 pre4a = {0.f, 0.f, 0.f, 0.f}, pre4b = {0.f, 0.f, 0.f, 0.f};
  if (epi) {
    if constexpr (MODE == EP_LSTM_FWD) {
      b = e_row0 + (e_l >> 2);
      n = (e_col0 >> 2) + (e_l & 3);                 // unit index
      e_ok = b < tk.B && e_col0 < tk.N;
      if (e_ok) {
        const int H = tk.N >> 2;
        len = tk.len ? tk.len[b] : tk.T;
        valid = t < len;
        tau = tk.reverse ? len - 1 - t : t;
        const long gb = tk.gather2 ? tk.gather2[b] : b;   // beam search: previous state lives in the parent's row
        pre0 = tk.p3[gb * H + n];                    // c_prev
        pre1 = tk.p4[gb * H + n];                    // h_prev
        if (valid) {
          if (tk.bias) pre4a = ld4(tk.bias + n * 4);
          if (tk.s2) pre4b = ld4(tk.p0 + (((long)b * tk.T + tau) * H + n) * 4);
        }
      }
    } else if constexpr (MODE == EP_LSTM_BWD) {
      b = e_row0 + (e_l >> 4);
      n = e_col0 + (e_l & 15);
      e_ok = b < tk.B && n < tk.N;
      if (e_ok) {
        const int H = tk.N;
        len = tk.len ? tk.len[b] : tk.T;
        valid = t < len;
        tau = tk.reverse ? len - 1 - t : t;
        const long bh = (long)b * H + n;
        pre0 = tk.p4[bh];                            // dc_in
        pre1 = tk.p6 ? tk.p6[bh] : 0.f;              // dh carry
        if (valid) {
          const long bt = (long)b * tk.T + tau;
          pre4a = ld4(tk.p0 + (bt * H + n) * 4);     // gates i, j, f, o
          pre2 = tk.p1[bt * H + n];                  // c
          if (t == 0) pre3 = tk.bias ? tk.bias[bh] : 0.f;
          else pre3 = tk.p1[(bt + (tk.reverse ? 1 : -1)) * H + n];   // c_prev
          if (tk.p8) pre2b = tk.p8[(long)b * tk.s0 + (long)tau * tk.s1 + n];   // external d out
          // + per-chunk partial gradients of the attention query (attention backward emits one partial per memory chunk):
          // summed here, in chunk order, instead of by a reduction launch in the decoder's sequential chain.
          // pm / nslab = first partial set [nslab][B][H], pl / pad0 = second set; unconditional loads (out of range = 0).
          if (tk.pm) {
            const __amdgpu_buffer_rsrc_t r0 = make_rsrc(tk.pm), r1 = make_rsrc(tk.pl);
            const int o = (b * H + n) * 4, cs = tk.B * H * 4;
            float v0[STEP_MAX_SLAB], v1[STEP_MAX_SLAB];
#pragma unroll
            for (int c = 0; c < STEP_MAX_SLAB; ++c) {
              v0[c] = ldb1(r0, c < tk.nslab ? o + c * cs : P_OOB);
              v1[c] = ldb1(r1, (tk.pl && c < tk.pad0) ? o + c * cs : P_OOB);
            }
#pragma unroll
            for (int c = 0; c < STEP_MAX_SLAB; ++c) pre2b += v0[c];
#pragma unroll
            for (int c = 0; c < STEP_MAX_SLAB; ++c) pre2b += v1[c];
          }
        }
      }
    } else if constexpr (MODE == EP_LINEAR) {
      b = e_row0 + (e_l >> 4);
      n = e_col0 + (e_l & 15);
      e_ok = b < tk.B && n < tk.N;
      if (e_ok) {
        if (tk.bias) pre0 = tk.bias[n];
        if (tk.p1) pre0 += tk.p1[(long)b * tk.s1 + n];
        valid = !(tk.len && t >= tk.len[b]);
      }
    }
  }

  const int row0 = (int)blockIdx.y * TM * 16, col0 = ((int)blockIdx.x * TN + nt) * 16;
  constexpr int UN = (TM >= 4) ? 2 : (TM == 2 ? 4 : 8);
  f32x4 acc[TM];
  mm16_partial<KP, UN, TM, MODE == EP_LSTM_BWD || MODE == EP_GRU_BWD_CAND>(tk, row0, col0, kp, blockIdx.x == 0 && nt == 0, acc);
  // C/D layout of mfma 16x16: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[kp][m * TN + nt][(lane >> 4) * 4 + r][lane & 15] = acc[m][r];
  __syncthreads();
#ifdef PROBE_NO_EPI
  if (tid < 256 && row0 + (tid >> 4) < tk.B) tk.p5[(long)(row0 + (tid >> 4)) * 16 + (tid & 15)] = red[0][0][tid >> 4][tid & 15] + red[KP - 1][NTILE - 1][tid >> 4][tid & 15];
  return;
#endif
  if constexpr (MODE >= EP_GRU_GATES) {
    // ===================== GRU (rnn_cell_impl.GRUCell, cells.py:25-29) =====================
    //   [r,u] = sigmoid([x,h] Wg + bg);  c~ = tanh([x, r*h] Wc + bc);  h' = u*h + (1-u)*c~
    // Two launches per time step forward (gates, candidate) and backward (candidate path, gate path); gate columns are
    // unit-interleaved (col = 2*unit + {0:r, 1:u}).  Records: gates [B,T,H,2], c~ [B,T,H], r*h [B,T,H].
    constexpr int GEPT = (MODE == EP_GRU_GATES) ? 128 : 256;
    if (tid >= NTILE * GEPT) return;
    const int gt = tid / GEPT, gl = tid % GEPT;
    const int g_row0 = ((int)blockIdx.y * TM + gt / TN) * 16, g_col0 = ((int)blockIdx.x * TN + gt % TN) * 16;
    if constexpr (MODE == EP_GRU_GATES) {
      // p0 gates record (+ hoisted x.Wg if s2), p4 h_in, p1 r*h out [B,H], p2 r*h record [B,T,H]
      const int r = gl >> 3, ul = gl & 7;
      const int bb = g_row0 + r, H = tk.N >> 1, u = (g_col0 >> 1) + ul;
      if (bb >= tk.B || u >= H) return;
      const int ln = tk.len ? tk.len[bb] : tk.T;
      if (!(t < ln)) return;
      const int ta = tk.reverse ? ln - 1 - t : t;
      const long bt = (long)bb * tk.T + ta;
      float zr = 0.f, zu = 0.f;
#pragma unroll
      for (int w = 0; w < KP; ++w) { zr += red[w][gt][r][2 * ul]; zu += red[w][gt][r][2 * ul + 1]; }
      if (tk.bias) { zr += tk.bias[2 * u]; zu += tk.bias[2 * u + 1]; }
      float* gp = tk.p0 + (bt * H + u) * 2;
      if (tk.s2) { zr += gp[0]; zu += gp[1]; }
      const float rr = fsigmoid(zr), uu = fsigmoid(zu);
      gp[0] = rr; gp[1] = uu;
      const float rh = rr * tk.p4[(long)(tk.gather2 ? tk.gather2[bb] : bb) * H + u];
      tk.p1[(long)bb * H + u] = rh;
      if (tk.p2) tk.p2[bt * H + u] = rh;
      return;
    }
    const int bb = g_row0 + (gl >> 4), nn = g_col0 + (gl & 15), H = tk.N;
    if (bb >= tk.B || nn >= H) return;
    const int ln = tk.len ? tk.len[bb] : tk.T;
    const bool ok = t < ln;
    const int ta = tk.reverse ? ln - 1 - t : t;
    const long bt = (long)bb * tk.T + ta, bh = (long)bb * H + nn;
    float z = 0.f, zB = 0.f;
    if (MODE == EP_GRU_BWD_CAND && tk.nsrc > 1) {
#pragma unroll
      for (int w = 0; w < KP / 2; ++w) z += red[w][gt][gl >> 4][gl & 15];
#pragma unroll
      for (int w = KP / 2; w < KP; ++w) zB += red[w][gt][gl >> 4][gl & 15];
    } else {
#pragma unroll
      for (int w = 0; w < KP; ++w) z += red[w][gt][gl >> 4][gl & 15];
    }
    if constexpr (MODE == EP_GRU_CAND) {
      // p0 c~ record (+ hoisted x.Wc if s2), p1 gates record, p2 seq_out (s0,s1), p4 h_in, p6 h_out, p9/p10/p11 as LSTM fwd
      const float hprev = tk.p4[(long)(tk.gather2 ? tk.gather2[bb] : bb) * H + nn];
      if (ok) {
        if (tk.bias) z += tk.bias[nn];
        if (tk.s2) z += tk.p0[bt * H + nn];
        const float cand = ftanh(z), uu = tk.p1[(bt * H + nn) * 2 + 1];
        const float h = uu * hprev + (1.f - uu) * cand;
        tk.p0[bt * H + nn] = cand;
        const uint32_t oidx = (uint32_t)(bt * H + nn);
        float ho = h * drop_scale(tk.seed, tk.r_out, oidx, tk.k_out);
        // ResidualWrapper (cells.py:91-92): emitted output = (dropped) cell output + the layer's raw input (the lower layer's output
        // record: p8 base, pad0 batch stride, pad1 time stride), as in the LSTM epilogue
        if (tk.p8) ho += tk.p8[(long)bb * tk.pad0 + (long)ta * tk.pad1 + nn];
        const float hs = h * drop_scale(tk.seed, tk.r_st, oidx, tk.k_st);
        if (tk.p2) tk.p2[(long)bb * tk.s0 + (long)ta * tk.s1 + nn] = ho;
        tk.p6[bh] = hs;
        if (tk.p9) tk.p9[(long)bb * tk.s4 + (long)ta * tk.s5 + nn] = hs;
        if (tk.p10 || tk.p11) {
          const float xn = ho * drop_scale(tk.seed, tk.r_in, (uint32_t)(bt * tk.in_W + tk.in_coff + nn), tk.k_in);
          if (tk.p10) tk.p10[bh] = xn;
          if (tk.p11) tk.p11[(long)bb * tk.s4 + (long)ta * tk.s5 + nn] = xn;
        }
      } else {
        tk.p6[bh] = hprev;
        if (t < tk.T) {
          if (tk.p2) tk.p2[(long)bb * tk.s0 + (long)t * tk.s1 + nn] = 0.f;
          if (tk.p9) tk.p9[(long)bb * tk.s4 + (long)t * tk.s5 + nn] = 0.f;
          if (tk.p11) tk.p11[(long)bb * tk.s4 + (long)t * tk.s5 + nn] = 0.f;
        }
      }
    } else if constexpr (MODE == EP_GRU_BWD_CAND) {
      // z = d(state h) from the gate path of step t+1 (source 0) [+ zB = d(output) from the layer above]
      // p0 gates rec, p1 c~ rec, p2 h_prev sequence (s2 batch stride, s3 time stride; index tau), p4 carry_in, p8 d out (s0,s1),
      // p5 d(c~ pre-act) rolling out [B,H], p3 d(c~ pre-act) record, p6 tmp d(u pre-act) [B,H], p7 tmp dh*u [B,H]
      const float carry = tk.p4[bh];
      if (ok) {
        const uint32_t oidx = (uint32_t)(bt * H + nn);
        float dout = zB * drop_scale(tk.seed, tk.r_in, (uint32_t)(bt * tk.in_W + tk.in_coff + nn), tk.k_in);
        if (tk.p8) dout += tk.p8[(long)bb * tk.s0 + (long)ta * tk.s1 + nn];
        if (tk.p10) tk.p10[bh] = dout;          // ResidualWrapper: the same gradient also reaches the lower layer's output
        const float dh = dout * drop_scale(tk.seed, tk.r_out, oidx, tk.k_out) + (z + carry) * drop_scale(tk.seed, tk.r_st, oidx, tk.k_st);
        const float uu = tk.p0[(bt * H + nn) * 2 + 1], cand = tk.p1[bt * H + nn];
        const float hprev = tk.p2[(long)bb * tk.s2 + (long)ta * tk.s3 + nn];
        const float dpc = dh * (1.f - uu) * (1.f - cand * cand);
        tk.p5[bh] = dpc;
        tk.p3[bt * H + nn] = dpc;
        tk.p6[bh] = dh * (hprev - cand) * uu * (1.f - uu);
        tk.p7[bh] = dh * uu;
      } else {
        tk.p5[bh] = 0.f;
        if (t < tk.T) tk.p3[((long)bb * tk.T + t) * H + nn] = 0.f;
        tk.p6[bh] = 0.f;
        tk.p7[bh] = carry;
      }
    } else {
      // EP_GRU_BWD_GATES: z = d(r*h) = d(c~ pre-act) . Wc_h^T.  p0 gates rec, p2 h_prev sequence (s2,s3), p6 tmp d(u pre-act),
      // p7 tmp dh*u, p5 carry_out [B,H], p3 d(gate pre-act) record [B,T,H,2], p1 d(gate pre-act) rolling out [B,2H]
      float dr = 0.f, du = 0.f;
      const float dhdir = tk.p7[bh];
      if (ok) {
        const float rr = tk.p0[(bt * H + nn) * 2];
        const float hprev = tk.p2[(long)bb * tk.s2 + (long)ta * tk.s3 + nn];
        dr = z * hprev * rr * (1.f - rr);
        du = tk.p6[bh];
        tk.p5[bh] = dhdir + z * rr;
        float* rec = tk.p3 + (bt * H + nn) * 2;
        rec[0] = dr; rec[1] = du;
      } else {
        tk.p5[bh] = dhdir;
        if (t < tk.T) { float* rec = tk.p3 + (((long)bb * tk.T + t) * H + nn) * 2; rec[0] = 0.f; rec[1] = 0.f; }
      }
      tk.p1[bh * 2] = dr; tk.p1[bh * 2 + 1] = du;
    }
    return;
  }
  if (!epi || !e_ok) return;

  if constexpr (MODE == EP_LINEAR) {
    // p0 out (row stride s0), p1 add (row stride s1)
    float z = 0.f;
#pragma unroll
    for (int w = 0; w < KP; ++w) z += red[w][e_tile][e_l >> 4][e_l & 15];
    const uint32_t midx = (uint32_t)(((long)b * tk.T + t) * tk.in_W + tk.in_coff + n);
    if (tk.act & 8) z *= drop_scale(tk.seed, tk.r_in, midx, tk.k_in);   // gradient through an input-dropout mask
    z = apply_act(z + pre0, tk.act);
    if (!valid) z = 0.f;                     // dynamic_rnn / impute_finished: zero output past the valid length
    tk.p0[(long)b * tk.s0 + n] = z;
    if (tk.p9) tk.p9[(long)b * tk.s4 + n] = z * drop_scale(tk.seed, tk.r_in, midx, tk.k_in);   // pre-dropped copy for the consumer
  } else if constexpr (MODE == EP_LSTM_BWD) {
    // n = unit.  p0 gates, p1 cs, p2 dgates record, p3 dG rolling out [B,4H], p4 dc_in, p5 dc_out,
    // p6 dh_carry_in, p7 dh_carry_out, p8 dout (s0 batch stride, s1 time stride), bias = c_init [B,H]
    const int H = tk.N;
    const long bh = (long)b * H + n;
    f32x4 dg = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      // zA: recurrent path (gradient of the state h), zB: gradient of the emitted output from upper layer / attention
      float zA = 0.f, zB = 0.f;
      if (tk.nsrc > 1) {
#pragma unroll
        for (int w = 0; w < KP / 2; ++w) zA += red[w][e_tile][e_l >> 4][e_l & 15];
#pragma unroll
        for (int w = KP / 2; w < KP; ++w) zB += red[w][e_tile][e_l >> 4][e_l & 15];
      } else {
#pragma unroll
        for (int w = 0; w < KP; ++w) zA += red[w][e_tile][e_l >> 4][e_l & 15];
      }
      const uint32_t oidx = (uint32_t)(((long)b * tk.T + tau) * H + n);
      const uint32_t iidx = (uint32_t)(((long)b * tk.T + tau) * tk.in_W + tk.in_coff + n);
      const float dout = pre2b + zB * drop_scale(tk.seed, tk.r_in, iidx, tk.k_in);
      const float dh = dout * drop_scale(tk.seed, tk.r_out, oidx, tk.k_out) +
                       (zA + pre1) * drop_scale(tk.seed, tk.r_st, oidx, tk.k_st);
      if (tk.p10) tk.p10[bh] = dout;          // ResidualWrapper: the same gradient also reaches the lower layer's output
      const f32x4 g = pre4a;
      const float c = pre2, cprev = pre3;
      const float tc = ftanh(c);
      float dc = dh * g[3] * (1.f - tc * tc) + pre0;
      if (!(fabsf(c) < 1.0f)) dc = 0.f;      // cell_clip=1.0: no gradient through a clipped cell
      dg[3] = dh * tc * g[3] * (1.f - g[3]);
      dg[0] = dc * g[1] * g[0] * (1.f - g[0]);
      dg[1] = dc * g[0] * (1.f - g[1] * g[1]);
      dg[2] = dc * cprev * g[2] * (1.f - g[2]);
      tk.p5[bh] = dc * g[2];
      if (tk.p7) tk.p7[bh] = 0.f;
      st4(tk.p2 + (((long)b * tk.T + tau) * H + n) * 4, dg);
    } else {
      tk.p5[bh] = pre0;
      if (tk.p7) tk.p7[bh] = pre1;
      if (t < tk.T) st4(tk.p2 + (((long)b * tk.T + t) * H + n) * 4, dg);  // padding position t: zero record
    }
    st4(tk.p3 + bh * 4, dg);
  } else {
    // EP_LSTM_FWD: p0 gates/zpre [B,T,H,4], p1 cs [B,T,H], p2 seq_out (s0 batch stride, s1 time stride),
    // p3 c_in, p4 h_in, p5 c_out, p6 h_out; s2 = has_zpre
    const int H = tk.N >> 2;
    const long bh = (long)b * H + n;
    const float cprev = pre0, hprev = pre1;
    if (valid) {
      const long bt = (long)b * tk.T + tau;
      f32x4 z = pre4a + pre4b;
      const int r = e_l >> 2, ul = e_l & 3;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        float zz = 0.f;
#pragma unroll
        for (int w = 0; w < KP; ++w) zz += red[w][e_tile][r][ul * 4 + gi];
        z[gi] += zz;
      }
      f32x4 g;
      g[0] = fsigmoid(z[0]);
      g[1] = ftanh(z[1]);
      g[2] = fsigmoid(z[2] + 1.0f);          // forget_bias = 1.0
      g[3] = fsigmoid(z[3]);
      float c = g[2] * cprev + g[0] * g[1];
      c = fminf(1.0f, fmaxf(-1.0f, c));      // cell_clip = 1.0 (cells.py:16)
      const float h = g[3] * ftanh(c);
      st4(tk.p0 + (bt * H + n) * 4, g);
      tk.p1[bt * H + n] = c;
      // DropoutWrapper: emitted output and recurrent h carry independent masks; c is never dropped
      const uint32_t oidx = (uint32_t)(bt * H + n);
      float ho = h * drop_scale(tk.seed, tk.r_out, oidx, tk.k_out);
      // ResidualWrapper (cells.py:91-92): emitted output = (dropped) cell output + the layer's raw input, read from the
      // lower layer's output record: p8 base, pad0 batch stride, pad1 time stride
      if (tk.p8) ho += tk.p8[(long)b * tk.pad0 + (long)tau * tk.pad1 + n];
      const float hs = h * drop_scale(tk.seed, tk.r_st, oidx, tk.k_st);
      if (tk.p2) tk.p2[(long)b * tk.s0 + (long)tau * tk.s1 + n] = ho;
      tk.p5[bh] = c;
      tk.p6[bh] = hs;
      if (tk.p9) tk.p9[(long)b * tk.s4 + (long)tau * tk.s5 + n] = hs;          // state-dropped h sequence (dWh operand)
      if (tk.p10 || tk.p11) {                                                   // consumer's (input-dropped) view of the output
        const float xn = ho * drop_scale(tk.seed, tk.r_in, (uint32_t)(bt * tk.in_W + tk.in_coff + n), tk.k_in);
        if (tk.p10) tk.p10[bh] = xn;
        if (tk.p11) tk.p11[(long)b * tk.s4 + (long)tau * tk.s5 + n] = xn;
      }
    } else {
      tk.p5[bh] = cprev;
      tk.p6[bh] = hprev;
      if (t < tk.T) {                                                            // zero output past len
        if (tk.p2) tk.p2[(long)b * tk.s0 + (long)t * tk.s1 + n] = 0.f;
        if (tk.p9) tk.p9[(long)b * tk.s4 + (long)t * tk.s5 + n] = 0.f;
        if (tk.p11) tk.p11[(long)b * tk.s4 + (long)t * tk.s5 + n] = 0.f;
      }
    }
  }
}

}  // namespace avsr

extern "C" int avsr_step_launch_raw(const void* launch, void* stream) {
  using namespace avsr;
  const StepLaunch* L = (const StepLaunch*)launch;
  if (!L || L->ntask <= 0 || L->ntask > STEP_MAX_TASKS) return AVSR_ERR_ARG;
  int maxN = 0, maxB = 0;
  for (int i = 0; i < L->ntask; ++i) {
    if (L->task[i].N > maxN) maxN = L->task[i].N;
    if (L->task[i].B > maxB) maxB = L->task[i].B;
  }
  const int mode = L->task[0].mode;
  int ksum = 0;
  double flops = 0.0;                                    // algorithmic FLOPs of this launch (event profiler)
  for (int i = 0; i < L->ntask; ++i) {
    if (L->task[i].mode != mode) return AVSR_ERR_ARG;   // one epilogue kind per launch
    int k = 0;
    for (int j = 0; j < L->task[i].nsrc; ++j) k += L->task[i].src[j].K;
    if (k > ksum) ksum = k;
    flops += 2.0 * L->task[i].B * L->task[i].N * k;
  }
  hipStream_t s = (hipStream_t)stream;
  // geometry (TM, TN, KP): probe sweep on MI355X (tools/step_probe.hip, B=64 H=256 3-task wavefront):
  //   fwd (K<=768): (1,1,4) 8.1 us  vs (1,1,8) 11.0, (2,1,8) 10.1, 2x2-tile variants 11-15 us
  //   bwd (K>=1024): (1,1,8) 10.6 us vs (1,1,4) 12.1-13.6, (2,1,8) 15.5, (1,1,16) 12.2 us
  // i.e. many small workgroups win; sharing operands inside a workgroup costs more latency than it saves.
  int geo = (ksum >= 1024) ? 1 : 0;
  if (g_step_geo >= 0) geo = g_step_geo;
#define LAUNCH_(M, K_)                                                                                         \
  {                                                                                                            \
    ProfScope ps(K_, s, flops);                                                                                    \
    switch (geo) {                                                                                             \
      case 0: hipLaunchKernelGGL((step_kernel<M, 1, 1, 4>), GRID(1, 1), dim3(256), 0, s, *L); break;           \
      case 1: hipLaunchKernelGGL((step_kernel<M, 1, 1, 8>), GRID(1, 1), dim3(512), 0, s, *L); break;           \
      default: return AVSR_ERR_ARG;                                                                            \
    }                                                                                                          \
  }
#define GRID(tm, tn) dim3((maxN + 16 * (tn) - 1) / (16 * (tn)), (maxB + 16 * (tm) - 1) / (16 * (tm)), L->ntask)
  if (mode == EP_LSTM_FWD) LAUNCH_(EP_LSTM_FWD, PROF_STEP_LSTM_FWD)
  else if (mode == EP_LSTM_BWD) LAUNCH_(EP_LSTM_BWD, PROF_STEP_LSTM_BWD)
  else if (mode == EP_LINEAR) LAUNCH_(EP_LINEAR, PROF_STEP_LINEAR)
  else if (mode == EP_GRU_GATES) LAUNCH_(EP_GRU_GATES, PROF_STEP_LSTM_FWD)
  else if (mode == EP_GRU_CAND) LAUNCH_(EP_GRU_CAND, PROF_STEP_LSTM_FWD)
  else if (mode == EP_GRU_BWD_CAND) LAUNCH_(EP_GRU_BWD_CAND, PROF_STEP_LSTM_BWD)
  else if (mode == EP_GRU_BWD_GATES) LAUNCH_(EP_GRU_BWD_GATES, PROF_STEP_LSTM_BWD)
  else return AVSR_ERR_UNSUPPORTED;
#undef LAUNCH_
#undef GRID
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" void avsr_step_set_geometry(int geo) { avsr::g_step_geo = geo; }
