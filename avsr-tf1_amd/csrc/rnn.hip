// Host-side wavefront scheduler for the masked multi-layer RNN (encoder stacks).
//
// tf.nn.dynamic_rnn runs layer-by-layer inside one while_loop iteration, T iterations in sequence
// (encoder.py:80, :110).  Cell (layer l, time t) only needs (l-1, t) and (l, t-1), so all cells on
// an anti-diagonal are independent: launch s runs cell (l, t = s - l) of every stack at once.  That
// cuts the sequential chain from n_layers*T launches to T + n_layers - 1, and lets the video and
// audio encoders and both directions of a bidirectional stack share launches.  The backward pass
// (BPTT) runs the mirrored wavefront.
#include "step.h"
#include "avsr_hip.h"

extern "C" int avsr_step_launch_raw(const void* launch, void* stream);
int avsr_rnn_fwd_persistent(const avsr_rnn_stack* st, int32_t n, void* stream, int dry);
int avsr_rnn_bwd_persistent(const avsr_rnn_stack* st, int32_t n, void* stream, int dry);

// Persistent execution: all stacks in one launch when they fit together, else one launch per stack when every
// stack fits on its own (checked first: nothing runs unless everything can), else AVSR_ERR_UNSUPPORTED.
static int run_persistent(int (*fn)(const avsr_rnn_stack*, int32_t, void*, int), const avsr_rnn_stack* st, int32_t n, void* stream) {
  for (int i = 0; i < n; ++i)                      // ResidualWrapper'd layers run through the per-step launches
    for (int l = 0; l < st[i].n_layers && l < AVSR_MAX_LAYERS; ++l)
      if (st[i].layer[l].residual) return AVSR_ERR_UNSUPPORTED;
  int rc = fn(st, n, stream, 0);
  if (rc != AVSR_ERR_UNSUPPORTED || n == 1) return rc;
  for (int i = 0; i < n; ++i)
    if ((rc = fn(st + i, 1, stream, 1)) != AVSR_OK) return rc;
  for (int i = 0; i < n; ++i)
    if ((rc = fn(st + i, 1, stream, 0)) != AVSR_OK) return rc;
  return AVSR_OK;
}

namespace avsr {

static inline float* hbuf(const avsr_rnn_layer& l, int B, int parity) { return l.state + (long)parity * B * l.units; }
static inline float* cbuf(const avsr_rnn_layer& l, int B, int parity) { return l.state + (long)(2 + parity) * B * l.units; }
// dropout only: the layer's output as its consumer sees it (output mask x consumer's input mask), ping-pong
static inline float* xbuf(const avsr_rnn_layer& l, int B, int parity) { return l.state + (long)(4 + parity) * B * l.units; }
static inline void set_cell_dropout(StepTask& tk, const avsr_rnn_stack& S, int l) {
  if (!S.seed) return;
  const uint32_t cid = (uint32_t)(S.cell_id_base + l);
  tk.seed = S.seed;
  tk.k_st = S.keep_state; tk.k_out = S.keep_out; tk.k_in = 1.0f;
  tk.r_st = cid * 4 + 1; tk.r_out = cid * 4 + 2;
  if (l + 1 < S.n_layers) {            // consumer = next layer of the stack: mask over its [units_l] wide input
    tk.k_in = S.keep_in; tk.r_in = (cid + 1) * 4; tk.in_W = S.layer[l].units; tk.in_coff = 0;
  } else if (S.consumer_width > 0) {   // consumer = attention-wrapped layer above the stack (AV-Align)
    tk.k_in = S.consumer_keep; tk.r_in = (uint32_t)S.consumer_stream; tk.in_W = S.consumer_width; tk.in_coff = 0;
  }
}
// dstate: dG rolling [2][B][4H] | dc [2][B][H] | dh_carry [2][B][H]
static inline float* dgroll(const avsr_rnn_layer& l, int B, int parity) { return l.dstate + (long)parity * B * 4 * l.units; }
static inline float* dcbuf(const avsr_rnn_layer& l, int B, int parity) { return l.dstate + (long)(8 + parity) * B * l.units; }
static inline float* dhcarry(const avsr_rnn_layer& l, int B, int parity) { return l.dstate + (long)(10 + parity) * B * l.units; }
// residual layers only: d(emitted output) handed to the layer below, rolling [2][B][H] after the 12*B*H of the rest
static inline float* dresbuf(const avsr_rnn_layer& l, int B, int parity) { return l.dstate + (long)(12 + parity) * B * l.units; }
// GRU dstate: d(gate pre-act) rolling [2][B][2H] | d(cand pre-act) rolling [2][B][H] | carry [2][B][H] | tmp du | tmp dh*u
static inline float* g_dgg(const avsr_rnn_layer& l, int B, int parity) { return l.dstate + (long)parity * B * 2 * l.units; }
static inline float* g_dpc(const avsr_rnn_layer& l, int B, int parity) { return l.dstate + (long)(4 + parity) * B * l.units; }
static inline float* g_carry(const avsr_rnn_layer& l, int B, int parity) { return l.dstate + (long)(6 + parity) * B * l.units; }
static inline float* g_tmp(const avsr_rnn_layer& l, int B, int which) { return l.dstate + (long)(8 + which) * B * l.units; }

}  // namespace avsr

extern "C" int avsr_rnn_fwd(const avsr_rnn_stack* st, int32_t n, void* stream) {
  using namespace avsr;
  if (!st || n <= 0 || n > AVSR_MAX_STACKS) return AVSR_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  {
    int rc = AVSR_ERR_UNSUPPORTED;                                            // one launch for the whole sequence when it fits:
    rc = run_persistent(avsr_rnn_fwd_persistent, st, n, stream);             // 8-row groups per XCD / the agent-scope form
    if (rc != AVSR_ERR_UNSUPPORTED) return rc;
  }
  int nsteps = 0, ntask_max = 0;
  for (int i = 0; i < n; ++i) {
    const avsr_rnn_stack& S = st[i];
    if (S.cell != 0 && S.cell != 1) return AVSR_ERR_UNSUPPORTED;
    if (S.cell != st[0].cell) return AVSR_ERR_ARG;            // one cell kind per call
    if (S.n_layers <= 0 || S.n_layers > AVSR_MAX_LAYERS || S.B <= 0 || S.T <= 0) return AVSR_ERR_ARG;
    for (int l = 0; l < S.n_layers; ++l) {
      const avsr_rnn_layer& Ly = S.layer[l];
      if (Ly.units % 4 || Ly.in_dim % 4 || !Ly.wt || !Ly.gates || !Ly.cs || !Ly.state) return AVSR_ERR_ARG;
      if (Ly.hoisted && l != 0) return AVSR_ERR_ARG;
      if (Ly.residual && (l == 0 || Ly.in_dim != Ly.units || !S.layer[l - 1].out)) return AVSR_ERR_ARG;
      // zero initial state: h parity 0, c parity 0
      if (avsr::dev_zero(hbuf(Ly, S.B, 0), sizeof(float) * S.B * Ly.units, s) != hipSuccess) return AVSR_ERR_HIP;
      if (avsr::dev_zero(cbuf(Ly, S.B, 0), sizeof(float) * S.B * Ly.units, s) != hipSuccess) return AVSR_ERR_HIP;
    }
    nsteps = nsteps > S.T + S.n_layers - 1 ? nsteps : S.T + S.n_layers - 1;
    ntask_max += S.n_layers;
  }
  if (ntask_max > STEP_MAX_TASKS) return AVSR_ERR_UNSUPPORTED;

  static thread_local StepLaunch L;
  const bool gru = st[0].cell == 1;
  for (int step = 0; step < nsteps; ++step) {
   for (int phase = 0; phase < (gru ? 2 : 1); ++phase) {
    L.ntask = 0;
    for (int i = 0; i < n; ++i) {
      const avsr_rnn_stack& S = st[i];
      for (int l = 0; l < S.n_layers; ++l) {
        const int t = step - l;
        if (t < 0 || t >= S.T) continue;
        const avsr_rnn_layer& Ly = S.layer[l];
        StepTask& tk = L.task[L.ntask++];
        tk = StepTask{};
        const int H = Ly.units, in = Ly.in_dim;
        tk.nsrc = 0;
        if (gru) {
          if (!Ly.wt2 || !Ly.rh_seq) return AVSR_ERR_ARG;
          const float* wt = phase == 0 ? Ly.wt : Ly.wt2;
          if (!Ly.hoisted) {
            if (l == 0) return AVSR_ERR_UNSUPPORTED;
            const avsr_rnn_layer& Lo = S.layer[l - 1];
            StepSrc& x = tk.src[tk.nsrc++];
            x.a = (S.seed || Lo.residual) ? xbuf(Lo, S.B, (t + 1) & 1) : hbuf(Lo, S.B, (t + 1) & 1); x.sb = Lo.units; x.K = in; x.w = wt; x.ldw = in + H; x.kind = SRC_PLAIN;
          }
          StepSrc& h = tk.src[tk.nsrc++];
          h.a = phase == 0 ? hbuf(Ly, S.B, t & 1) : hbuf(Ly, S.B, 2) /* r*h */; h.sb = H; h.K = H; h.w = wt + in; h.ldw = in + H; h.kind = SRC_PLAIN;
          tk.B = S.B; tk.t = t; tk.T = S.T; tk.reverse = S.reverse; tk.len = S.len; tk.s2 = Ly.hoisted ? 1 : 0;
          tk.p4 = hbuf(Ly, S.B, t & 1);
          if (phase == 0) {
            tk.N = 2 * H; tk.mode = EP_GRU_GATES; tk.bias = Ly.bias;
            tk.p0 = Ly.gates; tk.p1 = hbuf(Ly, S.B, 2); tk.p2 = Ly.rh_seq;
          } else {
            tk.N = H; tk.mode = EP_GRU_CAND; tk.bias = Ly.bias2;
            tk.p0 = Ly.cs; tk.p1 = Ly.gates;
            tk.p2 = Ly.out ? Ly.out + Ly.ld_out + Ly.out_col : nullptr;
            tk.s0 = (long)(S.T + 2) * Ly.ld_out; tk.s1 = Ly.ld_out;
            tk.p6 = hbuf(Ly, S.B, (t + 1) & 1);
            if (Ly.residual) {   // + raw input = the lower layer's output record (slot 1 = time 0), as for the LSTM below
              const avsr_rnn_layer& Lo = S.layer[l - 1];
              tk.p8 = Lo.out + Lo.ld_out + Lo.out_col; tk.pad0 = (int)((long)(S.T + 2) * Lo.ld_out); tk.pad1 = (int)Lo.ld_out;
            }
            if (S.seed) {
              set_cell_dropout(tk, S, l);
              tk.s4 = (long)(S.T + 2) * H; tk.s5 = H;
              if (Ly.hs_seq) tk.p9 = Ly.hs_seq + H;
              if (l + 1 < S.n_layers) tk.p10 = xbuf(Ly, S.B, (t + 1) & 1);
              if (Ly.xt_seq) tk.p11 = Ly.xt_seq + H;
            } else if (Ly.residual) {
              if (l + 1 < S.n_layers) tk.p10 = xbuf(Ly, S.B, (t + 1) & 1);   // no masks: the plain residual sum, for the layer above
              if (Ly.hs_seq) { tk.p9 = Ly.hs_seq + H; tk.s4 = (long)(S.T + 2) * H; tk.s5 = H; }   // h itself: `out` holds h + x
            }
          }
          continue;
        }
        if (!Ly.hoisted) {
          if (l == 0) return AVSR_ERR_UNSUPPORTED;  // layer 0 input projection must be hoisted (avsr_gemm)
          const avsr_rnn_layer& Lo = S.layer[l - 1];
          StepSrc& x = tk.src[tk.nsrc++];
          // the lower layer's EMITTED output: its h state, unless dropout masks or a residual sum make the two differ
          x.a = (S.seed || Lo.residual) ? xbuf(Lo, S.B, (t + 1) & 1) : hbuf(Lo, S.B, (t + 1) & 1); x.sb = Lo.units; x.K = in; x.w = Ly.wt; x.ldw = in + H; x.kind = SRC_PLAIN;
          if (Ly.residual) {   // + raw input = the lower layer's output record (slot 1 = time 0)
            tk.p8 = Lo.out + Lo.ld_out + Lo.out_col; tk.pad0 = (int)((long)(S.T + 2) * Lo.ld_out); tk.pad1 = (int)Lo.ld_out;
          }
        }
        StepSrc& h = tk.src[tk.nsrc++];
        h.a = hbuf(Ly, S.B, t & 1); h.sb = H; h.K = H; h.w = Ly.wt + in; h.ldw = in + H; h.kind = SRC_PLAIN;
        tk.B = S.B; tk.N = 4 * H; tk.mode = EP_LSTM_FWD;
        tk.t = t; tk.T = S.T; tk.reverse = S.reverse; tk.len = S.len; tk.bias = Ly.bias;
        tk.p0 = Ly.gates; tk.p1 = Ly.cs;
        tk.p2 = Ly.out ? Ly.out + Ly.ld_out /* slot 1 = time 0 */ + Ly.out_col : nullptr;
        tk.s0 = (long)(S.T + 2) * Ly.ld_out; tk.s1 = Ly.ld_out; tk.s2 = Ly.hoisted ? 1 : 0;
        tk.p3 = cbuf(Ly, S.B, t & 1); tk.p4 = hbuf(Ly, S.B, t & 1);
        tk.p5 = cbuf(Ly, S.B, (t + 1) & 1); tk.p6 = hbuf(Ly, S.B, (t + 1) & 1);
        if (S.seed) {
          set_cell_dropout(tk, S, l);
          tk.s4 = (long)(S.T + 2) * H; tk.s5 = H;
          if (Ly.hs_seq) tk.p9 = Ly.hs_seq + H;                       // slot 1 = time 0
          if (l + 1 < S.n_layers) tk.p10 = xbuf(Ly, S.B, (t + 1) & 1);
          if (Ly.xt_seq) tk.p11 = Ly.xt_seq + H;
        } else if (Ly.residual) {
          if (l + 1 < S.n_layers) tk.p10 = xbuf(Ly, S.B, (t + 1) & 1);   // no masks: the plain residual sum, for the layer above
          if (Ly.hs_seq) { tk.p9 = Ly.hs_seq + H; tk.s4 = (long)(S.T + 2) * H; tk.s5 = H; }   // h itself (dWh operand): `out` holds h + x
        }
      }
    }
    if (L.ntask == 0) continue;
    int rc = avsr_step_launch_raw(&L, stream);
    if (rc) return rc;
   }
  }
  for (int i = 0; i < n; ++i) {
    const avsr_rnn_stack& S = st[i];
    for (int l = 0; l < S.n_layers; ++l) {
      const avsr_rnn_layer& Ly = S.layer[l];
      const size_t bytes = sizeof(float) * S.B * Ly.units;
      if (Ly.h_final && avsr::dev_copy(Ly.h_final, hbuf(Ly, S.B, S.T & 1), bytes, s) != hipSuccess)
        return AVSR_ERR_HIP;
      if (!gru && Ly.c_final && avsr::dev_copy(Ly.c_final, cbuf(Ly, S.B, S.T & 1), bytes, s) != hipSuccess)
        return AVSR_ERR_HIP;
    }
  }
  return AVSR_OK;
}

extern "C" int avsr_rnn_bwd(const avsr_rnn_stack* st, int32_t n, void* stream) {
  using namespace avsr;
  if (!st || n <= 0 || n > AVSR_MAX_STACKS) return AVSR_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int nsteps = 0, ntask_max = 0;
  for (int i = 0; i < n; ++i) {
    const avsr_rnn_stack& S = st[i];
    if ((S.cell != 0 && S.cell != 1) || S.cell != st[0].cell) return AVSR_ERR_UNSUPPORTED;
    if (S.n_layers <= 0 || S.n_layers > AVSR_MAX_LAYERS) return AVSR_ERR_ARG;
    for (int l = 0; l < S.n_layers; ++l) {
      const avsr_rnn_layer& Ly = S.layer[l];
      if (!Ly.w || !Ly.dgates || !Ly.dstate || !Ly.gates || !Ly.cs) return AVSR_ERR_ARG;
      const size_t bh = sizeof(float) * S.B * Ly.units;
      // rolling dG (both parities), dc / dh_carry at parity T&1 = gradient of the final state
      if (avsr::dev_zero(Ly.dstate, (Ly.residual ? 14 : 12) * bh, s) != hipSuccess) return AVSR_ERR_HIP;
      if (l == S.n_layers - 1) {
        if (S.cell == 1) {
          if (S.dh_final && avsr::dev_copy(g_carry(Ly, S.B, S.T & 1), S.dh_final, bh, s) != hipSuccess)
            return AVSR_ERR_HIP;
        } else {
          if (S.dc_final && avsr::dev_copy(dcbuf(Ly, S.B, S.T & 1), S.dc_final, bh, s) != hipSuccess)
            return AVSR_ERR_HIP;
          if (S.dh_final && avsr::dev_copy(dhcarry(Ly, S.B, S.T & 1), S.dh_final, bh, s) != hipSuccess)
            return AVSR_ERR_HIP;
        }
      }
    }
    nsteps = nsteps > S.T + S.n_layers - 1 ? nsteps : S.T + S.n_layers - 1;
    ntask_max += S.n_layers;
  }
  if (ntask_max > STEP_MAX_TASKS) return AVSR_ERR_UNSUPPORTED;
  {
    int rc = AVSR_ERR_UNSUPPORTED;                                            // one launch for the whole BPTT when it fits:
    rc = run_persistent(avsr_rnn_bwd_persistent, st, n, stream);
    if (rc != AVSR_ERR_UNSUPPORTED) return rc;
  }

  static thread_local StepLaunch L;
  const bool gru = st[0].cell == 1;
  for (int step = 0; step < nsteps; ++step) {
   for (int phase = 0; phase < (gru ? 2 : 1); ++phase) {
    L.ntask = 0;
    for (int i = 0; i < n; ++i) {
      const avsr_rnn_stack& S = st[i];
      const int nl = S.n_layers;
      for (int l = nl - 1; l >= 0; --l) {
        const int t = S.T - 1 - (step - (nl - 1 - l));
        if (t < 0 || t >= S.T) continue;
        const avsr_rnn_layer& Ly = S.layer[l];
        StepTask& tk = L.task[L.ntask++];
        tk = StepTask{};
        const int H = Ly.units, in = Ly.in_dim;
        if (gru) {
          if (!Ly.w2 || !Ly.dgates2) return AVSR_ERR_ARG;
          // h(t-1) as the cell consumed it: the state-dropped sequence under dropout, else the output sequence
          const bool own_h = S.seed || Ly.residual;        // (a residual layer's output record holds h + x)
          if (own_h && !Ly.hs_seq) return AVSR_ERR_ARG;
          const float* hseq = own_h ? Ly.hs_seq : Ly.out + Ly.out_col;
          const long hld = own_h ? H : Ly.ld_out;
          tk.B = S.B; tk.N = H; tk.t = t; tk.T = S.T; tk.reverse = S.reverse; tk.len = S.len;
          tk.p0 = Ly.gates;
          tk.p2 = const_cast<float*>(hseq) + (S.reverse ? 2 * hld : 0); tk.s2 = (long)(S.T + 2) * hld; tk.s3 = hld;
          tk.p6 = g_tmp(Ly, S.B, 0); tk.p7 = g_tmp(Ly, S.B, 1);
          if (phase == 0) {
            tk.mode = EP_GRU_BWD_CAND;
            StepSrc& a = tk.src[tk.nsrc++];
            a.a = g_dgg(Ly, S.B, (t + 1) & 1); a.sb = 2 * H; a.K = 2 * H; a.w = Ly.w + (long)in * 2 * H; a.ldw = 2 * H; a.kind = SRC_PLAIN;
            if (l + 1 < nl) {
              const avsr_rnn_layer& Up = S.layer[l + 1];
              StepSrc& u1 = tk.src[tk.nsrc++];
              u1.a = g_dgg(Up, S.B, t & 1); u1.sb = 2 * Up.units; u1.K = 2 * Up.units; u1.w = Up.w; u1.ldw = 2 * Up.units; u1.kind = SRC_PLAIN;
              StepSrc& u2 = tk.src[tk.nsrc++];
              u2.a = g_dpc(Up, S.B, t & 1); u2.sb = Up.units; u2.K = Up.units; u2.w = Up.w2; u2.ldw = Up.units; u2.kind = SRC_PLAIN;
            }
            tk.p1 = Ly.cs; tk.p4 = g_carry(Ly, S.B, (t + 1) & 1);
            tk.p5 = g_dpc(Ly, S.B, t & 1); tk.p3 = Ly.dgates2;
            if (Ly.dout) {
              tk.p8 = const_cast<float*>(Ly.dout) + Ly.ld_dout + Ly.dout_col;
              tk.s0 = (long)(S.T + 2) * Ly.ld_dout; tk.s1 = Ly.ld_dout;
            } else if (l + 1 < nl && S.layer[l + 1].residual) {
              tk.p8 = dresbuf(S.layer[l + 1], S.B, t & 1); tk.s0 = H; tk.s1 = 0;   // d(output) through the upper layer's residual sum
            }
            if (Ly.residual) tk.p10 = dresbuf(Ly, S.B, t & 1);
            if (S.seed) {
              set_cell_dropout(tk, S, l);
              if (l + 1 >= nl) tk.k_in = 1.0f;
            }
          } else {
            tk.mode = EP_GRU_BWD_GATES;
            StepSrc& a = tk.src[tk.nsrc++];
            a.a = g_dpc(Ly, S.B, t & 1); a.sb = H; a.K = H; a.w = Ly.w2 + (long)in * H; a.ldw = H; a.kind = SRC_PLAIN;
            tk.p5 = g_carry(Ly, S.B, t & 1); tk.p3 = Ly.dgates; tk.p1 = g_dgg(Ly, S.B, t & 1);
          }
          continue;
        }
        // dh[b,u] = dG_{t+1}(own) . Wh[u,:]  +  dG_t(upper layer) . Wx_upper[u,:]
        StepSrc& a = tk.src[0];
        a.a = dgroll(Ly, S.B, (t + 1) & 1); a.sb = 4 * H; a.K = 4 * H; a.w = Ly.w + (long)in * 4 * H; a.ldw = 4 * H; a.kind = SRC_PLAIN;
        tk.nsrc = 1;
        if (l + 1 < nl) {
          const avsr_rnn_layer& Up = S.layer[l + 1];
          StepSrc& u = tk.src[tk.nsrc++];
          u.a = dgroll(Up, S.B, t & 1); u.sb = 4 * Up.units; u.K = 4 * Up.units; u.w = Up.w; u.ldw = 4 * Up.units; u.kind = SRC_PLAIN;
        }
        tk.B = S.B; tk.N = H; tk.mode = EP_LSTM_BWD;
        tk.t = t; tk.T = S.T; tk.reverse = S.reverse; tk.len = S.len; tk.bias = nullptr;
        tk.p0 = Ly.gates; tk.p1 = Ly.cs; tk.p2 = Ly.dgates; tk.p3 = dgroll(Ly, S.B, t & 1);
        tk.p4 = dcbuf(Ly, S.B, (t + 1) & 1); tk.p5 = dcbuf(Ly, S.B, t & 1);
        tk.p6 = dhcarry(Ly, S.B, (t + 1) & 1); tk.p7 = dhcarry(Ly, S.B, t & 1);
        if (Ly.dout) {
          tk.p8 = const_cast<float*>(Ly.dout) + Ly.ld_dout + Ly.dout_col;  // slot 1 = time 0
          tk.s0 = (long)(S.T + 2) * Ly.ld_dout; tk.s1 = Ly.ld_dout;
        } else if (l + 1 < nl && S.layer[l + 1].residual) {
          tk.p8 = dresbuf(S.layer[l + 1], S.B, t & 1); tk.s0 = H; tk.s1 = 0;   // d(output) through the upper layer's residual sum
        }
        if (Ly.residual) tk.p10 = dresbuf(Ly, S.B, t & 1);
        if (S.seed) {
          set_cell_dropout(tk, S, l);
          if (l + 1 >= nl) tk.k_in = 1.0f;   // no upper layer inside the stack: external d out is already wrt the output
        }
      }
    }
    if (L.ntask == 0) continue;
    int rc = avsr_step_launch_raw(&L, stream);
    if (rc) return rc;
   }
  }
  return AVSR_OK;
}
