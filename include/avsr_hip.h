/* avsr_hip.h -- C ABI of libavsr_hip.so, the MI355X (gfx950) engine for the AVSR seq2seq hot path.
 *
 * The reference (georgesterpu/avsr-tf1) has no FFI / plugin interface: every op of the hot path is a
 * TensorFlow-1.13 graph node built from Python.  This header therefore draws the boundary at the
 * granularity of the TensorFlow sequence primitives the reference calls, one entry point per
 * primitive, cited below.  Host code (the avsr_tf1_amd package) stays Python and binds these with ctypes
 * (see INTEGRATION.md); PyTorch tensors are only the container for device memory.
 *
 * Conventions: plain pointers and sizes; all pointers are DEVICE pointers unless stated; fp32
 * row-major, batch-major [B, T, F]; lengths int32; `stream` is a hipStream_t passed as void*.
 * Every function returns 0 on success or a negative AVSR_ERR_* code; nothing throws or aborts, no
 * function allocates or synchronises (all are hipGraph-capturable).
 *
 * Internal weight layout ("engine layout", produced by avsr_tf1_amd/params.py from TF layout):
 *   LSTM kernel  W  [in+H][H][4]   = TF kernel [in+H][4H] with column (g*H + u) moved to (u*4 + g),
 *                                    gate order i, j, f, o kept (rnn_cell_impl.LSTMCell)
 *   LSTM kernel  Wt [H*4][in+H]    = transpose of W (forward operand)
 *   dense kernels: TF [in][out] ("w") and transposed [out][in] ("wt")
 */
#ifndef AVSR_HIP_H
#define AVSR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libavsr_hip.so is built with -fvisibility=hidden: exactly the functions declared in this header are exported. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define AVSR_OK 0
#define AVSR_ERR_ARG (-1)
#define AVSR_ERR_HIP (-2)
#define AVSR_ERR_UNSUPPORTED (-3)

#define AVSR_MAX_LAYERS 4
#define AVSR_MAX_MECH 4
#define AVSR_MAX_STACKS 4
#define AVSR_HAVE_ATTN 1
#define AVSR_MAX_TRANSPOSE 16
#define AVSR_MAX_SEGMENTS 32

int avsr_abi_version(void);
/* sizeof() of a struct of this header by name ("avsr_attn_rnn", ...), -1 if unknown: lets an FFI binding verify its
 * struct layouts against the library before the first launch (avsr_tf1_amd/_lib.py does). */
int64_t avsr_sizeof(const char* name);

/* ---------------------------------------------------------------------------------------------
 * Dense GEMM (fp32 MFMA).  Replaces tf.matmul / tf.layers.Dense over all B*T rows: hoisted
 * LSTMCell input projections (avsr/cells.py:14-18 via avsr/encoder.py:80,:110), attention
 * memory_layer (avsr/attention.py:26-72), output Dense (avsr/decoder_unimodal.py:112), and the
 * matching tf.gradients GEMMs (avsr/seq2seq.py:222).
 *   C = alpha * op(A) * op(B) + beta * C + bias
 * Row r of a stored matrix lives at ptr + (T ? (r / T) * ldo + (r % T) * ld : r * ld).
 * trans_a: A stored [K][M];  trans_b: B stored [N][K].
 * splitk > 1 needs workspace of batch*splitk*M*N floats (deterministic two-pass reduction). */
typedef struct avsr_mat {
  float* ptr;
  int64_t ld;
  int32_t T;
  int32_t pad_;
  int64_t ldo;
} avsr_mat;

typedef struct avsr_gemm_desc {
  avsr_mat A, B, C;
  const float* bias;
  int32_t M, N, K;
  int32_t trans_a, trans_b;
  float alpha, beta;
  int32_t batch;
  int64_t stride_a, stride_b, stride_c;
  const float* alpha_dev;       /* optional device scalar multiplied into alpha (learned attention scale g) */
  int32_t splitk;
  int32_t pad_;
  float* workspace;
  int64_t workspace_floats;
  /* optional: colsum[n] = colsum_beta * colsum[n] + sum_k op(B)[k][n], taken from the B tiles while they pass through the kernel (the bias
   * gradient of a layer whose weight gradient this GEMM is: seq2seq.py:222 -- the same d gates, one pass less over them).  trans_b == 0
   * and batch == 1 only; with split-K the workspace must hold splitk * N more floats. */
  float* colsum;
  float colsum_beta;
  int32_t pad2_;
} avsr_gemm_desc;

int avsr_gemm(const avsr_gemm_desc* d, void* stream);
/* n INDEPENDENT GEMMs (no output of one is an operand or the output of another; split-K workspaces disjoint) issued side by side:
 * entries of one operand-layout class share a launch (and one split-K reduction launch) -- the many few-workgroup matmuls of a train
 * step (state bridges decoder_bimodal.py:480-490, per-memory attention gradients, the row blocks of a cell kernel's gradient,
 * seq2seq.py:222) cost one dependent launch each when issued one by one.  Same per-entry semantics as avsr_gemm; n <= 64. */
int avsr_gemm_batch(const avsr_gemm_desc* descs, int32_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-layer masked RNN over a sequence.  Replaces tf.nn.dynamic_rnn(MultiRNNCell(LSTMCell...),
 * sequence_length=...) and each direction of tf.nn.bidirectional_dynamic_rnn
 * (avsr/encoder.py:80-88, :110-119; cells from avsr/cells.py:61-102).  Several independent stacks
 * (video/audio, forward/backward direction) advance in ONE launch per wavefront step.
 *
 * Semantics (tf rnn.py _rnn_step): for t >= len[b] the output row is zero and the state is copied
 * through; reverse=1 processes utterance b in the order len[b]-1 .. 0 (array_ops.reverse_sequence)
 * and stores results at their original time positions.
 *
 * Buffers per layer (caller-allocated):
 *   gates  [B][T][H][4]  forward: activated gates i,j,f,o.  If hoisted=1 it must hold x*Wx(+0 bias)
 *                        on entry (bias is added by the kernel).
 *   cs     [B][T][H]     cell states (post clip)
 *   out    [B][T+2][ld_out] (+out_col)  slot s = time s-1; caller keeps slot 0 and slots > len zero.
 *   state  scratch 4*B*H floats (h and c ping-pong)
 *   dgates [B][T][H][4]  backward: d(pre-activation)
 *   dstate scratch (2*4 + 2 + 2)*B*H floats
 *   dout   gradient wrt out (same slot layout/stride as out), top layer only (others NULL)
 */
typedef struct avsr_rnn_layer {
  int32_t units, in_dim, hoisted, out_col;
  const float* wt;
  const float* w;
  const float* bias;
  float* gates;
  float* cs;
  float* out;
  int64_t ld_out;
  float* state;
  float* h_final;
  float* c_final;
  float* dgates;
  float* dstate;
  const float* dout;
  int64_t ld_dout;
  int32_t dout_col;
  int32_t residual;             /* ResidualWrapper around this layer's cell (cells.py:91-92; layers > 0, LSTM or GRU, units == in_dim):
                                 * emitted output = cell output + layer input.  Runs through the per-step launches; `state`
                                 * must hold 6*B*units and `dstate` 14*B*units floats, and hs_seq must be given (it then records
                                 * the recurrent h even without dropout: `out` holds h + input). */
  /* DropoutWrapper buffers (NULL when dropout is off): both [B][T+2][units], slot s = time s-1 */
  float* hs_seq;                /* state-dropped h (what the next time step consumed): dWh operand */
  float* xt_seq;                /* output as seen by the consumer above (output mask x its input mask): dWx operand */
  /* GRU (stack.cell == 1; rnn_cell_impl.GRUCell, avsr/cells.py:25-29).  wt/w/bias = gate kernel [in+H][2H] with unit-
   * interleaved columns (2*unit + {0:r,1:u}) and its transpose; wt2/w2/bias2 = candidate kernel [in+H][H].
   * `gates` holds [B][T][H][2] (hoisted x.Wg_x on entry), `cs` the candidate record [B][T][H] (hoisted x.Wc_x on entry),
   * rh_seq [B][T][H] = r*h (dWc operand); backward: dgates [B][T][H][2], dgates2 [B][T][H]. */
  const float* wt2;
  const float* w2;
  const float* bias2;
  float* rh_seq;
  float* dgates2;
} avsr_rnn_layer;

typedef struct avsr_rnn_stack {
  int32_t B, T, reverse, n_layers;
  int32_t cell, pad_;             /* 0 = LSTM, 1 = GRU */
  const int32_t* len;
  const float* dh_final;          /* [B][H_top] gradient wrt the top layer's final h (may be NULL) */
  const float* dc_final;
  /* tf.contrib.rnn.DropoutWrapper(input/state/output keep prob, variational_recurrent=False) (avsr/cells.py:46-54).
   * Masks are stateless: bit = hash(*seed, stream, index) (csrc/common.h hash_u32), stream = cell_id*4 + {0 in,1 state,2 out},
   * cell_id = cell_id_base + layer, index = (b*T + t)*width + column.  seed == NULL disables dropout.  With dropout on,
   * `state` scratch must hold 6*B*H floats per layer.  consumer_*: input mask of whatever reads the TOP layer's output
   * through xt_seq (AV-Align attentive layer); consumer_width 0 = none. */
  const int32_t* seed;
  float keep_in, keep_state, keep_out, consumer_keep;
  int32_t cell_id_base, consumer_stream, consumer_width, pad2_;
  avsr_rnn_layer layer[AVSR_MAX_LAYERS];
} avsr_rnn_stack;

int avsr_rnn_fwd(const avsr_rnn_stack* stacks, int32_t n_stacks, void* stream);
int avsr_rnn_bwd(const avsr_rnn_stack* stacks, int32_t n_stacks, void* stream);

/* Opt-in persistent execution of avsr_rnn_fwd (LSTM stacks): ONE launch runs the whole layer/time wavefront,
 * weights and (c, h) state stay in registers, steps are ordered by device-side arrival counters instead of
 * launch boundaries.  `sync` = caller-owned int32 device scratch of `ints` words: word 0 is a sticky error flag
 * (a bounded wait expired; zero it before installing, read it back to check), the rest are counters
 * (needs 1 + sum over layers of ceil(B/16)*T).  NULL disables.  Configurations that do not fit (GRU,
 * units %% 8, in+units > 512, > 512 workgroups) silently use the per-step launches.  Same results either way. */
int avsr_rnn_set_persistent(int32_t* sync, int64_t ints);
/* Which persistent kernels may be used: bit 0 = agent-scope (any placement, 16-row tiles), bit 1 = XCD-local
 * (8-row groups bound to the XCD their workgroups actually run on; tried first) and the fused persistent BPTT,
 * bit 2 = split persistent BPTT (needs avsr_rnn_set_persistent_scratch; opt-in), bit 3 = pair-layout forward (16-row groups
 * on XCD pairs; opt-in).  Default 3: the two opt-in forms measured slower on the benchmark shape. */
int avsr_rnn_set_persistent_mode(int mode);
/* Float device scratch for the split persistent BPTT (bit 2 of the mode): one [B,T,units] operand per encoder cell that has a
 * layer above it.  NULL / too small: that form is skipped (the fused form or the per-step launches run instead). */
int avsr_rnn_set_persistent_scratch(float* scratch, int64_t floats);

/* ---------------------------------------------------------------------------------------------
 * Attention-wrapped LSTM over a sequence.  Replaces
 *   seq2seq.dynamic_decode(BasicDecoder(AttentionWrapper(LSTMCell, [mechanisms]), helper, Dense(V)))
 * for the unimodal and bimodal decoders (avsr/decoder_unimodal.py:299-352 train, :176-217 greedy;
 * avsr/decoder_bimodal.py:227-277, :279-326) and tf.nn.dynamic_rnn over the AttentionWrapper'd top
 * audio layer of AV-Align (avsr/encoder.py:265-290).
 *
 * Per step l (AttentionWrapper.call):  x' = [x_l, attention_{l-1}] -> LSTM -> cell_out;
 * for each mechanism m: alpha_m = softmax(score_m(cell_out)), ctx_m = alpha_m . values_m,
 * att_m = [cell_out, ctx_m] . W_att,m;  attention_l = concat_m att_m.
 * Steps with l >= steplen[b] are frozen (state copy-through, zero outputs).
 *
 * mode 0 (train / encoder): x_l . Wx is hoisted by the caller into `gates` (avsr_gemm).
 * mode 1 (greedy):          x_l = embedding[tok[b]]; after each step logits = out . Wout + bout,
 *                           ids[b,l] = argmax (0 once finished), steplen[b] is set when EOS is emitted
 *                           (GreedyEmbeddingHelper + impute_finished).  The caller initialises
 *                           steplen[b] = L and tok[b] = GO, and may run steps in slices [l0, l1).
 *
 * Buffers (A = n_mech * H):
 *   gates [B][L][H][4], cs [B][L][H], cell_out [B][L+1][H] (slot l+1 = step l; slot 0 = h0, written
 *   by the op), att [B][L+1][A] (slot l+1 = attention after step l; slot 0 zero, written by the op),
 *   state scratch 4*B*H.
 *   per mechanism: scores [B][L][T] raw scores (Luong: un-scaled), ctx [B][L][D], pq [B][L][H]
 *   (Bahdanau), pstat [L][2][nchunk][B], pctx scratch [nchunk][B][D], with nchunk = ceil(T / chunk).
 * Backward adds: dgates [B][L][H][4], dstate scratch 12*B*H, datt [B][L][A], dq [B][L][H];
 *   per mechanism dscores [B][L][T], dctx [B][L][D], dpq [B][L][H] (Bahdanau), pdq scratch
 *   [nchunk][B][H].  Inputs datt_ext [B][L][A] / dcell_ext [B][L][H]: gradient of the loss wrt the
 *   emitted attention / cell output of every step (from the output layer, or from the consumer of the
 *   encoder memory); either may be NULL.  Outputs dh0, dc0 [B][H].
 */
typedef struct avsr_attn_mech {
  int32_t type;                 /* 0 luong, 1 scaled_luong, 2 bahdanau, 3 normed_bahdanau */
  int32_t T, D, chunk;
  const int32_t* len;           /* [B] memory lengths */
  const float* keys;            /* [B][T][H] */
  const float* values;          /* row (b,t) at values + b*values_sb + t*values_st */
  int64_t values_sb, values_st;
  const float* g;               /* scalar: scaled_luong */
  const float* v;               /* [H] Bahdanau (normed: g*v/|v|) */
  const float* bq;              /* [H] normed_bahdanau bias */
  const float* wq_t;            /* [H][H]   query_layer^T */
  const float* wq;              /* [H][H]   query_layer */
  const float* watt_t;          /* [H][H+D] attention_layer^T */
  const float* watt;            /* [H+D][H] attention_layer */
  float* scores;
  float* ctx;
  float* pq;
  float* pstat;
  float* pctx;
  float* dscores;
  float* dctx;
  float* dpq;
  float* pdq;
} avsr_attn_mech;

/* One extra decoder layer above the attention-fed cell: the wrapped cell is a MultiRNNCell (cells.py:96-100,
 * decoder_unimodal.py:101-108 with len(decoder_units_per_layer) > 1).  LSTM or GRU (the block's cell kind), width = H of the block.  Layer j consumes
 * the emitted output of layer j-1 of the same step; the TOP layer's output is what the attention mechanisms query and
 * what `cell_out` records, so the top layer's `out` must be the block's cell_out.  State starts at zero
 * (decoder_unimodal.py:151-157).  Buffers: wt [4H][2H] / w [2H][4H] (rows: input part, then recurrent part), bias [H][4],
 * gates [B][L][H][4], cs [B][L][H], out [B][L+1][H] (slot l+1 = step l, slot 0 = 0), state 4*B*H, dgates [B][L][H][4],
 * dstate 12*B*H; dropout only: hs_seq [B][L+1][H] state-dropped h, xin_seq [B][L+1][H] = the input as this layer consumed it
 * (slot l+1 = step l: lower layer's output mask x this layer's input mask).
 * GRU layers (round 6): wt [2H][2H] / w [2H][2H] = the gate kernel (columns unit-interleaved r, u), bias [H][2], gates [B][L][H][2],
 * cs = the candidate record [B][L][H], dgates [B][L][H][2]; plus the candidate kernel wt2 [H][2H] / w2 [2H][H] (rows: input part, then
 * the r*h part), bias2 [H], rh_seq [B][L][H] (r*h as the candidate consumed it) and dgates2 [B][L][H] (d candidate pre-activation). */
typedef struct avsr_dec_layer {
  const float* wt;
  const float* w;
  const float* bias;
  float* gates;
  float* cs;
  float* out;
  float* state;
  float* hs_seq;
  float* xin_seq;
  float* dgates;
  float* dstate;
  int32_t cell_id, pad_;
  const float* wt2;             /* GRU only (NULL for LSTM layers) */
  const float* w2;
  const float* bias2;
  float* rh_seq;
  float* dgates2;
} avsr_dec_layer;
#define AVSR_MAX_DEC_EXTRA 3

typedef struct avsr_attn_rnn {
  int32_t B, L, H, E, n_mech, output_attention, V, mode;
  int32_t go_id, eos_id;
  int32_t* steplen;             /* [B] valid steps (labels_len / audio len); greedy: updated in place */
  const float* wt;              /* [4H][E + A + H] */
  const float* w;               /* [E + A + H][4H] */
  const float* bias;            /* [H][4] */
  float* gates;
  float* cs;
  float* cell_out;
  float* att;
  const float* h0;
  const float* c0;
  float* state;
  float* h_final;
  float* c_final;
  avsr_attn_mech mech[AVSR_MAX_MECH];
  /* greedy */
  const float* embedding;       /* [V][E] */
  const float* wout_t;          /* [V][O], O = A if output_attention else H */
  const float* bout;            /* [V] */
  float* logits;                /* [B][L][V] (greedy: per-step logits, zero once finished) */
  int32_t* ids;                 /* [B][L] */
  int32_t* tok;                 /* [B] */
  int32_t* n_unfinished;        /* [1] device counter, recomputed after every greedy step */
  /* backward */
  float* dgates;
  float* dstate;
  float* datt;
  float* dq;
  const float* datt_ext;
  const float* dcell_ext;
  float* dh0;
  float* dc0;
  const float* dh_final;        /* [B][H] gradient wrt the final cell state (AV-Align: from the decoder init) */
  const float* dc_final;
  /* DropoutWrapper on the wrapped cell (train only; seed NULL = off), stream = cell_id*4 + kind, input width E + A */
  const int32_t* seed;
  float keep_in, keep_state, keep_out, sampling_prob;
  int32_t cell_id, pad3_;
  float* hs_seq;                /* [B][L+1][H] state-dropped h, slot 0 = h0 (dropout only) */
  float* attd;                  /* [B][L+1][A] input-dropped attention fed to the next step, slot 0 = 0 (dropout only) */
  /* mode 2 = train with scheduled sampling (ScheduledEmbeddingTrainingHelper, avsr/decoder_unimodal.py:304-309):
   * x_l = xs[b][l][:] (caller fills slot 0 = dropout(emb[GO])); after step l the op computes logits[b][l][:], draws
   * select ~ Bernoulli(sampling_prob) and a categorical sample with the stateless RNG (streams 1000 / 1001, index b*L+l)
   * and writes fed[b][l+1] = select ? sample : labels[b][l], xs[b][l+1][:] = dropout(emb[fed[b][l+1]]). */
  float* xs;                    /* [B][L][E] */
  const int32_t* labels;        /* [B][L] */
  int32_t* fed;                 /* [B][L] tokens actually fed (fed[b][0] = GO set by the caller) */
  /* GRU cell (cell == 1): same conventions as avsr_rnn_layer; gates [B][L][H][2], cs = candidate record [B][L][H] */
  int32_t cell, pad4_;
  const float* wt2;
  const float* w2;
  const float* bias2;
  float* rh_seq;
  float* dgates2;
  /* mode 3 = beam search (contrib.seq2seq.BeamSearchDecoder, avsr/decoder_unimodal.py:248-271, avsr/decoder_bimodal.py:358-381):
   * the B rows are beam_width consecutive hypotheses per utterance over tile_batch'ed memories.  After every step the op
   * scores log_softmax(logits) + beam log-prob with the length penalty ((5+len)/6)^w, keeps the top beam_width of the
   * beam_width*V continuations per utterance (ties -> lower index), and records step_ids / parent_ids [L][B].  The next
   * step reads its previous state through parent_rows (no state copies).  Caller initialises tok = GO, parent_rows = identity,
   * beam_logp[0] = {0, -inf, ...} per utterance, beam_fin = beam_len = 0, steplen = L.  In this mode n_unfinished is an
   * [L] array: entry l = number of unfinished beams after step l (dynamic_decode stops at the first 0).
   * mem_shared != 0: the attention memories are NOT tiled -- keys / values / len of every mechanism hold one entry per UTTERANCE
   * (B / beam_width rows) and hypothesis row b attends memory row b / beam_width: the beam_width hypotheses of an utterance read the
   * same bytes (L2 / Infinity-Cache hits) instead of beam_width copies streamed from HBM every step. */
  int32_t beam_width, mem_shared;
  float length_penalty;
  float pad6_;
  float* beam_logp;             /* [2][B] ping-pong */
  int32_t* beam_fin;            /* [2][B] */
  int32_t* beam_len;            /* [2][B] */
  int32_t* step_ids;            /* [L][B] */
  int32_t* parent_ids;          /* [L][B] */
  int32_t* parent_rows;         /* [B] */
  /* multi-layer decoder cell: n_extra layers above the attention-fed one (0 = the plain single cell).  out0 = output
   * record of the attention-fed layer [B][L+1][H] (slot 0 = h0), needed because cell_out then belongs to the top layer. */
  int32_t n_extra, prof_tag;    /* prof_tag: 1 = this block is the AV-Align attentive encoder layer (encoder.py:265-290): its fused launches are
                                 * timed as their own avsr_prof classes (13 / 14) instead of the decoder's (8 / 9); no effect on results */
  float* out0;
  avsr_dec_layer extra[AVSR_MAX_DEC_EXTRA];
  /* Scratch of the fused persistent decode kernel (csrc/dec_persist.hip; avsr/decoder_bimodal.py:241-275,
   * avsr/decoder_unimodal.py:320-350 as ONE launch per call): per-quarter softmax partials and split-K logits.
   * >= avsr_attn_rnn_fused_ws_floats(B, n_mech, Dmax) floats; NULL = always use one launch per step and phase. */
  float* fused_ws;
  int64_t fused_ws_floats;
} avsr_attn_rnn;
int64_t avsr_attn_rnn_fused_ws_floats(int32_t B, int32_t n_mech, int32_t Dmax);
/* 1 if the fused persistent decode kernels are built for this descriptor (the block is inside their plan), else 0. */
int avsr_attn_rnn_fused_eligible(const avsr_attn_rnn* d);
/* 1 if avsr_attn_rnn_fwd(d, ...) WILL run the fused forward kernel right now: eligible, forward fusion switched on
 * (avsr_attn_rnn_set_fused 1 or 2) and the persistent sync scratch registered and large enough.  Greedy decode
 * (avsr/decoder_unimodal.py:176-217) issues all maximum_iterations steps as one call only then; otherwise it keeps its
 * host check every few steps. */
int avsr_attn_rnn_fused_fwd_active(const avsr_attn_rnn* d);
/* avsr_attn_rnn_bwd runs the same block's BPTT loop as one launch of the fused persistent backward kernel (csrc/dec_persist_bwd.hip)
 * under the same conditions.  Process-wide switch: 0 per-step launches, 1 (default) fused forward and backward, 2 forward only,
 * 3 backward only; the path also needs avsr_rnn_set_persistent's sync scratch. */
int avsr_attn_rnn_set_fused(int32_t on);
/* Beam search (mode 3): which kernels run the B * K-row steps.  1 (default; AVSR_ATTN_BEAM / AVSR_BEAM_DENSE in the environment set
 * the initial values) = all beam-shaped kernels: the per-step attention as one workgroup per (utterance, chunk) serving all K
 * hypotheses over the shared memories, scores and contexts on the matrix pipe with the hypotheses as one MFMA row tile
 * (attn_fwd_beam_mfma_kernel), and the LSTM cell / attention layers as 64 x 64-tiled products with row-gathered operands
 * (csrc/beam_gemm.hip); 2 = the scalar K-hypotheses attention kernel (attn_fwd_beam_kernel) only, dense steps through the small-tile step kernel;
 * 0 = the general kernels everywhere.  0 and 2 give bit-identical scores, statistics and contexts; 1 differs from them by the
 * summation order of the dense products (tests/test_gpu_beam.py checks all three against the oracle). */
int avsr_attn_rnn_set_beam_kernel(int32_t on);

int avsr_attn_rnn_fwd(const avsr_attn_rnn* d, int32_t l_begin, int32_t l_end, void* stream);
int avsr_attn_rnn_bwd(const avsr_attn_rnn* d, void* stream);
/* gather_tree over the recorded beam steps: out[b][t][k] for t < T (ids after the first EOS and past the longest beam = EOS) */
int avsr_beam_gather_tree(const int32_t* step_ids, const int32_t* parent_ids, const int32_t* beam_len, int32_t* out,
                          int32_t n_utt, int32_t beam_width, int32_t T, int32_t eos_id, void* stream);

/* One BeamSearchDecoder step on GIVEN logits [n_utt * beam_width][V] -- the selection avsr_attn_rnn_fwd (mode 3) runs after its output
 * layer, as an entry point of its own (contrib.seq2seq `_beam_search_step` as decoder_unimodal.py:248-271 / decoder_bimodal.py:358-381
 * reach it through BeamSearchDecoder): log_softmax, finished beams continue with EOS at log-probability 0, scores =
 * accumulated log-probability / ((5 + len) / 6)^w with an EOS continuation not counted in len, top beam_width of the beam_width * V
 * continuations per utterance (best first, ties -> lower index).  State in / out: logp, fin, len [n_utt * beam_width] (start:
 * logp = {0, -inf, ...} per utterance, fin = len = 0); tok / parent_rows [n_utt * beam_width] = the kept symbols and the global rows of their
 * parents; step_ids / parent_ids [L][n_utt * beam_width] and n_unfinished [L] (zeroed by the caller) are written at index `step`
 * (a step after one that left n_unfinished == 0 hands the state through unchanged).  beam_width * V <= 1024.
 * x != NULL: the decoder's output layer runs inside the step as it does in the default evaluation path (seq2seq.py:339 Dense(vocab)):
 * logits [row][v] = x[row * x_stride + .] (O inputs) . wout_t[v][.] + bout[v] are COMPUTED (and written to `logits`) before the selection;
 * needs beam_width <= 16, V <= 64, O % 256 == 0, 16-byte aligned x / wout_t, else AVSR_ERR_UNSUPPORTED.  x == NULL: `logits` is the input.
 * tests/test_gpu_beam.py replays TensorFlow's own trace of the reference's sample search (avsr/visualise/00025.html) through it. */
int avsr_beam_search_step(float* logits, int32_t n_utt, int32_t beam_width, int32_t V, int32_t step, int32_t eos_id,
                          float length_penalty_weight, const float* logp_in, const int32_t* fin_in, const int32_t* len_in,
                          float* logp_out, int32_t* fin_out, int32_t* len_out, int32_t* tok, int32_t* parent_rows,
                          int32_t* step_ids, int32_t* parent_ids, int32_t* n_unfinished, const float* x, int64_t x_stride,
                          int32_t O, const float* wout_t, const float* bout, void* stream);

/* Post-loop helpers of the attention backward (see csrc/attention.hip). */
int avsr_attn_alpha_rows(float* scores, const float* dscores, const int32_t* len, const int32_t* steplen,
                         const float* g, float* rowdot, int32_t B, int32_t L, int32_t T, void* stream);
int avsr_bahdanau_dkeys(const float* keys, const float* pq, int64_t pq_sb, int64_t pq_sl, const float* dscores,
                        const float* v, const float* bq, const int32_t* len, float* dkeys, float* dv_part,
                        int32_t B, int32_t L, int32_t T, int32_t H, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Memory-bound helpers (csrc/elementwise.hip).  scratch buffers are caller-provided device floats. */
typedef struct avsr_transpose_job {
  const float* src;             /* [rows][cols] */
  float* dst;                   /* [cols][rows] */
  int32_t rows, cols;
} avsr_transpose_job;
/* jobs is a HOST array (copied into kernel arguments). */
int avsr_transpose(const avsr_transpose_job* jobs, int32_t n, void* stream);

/* out[f] = alpha * sum_r a[r][f] * (b ? b[r][f] : 1) + beta*out[f]   (bias / BN gradients); scratch >= 2*F floats */
int avsr_colsum(const avsr_mat* a, const avsr_mat* b, int32_t rows, int32_t F, float alpha, float beta, float* out,
                float* scratch, int64_t scratch_floats, void* stream);

/* tf.layers.batch_normalization(axis=-1, momentum=.99, eps=1e-3) over `rows` = B*T rows incl. padding
 * (avsr/encoder.py:44-50).  training=1: batch statistics + moving-average update; 0: moving statistics.
 * The input is rank 3, so TF 1.13 drops `fused=True` and the moving variance takes the BIASED batch variance.
 * moving_mean / moving_var may be NULL in training (no update: seq2seq.py:241-250 runs UPDATE_OPS only under
 * batch_normalisation=True). */
int avsr_batchnorm_fwd(const float* x, float* y, int32_t rows, int32_t F, const float* gamma, const float* beta,
                       float* moving_mean, float* moving_var, float* save_mean, float* save_invstd, int32_t training,
                       float* scratch, int64_t scratch_floats, void* stream);
/* As avsr_batchnorm_fwd with explicit epsilon / momentum and an optional fused ReLU: batch_norm_relu of the lip-crop CNN
 * (avsr/video.py:4-14: epsilon 1e-5, momentum 0.98) and any tf.layers.batch_normalization(axis=-1) over [rows, F].
 * bessel=1 restates the fused kernel (rank-4 inputs): the moving variance takes var * rows / (rows - 1). */
int avsr_batchnorm_fwd_ex(const float* x, float* y, int32_t rows, int32_t F, const float* gamma, const float* beta,
                          float* moving_mean, float* moving_var, float* save_mean, float* save_invstd, int32_t training,
                          float eps, float momentum, int32_t relu, int32_t bessel, float* scratch, int64_t scratch_floats,
                          void* stream);
/* Sync batch-norm across data-parallel ranks (SURVEY 8(e) collective (3); statistics of encoder.py:44-50 over the
 * GLOBAL batch).  Three local phases; the host all-reduces sum_out (with the row count) after the first and sq_out
 * after the second:  (1) sum_out[f] = sum_rows x;  (2) mean_out = sum_global / total_rows[0], sq_out[f] = sum_rows
 * (x - mean)^2;  (3) invstd = rsqrt(sq_global / total + eps), moving-average update with the (biased, rank-3
 * non-fused path) global variance, y = (x - mean) * invstd * gamma + beta.  total_rows is a DEVICE float (the all-reduced count). */
int avsr_batchnorm_sync_sum(const float* x, int32_t rows, int32_t F, float* sum_out, float* scratch, int64_t scratch_floats,
                            void* stream);
int avsr_batchnorm_sync_sqsum(const float* x, int32_t rows, int32_t F, const float* sum_global, const float* total_rows,
                              float* mean_out, float* sq_out, float* scratch, int64_t scratch_floats, void* stream);
/* Data-parallel step, ONE small collective (the loss normalisers and the sync-batch-norm statistics fused):
 *   avsr_batchnorm_sync_moments: out64[0..F) = sum x, out64[F..2F) = sum x^2 over this rank's rows, fp64 (scratch 8-byte aligned,
 *     >= 4*F*min(1024, ceil(rows/64)) floats); the caller packs them with its row count and the two loss normalisers into one fp64
 *     buffer and all-reduces that (sum);
 *   avsr_dp_sync_unpack: the reduced buffer -> dp_norm[0..1] and per stream mean / centred squares (sum x^2 - rows*mean^2, fp64) /
 *     rows as the float32 operands avsr_batchnorm_sync_apply and the loss kernels read.  encoder.py:44-50 statistics over the GLOBAL
 *     batch; no reference counterpart for the transport. */
int avsr_batchnorm_sync_moments(const float* x, int32_t rows, int32_t F, double* out64, float* scratch, int64_t scratch_floats, void* stream);
int avsr_dp_sync_unpack(const double* buf, float* dp_norm, int32_t nstream, const int32_t* off, const int32_t* F, float* const* mean,
                        float* const* sq, float* const* rows, void* stream);
int avsr_batchnorm_sync_apply(const float* x, float* y, int32_t rows, int32_t F, const float* gamma, const float* beta,
                              float* moving_mean, float* moving_var, const float* mean, const float* sq_global,
                              const float* total_rows, float* invstd_out, float eps, float momentum, int32_t relu, void* stream);
/* Gradient of batch normalisation with training statistics (tf.gradients through fused batch norm), optionally through
 * the ReLU after it: dy' = dy * [gamma*xhat + beta > 0];  dbeta = sum dy';  dgamma = sum dy'*xhat;
 * dx = gamma*invstd*(dy' - (sum dy' + xhat*sum dy'*xhat)/rows), written as dx = that + dx_beta*dx.  dx / dgamma / dbeta may
 * be NULL.  scratch >= (rows/64 + 1) * max(1, 256/F) * 2F + 2F floats. */
int avsr_batchnorm_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean,
                       const float* invstd, float* dx, float* dgamma, float* dbeta, int32_t rows, int32_t F, int32_t relu,
                       float dx_beta, float* scratch, int64_t scratch_floats, void* stream);

/* ---- lip-crop CNN front-end (avsr/video.py:143-195 resnet_cnn): convolutions are im2col + avsr_gemm ----
 * im2col over NHWC maps with TF "SAME"/"VALID" geometry given explicitly: col[(n,ho,wo)][(i,j,c)] =
 * x[n, ho*stride - pad_t + i, wo*stride - pad_l + j, c] (0 outside); the TF kernel [kh,kw,cin,cout] is the GEMM's
 * [kh*kw*cin, cout] operand as stored.  col2im is its adjoint (gather form, deterministic): dx = col2im(dcol) + beta*dx. */
int avsr_im2col(const float* x, float* col, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t stride,
                int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, void* stream);
int avsr_col2im(const float* dcol, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t stride,
                int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, float beta, void* stream);
/* Direct 3x3 convolutions on NHWC maps for the shallow layers of the lip CNN (cin*cout <= 256; avsr_conv3x3_supported):
 * the same tf.layers.conv2d(3x3, SAME) and its gradients as the im2col + avsr_gemm route, without the 9x operand.
 *   avsr_conv3x3(flip=0): y = conv(x, w[3,3,Ci,Co]) + bias (+ beta*y).
 *   avsr_conv3x3(flip=1): stride-1 data gradient: x = dy [.,Ci = cout], y = dx [.,Co = cin], w = the FORWARD kernel [3,3,Co,Ci].
 *   avsr_conv3x3_bwd_data_s2: data gradient of a stride-2 conv (dx [N,H,W,Ci] from dy [N,Ho,Wo,Co]).
 *   avsr_conv3x3_bwd_weight: dw[3,3,Ci,Co] = beta*dw + sum x (x) dy; Co in {4, 8, 16}; scratch >= ceil(N/4) * (256/(9*Ci)) * 9*Ci*Co floats (fewer blocks if smaller). */
int avsr_conv3x3_supported(int32_t Ci, int32_t Co, int32_t H, int32_t W);
/* The three entry points below run the frame-resident MFMA kernels of csrc/conv_mfma.hip where they cover the shape (whole frames
 * staged in LDS, implicit GEMM on v_mfma_f32_16x16x4_f32) and the direct VALU kernels otherwise; this switch (default 1) forces the
 * latter (A/B timing, tests). */
int avsr_conv_set_mfma(int32_t on);
/* Frame-resident MFMA convolutions of the lip CNN through ONE descriptor (csrc/conv_mfma.hip; tf.layers.conv2d of video.py:18-30 with
 * k = 1 (the stride-2 projection shortcut, video.py:70-75) or 3, stride 1 / 2, Ci in {1..3, 4n}, Co = 4n <= 64):
 *   bn_scale / bn_shift (may be NULL): the input map is the PRE-normalisation map x and the kernels apply the consumer-side
 *     batch_norm_relu (video.py:4-14) max(x*scale[c] + shift[c], 0) while staging frames in LDS -- the normalised map is never
 *     written (forward and weight gradient; the zero padding is applied after the BN-ReLU, as in the graph).
 *   avsr_conv_fwd: y = conv(x) + bias (+ res: the residual of video.py:84 `tf.add`; res_scale/res_shift != NULL: the residual is
 *     itself a lazily normalised map); stats != NULL (>= 512*2*Co floats): per-workgroup partial sums / sums of squares of y
 *     [*nparts][2*Co] for the batch norm that consumes y; avsr_bn_finalize merges them in fp64 into mean / inverse std (+ the
 *     moving averages with the Bessel-corrected variance of the fused rank-4 TF kernel) and the scale / shift vectors above.
 *   avsr_conv_bwd_data: dx = beta*dx + conv_transpose(dy); a stride-2 1x1 kernel reaches only the even pixels, so beta must be 1.
 *   avsr_conv_bwd_weight: dw = beta*dw + sum x (x) dy, dbias (may be NULL) = beta*dbias + column sums of dy (same pass over dy);
 *     scratch >= 256 * (k*k*Ci*Co + Co) floats.
 * AVSR_ERR_UNSUPPORTED (-3) for shapes outside the kernels' register / LDS budget (avsr_conv_supported == 0): the caller then uses
 * im2col + avsr_gemm. */
typedef struct avsr_conv_desc {
  int32_t N, H, W, Ci, Co, k, stride, pad_t, pad_l, Ho, Wo, pad_;
  const float* bn_scale;
  const float* bn_shift;
} avsr_conv_desc;
int avsr_conv_supported(const avsr_conv_desc* c);
int avsr_conv_fwd(const avsr_conv_desc* c, const float* x, const float* w, const float* bias, const float* res, const float* res_scale,
                  const float* res_shift, float* y, float* stats, int32_t* nparts, void* stream);
int avsr_conv_bwd_data(const avsr_conv_desc* c, const float* dy, const float* w, float* dx, float beta, void* stream);
int avsr_conv_bwd_weight(const avsr_conv_desc* c, const float* x, const float* dy, float* dw, float* dbias, float beta, float* scratch,
                         int64_t scratch_floats, void* stream);
/* Data gradient with the fused forms of the lip CNN's backward pass (avsr/video.py:4-14 batch_norm_relu, :57-88 residual blocks):
 *   acc (may be NULL): dx = beta*acc + conv_transpose(dy) -- the gradient arriving over a residual connection is read where it lies
 *     instead of being copied into dx first;
 *   bn_x != NULL: dx is the gradient of y = relu(bn_x*bn_scale + bn_shift) (the lazily normalised input of the forward conv): the
 *     epilogue writes dz = dx * [y > 0] and the per-workgroup partial sums [*nparts][2*Ci] of (dz | dz*bn_x) into stats
 *     (>= 512*2*Ci floats) -- stage 1 of tf.layers.batch_normalization's backward without its own two-map pass;
 *   avsr_bn_bwd_finalize: d beta / d gamma (grad_beta = 1 accumulates) and the three coefficient vectors k [3*C];
 *   avsr_bn_bwd_apply: dx = beta*dx + k1*dz + k2*x + k3  ==  gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)).
 * AVSR_ERR_UNSUPPORTED where the data gradient needs several launches over disjoint pixel classes (avsr_conv_bwd_data_bn_supported == 0). */
int avsr_conv_bwd_data_bn(const avsr_conv_desc* c, const float* dy, const float* w, float* dx, float beta, const float* acc,
                          const float* bn_x, const float* bn_scale, const float* bn_shift, float* stats, int32_t* nparts, void* stream);
int avsr_conv_bwd_data_bn_supported(const avsr_conv_desc* c);
/* Weight (+ bias) gradient of a convolution whose OUTPUT y feeds a batch norm (video.py:4-14), with that batch norm's backward folded into
 * the operand fetch: the gradient of y, gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)) = k[0..C)*dz + k[C..2C)*y + k[2C..3C)
 * (avsr_bn_bwd_finalize's coefficient vectors), is evaluated where the kernel fetches it.  dx_out == NULL (a layer with no data gradient:
 * layer 0 of resnet_cnn, whose input are the lip crops): it is never stored; dx_out != NULL: every element is also written there by the
 * one lane that fetched it, for the layer's data gradient (issued after this call) -- either way the avsr_bn_bwd_apply pass over three
 * maps disappears.  avsr_conv_bwd_weight_bn_supported (single-launch forms only): 0 = use avsr_bn_bwd_apply + avsr_conv_bwd_weight. */
int avsr_conv_bwd_weight_bn(const avsr_conv_desc* c, const float* x, const float* dz, const float* y, const float* k, float* dx_out,
                            float* dw, float* dbias, float beta, float* scratch, int64_t scratch_floats, void* stream);
int avsr_conv_bwd_weight_bn_supported(const avsr_conv_desc* c);
int avsr_bn_bwd_finalize(const float* part, int32_t nparts, int32_t C, int64_t count, const float* mean, const float* invstd,
                         const float* gamma, float* dgamma, float* dbeta, float grad_beta, float* k, void* stream);
int avsr_bn_bwd_apply(const float* dz, const float* x, const float* k, float* dx, int64_t rows, int32_t C, float beta, void* stream);
/* Stage 1 of the batch-norm backward as a pass of its own, for a batch norm whose output gradient several launches assembled (no single
 * epilogue to fuse it into): dz = dy * [scale*x + shift > 0] (scale == NULL: [y > 0]), part [*nparts <= 512][2*C] = (sum dz | sum dz*x) --
 * the layout avsr_bn_bwd_finalize / avsr_bn_partials_f64 read.  dz may alias dy; C % 4 == 0, C <= 1024.  Used by sync_cnn_bn for the
 * layers avsr_conv_bwd_data_bn does not cover (video.py:4-14 differentiated over the global batch). */
int avsr_bn_bwd_stage1(const float* dy, const float* x, const float* scale, const float* shift, const float* y, float* dz, int64_t rows,
                       int32_t C, float* part, int32_t* nparts, void* stream);
int avsr_bn_finalize(const float* part, int32_t nparts, int32_t C, int64_t count, float eps, float momentum, float* mean, float* invstd,
                     float* mov_mean, float* mov_var, const float* gamma, const float* beta, float* scale, float* shift, void* stream);
/* The same two finalisations from GLOBAL statistics under data parallelism (video.py:4-14 `tf.layers.batch_normalization` over the whole
 * batch; opt-in, DataParallelTrainer(sync_cnn_bn=True)): avsr_bn_partials_f64 merges a convolution's partial rows into fp64 sums
 * out64 [2*C]; the caller all-reduces [sums | rows per channel] over the ranks; avsr_bn_finalize_f64 reads sums [2*C + 1] (count last);
 * avsr_bn_bwd_finalize_f64 takes this rank's sums `local` [2*C] for d gamma / d beta (the gradient all-reduce adds the ranks' shares)
 * and the all-reduced `global` [2*C + 1] for the coefficient vectors k [3*C]. */
int avsr_bn_partials_f64(const float* part, int32_t nparts, int32_t C, double* out64, void* stream);
int avsr_bn_finalize_f64(const double* sums, int32_t C, float eps, float momentum, float* mean, float* invstd, float* mov_mean,
                         float* mov_var, const float* gamma, const float* beta, float* scale, float* shift, void* stream);
int avsr_bn_bwd_finalize_f64(const double* local, const double* global, int32_t C, const float* mean, const float* invstd,
                             const float* gamma, float* dgamma, float* dbeta, float grad_beta, float* k, void* stream);
/* scale = gamma * rsqrt(moving_variance + eps), shift = beta - moving_mean * scale: the evaluation graph's batch norm as the
 * loader-applied affine of avsr_conv_desc (no normalised map is written in evaluation either). */
int avsr_bn_eval_affine(const float* gamma, const float* beta, const float* mov_mean, const float* mov_var, float eps, float* scale,
                        float* shift, int32_t C, void* stream);
int avsr_batchnorm_apply(const float* x, float* y, int32_t rows, int32_t F, const float* gamma, const float* beta, const float* mean,
                         const float* invstd, int32_t relu, void* stream);
int avsr_conv3x3(const float* x, const float* w, const float* bias, float* y, int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                 int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, int32_t flip, float beta, void* stream);
int avsr_conv3x3_bwd_data_s2(const float* dy, const float* w, float* dx, int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                             int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, float beta, void* stream);
int avsr_conv3x3_bwd_weight(const float* x, const float* dy, float* dw, int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                            int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, float beta, float* scratch,
                            int64_t scratch_floats, void* stream);
/* y = max(x, 0);  dx = dy * [y > 0];  out = a + b (tf.nn.relu / residual tf.add of video.py) */
int avsr_relu(const float* x, float* y, int64_t n, void* stream);
int avsr_relu_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream);
int avsr_add(const float* a, const float* b, float* out, int64_t n, void* stream);
/* tf.nn.selu and its gradient (dz = dy * selu'(z)): the optional input Dense stack of the encoders (encoder.py:148-171) */
int avsr_selu(const float* z, float* y, int64_t n, void* stream);
int avsr_selu_bwd(const float* z, const float* dy, float* dz, int64_t n, void* stream);

int avsr_batchnorm_xhat(const float* x, const float* mean, const float* invstd, float* xhat, int32_t rows, int32_t F,
                        void* stream);

/* embedding_lookup(labels_padded_GO) (avsr/decoder_unimodal.py:66-68, :170): fed[b][l] = l ? labels[b][l-1] : GO,
 * out[b][l][:] = emb[fed[b][l]] for l < n_steps (n_steps = L: all; 1: only the GO column).  Gradient: demb[v][:] = sum of
 * dx rows whose fed token is v (deterministic order; scratch >= ceil(B*L/256) * V * E floats). */
int avsr_embed_labels(const float* emb, const int32_t* labels, int32_t go_id, float* out, int32_t* fed, int32_t B,
                      int32_t L, int32_t E, int32_t n_steps, void* stream);
int avsr_embed_grad(const float* dx, const int32_t* fed, float* demb, int32_t B, int32_t L, int32_t E, int32_t V,
                    float* scratch, int64_t scratch_floats, void* stream);

/* y[r][c] = (accumulate ? y[r][c] : 0) + x[r][c] * mask(r, c) / keep for r < rows, c < cols;
 * mask index = r * idx_width + idx_coff + c (r = b*T + t), stream as in avsr_rnn_stack.  Used for input dropout of
 * hoisted operands and for the matching gradients.  seed NULL or keep >= 1 copies. */
int avsr_dropout_rows(const avsr_mat* x, const avsr_mat* y, int32_t rows, int32_t cols, const int32_t* seed,
                      int32_t stream_id, float keep, int32_t idx_width, int32_t idx_coff, int32_t accumulate, void* stream);

/* seq2seq.sequence_loss (avsr/seq2seq.py:165-171): row_loss[b*L+l] = CE * mask / (sum(mask) + 1e-12) and
 * d loss / d logits.  denom[0] = sum(mask) is computed when compute_denom=1 (single GPU) or supplied
 * (data parallel: all-reduced by the caller).  Logit rows of finished steps are zeroed in place
 * (dynamic_decode impute_finished=True, avsr/decoder_unimodal.py:344-350). */
int avsr_seq_loss(float* logits, const int32_t* labels, const int32_t* labels_len, float* denom,
                  int32_t compute_denom, float* row_loss, float* dlogits, int32_t B, int32_t L, int32_t V, void* stream);
/* The non-default per-step losses of avsr/seq2seq.py:147-163.  loss_fun: 0 sparse softmax cross-entropy (= avsr_seq_loss);
 * 1 label smoothing through tf.losses.softmax_cross_entropy (avsr/devel.py:54-61), whose default reduction turns the
 * sequence loss into the mean over ALL B*L rows -- denom is then the row count (compute_denom=1 sets B*L), padding rows
 * contribute log V each and no gradient; 2 focal_loss (gamma 2) and 3 mc_loss on the clipped softmax (avsr/devel.py:12-51). */
int avsr_seq_loss_fun(float* logits, const int32_t* labels, const int32_t* labels_len, float* denom, int32_t compute_denom,
                      float* row_loss, float* dlogits, int32_t B, int32_t L, int32_t V, int32_t loss_fun,
                      float label_smoothing, void* stream);

/* Per-utterance average of the step losses avsr_seq_loss left in row_loss (which carry the 1/denom factor):
 * out[b] = sum_l CE[b,l] w[b,l] / (sum_l w[b,l] + 1e-12) -- the language model's `average_log_likelihoods`
 * (avsr/lm.py:390-401: sequence_loss(average_across_batch=False, average_across_timesteps=True)). */
int avsr_seq_loss_per_utterance(const float* row_loss, const int32_t* labels_len, const float* denom, float* out, int32_t B,
                                int32_t L, void* stream);
/* tf.contrib.layers.instance_norm on the [B,T,F] encoder inputs (avsr/encoder.py:51-55): mean / variance over the T axis per
 * (utterance, feature), zero padding included, epsilon 1e-6, per-feature gamma / beta.  mean_out / invstd_out [B][F].
 * Backward: dx (may alias dy) and the per-utterance sums dgamma_part / dbeta_part [B][F] (column-sum them over B). */
int avsr_instnorm_fwd(const float* x, float* y, int32_t B, int32_t T, int32_t F, const float* gamma, const float* beta,
                      float* mean_out, float* invstd_out, float eps, void* stream);
int avsr_instnorm_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* invstd, float* dx,
                      float* dgamma_part, float* dbeta_part, int32_t B, int32_t T, int32_t F, void* stream);
/* tf.contrib.rnn.HighwayWrapper around an encoder cell (`highway_encoder=True`, avsr/cells.py:89-90), applied to a whole layer:
 * carry = sigmoid(carry_pre), y = x * carry + h * (1 - carry) for t < len[b], zero past the utterance.  x = the layer's raw
 * input, h = the cell's emitted output, carry_pre = x W_c + b_c (avsr_gemm).  Operands are [B*T, H] row views (avsr_mat with
 * T = rows per utterance).  Backward: dh = dy (1 - carry), dcarry_pre = dy (x - h) carry (1 - carry), dx (+)= dy carry. */
int avsr_highway_fwd(const avsr_mat* x, const avsr_mat* h, const avsr_mat* carry_pre, const avsr_mat* y, const int32_t* len,
                     int32_t B, int32_t T, int32_t H, void* stream);
int avsr_highway_bwd(const avsr_mat* x, const avsr_mat* h, const avsr_mat* carry_pre, const avsr_mat* dy, const avsr_mat* dh,
                     const avsr_mat* dcarry_pre, const avsr_mat* dx, const int32_t* len, int32_t B, int32_t T, int32_t H,
                     int32_t accumulate_dx, void* stream);
/* dst = src / dst = 0 over n 32-bit words, as KERNELS on the given stream.  The engine (and the host code around captured graphs)
 * never enqueues hipMemsetAsync / hipMemcpyAsync: as graph memset / memcpy nodes they were not reliably ordered against their
 * neighbours on ROCm 7.0.  No reference counterpart. */
int avsr_copy_words(void* dst, const void* src, int64_t n_words, void* stream);
int avsr_zero_words(void* dst, int64_t n_words, void* stream);
/* Any number of buffers zeroed eight per launch (the gradient buffers at the start of the backward pass), and
 * out[0] = a[0] + b on the device (the step's dropout / sampling RNG key = global step + per-rank offset, avsr/cells.py:46-54 masks). */
int avsr_zero_multi(void* const* ptrs, const int64_t* n_words, int32_t count, void* stream);
int avsr_add_int(const int32_t* a, int32_t b, int32_t* out, void* stream);
/* Many independent column sums (avsr_colsum semantics per job: out[f] = alpha * sum_r a[r][f] (* b[r][f]) + beta*out[f]) in two
 * launches: the bias gradients of a train step (seq2seq.py:222).  scratch >= sum_j ceil(rows_j / max(32, ceil(rows_j/256))) * F_j floats. */
typedef struct avsr_colsum_job {
  avsr_mat a;
  avsr_mat b;              /* b.ptr == NULL: plain column sums */
  float* out;
  int32_t rows, F;
  float alpha, beta;
} avsr_colsum_job;
int avsr_colsum_multi(const avsr_colsum_job* jobs, int32_t n, float* scratch, int64_t scratch_floats, void* stream);
/* Deferred weight-gradient reductions of the lip CNN (video.py:57-88 layers; seq2seq.py:222 tf.gradients of their kernels / biases):
 * between _begin and _end every avsr_conv_bwd_weight on this thread only RECORDS the final sum over its per-workgroup partial slabs;
 * _end runs all of them as one launch.  The caller passes each convolution its OWN scratch region (the slabs must survive until _end:
 * at most 512 * max(k*k*Ci*Co + Co, 12*Ci*16 + 16) floats per call) and must not read dw / dbias in between. */
int avsr_slab_defer_begin(void);
int avsr_slab_defer_end(void* stream);
/* AU regression loss (avsr/encoder.py:173-189); z = pre-sigmoid Dense(2) outputs [B][T][2]. */
int avsr_au_loss(const float* z, const float* aus, const int32_t* len, float* row_loss, float* dz, int32_t B, int32_t T,
                 float weight, void* stream);
/* Data-parallel form: the masked mean runs over the GLOBAL batch -- total_count[0] (device) = all-reduced 2 * sum_b min(len_b, T);
 * every rank then contributes its share of the numerator.  NULL = the local count (= avsr_au_loss). */
int avsr_au_loss_dp(const float* z, const float* aus, const int32_t* len, float* row_loss, float* dz, int32_t B, int32_t T,
                    float weight, const float* total_count, void* stream);

int avsr_normed_v(const float* v, const float* g, float* vn, int32_t H, void* stream);
int avsr_normed_v_bwd(const float* v, const float* g, const float* dvn, float* dv, float* dg, int32_t H, void* stream);

/* out[0] (+)= scale * (do_sqrt ? sqrt(sum part) : sum part) */
int avsr_reduce_scalar(const float* part, int32_t n, float* out, int32_t do_sqrt, int32_t accumulate, float scale,
                       void* stream);

/* l2_regularizer on the RNN kernels (avsr/seq2seq.py:175-178): grads += l2*w; loss_accum += 0.5*l2*sum w^2.
 * seg_off/seg_n are HOST arrays of flat-buffer segments; scratch >= 64*nseg floats. */
int avsr_l2_regularise(const int64_t* seg_off, const int64_t* seg_n, int32_t nseg, const float* params, float* grads,
                       float l2, float* loss_accum, float* scratch, void* stream);
/* norm_out[0] = grad_scale * ||grads||_2 ; scratch >= 1024 floats */
int avsr_global_norm(const float* grads, int64_t n, float grad_scale, float* norm_out, float* scratch, void* stream);
/* tf.clip_by_global_norm + tf.train.AdamOptimizer(eps=1e-8) + linear warm-up (avsr/seq2seq.py:195-199, :245-246,
 * :275-280).  step[0] (device int32) is read as global_step and incremented. clip_norm <= 0 disables clipping. */
int avsr_adam_step(float* params, float* grads, float* m, float* v, int64_t n, const float* global_norm, int32_t* step,
                   float lr, int32_t warmup_steps, float clip_norm, float grad_scale, void* stream);
/* Same with lr_decay=('cosine_restarts', first_decay_steps) (avsr/seq2seq.py:266-270: tf.train.cosine_decay_restarts
 * with its defaults t_mul=2, m_mul=1, alpha=0, evaluated at global_step; the warm-up factor multiplies the result).
 * first_decay_steps == 0: constant learning rate. */
int avsr_adam_step_decay(float* params, float* grads, float* m, float* v, int64_t n, const float* global_norm,
                         int32_t* step, float lr, int32_t warmup_steps, int32_t first_decay_steps, float clip_norm,
                         float grad_scale, void* stream);
/* The reference's other optimisers (avsr/seq2seq.py:195-218).  optimiser: 0 Adam (= avsr_adam_step_decay), 1 Nadam
 * (contrib.opt.NadamOptimizer: Adam with the Nesterov numerator beta1*m + (1-beta1)*g), 2 AdamW (contrib.opt.AdamWOptimizer:
 * var -= weight_decay * var, then Adam), 3 Momentum(0.9, use_nesterov=False) with the accumulator in m (v unused). */
int avsr_optimiser_step(float* params, float* grads, float* m, float* v, int64_t n, const float* global_norm, int32_t* step,
                        float lr, int32_t warmup_steps, int32_t first_decay_steps, float clip_norm, float grad_scale,
                        int32_t optimiser, float weight_decay, void* stream);

/* Optional per-launch HIP-event timing of the engine's own kernels (bench.py roofline figures).  Between
 * begin and end every gemm / step / attention launch is bracketed by an event pair on its stream; end
 * synchronises the device and returns per-kind launch counts and summed milliseconds.
 * kinds: 0 gemm, 1 LSTM-forward step, 2 LSTM-backward step, 3 dense step, 4 attention fwd, 5 attention bwd,
 * 6 persistent RNN forward, 7 persistent RNN backward, 8 / 9 fused persistent decoder forward / backward, 10 / 11 / 12 convolution forward / data gradient / weight gradient, 13 / 14 the AV-Align attentive layer's fused persistent forward / backward.  out_flops (may be NULL): algorithmic FLOPs summed per kind
 * where the launcher knows them (gemm, persistent RNN kernels), else 0. */
#define AVSR_PROF_NKIND 15
int avsr_prof_begin(int32_t max_launches);
int avsr_prof_end(int32_t* out_count, float* out_ms, double* out_flops);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* AVSR_HIP_H */
