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

#define AVSR_OK 0
#define AVSR_ERR_ARG (-1)
#define AVSR_ERR_HIP (-2)
#define AVSR_ERR_UNSUPPORTED (-3)

#define AVSR_MAX_LAYERS 4
#define AVSR_MAX_MECH 4
#define AVSR_MAX_STACKS 4

int avsr_abi_version(void);

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
  int32_t splitk;
  int32_t pad_;
  float* workspace;
  int64_t workspace_floats;
} avsr_gemm_desc;

int avsr_gemm(const avsr_gemm_desc* d, void* stream);

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
  int32_t dout_col, pad_;
} avsr_rnn_layer;

typedef struct avsr_rnn_stack {
  int32_t B, T, reverse, n_layers;
  int32_t cell, pad_;             /* 0 = LSTM */
  const int32_t* len;
  const float* dh_final;          /* [B][H_top] gradient wrt the top layer's final h (may be NULL) */
  const float* dc_final;
  avsr_rnn_layer layer[AVSR_MAX_LAYERS];
} avsr_rnn_stack;

int avsr_rnn_fwd(const avsr_rnn_stack* stacks, int32_t n_stacks, void* stream);
int avsr_rnn_bwd(const avsr_rnn_stack* stacks, int32_t n_stacks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AVSR_HIP_H */
