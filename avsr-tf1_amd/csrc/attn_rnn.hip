// Host-side scheduler for the attention-wrapped LSTM sequence (decoder train / greedy, AV-Align
// top encoder layer) -- see avsr_hip.h for the contract and the reference call sites.
//
// Forward chain per step:  [LSTM step] -> [Bahdanau query layer]* -> [attention partials] ->
//                          [attention layer (merges partials)] -> ([logits] -> [sample])^greedy
// Backward chain per step: [d attention] -> [d context] -> [attention backward] -> [partial sums]
//                          -> [LSTM backward]
// Every box is one launch covering all mechanisms; weight/keys gradients are deferred to post-loop
// GEMMs over all L steps (done by the caller with avsr_gemm).
#include "step.h"
#include "attn.h"

extern "C" int avsr_step_launch_raw(const void* launch, void* stream);
extern "C" int avsr_attn_launch_raw(const void* launch, int backward, void* stream);
namespace avsr {
bool beam_dense_on();
int beam_cell_launch(const avsr_attn_rnn& d, int l, hipStream_t s);
int beam_attention_layer_launch(const avsr_attn_rnn& d, int l, hipStream_t s);
}
int avsr_dec_persist_fwd(const avsr_attn_rnn* d, int32_t l_begin, int32_t l_end, void* stream);   // dec_persist.hip
int avsr_dec_persist_bwd(const avsr_attn_rnn* d, void* stream);                                      // dec_persist_bwd.hip

namespace avsr {

struct SlabJob {
  float* dst; long dst_sb;
  const float* add; long add_sb;
  const float* src[AVSR_MAX_MECH]; int nslab[AVSR_MAX_MECH];
  int nsrc, W;
};
struct SlabLaunch { int njob, B; SlabJob job[AVSR_MAX_MECH + 1]; };

// dst[b, k] = add[b, k] + sum_sets sum_c src[c][b][k]
__global__ void slab_sum_kernel(const SlabLaunch L) {
  const SlabJob& J = L.job[blockIdx.y];
  const long total = (long)L.B * J.W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / J.W), k = (int)(i % J.W);
    float s = J.add ? J.add[(long)b * J.add_sb + k] : 0.f;
    for (int j = 0; j < J.nsrc; ++j)
      for (int c = 0; c < J.nslab[j]; ++c) s += J.src[j][((long)c * L.B + b) * J.W + k];
    J.dst[(long)b * J.dst_sb + k] = s;
  }
}

// GreedyEmbeddingHelper.sample/next_inputs + dynamic_decode finished bookkeeping (one block).
__global__ void greedy_sample_kernel(const float* logits, long logits_sb, int V, int32_t* ids, long ids_sb, int32_t* tok,
                                     int32_t* steplen, int32_t* n_unfinished, int B, int l, int eos) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int local = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    int id = 0;
    if (l < steplen[b]) {
      const float* lg = logits + (long)b * logits_sb;
      float best = lg[0];
      for (int v = 1; v < V; ++v)
        if (lg[v] > best) { best = lg[v]; id = v; }   // first maximum (tf.argmax)
      tok[b] = id;
      if (id == eos) steplen[b] = l + 1; else local += 1;
    }
    ids[(long)b * ids_sb] = id;
  }
  atomicAdd(&cnt, local);
  __syncthreads();
  if (threadIdx.x == 0) n_unfinished[0] = cnt;
}

// ScheduledEmbeddingTrainingHelper.sample/next_inputs (contrib/seq2seq/python/ops/helper.py): per utterance draw
// select ~ Bernoulli(p); where selected feed the embedding of a categorical sample of this step's logits, else the
// ground-truth token.  One block per utterance; the draw is an fp32 inverse CDF in index order (oracle: categorical_f32).
__global__ void sched_sample_kernel(const float* logits, long logits_sb, int V, const int32_t* labels, int32_t* fed, float* xs,
                                    const float* emb, int B, int L, int E, int l, const int32_t* seed, float prob,
                                    float keep_in, uint32_t r_in, int in_W) {
  __shared__ int tok_s;
  extern __shared__ float lg_s[];          // this utterance's logits: fetched by all lanes together with the seed and the
  const int b = blockIdx.x;                // label, so the serial inverse-CDF below runs out of LDS (one memory round trip)
  if (l + 1 >= L) return;
  const uint32_t sd0 = (uint32_t)seed[0];
  const int label = labels[(long)b * L + l];
  {
    const float* lg = logits + (long)b * logits_sb;
    for (int v = threadIdx.x; v < V; v += blockDim.x) lg_s[v] = lg[v];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t sd = sd0;
    const uint32_t idx = (uint32_t)(b * L + l);
    int tok = label;
    if (prob > 0.f && uniform01(sd, 1000u, idx) < prob) {
      const float* lg = lg_s;
      float mx = lg[0];
      for (int v = 1; v < V; ++v) mx = fmaxf(mx, lg[v]);
      float tot = 0.f;
      for (int v = 0; v < V; ++v) tot += expf(lg[v] - mx);
      const float target = uniform01(sd, 1001u, idx) * tot;
      float run = 0.f;
      tok = V - 1;
      for (int v = 0; v < V; ++v) {
        run += expf(lg[v] - mx);
        if (run > target) { tok = v; break; }
      }
    }
    fed[(long)b * L + l + 1] = tok;
    tok_s = tok;
  }
  __syncthreads();
  const int tok = tok_s;
  const bool on = keep_in < 1.0f;
  const uint32_t sd = sd0;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float v = emb[(long)tok * E + e];
    if (on) v = uniform01(sd, r_in, (uint32_t)(((long)b * L + l + 1) * in_W + e)) < keep_in ? v / keep_in : 0.f;
    xs[((long)b * L + l + 1) * E + e] = v;
  }
}


// Output layer + ScheduledEmbeddingTrainingHelper step fused into ONE launch of the decoder's sequential chain (it replaces a
// dense step launch + sched_sample_kernel): block = utterance b.
//   logits[b, :] = x[b, :] . Wout + bout   (zero past the utterance's step length, as dynamic_decode's impute_finished)
//   then the draw of sched_sample_kernel above and the (input-dropped) embedding of the token fed to step l+1.
// Thread (vc = tid & 31, ks = tid >> 5): column v0 + vc, K slice ks of 8 (16-byte loads, Wout^T rows are K-contiguous).
__global__ __launch_bounds__(256) void logits_sample_kernel(const float* x, long x_sb, int O, const float* wout_t, const float* bout,
                                                            float* logits, long logits_sb, int V, const int32_t* steplen,
                                                            const int32_t* labels, int32_t* fed, float* xs, const float* emb, int B, int L,
                                                            int E, int l, const int32_t* seed, float prob, float keep_in, uint32_t r_in,
                                                            int in_W) {
  extern __shared__ float lg_s[];          // [V] logits of this utterance
  __shared__ float part[8][33];
  __shared__ int tok_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int vc = tid & 31, ks = tid >> 5;
  const uint32_t sd0 = seed ? (uint32_t)seed[0] : 0u;
  const int label = (l + 1 < L) ? labels[(long)b * L + l] : 0;
  const bool valid = !(steplen && l >= steplen[b]);
  const float* xr = x + (long)b * x_sb;
  const int kper = ((O + 31) / 32) * 4;    // K slice length, multiple of 4
  const int k0 = ks * kper, k1 = min(O, k0 + kper);
  for (int v0 = 0; v0 < V; v0 += 32) {
    const int v = v0 + vc;
    float a = 0.f;
    if (v < V) {
      const float* wr = wout_t + (long)v * O;
      for (int k = k0; k < k1; k += 4) {
        const f32x4 xv = ld4(xr + k), wv = ld4(wr + k);
        a += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
      }
    }
    part[ks][vc] = a;
    __syncthreads();
    if (ks == 0 && v < V) {
      float z = ((part[0][vc] + part[1][vc]) + (part[2][vc] + part[3][vc])) + ((part[4][vc] + part[5][vc]) + (part[6][vc] + part[7][vc]));
      z = valid ? z + (bout ? bout[v] : 0.f) : 0.f;
      lg_s[v] = z;
      logits[(long)b * logits_sb + v] = z;
    }
    __syncthreads();
  }
  if (l + 1 >= L) return;
  if (tid == 0) {
    const uint32_t idx = (uint32_t)(b * L + l);
    int tok = label;
    if (prob > 0.f && uniform01(sd0, 1000u, idx) < prob) {
      float mx = lg_s[0];
      for (int v = 1; v < V; ++v) mx = fmaxf(mx, lg_s[v]);
      float tot = 0.f;
      for (int v = 0; v < V; ++v) tot += expf(lg_s[v] - mx);
      const float target = uniform01(sd0, 1001u, idx) * tot;
      float run = 0.f;
      tok = V - 1;
      for (int v = 0; v < V; ++v) {
        run += expf(lg_s[v] - mx);
        if (run > target) { tok = v; break; }
      }
    }
    fed[(long)b * L + l + 1] = tok;
    tok_s = tok;
  }
  __syncthreads();
  const int tok = tok_s;
  const bool on = keep_in < 1.0f;
  for (int e = tid; e < E; e += blockDim.x) {
    float v = emb[(long)tok * E + e];
    if (on) v = uniform01(sd0, r_in, (uint32_t)(((long)b * L + l + 1) * in_W + e)) < keep_in ? v / keep_in : 0.f;
    xs[((long)b * L + l + 1) * E + e] = v;
  }
}

// One BeamSearchDecoder step for one utterance (one 256-thread workgroup per utterance): see avsr_hip.h mode 3 and
// oracle.beam_search_decode.  Everything the step reads comes in with ONE round of loads; the per-beam log-sum-exp is a wave reduction,
// its logf / the 2 K length penalties (powf) run one per lane side by side; every thread keeps its (at most two) candidates
// -- score as a sortable 32-bit key, accumulated log-probability -- in REGISTERS, and the top K are K rounds of
// {wave arg-max of a 64-bit (score key, ~index) word, four wave winners through LDS}: larger score first, LOWER index on ties
// (= tf.nn.top_k's order; -inf scores stay selectable, in index order).
// History, measured at c4 (B * K = 640 rows, V = 31; tools/beam_gemm_dissect.sh): round 2 one thread selecting serially 165 us;
// round 3 one wave, candidates in LDS, parents' flags loaded inside every round 22 us; rank by counting (every candidate against
// every other, broadcast LDS reads) 28 us -- 310 x 310 comparisons are more instructions than ten reductions.
// NCT > 0: the output layer runs here too (seq2seq.py:339 the decoder's Dense(vocab) under BeamSearchDecoder): the utterance's K rows of
// the attention output [K x O] times Wout^T [O x V] as NCT 16-column tiles of v_mfma_f32_16x16x4_f32, wave w taking the O / 4 inputs
// [w O / 4, (w + 1) O / 4), the four partial tiles summed through LDS -- one launch and one read-back of the logits less per step
// (the separate projection was a 7 us launch of 640 x 31 outputs).  Needs K <= 16, V <= 16 NCT, O % 256 == 0.  NCT == 0: logits given.
template <int NCT>
__global__ __launch_bounds__(256) void beam_step_kernel(float* logits, long logits_sb, int V, int K, int l, int eos, float w,
                                 const float* logp_in, const int32_t* fin_in, const int32_t* len_in,
                                 float* logp_out, int32_t* fin_out, int32_t* len_out, int32_t* tok, int32_t* parent_rows,
                                 int32_t* step_ids, int32_t* parent_ids, int32_t* n_unfinished,
                                 const float* xa, long xsb, int O, const float* wout_t, const float* bout) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // logits [K*V] | lse [K] | max [K] | sum [K] | penalties [K][2] | fin [K] | len [K] | logp [K] | alive | winners [2][4] x 64 bit
  const int n = K * V;
  float* lg_s = sm;
  float* lse_s = lg_s + n;
  float* mx_s = lse_s + K;
  float* sum_s = mx_s + K;
  float* pen = sum_s + K;
  int* fin_s = reinterpret_cast<int*>(pen + 2 * K);
  int* len_s = fin_s + K;
  float* logp_s = reinterpret_cast<float*>(len_s + K);
  int* alive_s = reinterpret_cast<int*>(logp_s + K);
  unsigned long long* win = reinterpret_cast<unsigned long long*>(sm + ((n + 8 * K + 1 + 1) & ~1));      // 8-byte aligned
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float FMIN = -3.4028234663852886e38f;
  // Steps queued past the end of the search (the host reads the "all finished" counter a chunk late): dynamic_decode has stopped, so
  // the step hands its input state through unchanged.  Without this a beam that finished at the last real step would be re-scored with
  // len + 1 and the beams re-ordered -- harmless for gather_tree (it reads step_ids / parent_ids [:T] and the maximum length) but not for
  // a caller reading the per-beam state afterwards.  n_unfinished[-1] is the previous step's count, complete when this launch starts.
  if (l > 0 && n_unfinished[-1] == 0) {
    if (tid < K) {
      const int r = b * K + tid;
      logp_out[r] = logp_in[r]; fin_out[r] = fin_in[r]; len_out[r] = len_in[r];
      tok[r] = eos; parent_rows[r] = r; step_ids[r] = eos; parent_ids[r] = tid;
    }
    return;
  }
  if (tid < K) { fin_s[tid] = fin_in[b * K + tid]; len_s[tid] = len_in[b * K + tid]; logp_s[tid] = logp_in[b * K + tid]; }
  if (tid == 0) alive_s[0] = 0;
  if constexpr (NCT > 0) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    float* part = reinterpret_cast<float*>(win + 8);               // [4 waves][16 rows][16 NCT columns]
    const int i16 = lane & 15, g = lane >> 4, kw = O >> 2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const float* xr = xa + (long)(b * K + (i16 < K ? i16 : 0)) * xsb + wave * kw + 4 * g;
    const float* wr[NCT];
    f32x4 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const int v = c * 16 + i16;
      wr[c] = wout_t + (long)(v < V ? v : 0) * O + wave * kw + 4 * g;
      acc[c] = zero;
    }
    const bool arow = i16 < K;
#pragma unroll 1
    for (int kk = 0; kk < kw; kk += 64) {                         // lane (i16, g) holds inputs kk + 16 u + 4 g .. + 3 of row / column i16
      f32x4 a[4], bv[4][NCT];                                     // (all loads of four 16-input groups in flight, then their products)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = *reinterpret_cast<const f32x4*>(xr + kk + 16 * u);
#pragma unroll
        for (int c = 0; c < NCT; ++c) bv[u][c] = *reinterpret_cast<const f32x4*>(wr[c] + kk + 16 * u);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!arow) a[u] = zero;
#pragma unroll
        for (int c = 0; c < NCT; ++c) if (c * 16 + i16 >= V) bv[u][c] = zero;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < NCT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], bv[u][c][e], acc[c], 0, 0, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(wave * 16 + 4 * g + r) * (16 * NCT) + c * 16 + i16] = acc[c][r];
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
      const int k = i / V, v = i - k * V, o = k * (16 * NCT) + v;
      const float sv = ((part[o] + part[256 * NCT + o]) + (part[512 * NCT + o] + part[768 * NCT + o])) + bout[v];
      lg_s[i] = sv;
      logits[(long)(b * K + k) * logits_sb + v] = sv;
    }
  } else {
    for (int i = tid; i < n; i += 256) {
      const int k = i / V, v = i - k * V;
      lg_s[i] = logits[(long)(b * K + k) * logits_sb + v];
    }
  }
  __syncthreads();
#if defined(BS_STOP) && BS_STOP <= 1
  if (tid == 0) n_unfinished[0] = (int)lg_s[0];
  return;
#endif
  for (int k = wave; k < K; k += 4) {                 // max and exp-sum of beam k: wave reductions over its V logits
    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, lg_s[k * V + v]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int v = lane; v < V; v += 64) sum += expf(lg_s[k * V + v] - mx);
    sum = wave_sum(sum);
    if (lane == 0) { mx_s[k] = mx; sum_s[k] = sum; }
  }
  __syncthreads();
  // one logf / powf per lane: K log-sum-exps on wave 0, the 2 K length penalties ((5 + len) / 6) ^ w of the two possible lengths of a
  // beam's continuations on waves 1-3 (the same powf values as one call per candidate)
  if (tid < K) lse_s[tid] = mx_s[tid] + logf(sum_s[tid]);
  for (int j = tid - 64; j >= 0 && j < 2 * K; j += 192) pen[j] = powf((5.0f + (float)(len_s[j >> 1] + (j & 1))) / 6.0f, w);
  __syncthreads();
#if defined(BS_STOP) && BS_STOP <= 2
  if (tid == 0) n_unfinished[0] = (int)lse_s[0];
  return;
#endif
  // ---- this thread's candidates i = tid + 256 c: key (0 = none / taken) and accumulated log-probability, in registers ----
  constexpr int NC = 4;                               // K * V <= 1024
  unsigned key[NC];
  float tot[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = tid + 256 * c;
    key[c] = 0u; tot[c] = 0.f;
    if (i < n) {
      const int k = i / V, v = i - k * V;
      const bool fin = fin_s[k] != 0;
      const float sl = fin ? (v == eos ? 0.f : FMIN) : lg_s[i] - lse_s[k];
      const float t = logp_s[k] + sl;
      const float sc = t / pen[2 * k + ((fin || v == eos) ? 0 : 1)];
      const unsigned u = __builtin_bit_cast(unsigned, sc);
      key[c] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);        // monotone in the score; -inf -> 0x007fffff (> 0 = none)
      tot[c] = t;
    }
  }
#if defined(BS_STOP) && BS_STOP <= 3
  if (tid == 0) n_unfinished[0] = (int)key[0];
  return;
#endif
  int alive = 0;
  for (int j = 0; j < K; ++j) {
    // this thread's best remaining candidate as one 64-bit word: (score key, ~index) -- larger is better, lower index wins ties
    unsigned long long best = 0ull;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const unsigned long long cand = key[c] ? (((unsigned long long)key[c] << 32) | (unsigned)(0xffffffffu - (unsigned)(tid + 256 * c))) : 0ull;
      best = cand > best ? cand : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned hi = (unsigned)__shfl_xor((int)(best >> 32), o, 64), lo = (unsigned)__shfl_xor((int)(unsigned)best, o, 64);
      const unsigned long long other = ((unsigned long long)hi << 32) | lo;
      best = other > best ? other : best;
    }
    unsigned long long* slot = win + (j & 1) * 4;
    if (lane == 0) slot[wave] = best;
    __syncthreads();
    unsigned long long g = slot[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) g = slot[q] > g ? slot[q] : g;
    const int idx = (int)(0xffffffffu - (unsigned)g);
    if ((idx & 255) == tid) {                           // the owner writes hypothesis j and retires the candidate
      const int c = idx >> 8;
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < NC; ++q) if (q == c) { t = tot[q]; key[q] = 0u; }
      const int word = idx % V, parent = idx / V, r = b * K + j, pr = b * K + parent;
      const bool pf = fin_s[parent] != 0;
      const int f = (pf || word == eos) ? 1 : 0;
      logp_out[r] = t;
      fin_out[r] = f;
      len_out[r] = len_s[parent] + (pf ? 0 : 1);
      tok[r] = word;
      parent_rows[r] = pr;
      step_ids[r] = word;
      parent_ids[r] = parent;
      if (!f) ++alive;
    }
  }
  if (alive) atomicAdd(alive_s, alive);
  __syncthreads();
  if (tid == 0 && alive_s[0]) atomicAdd(n_unfinished, alive_s[0]);
}

__global__ void beam_gather_tree_kernel(const int32_t* step_ids, const int32_t* parent_ids, const int32_t* beam_len, int32_t* out,
                                        int nutt, int K, int T, int eos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nutt * K) return;
  const int b = i / K, k = i % K, R = nutt * K;
  int ml = 0;
  for (int j = 0; j < K; ++j) ml = max(ml, beam_len[b * K + j]);
  ml = min(ml, T);
  int32_t* o = out + (long)b * T * K + k;              // out[b][t][k]
  for (int t = 0; t < T; ++t) o[(long)t * K] = eos;
  if (ml <= 0) return;
  int parent = parent_ids[(long)(ml - 1) * R + b * K + k];
  o[(long)(ml - 1) * K] = step_ids[(long)(ml - 1) * R + b * K + k];
  for (int level = ml - 2; level >= 0; --level) {
    o[(long)level * K] = step_ids[(long)level * R + b * K + parent];
    parent = parent_ids[(long)level * R + b * K + parent];
  }
  bool seen = false;
  for (int t = 0; t < ml; ++t) {
    if (seen) o[(long)t * K] = eos;
    else if (o[(long)t * K] == eos) seen = true;
  }
}

static inline float* hbuf(const avsr_attn_rnn& d, int p) { return d.state + (long)p * d.B * d.H; }
static inline float* cbuf(const avsr_attn_rnn& d, int p) { return d.state + (long)(2 + p) * d.B * d.H; }
static inline float* dgroll(const avsr_attn_rnn& d, int p) { return d.dstate + (long)p * d.B * 4 * d.H; }
static inline float* dcbuf(const avsr_attn_rnn& d, int p) { return d.dstate + (long)(8 + p) * d.B * d.H; }
static inline float* dhcarry(const avsr_attn_rnn& d, int p) { return d.dstate + (long)(10 + p) * d.B * d.H; }
// extra decoder layers (MultiRNNCell above the attention-fed cell): same buffer conventions as the block's own cell
static inline float* xhbuf(const avsr_attn_rnn& d, int j, int p) { return d.extra[j].state + (long)p * d.B * d.H; }
static inline float* xcbuf(const avsr_attn_rnn& d, int j, int p) { return d.extra[j].state + (long)(2 + p) * d.B * d.H; }
static inline float* xdgroll(const avsr_attn_rnn& d, int j, int p) { return d.extra[j].dstate + (long)p * d.B * 4 * d.H; }
static inline float* xdcbuf(const avsr_attn_rnn& d, int j, int p) { return d.extra[j].dstate + (long)(8 + p) * d.B * d.H; }
static inline float* xdhcarry(const avsr_attn_rnn& d, int j, int p) { return d.extra[j].dstate + (long)(10 + p) * d.B * d.H; }
// output record of decoder layer j (0 = the attention-fed cell): [B][L+1][H], slot l+1 = step l
// GRU extra layers: the same dstate layout as the block's own GRU cell (d gates rolling [2][B][2H] | d cand [2][B][H] | carry [2][B][H] | 2 tmp)
static inline float* xg_dgg(const avsr_attn_rnn& d, int j, int p) { return d.extra[j].dstate + (long)p * d.B * 2 * d.H; }
static inline float* xg_dpc(const avsr_attn_rnn& d, int j, int p) { return d.extra[j].dstate + (long)(4 + p) * d.B * d.H; }
static inline float* xg_carry(const avsr_attn_rnn& d, int j, int p) { return d.extra[j].dstate + (long)(6 + p) * d.B * d.H; }
static inline float* xg_tmp(const avsr_attn_rnn& d, int j, int w) { return d.extra[j].dstate + (long)(8 + w) * d.B * d.H; }
static inline float* layer_out(const avsr_attn_rnn& d, int j) { return j == 0 ? (d.n_extra > 0 ? d.out0 : d.cell_out) : d.extra[j - 1].out; }
static inline int nchunk(const avsr_attn_mech& m) { return (m.T + m.chunk - 1) / m.chunk; }
static inline bool is_bahdanau(const avsr_attn_mech& m) { return m.type >= ATT_BAHDANAU; }

static int validate(const avsr_attn_rnn* d) {
  if (!d || d->B <= 0 || d->L <= 0 || d->H <= 0 || d->H % 4 || d->E % 4 || d->n_mech < 0 || d->n_mech > AVSR_MAX_MECH)
    return AVSR_ERR_ARG;
  if (!d->wt || !d->gates || !d->cs || !d->cell_out || !d->state || !d->steplen) return AVSR_ERR_ARG;
  if (d->n_mech > 0 && !d->att) return AVSR_ERR_ARG;
  if (d->n_extra < 0 || d->n_extra > AVSR_MAX_DEC_EXTRA) return AVSR_ERR_ARG;
  if (d->n_extra > 0) {
    if (!d->out0 || d->extra[d->n_extra - 1].out != d->cell_out) return AVSR_ERR_ARG;
    for (int j = 0; j < d->n_extra; ++j) {
      const avsr_dec_layer& X = d->extra[j];
      if (!X.wt || !X.gates || !X.cs || !X.out || !X.state) return AVSR_ERR_ARG;
      if (d->cell == 1 && (!X.wt2 || !X.rh_seq)) return AVSR_ERR_ARG;  // GRU layers: candidate kernel and the r*h record
    }
  }
  for (int m = 0; m < d->n_mech; ++m) {
    const avsr_attn_mech& M = d->mech[m];
    if (M.T <= 0 || M.D % 4 || M.chunk <= 0 || M.chunk > ATTN_MAX_CHUNK || nchunk(M) > STEP_MAX_SLAB) return AVSR_ERR_ARG;
    if (M.D > 1024 || d->H > 1024) return AVSR_ERR_UNSUPPORTED;
    if (!M.keys || !M.values || !M.watt_t || !M.scores || !M.ctx || !M.pstat || !M.pctx) return AVSR_ERR_ARG;
    if (M.type < 0 || M.type > ATT_NORMED_BAHDANAU) return AVSR_ERR_UNSUPPORTED;
    if (M.type == ATT_SCALED_LUONG && !M.g) return AVSR_ERR_ARG;
    if (is_bahdanau(M) && (!M.v || !M.wq_t || !M.pq)) return AVSR_ERR_ARG;
  }
  return AVSR_OK;
}

static void fill_attn_launch(const avsr_attn_rnn& d, int l, AttnLaunch& AL) {
  const int B = d.B, H = d.H, L = d.L;
  AL = AttnLaunch{};
  AL.nmech = d.n_mech; AL.B = B;
  int off = 0;
  for (int m = 0; m < d.n_mech; ++m) {
    const avsr_attn_mech& M = d.mech[m];
    AttnMechDev& X = AL.m[m];
    const int nc = nchunk(M);
    AL.blk_off[m] = off;
    off += B * nc;
    X.keys = M.keys; X.values = M.values; X.values_sb = M.values_sb; X.values_st = M.values_st; X.len = M.len;
    if (is_bahdanau(M)) { X.query = M.pq + (long)l * H; X.query_sb = (long)L * H; }
    else { X.query = d.cell_out + (long)(l + 1) * H; X.query_sb = (long)(L + 1) * H; }
    X.g = M.g; X.v = M.v; X.bq = M.bq;
    X.scores = M.scores + (long)l * M.T; X.scores_sb = (long)L * M.T;
    X.pm = M.pstat + (long)(2 * l) * nc * B; X.pl = M.pstat + (long)(2 * l + 1) * nc * B;
    X.pctx = M.pctx;
    X.T = M.T; X.D = M.D; X.H = H; X.type = M.type; X.nchunk = nc; X.chunk = M.chunk;
    X.mem_div = (d.mode == 3 && d.mem_shared && d.beam_width > 1) ? d.beam_width : 1;
    if (M.dctx) { X.dctx = M.dctx + (long)l * M.D; X.dctx_sb = (long)L * M.D; }
    X.ctx = M.ctx + (long)l * M.D; X.ctx_sb = (long)L * M.D;
    if (M.dscores) { X.dscores = M.dscores + (long)l * M.T; X.dscores_sb = (long)L * M.T; }
    X.pdq = M.pdq;
  }
  AL.blk_off[d.n_mech] = off;
}

}  // namespace avsr

extern "C" int avsr_attn_rnn_fwd(const avsr_attn_rnn* dp, int32_t l_begin, int32_t l_end, void* stream) {
  using namespace avsr;
  int rc = validate(dp);
  if (rc) return rc;
  const avsr_attn_rnn& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  const int B = d.B, H = d.H, L = d.L, E = d.E, A = d.n_mech * H, KW = E + A + H;
  if (l_begin < 0 || l_end > L || l_begin > l_end) return AVSR_ERR_ARG;
  if (d.mode == 1 && (!d.embedding || !d.wout_t || !d.logits || !d.ids || !d.tok || !d.n_unfinished)) return AVSR_ERR_ARG;
  if (d.mode == 2 && (!d.embedding || !d.wout_t || !d.logits || !d.xs || !d.labels || !d.fed || !d.seed)) return AVSR_ERR_ARG;
  if (d.mode == 3 && (!d.embedding || !d.wout_t || !d.logits || !d.tok || !d.n_unfinished || d.beam_width <= 0 || B % d.beam_width ||
                      !d.beam_logp || !d.beam_fin || !d.beam_len || !d.step_ids || !d.parent_ids || !d.parent_rows)) return AVSR_ERR_ARG;
  if (d.mode == 3 && (d.beam_width * d.V > 1024 || d.beam_width > 64)) return AVSR_ERR_UNSUPPORTED;
  const bool feed = (d.mode == 1 || d.mode == 3);      // inputs come from the embedding of the previous prediction
  const bool gru = d.cell == 1;
  if (gru && (!d.wt2 || !d.rh_seq)) return AVSR_ERR_ARG;
  const bool drop = d.seed && d.mode != 1 && (d.keep_in < 1.f || d.keep_state < 1.f || d.keep_out < 1.f);
  if (drop && (!d.hs_seq || (A > 0 && !d.attd))) return AVSR_ERR_ARG;
  if (drop)
    for (int j = 0; j < d.n_extra; ++j)
      if (!d.extra[j].hs_seq || !d.extra[j].xin_seq) return AVSR_ERR_ARG;
  const uint32_t cid4 = (uint32_t)d.cell_id * 4;
  const size_t bh = sizeof(float) * B * H;

  if (l_begin == 0) {
    // initial state -> ping-pong parity 0, cell_out slot 0 = h0, attention slot 0 = 0: one launch (every operation reads the caller's
    // h0 / c0 or writes zeros: none reads another's result)
    avsr::DevBatch db(s);
    const size_t hrow = sizeof(float) * H, orow = sizeof(float) * (L + 1) * H;
    if (d.h0) { db.copy(hbuf(d, 0), d.h0, bh); db.copy_2d(layer_out(d, 0), orow, d.h0, hrow, hrow, B); }
    else { db.zero(hbuf(d, 0), bh); db.zero_2d(layer_out(d, 0), orow, hrow, B); }
    if (d.c0) db.copy(cbuf(d, 0), d.c0, bh);
    else db.zero(cbuf(d, 0), bh);
    for (int j = 0; j < d.n_extra; ++j) {            // layers above start from the zero state (decoder_unimodal.py:151-157)
      const avsr_dec_layer& X = d.extra[j];
      db.zero(X.state, 4 * bh);
      db.zero_2d(X.out, orow, hrow, B);
      if (drop && X.hs_seq) db.zero_2d(X.hs_seq, orow, hrow, B);
    }
    if (A > 0) db.zero_2d(d.att, sizeof(float) * (L + 1) * A, sizeof(float) * A, B);
    if (drop) {
      if (d.h0) db.copy_2d(d.hs_seq, orow, d.h0, hrow, hrow, B);
      else db.zero_2d(d.hs_seq, orow, hrow, B);
      if (A > 0) db.zero_2d(d.attd, sizeof(float) * (L + 1) * A, sizeof(float) * A, B);
    }
    if (db.flush() != hipSuccess) return AVSR_ERR_HIP;
  }

  static thread_local StepLaunch SL;
  static thread_local AttnLaunch AL;
  // the whole range as ONE persistent launch where the fused decode kernel covers the configuration (dec_persist.hip)
  int l_first = l_begin;
  {
    const int prc = avsr_dec_persist_fwd(dp, l_begin, l_end, stream);
    if (prc == AVSR_OK) l_first = l_end;
    else if (prc != AVSR_ERR_UNSUPPORTED) return prc;
  }
  // beam search: the per-step unfinished counters [L] of this call's range, zeroed once (not one fill per step)
  if (d.mode == 3 && l_first < l_end && avsr::dev_zero(d.n_unfinished + l_first, sizeof(int32_t) * (l_end - l_first), s) != hipSuccess)
    return AVSR_ERR_HIP;
  // beam search: the B * K-row cell and attention-layer steps as 64 x 64-tiled products with row-gathered operands (beam_gemm.hip)
  const bool beam_dense = d.mode == 3 && !gru && d.n_extra == 0 && !drop && beam_dense_on();
  for (int l = l_first; l < l_end; ++l) {
    // ---- K1: LSTM step -------------------------------------------------------------------
    bool cell_done = false;
    if (beam_dense) {
      const int brc = beam_cell_launch(d, l, s);
      if (brc == AVSR_OK) cell_done = true;
      else if (brc != AVSR_ERR_UNSUPPORTED) return brc;
    }
    for (int phase = 0; phase < (gru ? 2 : 1) && !cell_done; ++phase) {
      SL.ntask = 1;
      StepTask& tk = SL.task[0];
      tk = StepTask{};
      const float* wt = (gru && phase == 1) ? d.wt2 : d.wt;
      if (feed) {
        StepSrc& x = tk.src[tk.nsrc++];
        x.a = d.embedding; x.sb = E; x.K = E; x.w = wt; x.ldw = KW; x.kind = SRC_PLAIN;
        tk.gather = d.tok;
        if (d.mode == 3) tk.gather2 = d.parent_rows;
      } else if (d.mode == 2) {
        StepSrc& x = tk.src[tk.nsrc++];
        x.a = d.xs + (long)l * E; x.sb = (long)L * E; x.K = E; x.w = wt; x.ldw = KW; x.kind = SRC_PLAIN;
      }
      if (A > 0) {
        StepSrc& a = tk.src[tk.nsrc++];
        a.a = (drop ? d.attd : d.att) + (long)l * A; a.sb = (long)(L + 1) * A; a.K = A; a.w = wt + E; a.ldw = KW; a.kind = SRC_PLAIN;
      }
      StepSrc& h = tk.src[tk.nsrc++];
      h.a = (gru && phase == 1) ? hbuf(d, 2) /* r*h */ : hbuf(d, l & 1); h.sb = H; h.K = H; h.w = wt + E + A; h.ldw = KW;
      // r*h was written by the gate phase of THIS step into this hypothesis' own row: under beam search only the previous state
      // (h, and the attention fed back) lives in the parent's row
      h.kind = (gru && phase == 1) ? SRC_OWNROW : SRC_PLAIN;
      tk.B = B; tk.t = l; tk.T = L; tk.reverse = 0; tk.len = d.steplen;
      tk.s2 = (d.mode == 0) ? 1 : 0;
      tk.p4 = hbuf(d, l & 1);
      if (gru && phase == 0) {
        tk.N = 2 * H; tk.mode = EP_GRU_GATES; tk.bias = d.bias;
        tk.p0 = d.gates; tk.p1 = hbuf(d, 2); tk.p2 = d.rh_seq;
      } else {
        if (gru) { tk.N = H; tk.mode = EP_GRU_CAND; tk.bias = d.bias2; tk.p0 = d.cs; tk.p1 = d.gates; }
        else { tk.N = 4 * H; tk.mode = EP_LSTM_FWD; tk.bias = d.bias; tk.p0 = d.gates; tk.p1 = d.cs;
               tk.p3 = cbuf(d, l & 1); tk.p5 = cbuf(d, (l + 1) & 1); }
        tk.p2 = layer_out(d, 0) + H; tk.s0 = (long)(L + 1) * H; tk.s1 = H;
        tk.p6 = hbuf(d, (l + 1) & 1);
        if (drop) {
          tk.seed = d.seed; tk.k_st = d.keep_state; tk.k_out = d.keep_out; tk.k_in = 1.0f;
          tk.r_st = cid4 + 1; tk.r_out = cid4 + 2;
          tk.p9 = d.hs_seq + H; tk.s4 = (long)(L + 1) * H; tk.s5 = H;
          if (d.n_extra > 0) {   // what layer 1 consumes: this output under layer 1's input mask
            tk.p11 = d.extra[0].xin_seq + H; tk.k_in = d.keep_in; tk.r_in = (uint32_t)d.extra[0].cell_id * 4; tk.in_W = H; tk.in_coff = 0;
          }
        }
      }
      if ((rc = avsr_step_launch_raw(&SL, stream))) return rc;
    }
    // ---- K1b: the layers above (MultiRNNCell): layer j consumes layer j-1's output of this step -----------------------
    for (int j = 0; j < d.n_extra; ++j)
     for (int phase = 0; phase < (gru ? 2 : 1); ++phase) {
      const avsr_dec_layer& X = d.extra[j];
      SL.ntask = 1;
      StepTask& tk = SL.task[0];
      tk = StepTask{};
      const float* xwt = (gru && phase == 1) ? X.wt2 : X.wt;
      StepSrc& x = tk.src[tk.nsrc++];
      x.a = (drop ? X.xin_seq : layer_out(d, j)) + (long)(l + 1) * H; x.sb = (long)(L + 1) * H; x.K = H; x.w = xwt; x.ldw = 2 * H;
      x.kind = SRC_OWNROW;
      StepSrc& h = tk.src[tk.nsrc++];
      h.a = (gru && phase == 1) ? xhbuf(d, j, 2) /* r*h of this step, own row */ : xhbuf(d, j, l & 1);
      h.sb = H; h.K = H; h.w = xwt + H; h.ldw = 2 * H; h.kind = (gru && phase == 1) ? SRC_OWNROW : SRC_PLAIN;
      if (d.mode == 3) tk.gather2 = d.parent_rows;
      tk.B = B; tk.t = l; tk.T = L; tk.reverse = 0; tk.len = d.steplen;
      tk.p4 = xhbuf(d, j, l & 1);
      if (gru && phase == 0) {                       // gates of a GRU layer above the attention-fed one (same epilogues as the block's cell)
        tk.N = 2 * H; tk.mode = EP_GRU_GATES; tk.bias = X.bias;
        tk.p0 = X.gates; tk.p1 = xhbuf(d, j, 2); tk.p2 = X.rh_seq;
        if ((rc = avsr_step_launch_raw(&SL, stream))) return rc;
        continue;
      }
      if (gru) { tk.N = H; tk.mode = EP_GRU_CAND; tk.bias = X.bias2; tk.p0 = X.cs; tk.p1 = X.gates; }
      else { tk.N = 4 * H; tk.mode = EP_LSTM_FWD; tk.bias = X.bias; tk.p0 = X.gates; tk.p1 = X.cs;
             tk.p3 = xcbuf(d, j, l & 1); tk.p5 = xcbuf(d, j, (l + 1) & 1); }
      tk.p6 = xhbuf(d, j, (l + 1) & 1);
      tk.p2 = X.out + H; tk.s0 = (long)(L + 1) * H; tk.s1 = H;
      if (drop) {
        const uint32_t c4 = (uint32_t)X.cell_id * 4;
        tk.seed = d.seed; tk.k_st = d.keep_state; tk.k_out = d.keep_out; tk.k_in = 1.0f;
        tk.r_st = c4 + 1; tk.r_out = c4 + 2;
        tk.p9 = X.hs_seq + H; tk.s4 = (long)(L + 1) * H; tk.s5 = H;
        if (j + 1 < d.n_extra) {
          tk.p11 = d.extra[j + 1].xin_seq + H; tk.k_in = d.keep_in; tk.r_in = (uint32_t)d.extra[j + 1].cell_id * 4; tk.in_W = H; tk.in_coff = 0;
        }
      }
      if ((rc = avsr_step_launch_raw(&SL, stream))) return rc;
    }

    if (d.n_mech > 0) {
      // ---- Kq: Bahdanau processed query  pq = cell_out . Wq ---------------------------------
      SL.ntask = 0;
      for (int m = 0; m < d.n_mech; ++m) {
        const avsr_attn_mech& M = d.mech[m];
        if (!is_bahdanau(M)) continue;
        StepTask& tk = SL.task[SL.ntask++];
        tk = StepTask{};
        StepSrc& x = tk.src[tk.nsrc++];
        x.a = d.cell_out + (long)(l + 1) * H; x.sb = (long)(L + 1) * H; x.K = H; x.w = M.wq_t; x.ldw = H; x.kind = SRC_PLAIN;
        tk.B = B; tk.N = H; tk.mode = EP_LINEAR; tk.t = l; tk.T = L;
        tk.p0 = M.pq + (long)l * H; tk.s0 = (long)L * H;
      }
      if (SL.ntask && (rc = avsr_step_launch_raw(&SL, stream))) return rc;

      // ---- K2: attention partials ---------------------------------------------------------
      fill_attn_launch(d, l, AL);
      if ((rc = avsr_attn_launch_raw(&AL, 0, stream))) return rc;

      // ---- K3: attention layer  att_m = [cell_out, ctx_m] . W_att,m --------------------------
      bool layer_done = false;
      if (beam_dense) {
        const int brc = beam_attention_layer_launch(d, l, s);
        if (brc == AVSR_OK) layer_done = true;
        else if (brc != AVSR_ERR_UNSUPPORTED) return brc;
      }
      SL.ntask = 0;
      for (int m = 0; m < d.n_mech && !layer_done; ++m) {
        const avsr_attn_mech& M = d.mech[m];
        const int nc = nchunk(M);
        StepTask& tk = SL.task[SL.ntask++];
        tk = StepTask{};
        StepSrc& x = tk.src[tk.nsrc++];
        x.a = d.cell_out + (long)(l + 1) * H; x.sb = (long)(L + 1) * H; x.K = H; x.w = M.watt_t; x.ldw = H + M.D; x.kind = SRC_PLAIN;
        StepSrc& c = tk.src[tk.nsrc++];
        c.a = M.pctx; c.sb = M.D; c.K = M.D; c.w = M.watt_t + H; c.ldw = H + M.D; c.kind = SRC_SOFTMAX;
        tk.pm = M.pstat + (long)(2 * l) * nc * B; tk.pl = M.pstat + (long)(2 * l + 1) * nc * B;
        tk.nslab = nc; tk.slab_stride = (long)B * M.D;
        tk.ctx_save = M.ctx + (long)l * M.D; tk.ctx_sb = (long)L * M.D;
        tk.B = B; tk.N = H; tk.mode = EP_LINEAR; tk.t = l; tk.T = L; tk.len = d.steplen;
        tk.p0 = d.att + (long)(l + 1) * A + (long)m * H; tk.s0 = (long)(L + 1) * A;
        if (drop) {   // pre-dropped copy = the attention half of step l+1's cell input (mask index of time l+1)
          tk.seed = d.seed; tk.k_in = d.keep_in; tk.r_in = cid4; tk.in_W = E + A; tk.in_coff = (E + A) + E + m * H;
          tk.p9 = d.attd + (long)(l + 1) * A + (long)m * H; tk.s4 = (long)(L + 1) * A;
        }
      }
      if (!layer_done && (rc = avsr_step_launch_raw(&SL, stream))) return rc;
    }

    if (d.mode >= 1) {
      // ---- K4/K5: output layer + greedy / scheduled sample / beam step ------------------------
      const bool oa = d.output_attention && A > 0;
      const int O = oa ? A : H;
      const float* xa = oa ? d.att + (long)(l + 1) * A : d.cell_out + (long)(l + 1) * H;
      const long xsb = oa ? (long)(L + 1) * A : (long)(L + 1) * H;
      const bool fused_sample = d.mode == 2 && O % 4 == 0 && d.V <= 8192;
      if (fused_sample) {
        hipLaunchKernelGGL(logits_sample_kernel, dim3(B), dim3(256), sizeof(float) * d.V, s, xa, xsb, O, d.wout_t, d.bout,
                           d.logits + (long)l * d.V, (long)L * d.V, d.V, d.steplen, d.labels, d.fed, d.xs, d.embedding, B, L, E, l, d.seed,
                           d.sampling_prob, drop ? d.keep_in : 1.0f, cid4, E + A);
        AVSR_CHECK_LAUNCH();
        continue;
      }
      // beam search on the dense path: the output layer inside the beam step (beam_step_kernel<NCT>)
      const int bs_nct = (beam_dense && d.beam_width <= 16 && d.V <= 64 && O % 256 == 0 && xsb % 4 == 0 &&
                          (((uintptr_t)xa | (uintptr_t)d.wout_t) & 15) == 0) ? (d.V <= 32 ? 2 : 4) : 0;
      if (!bs_nct) {
        SL.ntask = 1;
        StepTask& tk = SL.task[0];
        tk = StepTask{};
        StepSrc& x = tk.src[tk.nsrc++];
        x.a = xa; x.sb = xsb;
        x.K = O; x.w = d.wout_t; x.ldw = O; x.kind = SRC_PLAIN;
        tk.B = B; tk.N = d.V; tk.mode = EP_LINEAR; tk.t = l; tk.T = L; tk.len = (d.mode == 3) ? nullptr : d.steplen; tk.bias = d.bout;
        tk.p0 = d.logits + (long)l * d.V; tk.s0 = (long)L * d.V;
        if ((rc = avsr_step_launch_raw(&SL, stream))) return rc;
      }
      if (d.mode == 1) {
        hipLaunchKernelGGL(greedy_sample_kernel, dim3(1), dim3(256), 0, s, d.logits + (long)l * d.V, (long)L * d.V, d.V,
                           d.ids + l, (long)L, d.tok, d.steplen, d.n_unfinished, B, l, d.eos_id);
      } else if (d.mode == 3) {
        const int K = d.beam_width, pin = l & 1, pout = (l + 1) & 1;
#define BS_GO(NCT_)                                                                                                                       \
  hipLaunchKernelGGL(beam_step_kernel<NCT_>, dim3(B / K), dim3(256), (K * d.V + 8 * K + 24 + 1024 * NCT_) * sizeof(float), s,               \
                     d.logits + (long)l * d.V, (long)L * d.V, d.V, K, l, d.eos_id, d.length_penalty, d.beam_logp + (long)pin * B,           \
                     d.beam_fin + (long)pin * B, d.beam_len + (long)pin * B, d.beam_logp + (long)pout * B, d.beam_fin + (long)pout * B,    \
                     d.beam_len + (long)pout * B, d.tok, d.parent_rows, d.step_ids + (long)l * B, d.parent_ids + (long)l * B,              \
                     d.n_unfinished + l, xa, xsb, O, d.wout_t, d.bout)
        if (bs_nct == 2) BS_GO(2); else if (bs_nct == 4) BS_GO(4); else BS_GO(0);
#undef BS_GO
      } else {
        hipLaunchKernelGGL(sched_sample_kernel, dim3(B), dim3(128), sizeof(float) * d.V, s, d.logits + (long)l * d.V, (long)L * d.V, d.V, d.labels,
                           d.fed, d.xs, d.embedding, B, L, E, l, d.seed, d.sampling_prob, drop ? d.keep_in : 1.0f, cid4, E + A);
      }
      AVSR_CHECK_LAUNCH();
    }
  }
  if (l_end == L) {
    avsr::DevBatch db(s);
    if (d.h_final) db.copy(d.h_final, hbuf(d, L & 1), bh);
    if (!gru && d.c_final) db.copy(d.c_final, cbuf(d, L & 1), bh);
    if (db.flush() != hipSuccess) return AVSR_ERR_HIP;
  }
  return AVSR_OK;
}

extern "C" int avsr_attn_rnn_bwd(const avsr_attn_rnn* dp, void* stream) {
  using namespace avsr;
  int rc = validate(dp);
  if (rc) return rc;
  const avsr_attn_rnn& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  const int B = d.B, H = d.H, L = d.L, E = d.E, A = d.n_mech * H;
  if (!d.w || !d.dgates || !d.dstate) return AVSR_ERR_ARG;
  if (A > 0 && !d.datt) return AVSR_ERR_ARG;
  int n_bah = 0, n_luong = 0;
  for (int m = 0; m < d.n_mech; ++m) {
    const avsr_attn_mech& M = d.mech[m];
    if (!M.watt || !M.dscores || !M.dctx || !M.pdq) return AVSR_ERR_ARG;
    if (is_bahdanau(M)) { if (!M.wq || !M.dpq) return AVSR_ERR_ARG; ++n_bah; } else ++n_luong;
  }
  if (1 + d.n_mech + n_bah > STEP_MAX_SRC) return AVSR_ERR_UNSUPPORTED;
  const bool drop = d.seed && (d.keep_in < 1.f || d.keep_state < 1.f || d.keep_out < 1.f);
  const uint32_t cid4 = (uint32_t)d.cell_id * 4;
  const bool gru = d.cell == 1;
  if (gru && (!d.w2 || !d.dgates2)) return AVSR_ERR_ARG;
  const int G = gru ? 2 : 4;      // gate pre-activations per unit
  // GRU dstate: dGg rolling [2][B][2H] | d(cand pre-act) rolling [2][B][H] | carry [2][B][H] | tmp du | tmp dh*u
  auto g_dgg = [&](int p) { return d.dstate + (long)p * B * 2 * H; };
  auto g_dpc = [&](int p) { return d.dstate + (long)(4 + p) * B * H; };
  auto g_carry = [&](int p) { return d.dstate + (long)(6 + p) * B * H; };
  auto g_tmp = [&](int w) { return d.dstate + (long)(8 + w) * B * H; };
  const bool use_dq = (n_luong > 0) || d.dcell_ext;
  if (use_dq && !d.dq) return AVSR_ERR_ARG;
  const size_t bh = sizeof(float) * B * H;
  const int NX = d.n_extra;
  if (NX > 0 && (d.dh_final || d.dc_final)) return AVSR_ERR_UNSUPPORTED;     // final-state gradients: single-cell blocks only
  {
    avsr::DevBatch db(s);                            // the zeroed rolling state of every layer: one launch
    for (int j = 0; j < NX; ++j) {
      if (!d.extra[j].w || !d.extra[j].dgates || !d.extra[j].dstate) return AVSR_ERR_ARG;
      db.zero(d.extra[j].dstate, 12 * bh);
    }
    db.zero(d.dstate, 12 * bh);
    if (db.flush() != hipSuccess) return AVSR_ERR_HIP;
    // ... then the final-state gradients into their slots of it (a second launch: they overwrite part of what the first zeroes)
    if (!gru && d.dc_final) db.copy(dcbuf(d, L & 1), d.dc_final, bh);
    if (d.dh_final) db.copy(gru ? g_carry(L & 1) : dhcarry(d, L & 1), d.dh_final, bh);
    if (db.flush() != hipSuccess) return AVSR_ERR_HIP;
  }

  static thread_local StepLaunch SL;
  static thread_local AttnLaunch AL;
  static thread_local SlabLaunch BL;
  // the whole loop as one persistent launch where the fused kernel covers the block (csrc/dec_persist_bwd.hip)
  bool fused = false;
  if (A > 0 && !gru && NX == 0) {                 // (Luong blocks, and one-memory Bahdanau blocks)
    const int frc = avsr_dec_persist_bwd(dp, stream);
    if (frc == AVSR_OK) fused = true;
    else if (frc != AVSR_ERR_UNSUPPORTED) return frc;
  }
  for (int l = fused ? -1 : L - 1; l >= 0; --l) {
    if (A > 0) {
      // ---- KB3: d attention_l = datt_ext[l] + dG_{l+1} . Wx_att^T ----------------------------
      SL.ntask = 1;
      {
        StepTask& tk = SL.task[0];
        tk = StepTask{};
        StepSrc& x = tk.src[tk.nsrc++];
        x.a = gru ? g_dgg((l + 1) & 1) : dgroll(d, (l + 1) & 1); x.sb = G * H; x.K = G * H; x.w = d.w + (long)E * G * H; x.ldw = G * H; x.kind = SRC_PLAIN;
        if (gru) {
          StepSrc& x2 = tk.src[tk.nsrc++];
          x2.a = g_dpc((l + 1) & 1); x2.sb = H; x2.K = H; x2.w = d.w2 + (long)E * H; x2.ldw = H; x2.kind = SRC_PLAIN;
        }
        tk.B = B; tk.N = A; tk.mode = EP_LINEAR; tk.t = l; tk.T = L;
        if (d.datt_ext) { tk.p1 = const_cast<float*>(d.datt_ext) + (long)l * A; tk.s1 = (long)L * A; }
        tk.p0 = d.datt + (long)l * A; tk.s0 = (long)L * A;
        if (drop) {   // the product is the gradient of step l+1's DROPPED attention input
          tk.act = 8; tk.seed = d.seed; tk.k_in = d.keep_in; tk.r_in = cid4; tk.in_W = E + A; tk.in_coff = (E + A) + E;
        }
      }
      if ((rc = avsr_step_launch_raw(&SL, stream))) return rc;
      // ---- KB4: d ctx_m = d att_m . W_att,m[H:, :]^T ----------------------------------------
      SL.ntask = 0;
      for (int m = 0; m < d.n_mech; ++m) {
        const avsr_attn_mech& M = d.mech[m];
        StepTask& tk = SL.task[SL.ntask++];
        tk = StepTask{};
        StepSrc& x = tk.src[tk.nsrc++];
        x.a = d.datt + (long)l * A + (long)m * H; x.sb = (long)L * A; x.K = H; x.w = M.watt + (long)H * H; x.ldw = H; x.kind = SRC_PLAIN;
        tk.B = B; tk.N = M.D; tk.mode = EP_LINEAR; tk.t = l; tk.T = L;
        tk.p0 = M.dctx + (long)l * M.D; tk.s0 = (long)L * M.D;
      }
      if ((rc = avsr_step_launch_raw(&SL, stream))) return rc;
      // ---- KB5: attention backward -----------------------------------------------------------
      fill_attn_launch(d, l, AL);
      if ((rc = avsr_attn_launch_raw(&AL, 1, stream))) return rc;
    }
    // ---- KB5b: reduce the per-chunk query gradients --------------------------------------------
    BL = SlabLaunch{};
    BL.B = B;
    // LSTM with only Luong mechanisms (<= 2): the cell-backward epilogue sums the partials itself (no launch here)
    const bool fold_dq = use_dq && !gru && n_bah == 0 && n_luong <= 2;
    if (use_dq && !fold_dq) {
      SlabJob& J = BL.job[BL.njob++];
      J.dst = d.dq + (long)l * H; J.dst_sb = (long)L * H; J.W = H;
      if (d.dcell_ext) { J.add = d.dcell_ext + (long)l * H; J.add_sb = (long)L * H; }
      for (int m = 0; m < d.n_mech; ++m)
        if (!is_bahdanau(d.mech[m])) { J.src[J.nsrc] = d.mech[m].pdq; J.nslab[J.nsrc] = nchunk(d.mech[m]); ++J.nsrc; }
    }
    for (int m = 0; m < d.n_mech; ++m) {
      const avsr_attn_mech& M = d.mech[m];
      if (!is_bahdanau(M)) continue;
      SlabJob& J = BL.job[BL.njob++];
      J.dst = M.dpq + (long)l * H; J.dst_sb = (long)L * H; J.W = H;
      J.src[0] = M.pdq; J.nslab[0] = nchunk(M); J.nsrc = 1;
    }
    if (BL.njob) {
      int gx = (B * H + 255) / 256;
      if (gx > 64) gx = 64;
      hipLaunchKernelGGL(slab_sum_kernel, dim3(gx, BL.njob), dim3(256), 0, s, BL);
      AVSR_CHECK_LAUNCH();
    }
    // ---- KB2: LSTM backward, top layer first.  The TOP layer's emitted output is what the attention mechanisms and the
    // attention layers consumed; every layer below receives d(output) from the layer above of the SAME step (through that
    // layer's input mask) next to its own recurrent path.
    for (int j = NX; j >= 0; --j) {
      SL.ntask = 1;
      StepTask& tk = SL.task[0];
      tk = StepTask{};
      const bool top = (j == NX);
      const uint32_t c4 = j == 0 ? cid4 : (uint32_t)d.extra[j - 1].cell_id * 4;
      StepSrc& a = tk.src[tk.nsrc++];
      if (j == 0) {
        a.a = gru ? g_dgg((l + 1) & 1) : dgroll(d, (l + 1) & 1); a.sb = G * H; a.K = G * H; a.w = d.w + (long)(E + A) * G * H; a.ldw = G * H;
      } else if (gru) {
        a.a = xg_dgg(d, j - 1, (l + 1) & 1); a.sb = 2 * H; a.K = 2 * H; a.w = d.extra[j - 1].w + (long)H * 2 * H; a.ldw = 2 * H;
      } else {
        a.a = xdgroll(d, j - 1, (l + 1) & 1); a.sb = 4 * H; a.K = 4 * H; a.w = d.extra[j - 1].w + (long)H * 4 * H; a.ldw = 4 * H;
      }
      a.kind = SRC_PLAIN;
      if (top) {
        for (int m = 0; m < d.n_mech; ++m) {
          const avsr_attn_mech& M = d.mech[m];
          StepSrc& x = tk.src[tk.nsrc++];
          x.a = d.datt + (long)l * A + (long)m * H; x.sb = (long)L * A; x.K = H; x.w = M.watt; x.ldw = H; x.kind = SRC_PLAIN;
          if (is_bahdanau(M)) {
            StepSrc& q = tk.src[tk.nsrc++];
            q.a = M.dpq + (long)l * H; q.sb = (long)L * H; q.K = H; q.w = M.wq; q.ldw = H; q.kind = SRC_PLAIN;
          }
        }
      } else if (gru) {   // GRU layer above: through its gate kernel and its candidate kernel (both phases of it ran a moment ago)
        StepSrc& x = tk.src[tk.nsrc++];
        x.a = xg_dgg(d, j, l & 1); x.sb = 2 * H; x.K = 2 * H; x.w = d.extra[j].w; x.ldw = 2 * H; x.kind = SRC_PLAIN;
        StepSrc& x2 = tk.src[tk.nsrc++];
        x2.a = xg_dpc(d, j, l & 1); x2.sb = H; x2.K = H; x2.w = d.extra[j].w2; x2.ldw = H; x2.kind = SRC_PLAIN;
      } else {   // d(output of layer j) = dG_{j+1}(l) . Wx_{j+1}^T  (dG of the layer above, computed a moment ago)
        StepSrc& x = tk.src[tk.nsrc++];
        x.a = xdgroll(d, j, l & 1); x.sb = 4 * H; x.K = 4 * H; x.w = d.extra[j].w; x.ldw = 4 * H; x.kind = SRC_PLAIN;
      }
      tk.B = B; tk.N = H; tk.t = l; tk.T = L; tk.reverse = 0; tk.len = d.steplen;
      if (gru && j == 0) {
        tk.mode = EP_GRU_BWD_CAND;
        tk.p0 = d.gates; tk.p1 = d.cs;
        tk.p2 = drop ? d.hs_seq : layer_out(d, 0); tk.s2 = (long)(L + 1) * H; tk.s3 = H;     // slot l = h consumed by step l
        tk.p4 = g_carry((l + 1) & 1); tk.p5 = g_dpc(l & 1); tk.p3 = d.dgates2; tk.p6 = g_tmp(0); tk.p7 = g_tmp(1);
      } else if (gru) {
        const avsr_dec_layer& X = d.extra[j - 1];
        tk.mode = EP_GRU_BWD_CAND;
        tk.p0 = X.gates; tk.p1 = X.cs;
        tk.p2 = drop ? X.hs_seq : X.out; tk.s2 = (long)(L + 1) * H; tk.s3 = H;
        tk.p4 = xg_carry(d, j - 1, (l + 1) & 1); tk.p5 = xg_dpc(d, j - 1, l & 1); tk.p3 = X.dgates2;
        tk.p6 = xg_tmp(d, j - 1, 0); tk.p7 = xg_tmp(d, j - 1, 1);
      } else if (j == 0) {
        tk.mode = EP_LSTM_BWD;
        tk.bias = d.c0;
        tk.p0 = d.gates; tk.p1 = d.cs; tk.p2 = d.dgates; tk.p3 = dgroll(d, l & 1);
        tk.p4 = dcbuf(d, (l + 1) & 1); tk.p5 = dcbuf(d, l & 1);
        tk.p6 = dhcarry(d, (l + 1) & 1); tk.p7 = dhcarry(d, l & 1);
      } else {
        const avsr_dec_layer& X = d.extra[j - 1];
        tk.mode = EP_LSTM_BWD;                       // c_init = 0: bias stays NULL
        tk.p0 = X.gates; tk.p1 = X.cs; tk.p2 = X.dgates; tk.p3 = xdgroll(d, j - 1, l & 1);
        tk.p4 = xdcbuf(d, j - 1, (l + 1) & 1); tk.p5 = xdcbuf(d, j - 1, l & 1);
        tk.p6 = xdhcarry(d, j - 1, (l + 1) & 1); tk.p7 = xdhcarry(d, j - 1, l & 1);
      }
      if (top) {
        if (fold_dq) {
          if (d.dcell_ext) { tk.p8 = const_cast<float*>(d.dcell_ext); tk.s0 = (long)L * H; tk.s1 = H; }
          int k = 0;
          for (int m = 0; m < d.n_mech; ++m) {
            if (is_bahdanau(d.mech[m])) continue;
            if (k == 0) { tk.pm = d.mech[m].pdq; tk.nslab = nchunk(d.mech[m]); }
            else { tk.pl = d.mech[m].pdq; tk.pad0 = nchunk(d.mech[m]); }
            ++k;
          }
        } else if (use_dq) { tk.p8 = d.dq; tk.s0 = (long)L * H; tk.s1 = H; }
      }
      if (drop) {
        tk.seed = d.seed; tk.k_st = d.keep_state; tk.k_out = d.keep_out; tk.k_in = 1.0f;
        tk.r_st = c4 + 1; tk.r_out = c4 + 2;
        if (!top) { tk.k_in = d.keep_in; tk.r_in = (uint32_t)d.extra[j].cell_id * 4; tk.in_W = H; tk.in_coff = 0; }
      }
      if ((rc = avsr_step_launch_raw(&SL, stream))) return rc;
      if (gru) {   // second phase of THIS layer: through r*h and the gate pre-activations (the layer below reads both results)
        SL.ntask = 1;
        StepTask& tg = SL.task[0];
        tg = StepTask{};
        StepSrc& ag = tg.src[tg.nsrc++];
        tg.B = B; tg.N = H; tg.mode = EP_GRU_BWD_GATES; tg.t = l; tg.T = L; tg.reverse = 0; tg.len = d.steplen;
        tg.s2 = (long)(L + 1) * H; tg.s3 = H;
        if (j == 0) {
          ag.a = g_dpc(l & 1); ag.sb = H; ag.K = H; ag.w = d.w2 + (long)(E + A) * H; ag.ldw = H; ag.kind = SRC_PLAIN;
          tg.p0 = d.gates; tg.p2 = drop ? d.hs_seq : layer_out(d, 0);
          tg.p6 = g_tmp(0); tg.p7 = g_tmp(1); tg.p5 = g_carry(l & 1); tg.p3 = d.dgates; tg.p1 = g_dgg(l & 1);
        } else {
          const avsr_dec_layer& X = d.extra[j - 1];
          if (!X.w2 || !X.dgates2) return AVSR_ERR_ARG;
          ag.a = xg_dpc(d, j - 1, l & 1); ag.sb = H; ag.K = H; ag.w = X.w2 + (long)H * H; ag.ldw = H; ag.kind = SRC_PLAIN;
          tg.p0 = X.gates; tg.p2 = drop ? X.hs_seq : X.out;
          tg.p6 = xg_tmp(d, j - 1, 0); tg.p7 = xg_tmp(d, j - 1, 1); tg.p5 = xg_carry(d, j - 1, l & 1); tg.p3 = X.dgates; tg.p1 = xg_dgg(d, j - 1, l & 1);
        }
        if ((rc = avsr_step_launch_raw(&SL, stream))) return rc;
      }
    }
  }
  // gradient wrt the initial state
  if (d.dh0) {
    SL.ntask = 1;
    StepTask& tk = SL.task[0];
    tk = StepTask{};
    StepSrc& a = tk.src[tk.nsrc++];
    a.a = gru ? g_dgg(0) : dgroll(d, 0); a.sb = G * H; a.K = G * H; a.w = d.w + (long)(E + A) * G * H; a.ldw = G * H; a.kind = SRC_PLAIN;
    tk.B = B; tk.N = H; tk.mode = EP_LINEAR; tk.t = 0; tk.T = L;
    tk.p1 = gru ? g_carry(0) : dhcarry(d, 0); tk.s1 = H;
    tk.p0 = d.dh0; tk.s0 = H;
    if ((rc = avsr_step_launch_raw(&SL, stream))) return rc;
  }
  if (!gru && d.dc0 && avsr::dev_copy(d.dc0, dcbuf(d, 0), bh, s) != hipSuccess) return AVSR_ERR_HIP;
  return AVSR_OK;
}

extern "C" int avsr_beam_gather_tree(const int32_t* step_ids, const int32_t* parent_ids, const int32_t* beam_len, int32_t* out,
                                     int32_t n_utt, int32_t beam_width, int32_t T, int32_t eos_id, void* stream) {
  using namespace avsr;
  if (!step_ids || !parent_ids || !beam_len || !out || n_utt <= 0 || beam_width <= 0 || T <= 0) return AVSR_ERR_ARG;
  const int n = n_utt * beam_width;
  hipLaunchKernelGGL(beam_gather_tree_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, step_ids, parent_ids, beam_len, out,
                     n_utt, beam_width, T, eos_id);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_beam_search_step(float* logits, int32_t n_utt, int32_t beam_width, int32_t V, int32_t step, int32_t eos_id,
                                     float length_penalty_weight, const float* logp_in, const int32_t* fin_in, const int32_t* len_in,
                                     float* logp_out, int32_t* fin_out, int32_t* len_out, int32_t* tok, int32_t* parent_rows,
                                     int32_t* step_ids, int32_t* parent_ids, int32_t* n_unfinished, const float* x, int64_t x_stride,
                                     int32_t O, const float* wout_t, const float* bout, void* stream) {
  using namespace avsr;
  if (!logits || !logp_in || !fin_in || !len_in || !logp_out || !fin_out || !len_out || !tok || !parent_rows || !step_ids || !parent_ids ||
      !n_unfinished || n_utt <= 0 || beam_width <= 0 || V <= 0 || step < 0 || eos_id < 0 || eos_id >= V)
    return AVSR_ERR_ARG;
  if ((long)beam_width * V > 1024) return AVSR_ERR_UNSUPPORTED;      // the kernel keeps a thread's candidates in registers (4 x 256)
  const int K = beam_width;
  const long B = (long)n_utt * K;
  hipStream_t s = (hipStream_t)stream;
  // the same launches avsr_attn_rnn_fwd (mode 3) issues: logits given (NCT = 0), or the output layer inside the step (x != NULL; its conditions below)
  int nct = 0;
  if (x) {
    if (!wout_t || !bout || O <= 0) return AVSR_ERR_ARG;
    if (K > 16 || V > 64 || O % 256 != 0 || x_stride % 4 != 0 || (((uintptr_t)x | (uintptr_t)wout_t) & 15) != 0) return AVSR_ERR_UNSUPPORTED;
    nct = V <= 32 ? 2 : 4;
  }
#define BS_GO(NCT_)                                                                                                                       \
  hipLaunchKernelGGL(beam_step_kernel<NCT_>, dim3(n_utt), dim3(256), (K * V + 8 * K + 24 + 1024 * NCT_) * sizeof(float), s, logits, (long)V, V, K,  \
                     step, eos_id, length_penalty_weight, logp_in, fin_in, len_in, logp_out, fin_out, len_out, tok, parent_rows,                     \
                     step_ids + (long)step * B, parent_ids + (long)step * B, n_unfinished + step, x, (long)x_stride, O, wout_t, bout)
  if (nct == 2) BS_GO(2); else if (nct == 4) BS_GO(4); else BS_GO(0);
#undef BS_GO
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}
