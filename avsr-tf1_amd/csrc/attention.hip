// Attention score / masked softmax / context kernels (forward and per-step backward).
//
// Restates contrib.seq2seq.{Luong,Bahdanau}Attention.__call__ + _compute_attention as selected by
// avsr/attention.py:25-72 and avsr/decoder_bimodal.py:383-445:
//   Luong     score[b,t] = g * sum_h keys[b,t,h] * q[b,h]              (g only for scaled_luong)
//   Bahdanau  score[b,t] = sum_h v[h] * tanh(keys[b,t,h] + pq[b,h] (+ b[h]))
//   alpha = softmax_t(score masked to -inf past len[b]);  ctx[b,:] = sum_t alpha[b,t] * values[b,t,:]
//
// This is the HBM-bound part of the decoder step: every step streams keys [B,T,H] and values
// [B,T,D] once.  One launch covers every mechanism of the AttentionWrapper (video + audio memory).
// Work item = (mechanism, utterance b, chunk of <= 128 time steps): flash-decoding style partials
// (chunk max m, chunk sum l, un-normalised partial context) so that >= 2 workgroups per CU stream
// concurrently; the consumer (the attention-layer step kernel, step.hip SRC_SOFTMAX) merges the
// partials while it loads its A operand, so no extra launch sits in the sequential chain.
//
// Access pattern: 16 lanes cooperate on one memory row (each lane 16-byte loads at 64-byte stride
// groups -> 256 contiguous bytes per row per instruction, 16 rows in flight per workgroup pass);
// the context phase assigns one float4 column per thread and walks rows, fully coalesced.
// Rows past len[b] are never loaded.
#include "attn.h"
#include "prof.h"
#include "persist.h"

namespace avsr {

__device__ __forceinline__ void locate(const AttnLaunch& L, int& m, int& b, int& c) {
  int blk = blockIdx.x;
  m = 0;
#pragma unroll
  for (int i = 1; i < AVSR_MAX_MECH; ++i)
    if (i < L.nmech && blk >= L.blk_off[i]) m = i;
  blk -= L.blk_off[m];
  c = blk % L.m[m].nchunk;
  b = blk / L.m[m].nchunk;
}


// ---- streaming helpers (Luong paths).  The memory rows are the HBM stream of the decoder step, and only ~1.25
// workgroups per CU run, so each thread must keep many 16-byte loads in flight: rows are fetched in unconditional
// batches through raw buffer loads (out-of-range offset = reads 0 for rows past the chunk), never one row per loop trip.

// 16 lanes per row: dot(row[r], vec) for the rows r = rg + 16*u of a chunk (u < 8); vec fragments v4[j] at k = 4*s16 + 64*j.
template <class Emit>
__device__ __forceinline__ void rows_dot16(__amdgpu_buffer_rsrc_t rs, int base_b, int row_stride_b, int W, int n,
                                           const f32x4 (&v4)[4], Emit&& emit) {
  const int s16 = threadIdx.x & 15, rg = threadIdx.x >> 4;
#pragma unroll
  for (int u0 = 0; u0 < 8; u0 += 4) {
    if (rg + 16 * u0 >= n) break;                       // uniform per 16-lane group; later rows are out of the chunk too
    f32x4 x[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = rg + 16 * (u0 + u);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * s16 + 64 * j;
        x[u][j] = ldb4(rs, (r < n && k < W) ? base_b + r * row_stride_b + k * 4 : P_OOB);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += x[u][j][0] * v4[j][0] + x[u][j][1] * v4[j][1] + x[u][j][2] * v4[j][2] + x[u][j][3] * v4[j][3];
      acc = group16_sum(acc);
      const int r = rg + 16 * (u0 + u);
      if (r < n) emit(r, acc);
    }
  }
}

// one float4 column per thread: sum_r w[r] * row[r][col] over the rows r = grp + G*u of the chunk, 8 rows in flight
__device__ __forceinline__ f32x4 rows_wsum(__amdgpu_buffer_rsrc_t rs, int base_b, int row_stride_b, int col4, int grp, int G,
                                           int n, const float* w, float scale) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r0 = grp; r0 < n; r0 += 8 * G) {
    f32x4 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = r0 + u * G;
      x[u] = ldb4(rs, r < n ? base_b + r * row_stride_b + col4 * 16 : P_OOB);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = r0 + u * G;
      acc += ((r < n) ? w[r] * scale : 0.f) * x[u];
    }
  }
  return acc;
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnLaunch L) {
  __shared__ float sc[ATTN_MAX_CHUNK];
  __shared__ float red[4];
  __shared__ __attribute__((aligned(16))) float cpart[1024];
  int mi, b, c;
  locate(L, mi, b, c);
  const AttnMechDev& M = L.m[mi];
  const int tid = threadIdx.x;
  const int mb = M.mem_div > 1 ? b / M.mem_div : b;          // memory row of this hypothesis (beam search shares the utterance's memory)
  const int len = min(M.len ? M.len[mb] : M.T, M.T);
  const int t0 = c * M.chunk;
  const int n = max(0, min(M.chunk, len - t0));  // valid rows in this chunk
  const int H = M.H, D = M.D;
  const float* keys = M.keys + (long)mb * M.T * H;
  const float* q = M.query + (long)b * M.query_sb;

  // ---- phase 1: scores for rows t0 .. t0+n-1 (16 lanes per row) ----
  const int s16 = tid & 15, rg = tid >> 4;
  const bool fast = M.type <= ATT_SCALED_LUONG && H <= 256 && (long)M.T * H * 4 < (1L << 31) && (long)M.T * M.values_st * 4 < (1L << 31);
  if (fast) {
    f32x4 q4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 4 * s16 + 64 * j;
      q4[j] = k < H ? ld4(q + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float gsc = (M.type == ATT_SCALED_LUONG) ? M.g[0] : 1.f;
    float* srow = M.scores + (long)b * M.scores_sb + t0;
    rows_dot16(make_rsrc(keys), t0 * H * 4, H * 4, H, n, q4, [&](int r, float acc) {
      if (s16 == 0) { srow[r] = acc; sc[r] = acc * gsc; }
    });
  } else
  for (int r = rg; r < n; r += 16) {
    const float* krow = keys + (long)(t0 + r) * H;
    float acc = 0.f;
    if (M.type <= ATT_SCALED_LUONG) {
      for (int k = 4 * s16; k < H; k += 64) {
        const f32x4 kv = ld4(krow + k), qv = ld4(q + k);
        acc += kv[0] * qv[0] + kv[1] * qv[1] + kv[2] * qv[2] + kv[3] * qv[3];
      }
    } else {
      for (int k = 4 * s16; k < H; k += 64) {
        f32x4 kv = ld4(krow + k) + ld4(q + k);
        if (M.bq) kv += ld4(M.bq + k);
        const f32x4 vv = ld4(M.v + k);
        acc += vv[0] * tanhf(kv[0]) + vv[1] * tanhf(kv[1]) + vv[2] * tanhf(kv[2]) + vv[3] * tanhf(kv[3]);
      }
    }
    acc = group16_sum(acc);
    if (s16 == 0) {
      M.scores[(long)b * M.scores_sb + t0 + r] = acc;
      sc[r] = (M.type == ATT_SCALED_LUONG) ? acc * M.g[0] : acc;
    }
  }
  __syncthreads();

  // ---- phase 2: chunk max / exp / sum ----
  float mloc = -INFINITY;
  for (int r = tid; r < n; r += 256) mloc = fmaxf(mloc, sc[r]);
  const float mx = block_max_256(mloc, red);
  float lloc = 0.f;
  for (int r = tid; r < n; r += 256) {
    const float p = expf(sc[r] - mx);
    sc[r] = p;
    lloc += p;
  }
  const float lsum = block_sum_256(lloc, red);  // (contains the barrier that publishes sc[])
  if (tid == 0) {
    M.pm[(long)c * L.B + b] = (n > 0) ? mx : -INFINITY;
    M.pl[(long)c * L.B + b] = lsum;
  }

  // ---- phase 3: partial context = sum_r p_r * values[t0+r, :] ----
  const int cols = D >> 2;                 // float4 columns
  const float* vals = M.values + (long)mb * M.values_sb;
  float* pout = M.pctx + ((long)c * L.B + b) * D;
  for (int cb = 0; cb < cols; cb += 256) {
    const int ccols = min(256, cols - cb);
    const int G = 256 / ccols;             // row groups
    const int col = tid % ccols, grp = tid / ccols;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (grp < G) {
      if (fast) {
        acc = rows_wsum(make_rsrc(vals), t0 * (int)M.values_st * 4, (int)M.values_st * 4, cb + col, grp, G, n, sc, 1.0f);
      } else {
        const float* vp = vals + 4 * (cb + col);
        for (int r = grp; r < n; r += G) acc += sc[r] * ld4(vp + (long)(t0 + r) * M.values_st);
      }
    }
    if (G == 1) {
      if (grp < G) st4(pout + 4 * (cb + col), acc);
    } else {
      __syncthreads();
      if (grp < G) st4(&cpart[4 * (grp * ccols + col)], acc);
      __syncthreads();
      if (tid < ccols) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int g2 = 0; g2 < G; ++g2) s += ld4(&cpart[4 * (g2 * ccols + tid)]);
        st4(pout + 4 * (cb + tid), s);
      }
    }
  }
}

// Beam search (avsr_attn_rnn.mem_shared): the K hypotheses of an utterance attend the SAME memory.  One workgroup per (mechanism,
// utterance, chunk of <= 64 frames) keeps the chunk's keys -- then its values -- in registers (64 per lane) and runs the K queries
// over them: the memory is read once per step instead of K times (attn_fwd_kernel with K = 10: 770 MB of L2 traffic per decode
// step, the whole 40 us).  Same arithmetic and summation order per hypothesis as attn_fwd_kernel (dot products by 16-lane groups,
// softmax partials by one wave, context rows in row order then row groups in order), same outputs (raw scores, chunk max / sum,
// un-normalised partial contexts by hypothesis row).  Luong / scaled Luong, H <= 256, chunk <= 64, D <= 1024, K <= 16.
#define ATTN_BEAM_KMAX 16
template <int PF>      // PF: value rows of the first block requested ahead of the key phase (0, 8 or 16 of its 16)
__global__ __launch_bounds__(256) void attn_fwd_beam_kernel(const AttnLaunch L) {
  __shared__ float sc[ATTN_BEAM_KMAX][64];
  __shared__ __attribute__((aligned(16))) float qs[ATTN_BEAM_KMAX][256];
  __shared__ __attribute__((aligned(16))) float cpart[8 * 1024];          // [8 queries][row groups x columns] float4
  int blk = blockIdx.x, mi = 0;
  for (int i = 0; i < L.nmech; ++i) {
    const int nb = (L.B / L.m[i].mem_div) * L.m[i].nchunk;
    if (blk < nb) { mi = i; break; }
    blk -= nb;
  }
  const AttnMechDev& M = L.m[mi];
  const int K = M.mem_div, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blk % M.nchunk, u = blk / M.nchunk;
  const int len = min(M.len ? M.len[u] : M.T, M.T);
  const int t0 = c * M.chunk;
  const int n = max(0, min(M.chunk, len - t0));
  const int H = M.H, D = M.D;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- values of the first column block / row pass, requested FIRST: they travel under the whole key / score / softmax phase (the
  //      two operand fetches of a step were two dependent bursts: 29 us for 75 MB at c4).  Hidden loads: the compiler would sink them
  //      back to their first use. ----
  const int cols = D >> 2;
  const int vst = (int)M.values_st * 4;
  f32x4 vr0[PF ? PF : 1];
  if constexpr (PF > 0) {
    const i32x4_ vrw = make_rsrc_words(M.values + (long)u * M.values_sb);
    const int ccols0 = min(256, cols), G0 = 256 / ccols0, col0 = tid % ccols0, grp0 = tid / ccols0;
#pragma unroll
    for (int uu = 0; uu < PF; ++uu) {
      const int r = grp0 + G0 * uu;
      ldb4_hidden(vr0[uu], vrw, (grp0 < G0 && r < n) ? (t0 + r) * vst + col0 * 16 : P_OOB);
    }
  }
  // ---- keys of the chunk: 16 lanes per row, rows rg + 16*u4 ----
  const int s16 = tid & 15, rg = tid >> 4;
  {
    const __amdgpu_buffer_rsrc_t krs = make_rsrc(M.keys + (long)u * M.T * H);
    f32x4 kr[4][4];
#pragma unroll
    for (int u4 = 0; u4 < 4; ++u4) {
      const int r = rg + 16 * u4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * s16 + 64 * j;
        kr[u4][j] = ldb4(krs, (r < n && k < H) ? ((t0 + r) * H + k) * 4 : P_OOB);
      }
    }
    const float gsc = (M.type == ATT_SCALED_LUONG) ? M.g[0] : 1.f;
    // the K queries go to the LDS in one round of loads (fetched one by one inside the loop each cost an L2 round trip)
    for (int e = tid * 4; e < K * 256; e += 1024) {
      const int kq = e >> 8, k = e & 255;
      st4(&qs[kq][k], k < H ? ld4(M.query + (long)(u * K + kq) * M.query_sb + k) : zero4);
    }
    __syncthreads();
    for (int kq = 0; kq < K; ++kq) {
      const int b = u * K + kq;
      f32x4 q4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) q4[j] = ld4(&qs[kq][4 * s16 + 64 * j]);
      float* srow = M.scores + (long)b * M.scores_sb + t0;
#pragma unroll
      for (int u4 = 0; u4 < 4; ++u4) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc += kr[u4][j][0] * q4[j][0] + kr[u4][j][1] * q4[j][1] + kr[u4][j][2] * q4[j][2] + kr[u4][j][3] * q4[j][3];
        acc = group16_sum(acc);
        const int r = rg + 16 * u4;
        if (s16 == 0 && r < n) { srow[r] = acc; sc[kq][r] = acc * gsc; }
      }
    }
  }
  __syncthreads();
  // ---- chunk max / exp / sum per hypothesis: one wave per query, one lane per row ----
  for (int kq = wave; kq < K; kq += 4) {
    const int b = u * K + kq;
    const float v = lane < n ? sc[kq][lane] : -INFINITY;
    const float mx = wave_max(v);
    const float pr = lane < n ? expf(v - mx) : 0.f;
    const float lsum = wave_sum(pr);
    sc[kq][lane] = pr;
    if (lane == 0) {
      M.pm[(long)c * L.B + b] = (n > 0) ? mx : -INFINITY;
      M.pl[(long)c * L.B + b] = lsum;
    }
  }
  __syncthreads();
  // ---- partial contexts: one float4 column per thread, G row groups; the chunk's values stay in registers for the K queries ----
  const __amdgpu_buffer_rsrc_t vrs = make_rsrc(M.values + (long)u * M.values_sb);
  if constexpr (PF > 0) {
    vm_wait_all();
#pragma unroll
    for (int uu = 0; uu < PF; ++uu) vm_landed(vr0[uu]);
  }
  for (int cb = 0; cb < cols; cb += 256) {
    const int ccols = min(256, cols - cb);
    const int G = 256 / ccols;                           // 64 columns (D = 256): four row groups of 16 rows
    const int col = tid % ccols, grp = tid / ccols;
    const bool act = grp < G;
    for (int r0 = 0; r0 < n; r0 += 16 * G) {            // (G >= 4 at D <= 256: one trip)
      f32x4 vr[16];
      const bool first = PF > 0 && cb == 0 && r0 == 0;   // (uniform) the block whose first PF rows were requested at the top of the kernel
#pragma unroll
      for (int uu = 0; uu < 16; ++uu) {
        const int r = r0 + grp + G * uu;
        if (uu < PF) vr[uu] = first ? vr0[uu < PF ? uu : 0] : ldb4(vrs, (act && r < n) ? (t0 + r) * vst + (cb + col) * 16 : P_OOB);
        else vr[uu] = ldb4(vrs, (act && r < n) ? (t0 + r) * vst + (cb + col) * 16 : P_OOB);
      }
      for (int k0 = 0; k0 < K; k0 += 8) {
        const int kn = min(8, K - k0);
        if (G > 1) __syncthreads();
        for (int kk = 0; kk < kn; ++kk) {
          const float* w = sc[k0 + kk];
          f32x4 acc = zero4;
#pragma unroll
          for (int uu = 0; uu < 16; ++uu) {
            const int r = r0 + grp + G * uu;
            acc += ((r < n) ? w[r] * 1.0f : 0.f) * vr[uu];
          }
          float* pout = M.pctx + ((long)c * L.B + (u * K + k0 + kk)) * D;
          if (G == 1) {                                   // (ccols > 128: threads beyond the columns hold zeros and must not store)
            if (act) {
              if (r0 == 0) st4(pout + 4 * (cb + col), acc);
              else st4(pout + 4 * (cb + col), ld4(pout + 4 * (cb + col)) + acc);
            }
          } else if (act) st4(&cpart[4 * ((kk * G + grp) * ccols + col)], acc);
        }
        if (G > 1) {
          __syncthreads();
          for (int it = tid; it < kn * ccols; it += 256) {
            const int kk = it / ccols, cc = it - kk * ccols;
            f32x4 sum = zero4;
            for (int g2 = 0; g2 < G; ++g2) sum += ld4(&cpart[4 * ((kk * G + g2) * ccols + cc)]);
            float* pout = M.pctx + ((long)c * L.B + (u * K + k0 + kk)) * D;
            if (r0 == 0) st4(pout + 4 * (cb + cc), sum);
            else st4(pout + 4 * (cb + cc), ld4(pout + 4 * (cb + cc)) + sum);
          }
        }
      }
    }
    if (n == 0) {                                       // an empty chunk still defines its (zero) partial contexts
      for (int it = tid; it < K * ccols; it += 256) {
        const int kk = it / ccols, cc = it - kk * ccols;
        st4(M.pctx + ((long)c * L.B + (u * K + kk)) * D + 4 * (cb + cc), zero4);
      }
    }
  }
}

// The same step on the matrix pipe (default under beam search; avsr_attn_rnn_set_beam_kernel(1)).  attn_fwd_beam_kernel above is
// instruction-bound, not memory-bound: per workgroup 2 x 164 k scalar FMAs plus 160 cross-lane reduction steps per thread -- 29.5 us per
// decode step at c4 whether or not its two operand fetches overlap (measured).  Both products are small
// GEMMs with the K <= 16 hypotheses as one MFMA row tile:
//   scores   S[q][r] = sum_h Q[q][h] K[r][h]      [16 x 256] x [256 x 64]   wave w takes frames 16 w .. 16 w + 15: 64 x v_mfma_f32_16x16x4_f32
//   contexts C[q][d] = sum_r p[q][r] V[r][d]      [16 x 64]  x [64 x D]     wave w takes the 64-column blocks w, w + 4, ...: 64 each
// (k permutation: lane group g = lane >> 4 owns k in [64 g, 64 g + 64) of the score product -- its key row piece is 16 contiguous
// 16-byte loads -- and rows [16 g, 16 g + 16) of the context product; instruction i takes element i of every group's range.)
// Same outputs as attn_fwd_beam_kernel (raw scores, chunk max / sum, un-normalised partial contexts by hypothesis row); the summation
// order differs (an MFMA is an fp32 fma chain over its k), i.e. results agree to rounding, not bit for bit: tests/test_gpu_beam.py checks
// this path against the oracle and the other two against each other.
__global__ __launch_bounds__(256) void attn_fwd_beam_mfma_kernel(const AttnLaunch L) {
  __shared__ __attribute__((aligned(16))) float sc[16][68];               // scaled scores, then probabilities [q][r]
  __shared__ __attribute__((aligned(16))) float qs[16][260];              // queries [q][h], zero rows / columns beyond K / H
  int blk = blockIdx.x, mi = 0;
  for (int i = 0; i < L.nmech; ++i) {
    const int nb = (L.B / L.m[i].mem_div) * L.m[i].nchunk;
    if (blk < nb) { mi = i; break; }
    blk -= nb;
  }
  const AttnMechDev& M = L.m[mi];
  const int K = M.mem_div, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int c = blk % M.nchunk, u = blk / M.nchunk;
  const int len = min(M.len ? M.len[u] : M.T, M.T);
  const int t0 = c * M.chunk;
  const int n = max(0, min(M.chunk, len - t0));
  const int H = M.H, D = M.D;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // ---- key row piece of this lane: frame 16 wave + i16, k in [64 g, 64 g + 64) ----
  f32x4 kr[16];
  {
    const __amdgpu_buffer_rsrc_t krs = make_rsrc(M.keys + (long)u * M.T * H);
    const int r = 16 * wave + i16;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = 64 * g + 4 * j;
      kr[j] = ldb4(krs, (r < n && k < H) ? ((t0 + r) * H + k) * 4 : P_OOB);
    }
  }
  // queries, then the value operands of this wave's first 64-column block: all three operand fetches of the step are in flight together
  // (in-order counters: the query wait below leaves the younger value loads outstanding; barriers here are LDS-only, __syncthreads()
  // would drain them).  A lane fetches 16 bytes = 4 consecutive d of its 16 rows; instruction e of a row takes component e, so the
  // column set of MFMA e is {64 blk + 4 i16 + e}: a permutation of the block's columns that costs nothing (the result of (reg, e = 0..3)
  // is 4 consecutive d, one 16-byte store) and makes every load 256 contiguous bytes per row.
  f32x4 qv[4];
  {
    const __amdgpu_buffer_rsrc_t qrs = make_rsrc(M.query + (long)u * K * M.query_sb);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = tid * 4 + 1024 * it, q = e >> 8, k = e & 255;
      qv[it] = ldb4(qrs, (q < K && k < H) ? (int)(((long)q * M.query_sb + k) * 4) : P_OOB);
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e = tid * 4 + 1024 * it;
    st4(&qs[e >> 8][e & 255], qv[it]);
  }
  lds_barrier();
  {
    f32x4 acc = zero4;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const f32x4 qa = *reinterpret_cast<const f32x4*>(&qs[i16][64 * g + 4 * j]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[e], kr[j][e], acc, 0, 0, 0);
    }
    // C layout: lane holds S[q = 4 g + reg][frame = 16 wave + i16]
    const float gsc = (M.type == ATT_SCALED_LUONG) ? M.g[0] : 1.f;
    const int r = 16 * wave + i16;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int q = 4 * g + reg;
      if (q < K && r < n) M.scores[(long)(u * K + q) * M.scores_sb + t0 + r] = acc[reg];
      sc[q][r] = acc[reg] * gsc;
    }
  }
  // value operands of this wave's first 64-column block, requested before the softmax phase (the key registers are dead).  A lane
  // fetches 16 bytes = 4 consecutive d of its 16 rows; instruction e of a row takes component e, so the column set of MFMA e is
  // {64 blk + 4 i16 + e}: a permutation of the block's columns that costs nothing (the result of (reg, e = 0..3) is 4 consecutive d,
  // one 16-byte store) and makes every load 256 contiguous bytes per row.  (Requested at the very top instead -- hidden loads, keys,
  // queries and values in one burst -- the kernel needs 159 registers and loses a workgroup per CU: 27.0 us against 23 us.)
  const __amdgpu_buffer_rsrc_t vrs = make_rsrc(M.values + (long)u * M.values_sb);
  const int vst = (int)M.values_st * 4;
  const int nblk = (D + 63) >> 6;
  f32x4 vq[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = 16 * g + i, d = 64 * wave + 4 * i16;
    vq[i] = ldb4(vrs, (wave < nblk && r < n && d < D) ? (t0 + r) * vst + d * 4 : P_OOB);
  }
  lds_barrier();
  // ---- chunk max / exp / sum per hypothesis: one wave per query, one lane per frame (rows q >= K: zero probabilities) ----
  for (int q = wave; q < 16; q += 4) {
    const float v = (lane < n && q < K) ? sc[q][lane] : -INFINITY;
    const float mx = wave_max(v);
    const float pr = (lane < n && q < K) ? expf(v - mx) : 0.f;
    const float lsum = wave_sum(pr);
    sc[q][lane] = pr;
    if (lane == 0 && q < K) {
      const int b = u * K + q;
      M.pm[(long)c * L.B + b] = (n > 0) ? mx : -INFINITY;
      M.pl[(long)c * L.B + b] = lsum;
    }
  }
  lds_barrier();
  // ---- partial contexts: 64-column blocks wave, wave + 4, ... ----
  for (int blk = wave; blk < nblk; blk += 4) {
    if (blk != wave) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = 16 * g + i, d = 64 * blk + 4 * i16;
        vq[i] = ldb4(vrs, (r < n && d < D) ? (t0 + r) * vst + d * 4 : P_OOB);
      }
    }
    f32x4 acc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 pa = *reinterpret_cast<const f32x4*>(&sc[i16][16 * g + 4 * j]);
#pragma unroll
      for (int ee = 0; ee < 4; ++ee)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[ee], vq[4 * j + ee][e], acc[e], 0, 0, 0);
    }
    const int d = 64 * blk + 4 * i16;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int q = 4 * g + reg;
      if (q < K && d < D) st4(M.pctx + ((long)c * L.B + (u * K + q)) * D + d, f32x4{acc[0][reg], acc[1][reg], acc[2][reg], acc[3][reg]});
    }
  }
}

// Per-step backward.  Inputs: d ctx [B,D] (gradient of this step's context), the forward context,
// the saved raw scores and softmax partial statistics of this step.  Outputs: d score [B,T] (wrt the
// softmax input) and per-chunk partial gradients of the query.
//   d alpha_t = values_t . dctx ;  d s_t = alpha_t * (d alpha_t - ctx . dctx)       [sum_j alpha_j dalpha_j = ctx.dctx]
//   Luong:    d q      += g * d s_t * keys_t
//   Bahdanau: d pq[h]  += d s_t * v[h] * (1 - tanh^2(keys_t[h] + pq[h] + b[h]))
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnLaunch L) {
  __shared__ float ds[ATTN_MAX_CHUNK];
  __shared__ float red[4];
  __shared__ __attribute__((aligned(16))) float cpart[1024];
  int mi, b, c;
  locate(L, mi, b, c);
  const AttnMechDev& M = L.m[mi];
  const int tid = threadIdx.x;
  const int len = min(M.len ? M.len[b] : M.T, M.T);
  const int t0 = c * M.chunk;
  const int n = max(0, min(M.chunk, len - t0));
  const int H = M.H, D = M.D;

  // global softmax statistics of this row from the forward partials
  float Mx = -INFINITY;
  for (int j = 0; j < M.nchunk; ++j) Mx = fmaxf(Mx, M.pm[(long)j * L.B + b]);
  float Ls = 0.f;
  for (int j = 0; j < M.nchunk; ++j) {
    const float pmj = M.pm[(long)j * L.B + b];
    if (pmj != -INFINITY) Ls += expf(pmj - Mx) * M.pl[(long)j * L.B + b];
  }
  const float invL = Ls > 0.f ? 1.f / Ls : 0.f;

  const float* dctx = M.dctx + (long)b * M.dctx_sb;
  const float* ctx = M.ctx + (long)b * M.ctx_sb;
  float cdl = 0.f;
  for (int k = tid; k < D; k += 256) cdl += ctx[k] * dctx[k];
  const float cd = block_sum_256(cdl, red);

  const float gscale = (M.type == ATT_SCALED_LUONG) ? M.g[0] : 1.f;
  const float* vals = M.values + (long)b * M.values_sb;
  const int s16 = tid & 15, rg = tid >> 4;
  const bool fast = M.type <= ATT_SCALED_LUONG && D <= 256 && (long)M.T * H * 4 < (1L << 31) && (long)M.T * M.values_st * 4 < (1L << 31);
  if (fast) {
    f32x4 d4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 4 * s16 + 64 * j;
      d4[j] = k < D ? ld4(dctx + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float* raws = M.scores + (long)b * M.scores_sb + t0;
    float* dsr = M.dscores + (long)b * M.dscores_sb + t0;
    rows_dot16(make_rsrc(vals), t0 * (int)M.values_st * 4, (int)M.values_st * 4, D, n, d4, [&](int r, float acc) {
      if (s16 == 0) {
        const float alpha = expf(raws[r] * gscale - Mx) * invL;
        const float d = alpha * (acc - cd);
        ds[r] = d;
        dsr[r] = d;
      }
    });
  } else
  for (int r = rg; r < n; r += 16) {
    const float* vrow = vals + (long)(t0 + r) * M.values_st;
    float acc = 0.f;
    for (int k = 4 * s16; k < D; k += 64) {
      const f32x4 a = ld4(vrow + k), d4 = ld4(dctx + k);
      acc += a[0] * d4[0] + a[1] * d4[1] + a[2] * d4[2] + a[3] * d4[3];
    }
    acc = group16_sum(acc);
    if (s16 == 0) {
      const float raw = M.scores[(long)b * M.scores_sb + t0 + r];
      const float alpha = expf(raw * gscale - Mx) * invL;
      const float d = alpha * (acc - cd);
      ds[r] = d;
      M.dscores[(long)b * M.dscores_sb + t0 + r] = d;
    }
  }
  // zero d score for the masked tail of this chunk (keeps the post-loop GEMMs exact)
  for (int r = n + tid; r < M.chunk && t0 + r < M.T; r += 256) M.dscores[(long)b * M.dscores_sb + t0 + r] = 0.f;
  __syncthreads();

  // partial d query over this chunk
  const int cols = H >> 2;
  const float* keys = M.keys + (long)b * M.T * H;
  const float* q = M.query + (long)b * M.query_sb;
  float* pout = M.pdq + ((long)c * L.B + b) * H;
  for (int cb = 0; cb < cols; cb += 256) {
    const int ccols = min(256, cols - cb);
    const int G = 256 / ccols;
    const int col = tid % ccols, grp = tid / ccols;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (grp < G) {
      const int k = 4 * (cb + col);
      if (fast) {
        acc = rows_wsum(make_rsrc(keys), t0 * H * 4, H * 4, cb + col, grp, G, n, ds, gscale);
      } else if (M.type <= ATT_SCALED_LUONG) {
        for (int r = grp; r < n; r += G) acc += (ds[r] * gscale) * ld4(keys + (long)(t0 + r) * H + k);
      } else {
        f32x4 pq = ld4(q + k);
        if (M.bq) pq += ld4(M.bq + k);
        const f32x4 vv = ld4(M.v + k);
        for (int r = grp; r < n; r += G) {
          const f32x4 kv = ld4(keys + (long)(t0 + r) * H + k) + pq;
          f32x4 th;
          th[0] = tanhf(kv[0]); th[1] = tanhf(kv[1]); th[2] = tanhf(kv[2]); th[3] = tanhf(kv[3]);
          acc += ds[r] * (vv * (1.f - th * th));
        }
      }
    }
    if (G == 1) {
      if (grp < G) st4(pout + 4 * (cb + col), acc);
    } else {
      __syncthreads();
      if (grp < G) st4(&cpart[4 * (grp * ccols + col)], acc);
      __syncthreads();
      if (tid < ccols) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int g2 = 0; g2 < G; ++g2) s += ld4(&cpart[4 * (g2 * ccols + tid)]);
        st4(pout + 4 * (cb + tid), s);
      }
    }
  }
}

// In-place: scores[b,l,:] (raw) -> alpha[b,l,:]; also rowdot[b*L+l] = sum_t dscores*raw (for d g).
// One workgroup per (b, l) row.
__global__ __launch_bounds__(256) void attn_alpha_rows_kernel(float* scores, const float* dscores, const int* len,
                                                              const int* steplen, const float* g, float* rowdot,
                                                              int B, int Lsteps, int T) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const int b = row / Lsteps, l = row % Lsteps;
  const int n = min(len ? len[b] : T, T);
  float* s = scores + (long)row * T;
  const float* d = dscores + (long)row * T;
  const bool step_valid = steplen ? (l < steplen[b]) : true;
  const float gs = g ? g[0] : 1.f;
  float mx = -INFINITY, dot = 0.f;
  for (int t = threadIdx.x; t < n; t += 256) {
    mx = fmaxf(mx, s[t] * gs);
    dot += d[t] * s[t];
  }
  mx = block_max_256(mx, red);
  dot = block_sum_256(dot, red);
  float sum = 0.f;
  for (int t = threadIdx.x; t < n; t += 256) sum += expf(s[t] * gs - mx);
  sum = block_sum_256(sum, red);
  const float inv = (sum > 0.f && step_valid) ? 1.f / sum : 0.f;
  for (int t = threadIdx.x; t < T; t += 256) s[t] = (t < n) ? expf(s[t] * gs - mx) * inv : 0.f;
  if (threadIdx.x == 0 && rowdot) rowdot[row] = step_valid ? dot : 0.f;
}

// Bahdanau post-loop: d keys[b,t,h] (+)= sum_l ds[b,l,t] * v[h] * (1 - tanh^2(keys + pq[b,l,h] + bq[h]))
//                     d v partial[blk][h] = sum_{t in blk, l} ds[b,l,t] * tanh(...)
// grid = (ceil(T/16), B); block = H threads rounded up (<= 1024)
__global__ void bahdanau_dkeys_kernel(const float* keys, const float* pq, long pq_sb, long pq_sl, const float* dscores,
                                      const float* v, const float* bq, const int* len, float* dkeys, float* dv_part,
                                      int B, int Lsteps, int T, int H) {
  const int b = blockIdx.y, tb = blockIdx.x * 16;
  const int n = min(len ? len[b] : T, T);
  for (int h = threadIdx.x; h < H; h += blockDim.x) {
    const float vh = v[h], bh = bq ? bq[h] : 0.f;
    float dvacc = 0.f;
    for (int t = tb; t < min(tb + 16, T); ++t) {
      float acc = 0.f;
      if (t < n) {
        const float k = keys[((long)b * T + t) * H + h] + bh;
        for (int l = 0; l < Lsteps; ++l) {
          const float d = dscores[((long)b * Lsteps + l) * T + t];
          if (d != 0.f) {
            const float th = tanhf(k + pq[(long)b * pq_sb + (long)l * pq_sl + h]);
            acc += d * (1.f - th * th);
            dvacc += d * th;
          }
        }
      }
      dkeys[((long)b * T + t) * H + h] = acc * vh;
    }
    dv_part[((long)blockIdx.y * gridDim.x + blockIdx.x) * H + h] = dvacc;
  }
}

}  // namespace avsr

namespace avsr { extern int g_beam_dense; bool beam_dense_on(); }
static int g_beam_on = -1;
extern "C" int avsr_attn_rnn_set_beam_kernel(int32_t on) { g_beam_on = on ? 1 : 0; avsr::g_beam_dense = on == 1 ? 1 : 0; return AVSR_OK; }

extern "C" int avsr_attn_launch_raw(const void* launch, int backward, void* stream) {
  using namespace avsr;
  const AttnLaunch* L = (const AttnLaunch*)launch;
  if (!L || L->nmech <= 0 || L->nmech > AVSR_MAX_MECH) return AVSR_ERR_ARG;
  const int nblk = L->blk_off[L->nmech];
  if (nblk <= 0) return AVSR_ERR_ARG;
  ProfScope ps(backward ? PROF_ATTN_BWD : PROF_ATTN_FWD, (hipStream_t)stream);
  // beam search over shared memories: one workgroup per (utterance, chunk) runs the K hypotheses (attn_fwd_beam_kernel)
  bool beam = !backward, mfma_ok = true;
  int nbeam = 0;
  if (g_beam_on < 0) { const char* e = getenv("AVSR_ATTN_BEAM"); g_beam_on = e ? (atoi(e) != 0) : 1; }
  const int beam_on = g_beam_on;
  for (int i = 0; i < L->nmech && beam; ++i) {
    const AttnMechDev& M = L->m[i];
    beam = beam_on && M.mem_div > 1 && M.mem_div <= ATTN_BEAM_KMAX && L->B % M.mem_div == 0 && M.type <= ATT_SCALED_LUONG && M.H <= 256 && M.H % 4 == 0 &&
           M.chunk <= 64 && M.D % 4 == 0 && M.D <= 1024 && (long)M.T * M.H * 4 < (1L << 31) && (long)M.T * M.values_st * 4 < (1L << 31);
    nbeam += (L->B / (M.mem_div > 0 ? M.mem_div : 1)) * M.nchunk;
    mfma_ok = mfma_ok && M.D % 4 == 0;
  }
  if (backward) hipLaunchKernelGGL(attn_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, *L);
  else if (beam && mfma_ok && avsr::beam_dense_on()) hipLaunchKernelGGL(attn_fwd_beam_mfma_kernel, dim3(nbeam), dim3(256), 0, (hipStream_t)stream, *L);
  else if (beam) {
    static const int pf = getenv("AVSR_ATTN_BEAM_PREFETCH") ? atoi(getenv("AVSR_ATTN_BEAM_PREFETCH")) : 0;   // measured: 46.6 vs 29.5 us (200 VGPRs, one workgroup fewer per CU): off
    if (pf >= 16) hipLaunchKernelGGL(attn_fwd_beam_kernel<16>, dim3(nbeam), dim3(256), 0, (hipStream_t)stream, *L);
    else if (pf >= 8) hipLaunchKernelGGL(attn_fwd_beam_kernel<8>, dim3(nbeam), dim3(256), 0, (hipStream_t)stream, *L);
    else hipLaunchKernelGGL(attn_fwd_beam_kernel<0>, dim3(nbeam), dim3(256), 0, (hipStream_t)stream, *L);
  }
  else hipLaunchKernelGGL(attn_fwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, *L);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_attn_alpha_rows(float* scores, const float* dscores, const int32_t* len, const int32_t* steplen,
                                    const float* g, float* rowdot, int32_t B, int32_t L, int32_t T, void* stream) {
  using namespace avsr;
  if (!scores || !dscores || B <= 0 || L <= 0 || T <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(attn_alpha_rows_kernel, dim3(B * L), dim3(256), 0, (hipStream_t)stream, scores, dscores, len,
                     steplen, g, rowdot, B, L, T);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_bahdanau_dkeys(const float* keys, const float* pq, int64_t pq_sb, int64_t pq_sl,
                                   const float* dscores, const float* v, const float* bq, const int32_t* len,
                                   float* dkeys, float* dv_part, int32_t B, int32_t L, int32_t T, int32_t H,
                                   void* stream) {
  using namespace avsr;
  if (!keys || !pq || !dscores || !v || !dkeys || !dv_part) return AVSR_ERR_ARG;
  int threads = ((H + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  hipLaunchKernelGGL(bahdanau_dkeys_kernel, dim3((T + 15) / 16, B), dim3(threads), 0, (hipStream_t)stream, keys, pq,
                     (long)pq_sb, (long)pq_sl, dscores, v, bq, len, dkeys, dv_part, B, L, T, H);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}
