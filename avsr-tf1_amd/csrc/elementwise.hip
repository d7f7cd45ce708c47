// Memory-bound helper kernels of the train step: weight transposes, batch-norm, column reductions,
// embedding lookup / gradient, masked sequence cross-entropy, AU regression loss, L2 + global-norm
// clip + Adam.  All are single-pass, 16-byte vectorised where the layout allows, and deterministic
// (two-stage reductions, no float atomics).
#include "common.h"
#include "avsr_hip.h"

namespace avsr {

// ---------------------------------------------------------------------------------------------
// batched 2-D transposes (derived "Wt" operands of the step kernels), up to 16 jobs per launch
struct TJob { const float* src; float* dst; int rows, cols; };
struct TLaunch { int njob; TJob job[AVSR_MAX_TRANSPOSE]; };

__global__ void transpose_kernel(const TLaunch L) {
  __shared__ float tile[32][33];
  const TJob& J = L.job[blockIdx.z];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int tiles_c = (J.cols + 31) / 32, tiles_r = (J.rows + 31) / 32;
  for (int tIdx = blockIdx.x; tIdx < tiles_c * tiles_r; tIdx += gridDim.x) {
    const int r0 = (tIdx / tiles_c) * 32, c0 = (tIdx % tiles_c) * 32;
    for (int j = ty; j < 32; j += 8) {
      const int r = r0 + j, c = c0 + tx;
      tile[j][tx] = (r < J.rows && c < J.cols) ? J.src[(long)r * J.cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
      const int c = c0 + j, r = r0 + tx;
      if (r < J.rows && c < J.cols) J.dst[(long)c * J.rows + r] = tile[tx][j];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// column reduction: out[f] = alpha * sum_r a[r][f] * (b ? b[r][f] : 1) + beta * out[f]
// two-level row addressing on a and b (same convention as avsr_gemm)
__device__ __forceinline__ long rowoff(int r, long ld, int T, long ldo) {
  return T ? (long)(r / T) * ldo + (long)(r % T) * ld : (long)r * ld;
}

// Narrow matrices (F < 256): G = 256 / F row sub-groups of F columns share a block; their sums are combined through LDS in
// group order, so every block emits ONE partial row whatever G is (deterministic).
__device__ __forceinline__ void block_group_reduce(float s, int idx, int F, int G, float* red, float* out_row) {
  if (G == 1) { if (idx < F) out_row[idx] = s; return; }
  if (idx < G * F) red[idx] = s;
  __syncthreads();
  if (idx < F) {
    float t = 0.f;
    for (int g = 0; g < G; ++g) t += red[g * F + idx];
    out_row[idx] = t;
  }
  __syncthreads();
}

__global__ void colsum_partial_kernel(const float* a, long lda, int Ta, long ldoa, const float* b, long ldb, int Tb,
                                      long ldob, float* part, int rows, int F, int rows_per_blk) {
  __shared__ float red[256];
  const int G = F < 256 ? 256 / F : 1;
  const int r0 = blockIdx.x * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  if (G > 1) {
    const int idx = threadIdx.x, f = idx % F, g = idx / F;
    float s = 0.f;
    if (idx < G * F)
      for (int r = r0 + g; r < r1; r += G) {
        const float x = a[rowoff(r, lda, Ta, ldoa) + f];
        s += b ? x * b[rowoff(r, ldb, Tb, ldob) + f] : x;
      }
    block_group_reduce(s, idx, F, G, red, part + (long)blockIdx.x * F);
    return;
  }
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float s = 0.f;
    for (int r = r0; r < r1; ++r) {
      const float x = a[rowoff(r, lda, Ta, ldoa) + f];
      s += b ? x * b[rowoff(r, ldb, Tb, ldob) + f] : x;
    }
    part[(long)blockIdx.x * F + f] = s;
  }
}

// one block per 32 columns: 32 row-groups x 32 columns of threads walk the partials (4 loads in flight), then an LDS tree
// columns f < split go to out[f], the rest to out2[f - split] (split = F: one destination)
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* part, long ld, int nblk, float* out, int F, float alpha, float beta,
                                                            float* out2 = nullptr, int split = 0x7fffffff) {
  __shared__ double red[32][33];
  const int fl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int f = blockIdx.x * 32 + fl;
  double s = 0.0;
  if (f < F) {
    int i = g;
    for (; i + 96 < nblk; i += 128) {
      const float a0 = part[(long)i * ld + f], a1 = part[(long)(i + 32) * ld + f];
      const float a2 = part[(long)(i + 64) * ld + f], a3 = part[(long)(i + 96) * ld + f];
      s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    }
    for (; i < nblk; i += 32) s += (double)part[(long)i * ld + f];
  }
  red[g][fl] = s;
  __syncthreads();
  if (g == 0 && f < F) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 32; ++j) t += red[j][fl];
    const float v = alpha * (float)t;
    float* const o = f < split ? out + f : out2 + (f - split);
    *o = beta != 0.f ? v + beta * *o : v;
  }
}

// ---------------------------------------------------------------------------------------------
// batch norm over rows (tf.layers.batch_normalization axis=-1, encoder.py:44-50): statistics over ALL
// B*T rows including zero padding.  Stage 1: partial sums.  Stage 2: partial centred squares.  Stage 3:
// normalise (+ moving-average update and saved mean / inv-std by block 0).
// Thread layout of the row-walking BN kernels: G = 256 / F row sub-groups of F columns (F < 256), so narrow feature
// vectors (80 audio / 128 video) still use the whole block; partials are per (block, sub-group).
__global__ void bn_partial_sum_kernel(const float* x, float* part, int rows, int F, int rows_per_blk) {
  __shared__ float red[256];
  const int G = F < 256 ? 256 / F : 1;
  const int r0 = blockIdx.x * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  for (int base = 0; base < (G > 1 ? 1 : F); base += blockDim.x) {      // G > 1: a single pass (G*F <= 256)
    const int idx = base + threadIdx.x;
    const int f = G > 1 ? idx % F : idx, g = G > 1 ? idx / F : 0;
    float s0 = 0.f, s1 = 0.f;
    if (G > 1 ? idx < G * F : idx < F) {
      int r = r0 + g;
      for (; r + G < r1; r += 2 * G) { s0 += x[(long)r * F + f]; s1 += x[(long)(r + G) * F + f]; }
      if (r < r1) s0 += x[(long)r * F + f];
    }
    if (G > 1) block_group_reduce(s0 + s1, idx, F, G, red, part + (long)blockIdx.x * F);
    else if (idx < F) part[(long)blockIdx.x * F + idx] = s0 + s1;
  }
}

// mean[f] = sum of partials / rows   (one launch, one block per 32 columns; npart = nblk * G)
// total != nullptr: the divisor is the (all-reduced) row count total[0] instead of `rows` (sync batch-norm).
__global__ __launch_bounds__(1024) void bn_mean_kernel(const float* part, int npart, float* mean, int rows, int F,
                                                       const float* total = nullptr) {
  if (total) rows = (int)total[0];
  __shared__ double red[32][33];
  const int fl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int f = blockIdx.x * 32 + fl;
  double s = 0.0;
  if (f < F)
    for (int i = g; i < npart; i += 32) s += (double)part[(long)i * F + f];
  red[g][fl] = s;
  __syncthreads();
  if (g == 0 && f < F) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 32; ++j) t += red[j][fl];
    mean[f] = (float)(t / rows);
  }
}

__global__ void bn_partial_sq_kernel(const float* x, const float* mean_v, float* part, int rows, int F, int rows_per_blk) {
  __shared__ float red[256];
  const int G = F < 256 ? 256 / F : 1;
  const int r0 = blockIdx.x * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  for (int base = 0; base < (G > 1 ? 1 : F); base += blockDim.x) {
    const int idx = base + threadIdx.x;
    const int f = G > 1 ? idx % F : idx, g = G > 1 ? idx / F : 0;
    float s0 = 0.f, s1 = 0.f;
    if (G > 1 ? idx < G * F : idx < F) {
      const float mean = mean_v[f];
      int r = r0 + g;
      for (; r + G < r1; r += 2 * G) {
        const float d0 = x[(long)r * F + f] - mean, d1 = x[(long)(r + G) * F + f] - mean;
        s0 += d0 * d0; s1 += d1 * d1;
      }
      if (r < r1) { const float d0 = x[(long)r * F + f] - mean; s0 += d0 * d0; }
    }
    if (G > 1) block_group_reduce(s0 + s1, idx, F, G, red, part + (long)blockIdx.x * F);
    else if (idx < F) part[(long)blockIdx.x * F + idx] = s0 + s1;
  }
}

// var from the centred-square partials, inverse std, moving-average update (one launch, one block per 32 columns)
__global__ __launch_bounds__(1024) void bn_var_kernel(const float* part, int npart, const float* mean_v, float* invstd_v, float* mov_mean,
                                                      float* mov_var, int rows, int F, float eps, float momentum,
                                                      int bessel, const float* total = nullptr) {
  if (total) rows = (int)total[0];
  __shared__ double red[32][33];
  const int fl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int f = blockIdx.x * 32 + fl;
  double s = 0.0;
  if (f < F)
    for (int i = g; i < npart; i += 32) s += (double)part[(long)i * F + f];
  red[g][fl] = s;
  __syncthreads();
  if (g == 0 && f < F) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 32; ++j) t += red[j][fl];
    const float var = (float)(t / rows), mean = mean_v[f];
    invstd_v[f] = rsqrtf(var + eps);
    if (mov_mean) {
      // bessel = 1: the fused kernel (rank-4 CNN maps) feeds the Bessel-corrected variance to the moving average;
      // 0: the non-fused path TF 1.13 takes for the rank-3 [B,T,F] encoder input feeds the biased tf.nn.moments variance
      const float unbiased = bessel ? var * ((float)rows / (float)max(1, rows - 1)) : var;
      mov_mean[f] = momentum * mov_mean[f] + (1.f - momentum) * mean;
      mov_var[f] = momentum * mov_var[f] + (1.f - momentum) * unbiased;
    }
  }
}

// y = (x - mean) * invstd * gamma + beta, flat over rows * F (F % 4 == 0: one float4 per thread-iteration)
__global__ void bn_apply_kernel(const float* x, const float* mean_v, const float* invstd_v, const float* mov_mean, const float* mov_var,
                                const float* gamma, const float* beta, float* y, long n4, int F, int training, float eps, int relu) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int f = (int)((i * 4) % F);
    const f32x4 xv = ld4(x + i * 4);
    f32x4 yv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float mean = training ? mean_v[f + e] : mov_mean[f + e];
      const float istd = training ? invstd_v[f + e] : rsqrtf(mov_var[f + e] + eps);
      yv[e] = (xv[e] - mean) * (gamma[f + e] * istd) + beta[f + e];
      if (relu) yv[e] = fmaxf(yv[e], 0.f);
    }
    st4(y + i * 4, yv);
  }
}

// xhat[r][f] = (x - mean) * invstd  (for d gamma = sum dy * xhat)
__global__ void bn_xhat_kernel(const float* x, const float* mean, const float* invstd, float* xhat, long n, int F) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    xhat[i] = (x[i] - mean[f]) * invstd[f];
  }
}

// ---------------------------------------------------------------------------------------------
// embedding lookup of the GO-prefixed label sequence (decoder_unimodal.py:66-68, :170)
__global__ void embed_labels_kernel(const float* emb, const int32_t* labels, int go, float* out, int32_t* fed, int B, int L,
                                    int E, int nsteps) {
  const int b = blockIdx.x / nsteps, l = blockIdx.x % nsteps;
  const int tok = (l == 0) ? go : labels[(long)b * L + l - 1];
  if (fed && threadIdx.x == 0) fed[(long)b * L + l] = tok;
  for (int e = threadIdx.x; e < E; e += blockDim.x) out[((long)b * L + l) * E + e] = emb[(long)tok * E + e];
}

// inverted dropout over a row-structured matrix (two-level row addressing like avsr_gemm)
__global__ void dropout_rows_kernel(const float* x, long ldx, int Tx, long ldox, float* y, long ldy, int Ty, long ldoy,
                                    int rows, int cols, const int32_t* seed, uint32_t stream, float keep, int idx_w,
                                    int idx_coff, int accumulate) {
  const long total = (long)rows * cols;
  const bool on = seed && keep < 1.0f;
  const uint32_t sd = on ? (uint32_t)seed[0] : 0u;
  const float inv = 1.0f / keep;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    float v = x[rowoff(r, ldx, Tx, ldox) + c];
    if (on) v = uniform01(sd, stream, (uint32_t)((long)r * idx_w + idx_coff + c)) < keep ? v * inv : 0.f;
    float* d = y + rowoff(r, ldy, Ty, ldoy) + c;
    *d = accumulate ? *d + v : v;
  }
}

// d emb[v, :] = sum over rows whose input token == v.  Block (v, chunk of 256 rows): the token ids of the chunk are
// staged in LDS, every thread owns embedding columns and walks the chunk in row order with UNCONDITIONAL loads (the
// select is on the value, so 8 rows are in flight); per-chunk partials [nchunk][V*E] are then summed in chunk order
// by colsum_final_kernel (deterministic summation order).
#define EG_ROWS 256
__global__ void embed_grad_partial_kernel(const float* dx, const int32_t* fed, float* part, int rows, int E, int V) {
  __shared__ int toks[EG_ROWS];
  const int v = blockIdx.x, base = blockIdx.y * EG_ROWS;
  const int n = min(EG_ROWS, rows - base);
  for (int j = threadIdx.x; j < EG_ROWS; j += blockDim.x) toks[j] = j < n ? fed[base + j] : -1;
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float a = 0.f;
    for (int j = 0; j < n; j += 8) {
      float x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = dx[(long)(base + min(j + k, n - 1)) * E + e];
#pragma unroll
      for (int k = 0; k < 8; ++k) a += (j + k < n && toks[j + k] == v) ? x[k] : 0.f;
    }
    part[((long)blockIdx.y * V + v) * E + e] = a;
  }
}

// ---------------------------------------------------------------------------------------------
// seq2seq.sequence_loss (seq2seq.py:165-171): masked sparse softmax CE averaged over sum(mask)+1e-12
__global__ void seqlen_sum_kernel(const int32_t* len, int B, int L, float* denom) {
  __shared__ float red[4];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) s += (float)min(max(len[b], 0), L);
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) denom[0] = s;
}

__global__ void set_scalar_kernel(float* p, float v) { p[0] = v; }

// mode 0: sparse softmax cross-entropy; 1: tf.losses.softmax_cross_entropy with label smoothing, whose default reduction makes
// the sequence loss the plain mean over ALL B*L rows (padding rows: imputed zero logits -> log V each, no gradient);
// 2: focal loss (gamma 2), 3: multi-class sigmoid-style cross-entropy on the clipped softmax (avsr/devel.py:12-51).
__global__ void seq_loss_kernel(float* logits, const int32_t* labels, const int32_t* len, const float* denom,
                                float* row_loss, float* dlogits, int B, int L, int V, int mode, float smooth) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= B * L) return;
  const int b = row / L, l = row % L;
  float* lg = logits + (long)row * V;
  float* dl = dlogits ? dlogits + (long)row * V : nullptr;
  const bool valid = l < len[b];
  const float inv = 1.f / (denom[0] + 1e-12f);     // mode 1: denom = number of rows B*L (of the GLOBAL batch under data parallelism)
  if (!valid) {
    row_loss[row] = mode == 1 ? logf((float)V) * inv : 0.f;
    for (int v = 0; v < V; ++v) lg[v] = 0.f;       // dynamic_decode(impute_finished=True): zero outputs once finished
    if (dl) for (int v = 0; v < V; ++v) dl[v] = 0.f;
    return;
  }
  float mx = lg[0];
  for (int v = 1; v < V; ++v) mx = fmaxf(mx, lg[v]);
  float s = 0.f;
  for (int v = 0; v < V; ++v) s += expf(lg[v] - mx);
  const float lse = mx + logf(s);
  const int y = labels[row];
  if (mode == 0) {
    row_loss[row] = (lse - lg[y]) * inv;
    if (dl) for (int v = 0; v < V; ++v) dl[v] = (expf(lg[v] - lse) - (v == y ? 1.f : 0.f)) * inv;
    return;
  }
  if (mode == 1) {
    const float off = smooth / (float)V, on = 1.f - smooth + off;
    float ce = 0.f;
    for (int v = 0; v < V; ++v) ce += (v == y ? on : off) * (lse - lg[v]);
    row_loss[row] = ce * inv;
    if (dl) for (int v = 0; v < V; ++v) dl[v] = (expf(lg[v] - lse) - (v == y ? on : off)) * inv;
    return;
  }
  // modes 2 / 3: loss = sum_v f_v(q_v), q = clip(softmax, 1e-7, 1 - 1e-7);  d/dz_k = p_k (a_k - sum_v a_v p_v), a_v = f_v'(q_v) [inside the clip]
  float loss = 0.f, dot = 0.f;
  for (int v = 0; v < V; ++v) {
    const float p = expf(lg[v] - lse);
    const float q = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
    const bool inside = (p >= 1e-7f) && (p <= 1.f - 1e-7f);
    float f, a;
    if (mode == 3) {
      f = (v == y) ? -logf(q) : -logf(1.f - q);
      a = (v == y) ? -1.f / q : 1.f / (1.f - q);
    } else {
      if (v == y) { f = -(1.f - q) * (1.f - q) * logf(q); a = 2.f * (1.f - q) * logf(q) - (1.f - q) * (1.f - q) / q; }
      else        { f = -q * q * logf(1.f - q);           a = -2.f * q * logf(1.f - q) + q * q / (1.f - q); }
    }
    loss += f;
    dot += inside ? a * p : 0.f;
  }
  row_loss[row] = loss * inv;
  if (dl)
    for (int v = 0; v < V; ++v) {
      const float p = expf(lg[v] - lse);
      const float q = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
      const bool inside = (p >= 1e-7f) && (p <= 1.f - 1e-7f);
      float a;
      if (mode == 3) a = (v == y) ? -1.f / q : 1.f / (1.f - q);
      else a = (v == y) ? 2.f * (1.f - q) * logf(q) - (1.f - q) * (1.f - q) / q : -2.f * q * logf(1.f - q) + q * q / (1.f - q);
      dl[v] = p * ((inside ? a : 0.f) - dot) * inv;
    }
}

// per-utterance average of the masked step losses: out[b] = sum_l row_loss[b,l] * (denom + 1e-12) / (min(len,L) + 1e-12)
// (sequence_loss(average_across_batch=False, average_across_timesteps=True), avsr/lm.py:390-401; row_loss carries 1/denom)
__global__ void seq_avg_kernel(const float* row_loss, const int32_t* len, const float* denom, float* out, int B, int L) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int l = 0; l < L; ++l) s += row_loss[(long)b * L + l];
  out[b] = s * (denom[0] + 1e-12f) / ((float)min(max(len[b], 0), L) + 1e-12f);
}

// ---------------------------------------------------------------------------------------------
// AU regression loss (encoder.py:173-189): pred = sigmoid(z), target = clip(aus,0,3)/3,
// loss = sum_w (pred - tgt)^2 / sum_w  over valid frames x 2 units;  dz = weight * 2 (pred-tgt) pred (1-pred) / sum_w
// total != nullptr: normalise by total[0] (the all-reduced count of valid frame-units of the GLOBAL batch) instead of the local count
__global__ void au_loss_kernel(const float* z, const float* aus, const int32_t* len, float* row_loss, float* dz,
                               int B, int T, float weight, const float* total) {
  __shared__ float red[4];
  float cnt = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) cnt += 2.f * (float)min(max(len[b], 0), T);
  cnt = block_sum_256(cnt, red);
  if (total) cnt = total[0];
  const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
  for (int row = blockIdx.x * 256 + threadIdx.x; row < B * T; row += gridDim.x * 256) {
    const int b = row / T, t = row % T;
    float l = 0.f;
    for (int k = 0; k < 2; ++k) {
      float d = 0.f;
      if (t < len[b]) {
        const float p = sigmoidf_(z[(long)row * 2 + k]);
        const float tg = fminf(fmaxf(aus[(long)row * 2 + k], 0.f), 3.f) / 3.f;
        l += (p - tg) * (p - tg) * inv;
        d = weight * 2.f * (p - tg) * p * (1.f - p) * inv;
      }
      if (dz) dz[(long)row * 2 + k] = d;
    }
    row_loss[row] = l * weight;
  }
}

// ---------------------------------------------------------------------------------------------
// normed Bahdanau score vector: vn = g * v / |v|   and its backward
__global__ void normed_v_kernel(const float* v, const float* g, float* vn, int H) {
  __shared__ float red[4];
  float s = 0.f;
  for (int h = threadIdx.x; h < H; h += 256) s += v[h] * v[h];
  s = block_sum_256(s, red);
  const float sc = g[0] * rsqrtf(s);
  for (int h = threadIdx.x; h < H; h += 256) vn[h] = v[h] * sc;
}
__global__ void normed_v_bwd_kernel(const float* v, const float* g, const float* dvn, float* dv, float* dg, int H) {
  __shared__ float red[4];
  float s = 0.f, d = 0.f;
  for (int h = threadIdx.x; h < H; h += 256) { s += v[h] * v[h]; d += v[h] * dvn[h]; }
  s = block_sum_256(s, red);
  d = block_sum_256(d, red);
  const float nrm = sqrtf(s);
  for (int h = threadIdx.x; h < H; h += 256) dv[h] = g[0] / nrm * (dvn[h] - d * v[h] / s);
  if (threadIdx.x == 0) dg[0] = d / nrm;
}

// ---------------------------------------------------------------------------------------------
// optimiser (seq2seq.py:175-178, :195-199, :245-246, :259-280)
// 1) L2: g += l2 * w on the RNN kernels, reg partial = 0.5 * l2 * sum w^2
// 2) partial sums of g^2  ->  global norm
// 3) Adam with clip-by-global-norm folded in; step counter and lr warm-up live on the device
struct Seg { long off, n; };
struct SegLaunch { int nseg; Seg seg[AVSR_MAX_SEGMENTS]; };

__global__ void l2_grad_kernel(const SegLaunch S, const float* w, float* g, float l2, float* part) {
  __shared__ float red[4];
  const Seg sg = S.seg[blockIdx.y];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < sg.n; i += (long)gridDim.x * 256) {
    const float x = w[sg.off + i];
    g[sg.off + i] += l2 * x;
    s += x * x;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = 0.5f * l2 * s;
}

__global__ void sumsq_partial_kernel(const float* g, long n, float* part) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += g[i] * g[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out[0] = scale * (sqrt ? sqrt(sum part) : sum part) [+ out[0] if accumulate]
__global__ void reduce_scalar_kernel(const float* part, int n, float* out, int do_sqrt, int accumulate, float scale) {
  __shared__ double dred[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)part[i];
  dred[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) dred[threadIdx.x] += dred[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double v = dred[0];
    if (do_sqrt) v = sqrt(v);
    v *= scale;
    out[0] = accumulate ? out[0] + (float)v : (float)v;
  }
}

// hyper: [0] step (as float-exact int32 via reinterpret), see avsr_adam_step
__global__ void adam_kernel(float* p, float* g, float* m, float* v, long n, const float* gnorm, int32_t* step,
                            float lr, int warmup, float clip, float b1, float b2, float eps, float grad_scale,
                            int decay_steps, int opt, float weight_decay) {
  const int t = step[0] + 1;
  float lr_now = lr;
  if (decay_steps > 0) {   // tf.train.cosine_decay_restarts(lr, global_step, first_decay_steps, t_mul=2, m_mul=1, alpha=0)
    const float frac = (float)(t - 1) / (float)decay_steps;
    const float i_restart = floorf(logf(1.0f + frac) / logf(2.0f));
    const float pw = exp2f(i_restart);
    const float within = (frac - (pw - 1.0f)) / pw;
    lr_now *= 0.5f * (1.0f + cosf(3.14159265358979323846f * within));
  }
  if (warmup > 0) lr_now *= fminf(1.0f, (float)t / (float)warmup);   // min(1, (global_step + 1) / warmup)
  const double c1 = 1.0 - exp((double)t * log((double)b1));
  const double c2 = 1.0 - exp((double)t * log((double)b2));
  const float lr_t = (float)((double)lr_now * sqrt(c2) / c1);
  float scale = grad_scale;
  if (clip > 0.f) scale *= clip / fmaxf(gnorm[0], clip);   // gnorm is the norm of the already-scaled gradient   // tf.clip_by_global_norm
  // opt: 0 Adam, 1 Nadam (ApplyAdam use_nesterov), 2 AdamW (decoupled decay: var -= weight_decay * var, then Adam),
  //      3 Momentum(0.9, no Nesterov): accum = 0.9 accum + g, var -= lr accum (accum lives in m)   (avsr/seq2seq.py:195-218)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * scale;
    if (opt == 3) {
      const float acc = 0.9f * m[i] + gi;
      m[i] = acc;
      p[i] -= lr_now * acc;
      continue;
    }
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float pi = p[i];
    if (opt == 2) pi -= weight_decay * pi;
    const float num = opt == 1 ? b1 * mi + (1.f - b1) * gi : mi;
    p[i] = pi - lr_t * num / (sqrtf(vi) + eps);
  }
}
__global__ void step_inc_kernel(int32_t* step) { step[0] += 1; }

// ---------------------------------------------------------------------------------------------
// tf.contrib.rnn.HighwayWrapper around an encoder cell (cells.py:89-90), applied per layer over the whole sequence:
//   carry = sigmoid(cpre),  y = x * carry + h * (1 - carry)  for t < len[b], 0 past the utterance (dynamic_rnn zero-fills)
// x = the layer's raw input, h = the (dropout-wrapped) cell output, cpre = x W_c + b_c.  All operands are row views [B*T, H].
struct HwView { float* p; long ld; int T; long ldo; };
__device__ __forceinline__ long hw_off(const HwView& v, int r) { return rowoff(r, v.ld, v.T, v.ldo); }

__global__ void highway_fwd_kernel(HwView x, HwView h, HwView cpre, HwView y, const int32_t* len, int rows, int H, int T) {
  const long total = (long)rows * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / H), c = (int)(i % H);
    float out = 0.f;
    if (!len || (r % T) < len[r / T]) {
      const float g = 1.f / (1.f + expf(-cpre.p[hw_off(cpre, r) + c]));
      out = x.p[hw_off(x, r) + c] * g + h.p[hw_off(h, r) + c] * (1.f - g);
    }
    y.p[hw_off(y, r) + c] = out;
  }
}

// dh = dy (1 - carry);  dcpre = dy (x - h) carry (1 - carry);  dx (+)= dy carry      (zero past the utterance)
__global__ void highway_bwd_kernel(HwView x, HwView h, HwView cpre, HwView dy, HwView dh, HwView dcpre, HwView dx, const int32_t* len,
                                   int rows, int H, int T, int accumulate_dx) {
  const long total = (long)rows * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / H), c = (int)(i % H);
    float vdh = 0.f, vdc = 0.f, vdx = 0.f;
    if (!len || (r % T) < len[r / T]) {
      const float g = 1.f / (1.f + expf(-cpre.p[hw_off(cpre, r) + c]));
      const float d = dy.p[hw_off(dy, r) + c];
      vdh = d * (1.f - g);
      vdc = d * (x.p[hw_off(x, r) + c] - h.p[hw_off(h, r) + c]) * g * (1.f - g);
      vdx = d * g;
    }
    dh.p[hw_off(dh, r) + c] = vdh;
    dcpre.p[hw_off(dcpre, r) + c] = vdc;
    float* px = dx.p + hw_off(dx, r) + c;
    *px = accumulate_dx ? *px + vdx : vdx;
  }
}

static inline int blocks_for(long n, int per = 256, int cap = 2048) {
  long b = (n + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace avsr

using namespace avsr;
#define S_(x) ((hipStream_t)(x))

extern "C" int avsr_transpose(const avsr_transpose_job* jobs, int32_t n, void* stream) {
  if (!jobs || n <= 0) return AVSR_ERR_ARG;
  for (int i0 = 0; i0 < n; i0 += AVSR_MAX_TRANSPOSE) {
    TLaunch L;
    L.njob = (n - i0) < AVSR_MAX_TRANSPOSE ? (n - i0) : AVSR_MAX_TRANSPOSE;
    int maxtiles = 1;
    for (int i = 0; i < L.njob; ++i) {
      const avsr_transpose_job& j = jobs[i0 + i];
      if (!j.src || !j.dst || j.rows <= 0 || j.cols <= 0) return AVSR_ERR_ARG;
      L.job[i] = TJob{j.src, j.dst, j.rows, j.cols};
      const int tiles = ((j.rows + 31) / 32) * ((j.cols + 31) / 32);
      if (tiles > maxtiles) maxtiles = tiles;
    }
    if (maxtiles > 1024) maxtiles = 1024;
    hipLaunchKernelGGL(transpose_kernel, dim3(maxtiles, 1, L.njob), dim3(256), 0, S_(stream), L);
    AVSR_CHECK_LAUNCH();
  }
  return AVSR_OK;
}

int avsr_colsum_final_launch(const float* part, int nblk, float* out, int F, float alpha, float beta, void* stream) {
  hipLaunchKernelGGL(colsum_final_kernel, dim3((F + 31) / 32), dim3(1024), 0, S_(stream), part, (long)F, nblk, out, F, alpha, beta);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

// one launch, two destinations: columns [0, split) -> out, [split, F) -> out2 (the weight and bias gradients of a convolution's slab)
int avsr_colsum_final_launch_split(const float* part, long ld, int nblk, float* out, float* out2, int split, int F, float alpha, float beta,
                                   void* stream) {
  hipLaunchKernelGGL(colsum_final_kernel, dim3((F + 31) / 32), dim3(1024), 0, S_(stream), part, ld, nblk, out, F, alpha, beta, out2, split);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

// ---------------------------------------------------------------------------------------------
// Deferred slab reductions.  Every weight-gradient launch of the lip CNN leaves per-workgroup partial slabs that a tiny kernel then sums
// (5 us of launch latency for ~1 MB, twelve times per step, each in line behind its convolution).  Nothing reads a weight gradient
// before the optimiser, so between avsr_slab_defer_begin() and avsr_slab_defer_end() those reductions are only RECORDED (the caller
// gives every convolution its own scratch region) and run as ONE launch at the end.  Same arithmetic per job as the kernels they replace.
#define SLABJ_MAX 32
struct SlabJob { const float* part; long ld; int nblk, F; float* out; float* out2; int split, kind, Ci, blk0; float alpha, beta; };
struct SlabLaunch { int n; SlabJob job[SLABJ_MAX]; };
__global__ __launch_bounds__(1024) void slab_final_multi_kernel(const SlabLaunch L) {
  __shared__ double red[32][33];
  int j = 0;
#pragma unroll 1
  for (int k = 1; k < L.n; ++k) if ((int)blockIdx.x >= L.job[k].blk0) j = k;
  j = __builtin_amdgcn_readfirstlane(j);
  const float* const part = L.job[j].part;
  const long ld = L.job[j].ld;
  const int nblk = L.job[j].nblk, F = L.job[j].F, Ci = L.job[j].Ci, kind = L.job[j].kind;
  const int fl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int f = ((int)blockIdx.x - L.job[j].blk0) * 32 + fl;
  long o0 = -1, o1 = -1;
  const int nw = 9 * Ci * 8;                          // (kind 1: kernel entries of the pixel-pair slab, then 8 bias entries)
  if (kind == 0) { if (f < F) o0 = f; }
  else if (f < nw) {
    const int co = f & 7, ci = (f >> 3) % Ci, t = (f >> 3) / Ci, ti = t / 3, tj = t - ti * 3;
    o0 = ((ti * 4 + tj) * Ci + ci) * 16 + co;
    o1 = ((ti * 4 + tj + 1) * Ci + ci) * 16 + 8 + co;
  } else if (f < nw + 8 && L.job[j].out2) { o0 = 12 * Ci * 16 + (f - nw); o1 = o0 + 8; }
  double s = 0.0;
  if (o0 >= 0) {
    int i = g;
    if (kind == 0) {
      for (; i + 96 < nblk; i += 128) {
        const float a0 = part[(long)i * ld + o0], a1 = part[(long)(i + 32) * ld + o0];
        const float a2 = part[(long)(i + 64) * ld + o0], a3 = part[(long)(i + 96) * ld + o0];
        s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      }
      for (; i < nblk; i += 32) s += (double)part[(long)i * ld + o0];
    } else {
      for (; i + 96 < nblk; i += 128) {
        const float a0 = part[(long)i * ld + o0], b0 = part[(long)i * ld + o1];
        const float a1 = part[(long)(i + 32) * ld + o0], b1 = part[(long)(i + 32) * ld + o1];
        const float a2 = part[(long)(i + 64) * ld + o0], b2 = part[(long)(i + 64) * ld + o1];
        const float a3 = part[(long)(i + 96) * ld + o0], b3 = part[(long)(i + 96) * ld + o1];
        s += (double)a0 + (double)b0;
        s += (double)a1 + (double)b1;
        s += (double)a2 + (double)b2;
        s += (double)a3 + (double)b3;
      }
      for (; i < nblk; i += 32) s += (double)part[(long)i * ld + o0] + (double)part[(long)i * ld + o1];
    }
  }
  red[g][fl] = s;
  __syncthreads();
  if (g == 0 && o0 >= 0) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][fl];
    const float beta = L.job[j].beta;
    float* o;
    float v;
    if (kind == 0) { v = L.job[j].alpha * (float)t; o = f < L.job[j].split ? L.job[j].out + f : L.job[j].out2 + (f - L.job[j].split); }
    else { v = (float)t; o = f < nw ? L.job[j].out + f : L.job[j].out2 + (f - nw); }
    *o = beta != 0.f ? v + beta * *o : v;
  }
}

namespace avsr {
static thread_local bool g_slab_defer = false;
static thread_local SlabLaunch g_slab_jobs;
// record a reduction instead of launching it; false: not deferring (the caller launches as before)
bool slab_defer_push(const float* part, long ld, int nblk, int F, float* out, float* out2, int split, int kind, int Ci, float alpha, float beta,
                     hipStream_t s);
bool slab_deferring() { return g_slab_defer; }
static int slab_flush(hipStream_t s) {
  SlabLaunch& L = g_slab_jobs;
  if (!L.n) return AVSR_OK;
  int blocks = 0;
  for (int j = 0; j < L.n; ++j) { L.job[j].blk0 = blocks; blocks += ((L.job[j].kind ? 9 * L.job[j].Ci * 8 + 8 : L.job[j].F) + 31) / 32; }
  hipLaunchKernelGGL(slab_final_multi_kernel, dim3(blocks), dim3(1024), 0, s, L);
  L.n = 0;
  return hipGetLastError() == hipSuccess ? AVSR_OK : AVSR_ERR_HIP;
}
bool slab_defer_push(const float* part, long ld, int nblk, int F, float* out, float* out2, int split, int kind, int Ci, float alpha, float beta,
                     hipStream_t s) {
  if (!g_slab_defer) return false;
  if (g_slab_jobs.n == SLABJ_MAX && slab_flush(s) != AVSR_OK) return false;
  SlabJob& J = g_slab_jobs.job[g_slab_jobs.n++];
  J = SlabJob{part, ld, nblk, F, out, out2, split, kind, Ci, 0, alpha, beta};
  return true;
}
}  // namespace avsr

extern "C" int avsr_slab_defer_begin(void) {
  avsr::g_slab_jobs.n = 0;                            // (a collection left open by an aborted pass is dropped)
  avsr::g_slab_defer = true;
  return AVSR_OK;
}
extern "C" int avsr_slab_defer_end(void* stream) {
  avsr::g_slab_defer = false;
  return avsr::slab_flush(S_(stream));
}

// partial rows `ld` floats apart (F <= ld): several column ranges of one slab are reduced to different destinations
int avsr_colsum_final_launch_ld(const float* part, long ld, int nblk, float* out, int F, float alpha, float beta, void* stream) {
  hipLaunchKernelGGL(colsum_final_kernel, dim3((F + 31) / 32), dim3(1024), 0, S_(stream), part, ld, nblk, out, F, alpha, beta);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

// ---------------------------------------------------------------------------------------------
// Many column sums in TWO launches (bias gradients of a train step: one per cell / layer, each a pass over a [B*T, F] record
// followed by a tiny reduction -- 12 + 33 launches of ~5-20 us before; seq2seq.py:222 tf.gradients of the `bias` variables).
// The jobs are independent; every block finds its job by its block index in the jobs' prefix sums.
#define CSM_MAX 32
struct CsmJob { const float* a; const float* b; float* out; float* part; long lda, ldoa, ldb, ldob; int Ta, Tb, rows, F, rpb, blk0, fblk0; float alpha, beta; };
struct CsmLaunch { int n; CsmJob job[CSM_MAX]; };

__global__ __launch_bounds__(256) void colsum_multi_partial_kernel(const CsmLaunch L) {
  __shared__ float red[256];
  int j = 0;
#pragma unroll 1
  for (int k = 1; k < L.n; ++k) if ((int)blockIdx.x >= L.job[k].blk0) j = k;
  const CsmJob& J = L.job[j];
  const int blk = blockIdx.x - J.blk0, F = J.F;
  const int G = F < 256 ? 256 / F : 1;
  const int r0 = blk * J.rpb, r1 = min(J.rows, r0 + J.rpb);
  const float* a = J.a; const float* b = J.b;
  if (G > 1) {
    const int idx = threadIdx.x, f = idx % F, g = idx / F;
    float s = 0.f;
    if (idx < G * F)
      for (int r = r0 + g; r < r1; r += G) {
        const float x = a[rowoff(r, J.lda, J.Ta, J.ldoa) + f];
        s += b ? x * b[rowoff(r, J.ldb, J.Tb, J.ldob) + f] : x;
      }
    block_group_reduce(s, idx, F, G, red, J.part + (long)blk * F);
    return;
  }
  // wide records (the [B*T, 4H] gate gradients: 131 MB each): 16-byte loads, eight rows in flight per thread -- a bandwidth stream,
  // not a latency chain (two 4-byte loads in flight per thread ran at 2.4 TB/s)
  const bool vec = (F & 3) == 0 && ((uintptr_t)a & 15) == 0 && (J.lda & 3) == 0 && (J.ldoa & 3) == 0 &&
                   (!b || (((uintptr_t)b & 15) == 0 && (J.ldb & 3) == 0 && (J.ldob & 3) == 0));
  if (vec) {
    for (int f = threadIdx.x * 4; f < F; f += 4 * blockDim.x) {
      f32x4 acc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int r = r0; r < r1; r += 8) {
        f32x4 x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool in = r + u < r1;
          x[u] = in ? ld4(a + rowoff(r + u, J.lda, J.Ta, J.ldoa) + f) : f32x4{0.f, 0.f, 0.f, 0.f};
          if (b) y[u] = in ? ld4(b + rowoff(r + u, J.ldb, J.Tb, J.ldob) + f) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += b ? x[u] * y[u] : x[u];
      }
      st4(J.part + (long)blk * F + f, ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])));
    }
    return;
  }
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float s0 = 0.f, s1 = 0.f;
    int r = r0;
    for (; r + 1 < r1; r += 2) {                        // two independent chains: two loads in flight per thread
      const float x0 = a[rowoff(r, J.lda, J.Ta, J.ldoa) + f], x1 = a[rowoff(r + 1, J.lda, J.Ta, J.ldoa) + f];
      s0 += b ? x0 * b[rowoff(r, J.ldb, J.Tb, J.ldob) + f] : x0;
      s1 += b ? x1 * b[rowoff(r + 1, J.ldb, J.Tb, J.ldob) + f] : x1;
    }
    if (r < r1) { const float x0 = a[rowoff(r, J.lda, J.Ta, J.ldoa) + f]; s0 += b ? x0 * b[rowoff(r, J.ldb, J.Tb, J.ldob) + f] : x0; }
    J.part[(long)blk * F + f] = s0 + s1;
  }
}

__global__ __launch_bounds__(1024) void colsum_multi_final_kernel(const CsmLaunch L) {
  __shared__ double red[32][33];
  int j = 0;
#pragma unroll 1
  for (int k = 1; k < L.n; ++k) if ((int)blockIdx.x >= L.job[k].fblk0) j = k;
  const CsmJob& J = L.job[j];
  const int F = J.F, nblk = (J.rows + J.rpb - 1) / J.rpb;
  const int fl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int f = (blockIdx.x - J.fblk0) * 32 + fl;
  const float* part = J.part;
  double s = 0.0;
  if (f < F)
    for (int i = g; i < nblk; i += 32) s += (double)part[(long)i * F + f];
  red[g][fl] = s;
  __syncthreads();
  if (g == 0 && f < F) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][fl];
    const float v = J.alpha * (float)t;
    J.out[f] = J.beta != 0.f ? v + J.beta * J.out[f] : v;
  }
}

extern "C" int avsr_colsum_multi(const avsr_colsum_job* jobs, int32_t n, float* scratch, int64_t scratch_floats, void* stream) {
  if (n <= 0) return AVSR_OK;
  if (!jobs || !scratch) return AVSR_ERR_ARG;
  for (int j0 = 0; j0 < n; j0 += CSM_MAX) {              // more than 32 jobs: consecutive launch pairs
    const int m = n - j0 < CSM_MAX ? n - j0 : CSM_MAX;
    CsmLaunch L = {};
    L.n = m;
    long used = 0;
    int blocks = 0, fblocks = 0;
    for (int k = 0; k < m; ++k) {
      const avsr_colsum_job& Q = jobs[j0 + k];
      if (!Q.a.ptr || !Q.out || Q.rows <= 0 || Q.F <= 0) return AVSR_ERR_ARG;
      CsmJob& J = L.job[k];
      J.a = Q.a.ptr; J.lda = Q.a.ld; J.Ta = Q.a.T; J.ldoa = Q.a.ldo;
      J.b = Q.b.ptr; J.ldb = Q.b.ld; J.Tb = Q.b.T; J.ldob = Q.b.ldo;
      J.out = Q.out; J.rows = Q.rows; J.F = Q.F; J.alpha = Q.alpha; J.beta = Q.beta;
      // <= 256 partial rows per job (the big records are [32000, 1024]: 125 rows per block), >= 32 rows per block
      int rpb = (Q.rows + 255) / 256;
      if (rpb < 32) rpb = 32;
      J.rpb = rpb;
      const int nblk = (Q.rows + rpb - 1) / rpb;
      J.part = scratch + used;
      used += (long)nblk * Q.F;
      J.blk0 = blocks; blocks += nblk;
      J.fblk0 = fblocks; fblocks += (Q.F + 31) / 32;
    }
    if (used > scratch_floats) return AVSR_ERR_ARG;
    hipLaunchKernelGGL(colsum_multi_partial_kernel, dim3(blocks), dim3(256), 0, S_(stream), L);
    AVSR_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_multi_final_kernel, dim3(fblocks), dim3(1024), 0, S_(stream), L);
    AVSR_CHECK_LAUNCH();
  }
  return AVSR_OK;
}

// Zero up to 8 buffers (32-bit words) per launch: the fills at the start of the backward pass as ONE engine kernel.
struct ZmLaunch { uint32_t* p[8]; long n[8]; int cnt; };
__global__ __launch_bounds__(256) void zero_multi_kernel(const ZmLaunch L) {
  const long stride = (long)gridDim.x * 256;
#pragma unroll 1
  for (int k = 0; k < L.cnt; ++k) {
    uint32_t* p = L.p[k];
    const long n = L.n[k], n4 = ((uintptr_t)p & 15) == 0 ? n >> 2 : 0;
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) reinterpret_cast<u32x4_*>(p)[i] = u32x4_{0u, 0u, 0u, 0u};
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = 0u;
  }
}
extern "C" int avsr_zero_multi(void* const* ptrs, const int64_t* n_words, int32_t count, void* stream) {
  if (count <= 0) return AVSR_OK;
  if (!ptrs || !n_words) return AVSR_ERR_ARG;
  for (int c0 = 0; c0 < count; c0 += 8) {
    ZmLaunch L = {};
    long tot = 0;
    for (int k = 0; k < 8 && c0 + k < count; ++k) {
      if (!ptrs[c0 + k] || n_words[c0 + k] < 0) return AVSR_ERR_ARG;
      L.p[k] = (uint32_t*)ptrs[c0 + k]; L.n[k] = n_words[c0 + k]; L.cnt = k + 1; tot += n_words[c0 + k];
    }
    long blocks = (tot / 4 + 256 * 4 - 1) / (256 * 4);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_multi_kernel, dim3((int)blocks), dim3(256), 0, S_(stream), L);
    AVSR_CHECK_LAUNCH();
  }
  return AVSR_OK;
}
// out[0] = a[0] + b on the device: the RNG key of the step's dropout / sampling masks = global step + per-rank offset
__global__ void add_int_kernel(const int32_t* a, int32_t b, int32_t* out) { out[0] = a[0] + b; }
extern "C" int avsr_add_int(const int32_t* a, int32_t b, int32_t* out, void* stream) {
  if (!a || !out) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(add_int_kernel, dim3(1), dim3(1), 0, S_(stream), a, b, out);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_colsum(const avsr_mat* a, const avsr_mat* b, int32_t rows, int32_t F, float alpha, float beta,
                           float* out, float* scratch, int64_t scratch_floats, void* stream) {
  if (!a || !a->ptr || !out || !scratch || rows <= 0 || F <= 0) return AVSR_ERR_ARG;
  const int maxblk = 2048;                                     // at most 2048 partial rows: the final pass stays one short launch
  int rpb = rows > 32 * maxblk ? (rows + maxblk - 1) / maxblk : 32;
  int nblk = (rows + rpb - 1) / rpb;
  if ((long)nblk * F > scratch_floats) {
    nblk = (int)(scratch_floats / F);
    if (nblk < 1) return AVSR_ERR_ARG;
    rpb = (rows + nblk - 1) / nblk;
    nblk = (rows + rpb - 1) / rpb;
  }
  const int th = 256;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(th), 0, S_(stream), a->ptr, (long)a->ld, a->T, (long)a->ldo,
                     b ? b->ptr : nullptr, b ? (long)b->ld : 0, b ? b->T : 0, b ? (long)b->ldo : 0, scratch, rows, F, rpb);
  AVSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_final_kernel, dim3((F + 31) / 32), dim3(1024), 0, S_(stream), scratch, (long)F, nblk, out, F, alpha, beta);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_batchnorm_fwd_ex(const float* x, float* y, int32_t rows, int32_t F, const float* gamma, const float* beta,
                                     float* moving_mean, float* moving_var, float* save_mean, float* save_invstd, int32_t training,
                                     float eps, float momentum, int32_t relu, int32_t bessel, float* scratch, int64_t scratch_floats, void* stream);

extern "C" int avsr_batchnorm_fwd(const float* x, float* y, int32_t rows, int32_t F, const float* gamma,
                                  const float* beta, float* moving_mean, float* moving_var, float* save_mean,
                                  float* save_invstd, int32_t training, float* scratch, int64_t scratch_floats,
                                  void* stream) {
  return avsr_batchnorm_fwd_ex(x, y, rows, F, gamma, beta, moving_mean, moving_var, save_mean, save_invstd, training, 1e-3f, 0.99f, 0,
                               0 /* rank-3 input: non-fused path, biased moving variance */, scratch, scratch_floats, stream);
}

extern "C" int avsr_batchnorm_fwd_ex(const float* x, float* y, int32_t rows, int32_t F, const float* gamma, const float* beta,
                                     float* moving_mean, float* moving_var, float* save_mean, float* save_invstd, int32_t training,
                                     float eps, float momentum, int32_t relu, int32_t bessel, float* scratch, int64_t scratch_floats, void* stream) {
  if (!x || !y || !gamma || !beta || rows <= 0 || F <= 0 || !scratch) return AVSR_ERR_ARG;
  if (F % 4) return AVSR_ERR_ARG;
  const int maxblk = 2048;
  int rpb = rows > 64 * maxblk ? (rows + maxblk - 1) / maxblk : 64;
  int nblk = (rows + rpb - 1) / rpb;
  if ((long)nblk * F + 2 * F > scratch_floats) {
    nblk = (int)((scratch_floats - 2 * F) / F);
    if (nblk < 1) return AVSR_ERR_ARG;
    rpb = (rows + nblk - 1) / nblk;
    nblk = (rows + rpb - 1) / rpb;
  }
  // scratch: partials [nblk][F] | mean [F] | invstd [F]  (mean / invstd go to save_mean / save_invstd when given)
  float* part = scratch;
  float* mean_v = save_mean ? save_mean : scratch + (long)nblk * F;
  float* invstd_v = save_invstd ? save_invstd : scratch + (long)nblk * F + F;
  if (training) {
    hipLaunchKernelGGL(bn_partial_sum_kernel, dim3(nblk), dim3(256), 0, S_(stream), x, part, rows, F, rpb);
    AVSR_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_mean_kernel, dim3((F + 31) / 32), dim3(1024), 0, S_(stream), part, nblk, mean_v, rows, F);
    AVSR_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_partial_sq_kernel, dim3(nblk), dim3(256), 0, S_(stream), x, mean_v, part, rows, F, rpb);
    AVSR_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_var_kernel, dim3((F + 31) / 32), dim3(1024), 0, S_(stream), part, nblk, mean_v, invstd_v, moving_mean,
                       moving_var, rows, F, eps, momentum, bessel);
    AVSR_CHECK_LAUNCH();
  } else if (!moving_mean || !moving_var) {
    return AVSR_ERR_ARG;
  }
  const long n4 = (long)rows * F / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, S_(stream), x, mean_v, invstd_v, moving_mean, moving_var, gamma, beta, y,
                     n4, F, training, eps, relu);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_batchnorm_apply(const float* x, float* y, int32_t rows, int32_t F, const float* gamma, const float* beta, const float* mean,
                                    const float* invstd, int32_t relu, void* stream) {
  if (!x || !y || !gamma || !beta || !mean || !invstd || rows <= 0 || F <= 0 || F % 4) return AVSR_ERR_ARG;
  const long n4 = (long)rows * F / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, S_(stream), x, mean, invstd, nullptr, nullptr, gamma, beta, y, n4, F, 1, 0.f, relu);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

// ---- sync batch-norm over data-parallel ranks: the three local phases around the two host-side all-reduces ----
static int bn_blocks(int rows, int F, int64_t scratch_floats, int* rpb_out) {
  const int maxblk = 2048;
  int rpb = rows > 64 * maxblk ? (rows + maxblk - 1) / maxblk : 64;
  int nblk = (rows + rpb - 1) / rpb;
  if ((long)nblk * F > scratch_floats) {
    nblk = (int)(scratch_floats / F);
    if (nblk < 1) return 0;
    rpb = (rows + nblk - 1) / nblk;
    nblk = (rows + rpb - 1) / rpb;
  }
  *rpb_out = rpb;
  return nblk;
}

extern "C" int avsr_batchnorm_sync_sum(const float* x, int32_t rows, int32_t F, float* sum_out, float* scratch,
                                       int64_t scratch_floats, void* stream) {
  if (!x || !sum_out || !scratch || rows <= 0 || F <= 0 || F % 4) return AVSR_ERR_ARG;
  int rpb;
  const int nblk = bn_blocks(rows, F, scratch_floats, &rpb);
  if (!nblk) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(bn_partial_sum_kernel, dim3(nblk), dim3(256), 0, S_(stream), x, scratch, rows, F, rpb);
  AVSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_mean_kernel, dim3((F + 31) / 32), dim3(1024), 0, S_(stream), scratch, nblk, sum_out, 1, F, nullptr);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_batchnorm_sync_sqsum(const float* x, int32_t rows, int32_t F, const float* sum_global,
                                         const float* total_rows, float* mean_out, float* sq_out, float* scratch,
                                         int64_t scratch_floats, void* stream) {
  if (!x || !sum_global || !total_rows || !mean_out || !sq_out || !scratch || rows <= 0 || F <= 0 || F % 4) return AVSR_ERR_ARG;
  int rpb;
  const int nblk = bn_blocks(rows, F, scratch_floats, &rpb);
  if (!nblk) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(bn_mean_kernel, dim3((F + 31) / 32), dim3(1024), 0, S_(stream), sum_global, 1, mean_out, 1, F, total_rows);
  AVSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_partial_sq_kernel, dim3(nblk), dim3(256), 0, S_(stream), x, mean_out, scratch, rows, F, rpb);
  AVSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_mean_kernel, dim3((F + 31) / 32), dim3(1024), 0, S_(stream), scratch, nblk, sq_out, 1, F, nullptr);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

// ---- ONE small collective per data-parallel step (SURVEY 8(e) collectives (2) + (3) fused) ----
// avsr_batchnorm_sync_moments: per-feature sum x | sum x^2 of this rank's rows in DOUBLE precision into out64 [2F] (the global
// variance is then E[x^2] - mean^2 evaluated in fp64: exact to ~1e-13 relative for feature-scale inputs, so the second,
// mean-dependent reduction -- and its all-reduce -- is not needed).
__global__ __launch_bounds__(256) void bn_moments_partial_kernel(const float* __restrict__ x, double* __restrict__ part, int rows, int F, int rpb) {
  const int r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
  for (int f = threadIdx.x; f < F; f += 256) {
    double s = 0.0, s2 = 0.0;
    for (int r = r0; r < r1; ++r) { const double v = (double)x[(long)r * F + f]; s += v; s2 += v * v; }
    part[(long)blockIdx.x * 2 * F + f] = s;
    part[(long)blockIdx.x * 2 * F + F + f] = s2;
  }
}
__global__ __launch_bounds__(256) void bn_moments_final_kernel(const double* __restrict__ part, int nblk, int F2, double* __restrict__ out) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F2) return;
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += part[(long)i * F2 + f];
  out[f] = s;
}
extern "C" int avsr_batchnorm_sync_moments(const float* x, int32_t rows, int32_t F, double* out64, float* scratch, int64_t scratch_floats,
                                           void* stream) {
  if (!x || !out64 || !scratch || rows <= 0 || F <= 0 || ((uintptr_t)scratch & 7)) return AVSR_ERR_ARG;
  int nblk = (rows + 63) / 64;
  if (nblk > 1024) nblk = 1024;
  while (nblk > 1 && (long)nblk * 2 * F * 2 > scratch_floats) nblk /= 2;
  if ((long)nblk * 2 * F * 2 > scratch_floats) return AVSR_ERR_ARG;
  const int rpb = (rows + nblk - 1) / nblk;
  nblk = (rows + rpb - 1) / rpb;
  double* part = reinterpret_cast<double*>(scratch);
  hipLaunchKernelGGL(bn_moments_partial_kernel, dim3(nblk), dim3(256), 0, S_(stream), x, part, rows, F, rpb);
  AVSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_moments_final_kernel, dim3((2 * F + 255) / 256), dim3(256), 0, S_(stream), part, nblk, 2 * F, out64);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}
// avsr_dp_sync_unpack: the all-reduced buffer [sum(mask), AU count | per stream: sum x [F], sum x^2 [F], rows] -> the float32 operands
// the step's kernels read: dp_norm[0..1]; per stream mean [F], centred squares sum (x - mean)^2 [F] = sum x^2 - rows*mean^2, rows.
struct DpUnpack { const double* buf; float* dp_norm; int nstream; int off[4]; int F[4]; float* mean[4]; float* sq[4]; float* rows[4]; };
__global__ void dp_sync_unpack_kernel(const DpUnpack U) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 2 && U.dp_norm) U.dp_norm[t] = (float)U.buf[t];
  for (int s = 0; s < U.nstream; ++s) {
    const double* b = U.buf + U.off[s];
    const int F = U.F[s];
    const double n = b[2 * F];
    if (t == 0) U.rows[s][0] = (float)n;
    if (t < F) {
      const double m = n > 0.0 ? b[t] / n : 0.0;
      double c = b[F + t] - n * m * m;
      if (c < 0.0) c = 0.0;
      U.mean[s][t] = (float)m;
      U.sq[s][t] = (float)c;
    }
  }
}
extern "C" int avsr_dp_sync_unpack(const double* buf, float* dp_norm, int32_t nstream, const int32_t* off, const int32_t* F,
                                   float* const* mean, float* const* sq, float* const* rows, void* stream) {
  if (!buf || nstream < 0 || nstream > 4) return AVSR_ERR_ARG;
  DpUnpack U = {};
  U.buf = buf; U.dp_norm = dp_norm; U.nstream = nstream;
  int maxF = 2;
  for (int s = 0; s < nstream; ++s) {
    if (!off || !F || !mean || !sq || !rows || !mean[s] || !sq[s] || !rows[s] || F[s] <= 0) return AVSR_ERR_ARG;
    U.off[s] = off[s]; U.F[s] = F[s]; U.mean[s] = mean[s]; U.sq[s] = sq[s]; U.rows[s] = rows[s];
    if (F[s] > maxF) maxF = F[s];
  }
  hipLaunchKernelGGL(dp_sync_unpack_kernel, dim3((maxF + 255) / 256), dim3(256), 0, S_(stream), U);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_batchnorm_sync_apply(const float* x, float* y, int32_t rows, int32_t F, const float* gamma,
                                         const float* beta, float* moving_mean, float* moving_var, const float* mean,
                                         const float* sq_global, const float* total_rows, float* invstd_out, float eps,
                                         float momentum, int32_t relu, void* stream) {
  if (!x || !y || !gamma || !beta || !mean || !sq_global || !total_rows || !invstd_out || rows <= 0 || F <= 0 || F % 4)
    return AVSR_ERR_ARG;
  hipLaunchKernelGGL(bn_var_kernel, dim3((F + 31) / 32), dim3(1024), 0, S_(stream), sq_global, 1, mean, invstd_out, moving_mean,
                     moving_var, 1, F, eps, momentum, 0, total_rows);
  AVSR_CHECK_LAUNCH();
  const long n4 = (long)rows * F / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, S_(stream), x, mean, invstd_out, moving_mean, moving_var, gamma, beta, y,
                     n4, F, 1, eps, relu);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_batchnorm_xhat(const float* x, const float* mean, const float* invstd, float* xhat, int32_t rows,
                                   int32_t F, void* stream) {
  if (!x || !mean || !invstd || !xhat) return AVSR_ERR_ARG;
  const long n = (long)rows * F;
  hipLaunchKernelGGL(bn_xhat_kernel, dim3(blocks_for(n)), dim3(256), 0, S_(stream), x, mean, invstd, xhat, n, F);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_embed_labels(const float* emb, const int32_t* labels, int32_t go_id, float* out, int32_t* fed,
                                 int32_t B, int32_t L, int32_t E, int32_t n_steps, void* stream) {
  if (!emb || !labels || !out || n_steps <= 0 || n_steps > L) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(embed_labels_kernel, dim3(B * n_steps), dim3(E >= 256 ? 256 : ((E + 63) / 64) * 64), 0, S_(stream),
                     emb, labels, go_id, out, fed, B, L, E, n_steps);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_embed_grad(const float* dx, const int32_t* fed, float* demb, int32_t B, int32_t L, int32_t E,
                               int32_t V, float* scratch, int64_t scratch_floats, void* stream) {
  if (!dx || !fed || !demb || !scratch || B <= 0 || L <= 0 || E <= 0 || V <= 0) return AVSR_ERR_ARG;
  const int rows = B * L;
  const int nchunk = (rows + EG_ROWS - 1) / EG_ROWS;
  if ((long)nchunk * V * E > scratch_floats || nchunk > 65535) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(embed_grad_partial_kernel, dim3(V, nchunk), dim3(E >= 256 ? 256 : ((E + 63) / 64) * 64), 0, S_(stream), dx, fed,
                     scratch, rows, E, V);
  AVSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_final_kernel, dim3((V * E + 31) / 32), dim3(1024), 0, S_(stream), scratch, (long)V * E, nchunk, demb, V * E, 1.0f, 0.0f);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_dropout_rows(const avsr_mat* x, const avsr_mat* y, int32_t rows, int32_t cols, const int32_t* seed,
                                 int32_t stream_id, float keep, int32_t idx_width, int32_t idx_coff, int32_t accumulate,
                                 void* stream) {
  if (!x || !y || !x->ptr || !y->ptr || rows <= 0 || cols <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(dropout_rows_kernel, dim3(blocks_for((long)rows * cols)), dim3(256), 0, S_(stream), x->ptr, (long)x->ld,
                     x->T, (long)x->ldo, y->ptr, (long)y->ld, y->T, (long)y->ldo, rows, cols, seed, (uint32_t)stream_id, keep,
                     idx_width, idx_coff, accumulate);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_seq_loss(float* logits, const int32_t* labels, const int32_t* labels_len, float* denom,
                             int32_t compute_denom, float* row_loss, float* dlogits, int32_t B, int32_t L, int32_t V,
                             void* stream) {
  return avsr_seq_loss_fun(logits, labels, labels_len, denom, compute_denom, row_loss, dlogits, B, L, V, 0, 0.f, stream);
}

extern "C" int avsr_seq_loss_fun(float* logits, const int32_t* labels, const int32_t* labels_len, float* denom,
                                 int32_t compute_denom, float* row_loss, float* dlogits, int32_t B, int32_t L, int32_t V,
                                 int32_t loss_fun, float label_smoothing, void* stream) {
  if (!logits || !labels || !labels_len || !denom || !row_loss || loss_fun < 0 || loss_fun > 3) return AVSR_ERR_ARG;
  if (compute_denom) {
    if (loss_fun == 1) hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, S_(stream), denom, (float)B * (float)L);
    else hipLaunchKernelGGL(seqlen_sum_kernel, dim3(1), dim3(256), 0, S_(stream), labels_len, B, L, denom);
    AVSR_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(seq_loss_kernel, dim3((B * L + 127) / 128), dim3(128), 0, S_(stream), logits, labels, labels_len,
                     denom, row_loss, dlogits, B, L, V, loss_fun, label_smoothing);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

static inline avsr::HwView hwv(const avsr_mat* m) { return avsr::HwView{m->ptr, (long)m->ld, m->T, (long)m->ldo}; }

extern "C" int avsr_highway_fwd(const avsr_mat* x, const avsr_mat* h, const avsr_mat* carry_pre, const avsr_mat* y, const int32_t* len,
                                int32_t B, int32_t T, int32_t H, void* stream) {
  if (!x || !h || !carry_pre || !y || !x->ptr || !h->ptr || !carry_pre->ptr || !y->ptr || B <= 0 || T <= 0 || H <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(highway_fwd_kernel, dim3(blocks_for((long)B * T * H)), dim3(256), 0, S_(stream), hwv(x), hwv(h), hwv(carry_pre), hwv(y),
                     len, B * T, H, T);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_highway_bwd(const avsr_mat* x, const avsr_mat* h, const avsr_mat* carry_pre, const avsr_mat* dy, const avsr_mat* dh,
                                const avsr_mat* dcarry_pre, const avsr_mat* dx, const int32_t* len, int32_t B, int32_t T, int32_t H,
                                int32_t accumulate_dx, void* stream) {
  if (!x || !h || !carry_pre || !dy || !dh || !dcarry_pre || !dx || !x->ptr || !h->ptr || !carry_pre->ptr || !dy->ptr || !dh->ptr ||
      !dcarry_pre->ptr || !dx->ptr || B <= 0 || T <= 0 || H <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(highway_bwd_kernel, dim3(blocks_for((long)B * T * H)), dim3(256), 0, S_(stream), hwv(x), hwv(h), hwv(carry_pre), hwv(dy),
                     hwv(dh), hwv(dcarry_pre), hwv(dx), len, B * T, H, T, accumulate_dx);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_copy_words(void* dst, const void* src, int64_t n_words, void* stream) {
  if (!dst || !src || n_words < 0) return AVSR_ERR_ARG;
  return avsr::dev_copy(dst, src, (size_t)n_words * 4, S_(stream)) == hipSuccess ? AVSR_OK : AVSR_ERR_HIP;
}

extern "C" int avsr_zero_words(void* dst, int64_t n_words, void* stream) {
  if (!dst || n_words < 0) return AVSR_ERR_ARG;
  return avsr::dev_zero(dst, (size_t)n_words * 4, S_(stream)) == hipSuccess ? AVSR_OK : AVSR_ERR_HIP;
}

extern "C" int avsr_seq_loss_per_utterance(const float* row_loss, const int32_t* labels_len, const float* denom, float* out,
                                           int32_t B, int32_t L, void* stream) {
  if (!row_loss || !labels_len || !denom || !out || B <= 0 || L <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(seq_avg_kernel, dim3((B + 63) / 64), dim3(64), 0, S_(stream), row_loss, labels_len, denom, out, B, L);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_au_loss(const float* z, const float* aus, const int32_t* len, float* row_loss, float* dz, int32_t B,
                            int32_t T, float weight, void* stream) {
  return avsr_au_loss_dp(z, aus, len, row_loss, dz, B, T, weight, nullptr, stream);
}

extern "C" int avsr_au_loss_dp(const float* z, const float* aus, const int32_t* len, float* row_loss, float* dz, int32_t B,
                               int32_t T, float weight, const float* total_count, void* stream) {
  if (!z || !aus || !len || !row_loss) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(au_loss_kernel, dim3(blocks_for((long)B * T, 256, 256)), dim3(256), 0, S_(stream), z, aus, len,
                     row_loss, dz, B, T, weight, total_count);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_normed_v(const float* v, const float* g, float* vn, int32_t H, void* stream) {
  if (!v || !g || !vn) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(normed_v_kernel, dim3(1), dim3(256), 0, S_(stream), v, g, vn, H);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}
extern "C" int avsr_normed_v_bwd(const float* v, const float* g, const float* dvn, float* dv, float* dg, int32_t H,
                                 void* stream) {
  if (!v || !g || !dvn || !dv || !dg) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(normed_v_bwd_kernel, dim3(1), dim3(256), 0, S_(stream), v, g, dvn, dv, dg, H);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_reduce_scalar(const float* part, int32_t n, float* out, int32_t do_sqrt, int32_t accumulate,
                                  float scale, void* stream) {
  if (!part || !out || n <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(reduce_scalar_kernel, dim3(1), dim3(256), 0, S_(stream), part, n, out, do_sqrt, accumulate, scale);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_l2_regularise(const int64_t* seg_off, const int64_t* seg_n, int32_t nseg, const float* params,
                                  float* grads, float l2, float* loss_accum, float* scratch, void* stream) {
  if (nseg <= 0) return AVSR_OK;
  if (!seg_off || !seg_n || !params || !grads || !scratch || nseg > AVSR_MAX_SEGMENTS) return AVSR_ERR_ARG;
  SegLaunch S;
  S.nseg = nseg;
  for (int i = 0; i < nseg; ++i) S.seg[i] = Seg{(long)seg_off[i], (long)seg_n[i]};
  const int gx = 64;
  hipLaunchKernelGGL(l2_grad_kernel, dim3(gx, nseg), dim3(256), 0, S_(stream), S, params, grads, l2, scratch);
  AVSR_CHECK_LAUNCH();
  if (loss_accum) {
    hipLaunchKernelGGL(reduce_scalar_kernel, dim3(1), dim3(256), 0, S_(stream), scratch, gx * nseg, loss_accum, 0, 1, 1.0f);
    AVSR_CHECK_LAUNCH();
  }
  return AVSR_OK;
}

extern "C" int avsr_global_norm(const float* grads, int64_t n, float grad_scale, float* norm_out, float* scratch,
                                void* stream) {
  if (!grads || !norm_out || !scratch || n <= 0) return AVSR_ERR_ARG;
  const int gx = blocks_for(n, 1024, 1024);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(gx), dim3(256), 0, S_(stream), grads, (long)n, scratch);
  AVSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(reduce_scalar_kernel, dim3(1), dim3(256), 0, S_(stream), scratch, gx, norm_out, 1, 0, grad_scale);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_adam_step(float* params, float* grads, float* m, float* v, int64_t n, const float* global_norm,
                              int32_t* step, float lr, int32_t warmup_steps, float clip_norm, float grad_scale,
                              void* stream) {
  return avsr_adam_step_decay(params, grads, m, v, n, global_norm, step, lr, warmup_steps, 0, clip_norm, grad_scale, stream);
}

extern "C" int avsr_adam_step_decay(float* params, float* grads, float* m, float* v, int64_t n, const float* global_norm,
                                    int32_t* step, float lr, int32_t warmup_steps, int32_t first_decay_steps,
                                    float clip_norm, float grad_scale, void* stream) {
  return avsr_optimiser_step(params, grads, m, v, n, global_norm, step, lr, warmup_steps, first_decay_steps, clip_norm, grad_scale, 0, 0.f,
                             stream);
}

extern "C" int avsr_optimiser_step(float* params, float* grads, float* m, float* v, int64_t n, const float* global_norm,
                                   int32_t* step, float lr, int32_t warmup_steps, int32_t first_decay_steps, float clip_norm,
                                   float grad_scale, int32_t optimiser, float weight_decay, void* stream) {
  if (!params || !grads || !m || !v || !step || n <= 0 || first_decay_steps < 0 || optimiser < 0 || optimiser > 3) return AVSR_ERR_ARG;
  if (clip_norm > 0.f && !global_norm) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n, 1024, 2048)), dim3(256), 0, S_(stream), params, grads, m, v, (long)n,
                     global_norm, step, lr, warmup_steps, clip_norm, 0.9f, 0.999f, 1e-8f, grad_scale, first_decay_steps, optimiser, weight_decay);
  AVSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, S_(stream), step);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}
