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

namespace avsr {

struct TileCtx {
  int row0, col0;
};

// ---- the shared MFMA core: returns this wave's partial 16x16 tile -------------------------------
__device__ __forceinline__ f32x4 mm16_partial(const StepTask& tk, int row0, int col0, bool save_ctx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int arow = row0 + i;
  const int wcol = col0 + i;
  const bool arow_ok = arow < tk.B;
  const bool wcol_ok = wcol < tk.N;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  int cbase = 0;
  for (int s = 0; s < tk.nsrc; ++s) {
    const StepSrc& S = tk.src[s];
    const int nch = (S.K + 15) >> 4;
    const float* wp = S.w + (long)wcol * S.ldw;
    int c = (wave - cbase) & 3;  // first chunk of this source owned by this wave
    if (S.kind == SRC_PLAIN) {
      long rb = arow;
      if (s == 0 && tk.gather && arow_ok) rb = tk.gather[arow];
      const float* ap = S.a + rb * S.sb;
#pragma unroll 4
      for (; c < nch; c += 4) {
        const int k = (c << 4) + (q << 2);
        f32x4 av = {0.f, 0.f, 0.f, 0.f}, wv = {0.f, 0.f, 0.f, 0.f};
        if (k < S.K) {
          if (arow_ok) av = ld4(ap + k);
          if (wcol_ok) wv = ld4(wp + k);
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], wv[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], wv[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], wv[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], wv[3], acc1, 0, 0, 0);
      }
    } else {
      // A row = sum_j wgt_j * slab_j[row]  (attention context from per-chunk softmax partials, or a plain sum)
      float wgt[STEP_MAX_SLAB];
      const int ns = tk.nslab;
      if (S.kind == SRC_SOFTMAX) {
        float M = -INFINITY;
#pragma unroll
        for (int j = 0; j < STEP_MAX_SLAB; ++j) {
          wgt[j] = -INFINITY;
          if (j < ns && arow_ok) wgt[j] = tk.pm[(long)j * tk.B + arow];
          M = fmaxf(M, wgt[j]);
        }
        float L = 0.f;
#pragma unroll
        for (int j = 0; j < STEP_MAX_SLAB; ++j) {
          const float e = (wgt[j] == -INFINITY) ? 0.f : expf(wgt[j] - M);
          if (j < ns && arow_ok) L += e * tk.pl[(long)j * tk.B + arow];
          wgt[j] = e;
        }
        const float inv = L > 0.f ? 1.0f / L : 0.f;
#pragma unroll
        for (int j = 0; j < STEP_MAX_SLAB; ++j) wgt[j] *= inv;
      } else {
#pragma unroll
        for (int j = 0; j < STEP_MAX_SLAB; ++j) wgt[j] = 1.0f;
      }
      const float* ap = S.a + (long)arow * S.sb;
      for (; c < nch; c += 4) {
        const int k = (c << 4) + (q << 2);
        f32x4 av = {0.f, 0.f, 0.f, 0.f}, wv = {0.f, 0.f, 0.f, 0.f};
        if (k < S.K) {
          if (arow_ok) {
#pragma unroll
            for (int j = 0; j < STEP_MAX_SLAB; ++j) {
              if (j < ns) {
                const f32x4 p = ld4(ap + (long)j * tk.slab_stride + k);
                av += wgt[j] * p;
              }
            }
            if (save_ctx && tk.ctx_save) st4(tk.ctx_save + (long)arow * tk.ctx_sb + k, av);
          }
          if (wcol_ok) wv = ld4(wp + k);
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], wv[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], wv[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], wv[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], wv[3], acc1, 0, 0, 0);
      }
    }
    cbase += nch;
  }
  return acc0 + acc1;
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return tanhf(v);
  if (act == 2) return sigmoidf_(v);
  return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void step_kernel(const StepLaunch L) {
  __shared__ __attribute__((aligned(16))) float red[4][16][16];
  const StepTask& tk = L.task[blockIdx.z];
  const int col0 = blockIdx.x * 16, row0 = blockIdx.y * 16;
  if (col0 >= tk.N || row0 >= tk.B) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  const f32x4 acc = mm16_partial(tk, row0, col0, blockIdx.x == 0);
  // C/D layout of mfma 16x16: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(lane >> 4) * 4 + r][lane & 15] = acc[r];
  __syncthreads();

  const int t = tk.t;
  if constexpr (MODE == EP_LINEAR || MODE == EP_LSTM_BWD) {
    const int r = tid >> 4, cidx = tid & 15;
    const int b = row0 + r, n = col0 + cidx;
    if (b >= tk.B || n >= tk.N) return;
    float z = red[0][r][cidx] + red[1][r][cidx] + red[2][r][cidx] + red[3][r][cidx];
    if constexpr (MODE == EP_LINEAR) {
      // p0 out (row stride s0), p1 add (row stride s1)
      if (tk.bias) z += tk.bias[n];
      if (tk.p1) z += tk.p1[(long)b * tk.s1 + n];
      z = apply_act(z, tk.act);
      if (tk.len && t >= tk.len[b]) z = 0.f;   // dynamic_rnn / impute_finished: zero output past the valid length
      tk.p0[(long)b * tk.s0 + n] = z;
      return;
    }
    // ---- EP_LSTM_BWD: n = unit.  p0 gates, p1 cs, p2 dgates record, p3 dG rolling out [B,4H],
    //      p4 dc_in, p5 dc_out, p6 dh_carry_in, p7 dh_carry_out, p8 dout (s0 batch stride, s1 time stride),
    //      bias = c_init [B,H] (state before step 0; null = zeros)
    const int H = tk.N;
    const int len = tk.len ? tk.len[b] : tk.T;
    const bool valid = t < len;
    const long bh = (long)b * H + n;
    const float dc_in = tk.p4[bh];
    const float carry = tk.p6 ? tk.p6[bh] : 0.f;
    f32x4 dg = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      const int tau = tk.reverse ? len - 1 - t : t;
      const long bt = (long)b * tk.T + tau;
      float dh = z + carry;
      if (tk.p8) dh += tk.p8[(long)b * tk.s0 + (long)tau * tk.s1 + n];
      const f32x4 g = ld4(tk.p0 + (bt * H + n) * 4);  // i, j, f, o (activated)
      const float c = tk.p1[bt * H + n];
      float cprev;
      if (t == 0) cprev = tk.bias ? tk.bias[bh] : 0.f;
      else cprev = tk.p1[(bt + (tk.reverse ? 1 : -1)) * H + n];
      const float tc = tanhf(c);
      float dc = dh * g[3] * (1.f - tc * tc) + dc_in;
      if (!(fabsf(c) < 1.0f)) dc = 0.f;            // cell_clip=1.0: no gradient through a clipped cell
      dg[3] = dh * tc * g[3] * (1.f - g[3]);
      dg[0] = dc * g[1] * g[0] * (1.f - g[0]);
      dg[1] = dc * g[0] * (1.f - g[1] * g[1]);
      dg[2] = dc * cprev * g[2] * (1.f - g[2]);
      tk.p5[bh] = dc * g[2];
      if (tk.p7) tk.p7[bh] = 0.f;
      st4(tk.p2 + (bt * H + n) * 4, dg);
    } else {
      tk.p5[bh] = dc_in;
      if (tk.p7) tk.p7[bh] = carry;
      if (t < tk.T) st4(tk.p2 + (((long)b * tk.T + t) * H + n) * 4, dg);  // padding position t: zero record
    }
    st4(tk.p3 + bh * 4, dg);
    return;
  }

  if constexpr (MODE == EP_LSTM_FWD) {
    // p0 gates/zpre [B,T,H,4], p1 cs [B,T,H], p2 seq_out (s0 batch stride, s1 time stride),
    // p3 c_in, p4 h_in, p5 c_out, p6 h_out; s2 = has_zpre
    if (tid >= 64) return;
    const int r = tid >> 2, ul = tid & 3;
    const int b = row0 + r;
    if (b >= tk.B) return;
    const int H = tk.N >> 2;
    const int u = (col0 >> 2) + ul;
    const int len = tk.len ? tk.len[b] : tk.T;
    const bool valid = t < len;
    const long bh = (long)b * H + u;
    const float cprev = tk.p3[bh], hprev = tk.p4[bh];
    if (valid) {
      const int tau = tk.reverse ? len - 1 - t : t;
      const long bt = (long)b * tk.T + tau;
      f32x4 z;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const int cc = ul * 4 + gi;
        z[gi] = red[0][r][cc] + red[1][r][cc] + red[2][r][cc] + red[3][r][cc];
      }
      if (tk.bias) z += ld4(tk.bias + u * 4);
      float* gp = tk.p0 + (bt * H + u) * 4;
      if (tk.s2) z += ld4(gp);
      f32x4 g;
      g[0] = sigmoidf_(z[0]);
      g[1] = tanhf(z[1]);
      g[2] = sigmoidf_(z[2] + 1.0f);   // forget_bias = 1.0
      g[3] = sigmoidf_(z[3]);
      float c = g[2] * cprev + g[0] * g[1];
      c = fminf(1.0f, fmaxf(-1.0f, c));  // cell_clip = 1.0 (cells.py:16)
      const float h = g[3] * tanhf(c);
      st4(gp, g);
      tk.p1[bt * H + u] = c;
      if (tk.p2) tk.p2[(long)b * tk.s0 + (long)tau * tk.s1 + u] = h;
      tk.p5[bh] = c;
      tk.p6[bh] = h;
    } else {
      tk.p5[bh] = cprev;
      tk.p6[bh] = hprev;
      if (tk.p2 && t < tk.T) tk.p2[(long)b * tk.s0 + (long)t * tk.s1 + u] = 0.f;  // zero output past len
    }
    return;
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
  dim3 grid((maxN + 15) / 16, (maxB + 15) / 16, L->ntask);
  const int mode = L->task[0].mode;
  for (int i = 1; i < L->ntask; ++i)
    if (L->task[i].mode != mode) return AVSR_ERR_ARG;   // one epilogue kind per launch
  hipStream_t s = (hipStream_t)stream;
  if (mode == EP_LSTM_FWD) {
    ProfScope ps(PROF_STEP_LSTM_FWD, s);
    hipLaunchKernelGGL(step_kernel<EP_LSTM_FWD>, grid, dim3(256), 0, s, *L);
  } else if (mode == EP_LSTM_BWD) {
    ProfScope ps(PROF_STEP_LSTM_BWD, s);
    hipLaunchKernelGGL(step_kernel<EP_LSTM_BWD>, grid, dim3(256), 0, s, *L);
  } else if (mode == EP_LINEAR) {
    ProfScope ps(PROF_STEP_LINEAR, s);
    hipLaunchKernelGGL(step_kernel<EP_LINEAR>, grid, dim3(256), 0, s, *L);
  } else {
    return AVSR_ERR_UNSUPPORTED;
  }
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}
