// Lip-crop CNN front-end pieces (video.resnet_cnn, avsr/video.py:143-195): NHWC im2col / col2im around the fp32 MFMA GEMM,
// batch-norm backward (optionally through the ReLU that follows it), ReLU and residual-add helpers.
//
// Convolutions are lowered to GEMMs: col[(n,ho,wo)][(i,j,c)] = x[n, ho*s - pad_t + i, wo*s - pad_l + j, c] (zero outside)
// times the TF kernel [kh, kw, cin, cout] read as a [kh*kw*cin, cout] matrix (HWIO is already that matrix, row-major).
// The data gradient is the transposed GEMM followed by the gather-form col2im below (no atomics, deterministic).
#include "common.h"
#include "avsr_hip.h"

namespace avsr {

__global__ void im2col_kernel(const float* x, float* col, int N, int H, int W, int C, int kh, int kw, int s, int pt, int pl,
                              int Ho, int Wo) {
  const long K = (long)kh * kw * C;
  const long total = (long)N * Ho * Wo * K;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    long r = idx / C;
    const int j = (int)(r % kw); r /= kw;
    const int i = (int)(r % kh); r /= kh;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const int h = ho * s - pt + i, w = wo * s - pl + j;
    col[idx] = (h >= 0 && h < H && w >= 0 && w < W) ? x[(((long)n * H + h) * W + w) * C + c] : 0.f;
  }
}

// dx[n,h,w,c] = sum over kernel taps (i,j) of dcol[(n,ho,wo)][(i,j,c)] with ho*s - pt + i == h, wo*s - pl + j == w
__global__ void col2im_kernel(const float* dcol, float* dx, int N, int H, int W, int C, int kh, int kw, int s, int pt, int pl,
                              int Ho, int Wo, float beta) {
  const long K = (long)kh * kw * C;
  const long total = (long)N * H * W * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    long r = idx / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    float acc = 0.f;
    for (int i = 0; i < kh; ++i) {
      const int hn = h + pt - i;
      if (hn < 0 || hn % s) continue;
      const int ho = hn / s;
      if (ho >= Ho) continue;
      for (int j = 0; j < kw; ++j) {
        const int wn = w + pl - j;
        if (wn < 0 || wn % s) continue;
        const int wo = wn / s;
        if (wo >= Wo) continue;
        acc += dcol[(((long)n * Ho + ho) * Wo + wo) * K + ((long)i * kw + j) * C + c];
      }
    }
    dx[idx] = beta != 0.f ? acc + beta * dx[idx] : acc;
  }
}

// ---- batch-norm backward (training statistics), optionally through a following ReLU -------------------------------
// dy' = dy * [bn(x) > 0] (relu) ;  d beta = sum dy' ;  d gamma = sum dy' * xhat ;
// dx = gamma * invstd * (dy' - (sum dy' + xhat * sum dy' xhat) / rows)
__global__ void bn_bwd_partial_kernel(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean,
                                      const float* invstd, float* part, int rows, int F, int rows_per_blk, int relu) {
  __shared__ float red[512];
  const int G = F < 256 ? 256 / F : 1;
  const int r0 = blockIdx.x * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  float* prow = part + (long)blockIdx.x * 2 * F;
  for (int base = 0; base < (G > 1 ? 1 : F); base += blockDim.x) {      // G > 1: a single pass (G*F <= 256)
    const int idx = base + threadIdx.x;
    const bool on = G > 1 ? idx < G * F : idx < F;
    const int f = G > 1 ? idx % F : idx, g = G > 1 ? idx / F : 0;
    float s1 = 0.f, s2 = 0.f;
    if (on) {
      const float m = mean[f], is = invstd[f], ga = gamma[f], be = beta[f];
      for (int r = r0 + g; r < r1; r += G) {
        const float xh = (x[(long)r * F + f] - m) * is;
        float d = dy[(long)r * F + f];
        if (relu && !(xh * ga + be > 0.f)) d = 0.f;
        s1 += d;
        s2 += d * xh;
      }
    }
    if (G > 1) {                                     // sub-groups combined through LDS in group order: one partial row per block
      if (on) { red[idx] = s1; red[256 + idx] = s2; }
      __syncthreads();
      if (idx < F) {
        float t1 = 0.f, t2 = 0.f;
        for (int gg = 0; gg < G; ++gg) { t1 += red[gg * F + idx]; t2 += red[256 + gg * F + idx]; }
        prow[idx] = t1; prow[F + idx] = t2;
      }
    } else if (on) { prow[f] = s1; prow[F + f] = s2; }
  }
}

__global__ void bn_bwd_apply_kernel(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean,
                                    const float* invstd, const float* sum1, const float* sum2, float* dx, long n, int rows, int F, int relu,
                                    float dx_beta) {
  const float inv_rows = 1.0f / (float)rows;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    const float is = invstd[f], ga = gamma[f];
    const float xh = (x[i] - mean[f]) * is;
    float d = dy[i];
    if (relu && !(xh * ga + beta[f] > 0.f)) d = 0.f;
    const float v = ga * is * (d - (sum1[f] + xh * sum2[f]) * inv_rows);
    dx[i] = dx_beta != 0.f ? v + dx_beta * dx[i] : v;
  }
}

// 16-byte versions for F | 1024 (every channel count of the lip CNN): a thread owns FOUR fixed channels -- the grid stride (1024 floats
// per block) is a multiple of F -- so the per-channel constants are loaded once and there is no per-element modulo.
// (The scalar kernels above spent a 64-bit modulo and six table loads per element: 4 TB/s on 600 MB maps.)
__global__ __launch_bounds__(256) void bn_bwd_partial4_kernel(const float* x, const float* dy, const float* gamma, const float* beta,
                                                               const float* mean, const float* invstd, float* part, long n4, int F, int relu) {
  __shared__ f32x4 red[2][256];
  const int f = (int)((threadIdx.x * 4) % F);
  const f32x4 m = ld4(mean + f), is = ld4(invstd + f), ga = ld4(gamma + f), be = ld4(beta + f);
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 xv = ld4(x + i * 4), dv = ld4(dy + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xv[e] - m[e]) * is[e];
      float d = dv[e];
      if (relu && !(xh * ga[e] + be[e] > 0.f)) d = 0.f;
      s1[e] += d;
      s2[e] += d * xh;
    }
  }
  red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
  __syncthreads();
  const int tpf = F >> 2;                               // threads per channel period
  if ((int)threadIdx.x < tpf) {                         // threads t, t + tpf, t + 2 tpf, ... own the same four channels
    f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = t1;
    for (int k = threadIdx.x; k < 256; k += tpf) { t1 += red[0][k]; t2 += red[1][k]; }
    st4(part + (long)blockIdx.x * 2 * F + 4 * threadIdx.x, t1);
    st4(part + (long)blockIdx.x * 2 * F + F + 4 * threadIdx.x, t2);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const float* x, const float* dy, const float* gamma, const float* beta,
                                                             const float* mean, const float* invstd, const float* sum1, const float* sum2, float* dx,
                                                             long n4, int rows, int F, int relu, float dx_beta) {
  const int f = (int)((threadIdx.x * 4) % F);
  const float inv_rows = 1.0f / (float)rows;
  const f32x4 m = ld4(mean + f), is = ld4(invstd + f), ga = ld4(gamma + f), be = ld4(beta + f);
  f32x4 c1 = ld4(sum1 + f), c2 = ld4(sum2 + f);
#pragma unroll
  for (int e = 0; e < 4; ++e) { c1[e] *= inv_rows; c2[e] *= inv_rows; }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 xv = ld4(x + i * 4), dv = ld4(dy + i * 4);
    f32x4 o;
    if (dx_beta != 0.f) o = ld4(dx + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xv[e] - m[e]) * is[e];
      float d = dv[e];
      if (relu && !(xh * ga[e] + be[e] > 0.f)) d = 0.f;
      const float v = ga[e] * is[e] * (d - (c1[e] + xh * c2[e]));
      o[e] = dx_beta != 0.f ? v + dx_beta * o[e] : v;
    }
    st4(dx + i * 4, o);
  }
}

__global__ void relu_kernel(const float* x, float* y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = fmaxf(x[i], 0.f);
}
// dx = dy * [y > 0]
__global__ void relu_bwd_kernel(const float* y, const float* dy, float* dx, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
// tf.nn.selu: scale * (x > 0 ? x : alpha * (exp(x) - 1))   (input Dense layers, encoder.py:148-171)
#define SELU_SCALE 1.0507009873554805f
#define SELU_ALPHA 1.6732632423543772f
__global__ void selu_kernel(const float* z, float* y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = z[i];
    y[i] = SELU_SCALE * (x > 0.f ? x : SELU_ALPHA * (expf(x) - 1.0f));
  }
}
// dz = dy * selu'(z)
__global__ void selu_bwd_kernel(const float* z, const float* dy, float* dz, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = z[i];
    dz[i] = dy[i] * SELU_SCALE * (x > 0.f ? 1.0f : SELU_ALPHA * expf(x));
  }
}
// out = a + b
__global__ void add_kernel(const float* a, const float* b, float* out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}

static inline int blocks_for_n(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace avsr

// out[f] = alpha * sum_i part[i][f] + beta * out[f]  (elementwise.hip)
int avsr_colsum_final_launch(const float* part, int nblk, float* out, int F, float alpha, float beta, void* stream);
int avsr_colsum_final_launch_split(const float* part, long ld, int nblk, float* out, float* out2, int split, int F, float alpha, float beta,
                                   void* stream);

using namespace avsr;
#define S_(x) ((hipStream_t)(x))

extern "C" int avsr_im2col(const float* x, float* col, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw,
                           int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, void* stream) {
  if (!x || !col || N <= 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || Ho <= 0 || Wo <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(im2col_kernel, dim3(blocks_for_n((long)N * Ho * Wo * kh * kw * C)), dim3(256), 0, S_(stream), x, col, N, H, W, C, kh,
                     kw, stride, pad_t, pad_l, Ho, Wo);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_col2im(const float* dcol, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw,
                           int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, float beta, void* stream) {
  if (!dcol || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || Ho <= 0 || Wo <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(col2im_kernel, dim3(blocks_for_n((long)N * H * W * C)), dim3(256), 0, S_(stream), dcol, dx, N, H, W, C, kh, kw,
                     stride, pad_t, pad_l, Ho, Wo, beta);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_batchnorm_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean,
                                  const float* invstd, float* dx, float* dgamma, float* dbeta, int32_t rows, int32_t F, int32_t relu,
                                  float dx_beta, float* scratch, int64_t scratch_floats, void* stream) {
  if (!x || !dy || !gamma || !beta || !mean || !invstd || !scratch || rows <= 0 || F <= 0) return AVSR_ERR_ARG;
  const int maxblk = 2048;
  int rpb = rows > 64 * maxblk ? (rows + maxblk - 1) / maxblk : 64;
  int nblk = (rows + rpb - 1) / rpb;
  if ((long)nblk * 2 * F + 2 * F > scratch_floats) {
    nblk = (int)((scratch_floats - 2 * F) / (2L * F));
    if (nblk < 1) return AVSR_ERR_ARG;
    rpb = (rows + nblk - 1) / nblk;
    nblk = (rows + rpb - 1) / rpb;
  }
  float* part = scratch;
  const long n = (long)rows * F;
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  const bool vec = F >= 4 && 1024 % F == 0 && al16(x) && al16(dy) && (!dx || al16(dx)) && al16(gamma) && al16(beta) && al16(mean) && al16(invstd) &&
                   al16(scratch) && rows >= 4096;
  if (vec) {
    int vb = (int)((n / 4 + 255) / 256);
    if (vb > 1024) vb = 1024;
    if ((long)vb * 2 * F + 2 * F > scratch_floats) vb = (int)((scratch_floats - 2 * F) / (2L * F));
    if (vb < 1) return AVSR_ERR_ARG;
    nblk = vb;
  }
  float* sums = scratch + (long)nblk * 2 * F;          // [2F]: sum dy' | sum dy' xhat
  if (vec) hipLaunchKernelGGL(bn_bwd_partial4_kernel, dim3(nblk), dim3(256), 0, S_(stream), x, dy, gamma, beta, mean, invstd, part, n / 4, F, relu);
  else hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nblk), dim3(256), 0, S_(stream), x, dy, gamma, beta, mean, invstd, part, rows, F, rpb, relu);
  AVSR_CHECK_LAUNCH();
  // the reduced sums ARE d beta | d gamma: one reduction launch writes them where the caller wants them and the apply pass reads
  // them from there (16-byte aligned destinations for the vector kernel; else through the scratch + two copies)
  const bool direct = dbeta && dgamma && al16(dbeta) && al16(dgamma);
  float* const s1 = direct ? dbeta : sums;
  float* const s2 = direct ? dgamma : sums + F;
  { const int rc = avsr_colsum_final_launch_split(part, 2L * F, nblk, s1, s2, F, 2 * F, 1.0f, 0.0f, stream); if (rc) return rc; }
  if (dx) {
    if (vec) {
      int ab = (int)((n / 4 + 255) / 256);
      if (ab > 4096) ab = 4096;
      hipLaunchKernelGGL(bn_bwd_apply4_kernel, dim3(ab), dim3(256), 0, S_(stream), x, dy, gamma, beta, mean, invstd, s1, s2, dx, n / 4, rows, F, relu,
                         dx_beta);
    } else {
      hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for_n(n)), dim3(256), 0, S_(stream), x, dy, gamma, beta, mean, invstd, s1, s2, dx, n, rows,
                         F, relu, dx_beta);
    }
    AVSR_CHECK_LAUNCH();
  }
  if (!direct) {
    if (dbeta && avsr::dev_copy(dbeta, sums, sizeof(float) * F, S_(stream)) != hipSuccess) return AVSR_ERR_HIP;
    if (dgamma && avsr::dev_copy(dgamma, sums + F, sizeof(float) * F, S_(stream)) != hipSuccess) return AVSR_ERR_HIP;
  }
  return AVSR_OK;
}

extern "C" int avsr_relu(const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(relu_kernel, dim3(blocks_for_n(n)), dim3(256), 0, S_(stream), x, y, (long)n);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}
extern "C" int avsr_relu_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream) {
  if (!y || !dy || !dx || n <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks_for_n(n)), dim3(256), 0, S_(stream), y, dy, dx, (long)n);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}
extern "C" int avsr_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
  if (!a || !b || !out || n <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(add_kernel, dim3(blocks_for_n(n)), dim3(256), 0, S_(stream), a, b, out, (long)n);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_selu(const float* z, float* y, int64_t n, void* stream) {
  if (!z || !y || n <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(selu_kernel, dim3(blocks_for_n(n)), dim3(256), 0, S_(stream), z, y, (long)n);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}
extern "C" int avsr_selu_bwd(const float* z, const float* dy, float* dz, int64_t n, void* stream) {
  if (!z || !dy || !dz || n <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(selu_bwd_kernel, dim3(blocks_for_n(n)), dim3(256), 0, S_(stream), z, dy, dz, (long)n);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}


// ---------------------------------------------------------------------------------------------
// tf.contrib.layers.instance_norm on [B, T, F] encoder inputs (avsr/encoder.py:51-55): statistics over the time axis per
// (utterance, feature) -- zero padding included -- epsilon 1e-6, learned gamma / beta per feature.
// One workgroup per (utterance, 64 features): 256 threads = 64 features x 4 time lanes.
namespace avsr {

__global__ __launch_bounds__(256) void instnorm_fwd_kernel(const float* x, float* y, const float* gamma, const float* beta, float* mean_o,
                                                           float* invstd_o, int T, int F, float eps) {
  __shared__ float red[4][64];
  const int b = blockIdx.x, f = blockIdx.y * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  const bool ok = f < F;
  const float* xb = x + (long)b * T * F + f;
  float s = 0.f;
  if (ok) for (int t = g; t < T; t += 4) s += xb[(long)t * F];
  red[g][threadIdx.x & 63] = s;
  __syncthreads();
  const float mean = (red[0][threadIdx.x & 63] + red[1][threadIdx.x & 63] + red[2][threadIdx.x & 63] + red[3][threadIdx.x & 63]) / (float)T;
  __syncthreads();
  float q = 0.f;
  if (ok) for (int t = g; t < T; t += 4) { const float d = xb[(long)t * F] - mean; q += d * d; }
  red[g][threadIdx.x & 63] = q;
  __syncthreads();
  const float var = (red[0][threadIdx.x & 63] + red[1][threadIdx.x & 63] + red[2][threadIdx.x & 63] + red[3][threadIdx.x & 63]) / (float)T;
  const float istd = rsqrtf(var + eps);
  if (!ok) return;
  if (g == 0) { mean_o[(long)b * F + f] = mean; invstd_o[(long)b * F + f] = istd; }
  const float ga = gamma[f] * istd, be = beta[f];
  float* yb = y + (long)b * T * F + f;
  for (int t = g; t < T; t += 4) yb[(long)t * F] = (xb[(long)t * F] - mean) * ga + be;
}

// dx may alias dy.  dgamma_part / dbeta_part [B][F]: per-utterance sums (the caller column-sums them over B).
__global__ __launch_bounds__(256) void instnorm_bwd_kernel(const float* x, const float* dy, const float* gamma, const float* mean_i,
                                                           const float* invstd_i, float* dx, float* dgamma_part, float* dbeta_part,
                                                           int T, int F) {
  __shared__ float red[2][4][64];
  const int b = blockIdx.x, fl = threadIdx.x & 63, f = blockIdx.y * 64 + fl, g = threadIdx.x >> 6;
  const bool ok = f < F;
  const long base = (long)b * T * F + f;
  const float mean = ok ? mean_i[(long)b * F + f] : 0.f, istd = ok ? invstd_i[(long)b * F + f] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  if (ok)
    for (int t = g; t < T; t += 4) {
      const float d = dy[base + (long)t * F];
      s1 += d;
      s2 += d * (x[base + (long)t * F] - mean) * istd;
    }
  red[0][g][fl] = s1; red[1][g][fl] = s2;
  __syncthreads();
  s1 = red[0][0][fl] + red[0][1][fl] + red[0][2][fl] + red[0][3][fl];
  s2 = red[1][0][fl] + red[1][1][fl] + red[1][2][fl] + red[1][3][fl];
  if (!ok) return;
  if (g == 0) { dgamma_part[(long)b * F + f] = s2; dbeta_part[(long)b * F + f] = s1; }
  const float k = gamma[f] * istd, m1 = s1 / (float)T, m2 = s2 / (float)T;
  for (int t = g; t < T; t += 4) {
    const float xh = (x[base + (long)t * F] - mean) * istd;
    dx[base + (long)t * F] = k * (dy[base + (long)t * F] - m1 - xh * m2);
  }
}

}  // namespace avsr

extern "C" int avsr_instnorm_fwd(const float* x, float* y, int32_t B, int32_t T, int32_t F, const float* gamma, const float* beta,
                                 float* mean_out, float* invstd_out, float eps, void* stream) {
  if (!x || !y || !gamma || !beta || !mean_out || !invstd_out || B <= 0 || T <= 0 || F <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(avsr::instnorm_fwd_kernel, dim3(B, (F + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, y, gamma, beta, mean_out,
                     invstd_out, T, F, eps);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_instnorm_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* invstd, float* dx,
                                 float* dgamma_part, float* dbeta_part, int32_t B, int32_t T, int32_t F, void* stream) {
  if (!x || !dy || !gamma || !mean || !invstd || !dx || !dgamma_part || !dbeta_part || B <= 0 || T <= 0 || F <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(avsr::instnorm_bwd_kernel, dim3(B, (F + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, dy, gamma, mean, invstd, dx,
                     dgamma_part, dbeta_part, T, F);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}
