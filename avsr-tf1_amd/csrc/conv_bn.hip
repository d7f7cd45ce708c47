// Batch-norm kernels around the frame-resident convolutions (split out of conv_mfma.hip in round 6): finalisation of the statistics the
// convolution epilogues emit (forward), of the fused backward sums, the affine apply passes, and the fp64 forms of the data-parallel
// synchronised batch norms (avsr/video.py:4-14 batch_norm_relu; DataParallelTrainer(sync_cnn_bn=True)).
#include "conv_mfma.h"

using namespace avsr;

// finalise batch-norm statistics from the per-workgroup partial sums the convolution epilogue wrote: part [nparts][2*C] (sum | sum of
// squares), count = rows per channel.  fp64 merge; the moving averages take the Bessel-corrected variance (fused rank-4 path).
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* part, int nparts, int C, double count, float eps, float momentum,
                                                          float* mean, float* invstd, float* mov_mean, float* mov_var, const float* gamma,
                                                          const float* beta, float* scale, float* shift) {
  // one workgroup per 16 channels: 16 lanes read 16 consecutive channels of a partial row (64 B segments), 64 row groups stride the rows
  // (1024 threads: the 512 partial rows are eight loads per thread -- the kernel is a latency chain, not a bandwidth one)
  __shared__ double red[2][64][17];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  double s = 0.0, s2 = 0.0;
  if (c < C)
    for (int p = rg; p < nparts; p += 64) { s += (double)part[(long)p * 2 * C + c]; s2 += (double)part[(long)p * 2 * C + C + c]; }
  red[0][rg][cl] = s; red[1][rg][cl] = s2;
  __syncthreads();
  if (threadIdx.x >= 16 || c >= C) return;
  s = 0.0; s2 = 0.0;
  for (int r = 0; r < 64; ++r) { s += red[0][r][cl]; s2 += red[1][r][cl]; }
  const double m = s / count;
  double var = s2 / count - m * m;
  if (var < 0.0) var = 0.0;
  const float is = rsqrtf((float)var + eps);
  mean[c] = (float)m;
  invstd[c] = is;
  if (scale) {                                          // y = x * scale + shift  ==  (x - mean) * invstd * gamma + beta
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - (float)m * sc;
  }
  if (mov_mean) {
    const float unbiased = (float)(var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
    mov_mean[c] = momentum * mov_mean[c] + (1.f - momentum) * (float)m;
    mov_var[c] = momentum * mov_var[c] + (1.f - momentum) * unbiased;
  }
}

// evaluation graph (training=False, video.py:8-12): scale / shift of the loader-applied batch norm from the MOVING statistics
__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* mov_mean, const float* mov_var, float eps, float* scale,
                                      float* shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] * rsqrtf(mov_var[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - mov_mean[c] * sc;
}
extern "C" int avsr_bn_eval_affine(const float* gamma, const float* beta, const float* mov_mean, const float* mov_var, float eps, float* scale,
                                   float* shift, int32_t C, void* stream) {
  if (!gamma || !beta || !mov_mean || !mov_var || !scale || !shift || C <= 0) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 63) / 64), dim3(64), 0, S_(stream), gamma, beta, mov_mean, mov_var, eps, scale, shift, C);
  if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
  return AVSR_OK;
}

extern "C" int avsr_bn_finalize(const float* part, int32_t nparts, int32_t C, int64_t count, float eps, float momentum, float* mean,
                                float* invstd, float* mov_mean, float* mov_var, const float* gamma, const float* beta, float* scale,
                                float* shift, void* stream) {
  if (!part || nparts <= 0 || C <= 0 || count <= 0 || !mean || !invstd || (scale && (!gamma || !beta || !shift))) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, S_(stream), part, nparts, C, (double)count, eps, momentum, mean, invstd,
                     mov_mean, mov_var, gamma, beta, scale, shift);
  if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
  return AVSR_OK;
}

// Batch-norm backward, stage 2 (after avsr_conv_bwd_data_bn wrote dz and the partial sums [nparts][2*C] = (sum dz | sum dz*x)):
//   d beta (+)= sum dz;  d gamma (+)= invstd * (sum dz*x - mean * sum dz);
//   k[0..C) = gamma*invstd, k[C..2C) = -gamma*invstd^2 * b, k[2C..3C) = -gamma*invstd*a + gamma*invstd^2 * b * mean
// with a = sum dz / count, b = invstd * (sum dz*x - mean * sum dz) / count, so that dx = k1*dz + k2*x + k3 (avsr_bn_bwd_apply) is
// gamma*invstd * (dz - a - xhat*b).  fp64 merge of the partials.
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* part, int nparts, int C, double count, const float* mean,
                                                              const float* invstd, const float* gamma, float* dgamma, float* dbeta,
                                                              float grad_beta, float* k) {
  __shared__ double red[2][64][17];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  double s = 0.0, s2 = 0.0;
  if (c < C)
    for (int p = rg; p < nparts; p += 64) { s += (double)part[(long)p * 2 * C + c]; s2 += (double)part[(long)p * 2 * C + C + c]; }
  red[0][rg][cl] = s; red[1][rg][cl] = s2;
  __syncthreads();
  if (threadIdx.x >= 16 || c >= C) return;
  s = 0.0; s2 = 0.0;
  for (int r = 0; r < 64; ++r) { s += red[0][r][cl]; s2 += red[1][r][cl]; }
  const double m = mean[c], is = invstd[c], g = gamma[c];
  const double sxh = is * (s2 - m * s);                  // sum dz * xhat
  if (dbeta) dbeta[c] = (grad_beta != 0.f ? grad_beta * dbeta[c] : 0.f) + (float)s;
  if (dgamma) dgamma[c] = (grad_beta != 0.f ? grad_beta * dgamma[c] : 0.f) + (float)sxh;
  const double a = s / count, b = sxh / count;
  k[c] = (float)(g * is);
  k[C + c] = (float)(-g * is * is * b);
  k[2 * C + c] = (float)(-g * is * a + g * is * is * b * m);
}

extern "C" int avsr_bn_bwd_finalize(const float* part, int32_t nparts, int32_t C, int64_t count, const float* mean, const float* invstd,
                                    const float* gamma, float* dgamma, float* dbeta, float grad_beta, float* k, void* stream) {
  if (!part || nparts <= 0 || C <= 0 || count <= 0 || !mean || !invstd || !gamma || !k) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, S_(stream), part, nparts, C, (double)count, mean, invstd, gamma,
                     dgamma, dbeta, grad_beta, k);
  if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
  return AVSR_OK;
}

// dx = beta*dx + k1[c]*dz + k2[c]*x + k3[c] over [rows][C] maps, C % 4 == 0 (16-byte accesses, one channel quad per lane)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ k,
                                                           float* __restrict__ dx, long n4, int C4, int C, float beta) {
  const long stride = (long)gridDim.x * 256;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n4; idx += stride) {
    const int c = (int)(idx % C4) * 4;
    const f32x4 k1 = ld4(k + c), k2 = ld4(k + C + c), k3 = ld4(k + 2 * C + c);
    const f32x4 a = ld4(dz + idx * 4), b = ld4(x + idx * 4);
    f32x4 v = k1 * a + k2 * b + k3;
    if (beta != 0.f) v += beta * ld4(dx + idx * 4);
    st4(dx + idx * 4, v);
  }
}

extern "C" int avsr_bn_bwd_apply(const float* dz, const float* x, const float* k, float* dx, int64_t rows, int32_t C, float beta, void* stream) {
  if (!dz || !x || !k || !dx || rows <= 0 || C <= 0 || C % 4) return AVSR_ERR_ARG;
  const long n4 = rows * (C / 4);
  long blocks = (n4 + 256 * 8 - 1) / (256 * 8);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((int)blocks), dim3(256), 0, S_(stream), dz, x, k, dx, n4, C / 4, C, beta);
  if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
  return AVSR_OK;
}

// Batch-norm backward, stage 1 on its own (the form avsr_conv_bwd_data_bn fuses into a single-launch data gradient's epilogue), for a
// batch norm whose output gradient was assembled by several launches (the per-class 3x3/2 data gradient of a wide layer):
//   dz = dy * [relu(scale*x + shift) > 0]   (or [y > 0] when the batch-norm output map was written),  part [nparts][2C] = (sum dz | sum dz*x)
// dz may alias dy.  C % 4 == 0, C <= 1024; *nparts <= 512 blocks, each over a contiguous run of rows.
__global__ __launch_bounds__(256) void bn_bwd_stage1_kernel(const float* dy, const float* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ y, float* dz, long rows,
                                                            int C, long rows_per_block, float* __restrict__ part) {
  __shared__ float red[256][9];
  const int C4 = C / 4, RL = 256 / C4, q = threadIdx.x % C4, rl = threadIdx.x / C4;
  const long r0 = blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, sx = s;
  f32x4 sc = s, sh = s;
  if (scale && rl < RL) { sc = ld4(scale + 4 * q); sh = ld4(shift + 4 * q); }
  if (rl < RL)
    for (long r = r0 + rl; r < r1; r += RL) {
      const long o = r * C + 4 * q;
      const f32x4 g = ld4(dy + o), xv = ld4(x + o);
      f32x4 v;
      if (scale) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(xv[e], sc[e], sh[e]) > 0.f ? g[e] : 0.f;
      } else {
        const f32x4 yv = ld4(y + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = yv[e] > 0.f ? g[e] : 0.f;
      }
      st4(dz + o, v);
      s += v; sx += v * xv;
    }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[threadIdx.x][e] = s[e]; red[threadIdx.x][4 + e] = sx[e]; }
  __syncthreads();
  if (threadIdx.x >= C4) return;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < RL; ++r)
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += red[r * C4 + q][e];
  float* p = part + (long)blockIdx.x * 2 * C;
#pragma unroll
  for (int e = 0; e < 4; ++e) { p[4 * q + e] = a[e]; p[C + 4 * q + e] = a[4 + e]; }
}
extern "C" int avsr_bn_bwd_stage1(const float* dy, const float* x, const float* scale, const float* shift, const float* y, float* dz, int64_t rows,
                                  int32_t C, float* part, int32_t* nparts, void* stream) {
  if (!dy || !x || !dz || !part || !nparts || rows <= 0 || C <= 0 || C % 4 || C > 1024 || (!scale && !y) || (scale && !shift)) return AVSR_ERR_ARG;
  const int RL = 256 / (C / 4);
  long per = (rows + 511) / 512;
  if (per < 8L * RL) per = 8L * RL;                          // at least eight passes of a block's row lanes
  const int blocks = (int)((rows + per - 1) / per);
  *nparts = blocks;
  hipLaunchKernelGGL(bn_bwd_stage1_kernel, dim3(blocks), dim3(256), 0, S_(stream), dy, x, scale, shift, y, dz, (long)rows, C, per, part);
  if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
  return AVSR_OK;
}

// ---- batch-norm statistics across data-parallel ranks (opt-in: DataParallelTrainer(sync_cnn_bn=True)) --------------------------------
// The partial sums a convolution epilogue wrote are merged into fp64 per-channel sums, the host all-reduces that small buffer (with the
// rank's row count behind it), and the finalisation reads the GLOBAL sums: mean / variance / moving averages / loader affine of the
// whole batch on every rank (video.py:4-14 over the global batch).  Same arithmetic as bn_finalize_kernel / bn_bwd_finalize_kernel.
__global__ __launch_bounds__(1024) void bn_partials_f64_kernel(const float* part, int nparts, int C, double* out) {
  __shared__ double red[2][64][17];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  double s = 0.0, s2 = 0.0;
  if (c < C)
    for (int p = rg; p < nparts; p += 64) { s += (double)part[(long)p * 2 * C + c]; s2 += (double)part[(long)p * 2 * C + C + c]; }
  red[0][rg][cl] = s; red[1][rg][cl] = s2;
  __syncthreads();
  if (threadIdx.x >= 16 || c >= C) return;
  s = 0.0; s2 = 0.0;
  for (int r = 0; r < 64; ++r) { s += red[0][r][cl]; s2 += red[1][r][cl]; }
  out[c] = s; out[C + c] = s2;
}
extern "C" int avsr_bn_partials_f64(const float* part, int32_t nparts, int32_t C, double* out64, void* stream) {
  if (!part || nparts <= 0 || C <= 0 || !out64) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(bn_partials_f64_kernel, dim3((C + 15) / 16), dim3(1024), 0, S_(stream), part, nparts, C, out64);
  if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
  return AVSR_OK;
}
// sums [2C + 1]: sum | sum of squares | rows per channel (all ranks)
__global__ void bn_finalize_f64_kernel(const double* sums, int C, float eps, float momentum, float* mean, float* invstd, float* mov_mean,
                                       float* mov_var, const float* gamma, const float* beta, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double count = sums[2 * C], m = sums[c] / count;
  double var = sums[C + c] / count - m * m;
  if (var < 0.0) var = 0.0;
  const float is = rsqrtf((float)var + eps);
  mean[c] = (float)m;
  invstd[c] = is;
  if (scale) {
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - (float)m * sc;
  }
  if (mov_mean) {
    const float unbiased = (float)(var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
    mov_mean[c] = momentum * mov_mean[c] + (1.f - momentum) * (float)m;
    mov_var[c] = momentum * mov_var[c] + (1.f - momentum) * unbiased;
  }
}
extern "C" int avsr_bn_finalize_f64(const double* sums, int32_t C, float eps, float momentum, float* mean, float* invstd, float* mov_mean,
                                    float* mov_var, const float* gamma, const float* beta, float* scale, float* shift, void* stream) {
  if (!sums || C <= 0 || !mean || !invstd || (scale && (!gamma || !beta || !shift))) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(bn_finalize_f64_kernel, dim3((C + 63) / 64), dim3(64), 0, S_(stream), sums, C, eps, momentum, mean, invstd, mov_mean, mov_var,
                     gamma, beta, scale, shift);
  if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
  return AVSR_OK;
}
// local [2C]: this rank's (sum dz | sum dz*x); global [2C + 1]: the all-reduced sums and the global row count.  d gamma / d beta take the
// LOCAL sums (the gradient all-reduce adds the ranks' shares), the coefficient vectors of dx = k1*dz + k2*x + k3 the GLOBAL means.
__global__ void bn_bwd_finalize_f64_kernel(const double* local, const double* global, int C, const float* mean, const float* invstd,
                                           const float* gamma, float* dgamma, float* dbeta, float grad_beta, float* k) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = mean[c], is = invstd[c], g = gamma[c], count = global[2 * C];
  const double sl = local[c], sxh_l = is * (local[C + c] - m * sl);
  if (dbeta) dbeta[c] = (grad_beta != 0.f ? grad_beta * dbeta[c] : 0.f) + (float)sl;
  if (dgamma) dgamma[c] = (grad_beta != 0.f ? grad_beta * dgamma[c] : 0.f) + (float)sxh_l;
  const double s = global[c], sxh = is * (global[C + c] - m * s);
  const double a = s / count, b = sxh / count;
  k[c] = (float)(g * is);
  k[C + c] = (float)(-g * is * is * b);
  k[2 * C + c] = (float)(-g * is * a + g * is * is * b * m);
}
extern "C" int avsr_bn_bwd_finalize_f64(const double* local, const double* global, int32_t C, const float* mean, const float* invstd,
                                        const float* gamma, float* dgamma, float* dbeta, float grad_beta, float* k, void* stream) {
  if (!local || !global || C <= 0 || !mean || !invstd || !gamma || !k) return AVSR_ERR_ARG;
  hipLaunchKernelGGL(bn_bwd_finalize_f64_kernel, dim3((C + 63) / 64), dim3(64), 0, S_(stream), local, global, C, mean, invstd, gamma, dgamma, dbeta,
                     grad_beta, k);
  if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
  return AVSR_OK;
}
