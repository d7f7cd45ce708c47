// Direct 3x3 convolutions for the wide, shallow layers of the lip-crop CNN (video.resnet_cnn, avsr/video.py:143-195):
// 36x36 and 18x18 maps with 3..16 channels.  As GEMMs these layers are [N*H*W, 9*cin] x [9*cin, cout<=16]: a 128x128 MFMA
// tile is >87 % padding and the im2col operand is 9x the map (1.8 GB per layer at 4800 frames), so here they run on
// the fp32 VALU instead (same 157 TF/s peak as the f32-input MFMA on gfx950), straight from the NHWC maps:
//   forward / stride-1 data gradient : one thread per output pixel, all taps x input channels, weights broadcast from LDS
//                                      (the data gradient is the same kernel on dy with the kernel flipped and transposed)
//   stride-2 data gradient           : one thread per INPUT pixel, gather over the taps whose parity matches
//   weight gradient                  : one block per group of frames; x (with a zero halo) and dy staged in LDS, one thread
//                                      per (tap, cin, cout) element walking the pixels; per-block partials are summed in block
//                                      order by the column-sum finaliser (deterministic)
// Layers with more channels (32, 64: 9x9 and 5x5 maps, few rows) stay on im2col + avsr_gemm (conv.hip).
#include "common.h"
#include "prof.h"
#include "avsr_hip.h"

namespace avsr {

// y[n,ho,wo,co] = bias[co] + sum_{i,j,ci} x[n, ho*s - pt + i, wo*s - pl + j, ci] * wk(i,j,ci,co)  (+ beta * y)
// flip = 0: wk = w[((i*3+j)*Ci + ci)*Co + co]                      (TF kernel [3,3,Ci,Co])
// flip = 1: wk = w[(((2-i)*3+(2-j))*Co + co)*Ci + ci]              (data gradient: w is the forward kernel [3,3,Co,Ci])
template <int COT>
__global__ __launch_bounds__(256) void conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ y, int N, int H, int W, int Ci, int Co, int s, int pt, int pl,
                                                      int Ho, int Wo, int flip, float beta) {
  extern __shared__ __attribute__((aligned(16))) float ws[];     // [9][Ci][Co]
  const int nw = 9 * Ci * Co;
  for (int idx = threadIdx.x; idx < nw; idx += 256) {
    const int co = idx % Co, ci = (idx / Co) % Ci, t = idx / (Co * Ci);
    ws[idx] = flip ? w[((8 - t) * Co + co) * Ci + ci] : w[idx];
  }
  __syncthreads();
  // one thread = two horizontally adjacent output pixels: every weight read from LDS feeds both
  const int Wh = (Wo + 1) >> 1;
  const long pr = (long)blockIdx.x * 256 + threadIdx.x;
  if (pr >= (long)N * Ho * Wh) return;
  const int co0 = blockIdx.y * COT;
  const int wo = 2 * (int)(pr % Wh), ho = (int)((pr / Wh) % Ho), n = (int)(pr / ((long)Wh * Ho));
  const bool two = wo + 1 < Wo;
  float acc0[COT], acc1[COT];
#pragma unroll
  for (int c = 0; c < COT; ++c) { acc0[c] = 0.f; acc1[c] = 0.f; }
  for (int i = 0; i < 3; ++i) {
    const int h = ho * s - pt + i;
    if (h < 0 || h >= H) continue;
    const float* xrow = x + ((long)n * H + h) * W * Ci;
    for (int j = 0; j < 3; ++j) {
      const int w0 = wo * s - pl + j, w1 = w0 + s;
      const bool v0 = w0 >= 0 && w0 < W, v1 = two && w1 >= 0 && w1 < W;
      if (!v0 && !v1) continue;
      const float* x0 = xrow + (long)(v0 ? w0 : 0) * Ci;
      const float* x1 = xrow + (long)(v1 ? w1 : 0) * Ci;
      const float* wp = ws + (i * 3 + j) * Ci * Co + co0;
      if (Ci % 4 == 0) {
        for (int ci = 0; ci < Ci; ci += 4) {
          f32x4 a = ld4(x0 + ci), b = ld4(x1 + ci);
          if (!v0) a = f32x4{0.f, 0.f, 0.f, 0.f};
          if (!v1) b = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float* wr = wp + (ci + e) * Co;
#pragma unroll
            for (int c = 0; c < COT; c += 4) {
              const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + c);
#pragma unroll
              for (int k = 0; k < 4; ++k) { acc0[c + k] += a[e] * wv[k]; acc1[c + k] += b[e] * wv[k]; }
            }
          }
        }
      } else {
        for (int ci = 0; ci < Ci; ++ci) {
          const float a = v0 ? x0[ci] : 0.f, b = v1 ? x1[ci] : 0.f;
          const float* wr = wp + ci * Co;
#pragma unroll
          for (int c = 0; c < COT; c += 4) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc0[c + k] += a * wv[k]; acc1[c + k] += b * wv[k]; }
          }
        }
      }
    }
  }
  float* yp = y + (((long)n * Ho + ho) * Wo + wo) * Co + co0;
#pragma unroll
  for (int c = 0; c < COT; c += 4) {
    f32x4 v0 = {acc0[c], acc0[c + 1], acc0[c + 2], acc0[c + 3]}, v1 = {acc1[c], acc1[c + 1], acc1[c + 2], acc1[c + 3]};
    if (bias) { const f32x4 bv = ld4(bias + co0 + c); v0 += bv; v1 += bv; }
    if (beta != 0.f) { v0 += beta * ld4(yp + c); if (two) v1 += beta * ld4(yp + Co + c); }
    st4(yp + c, v0);
    if (two) st4(yp + Co + c, v1);
  }
}

// stride-2 data gradient: dx[n,h,w,ci] = sum over taps (i,j) with (h + pt - i), (w + pl - j) even and in range, over co, of
// dy[n, (h+pt-i)/2, (w+pl-j)/2, co] * w[i,j,ci,co]      (+ beta * dx)
template <int CIT>
__global__ __launch_bounds__(256) void conv3x3_bwd_data_s2_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                                  int N, int H, int W, int Ci, int Co, int pt, int pl, int Ho, int Wo,
                                                                  float beta) {
  extern __shared__ float ws[];                        // [9][Ci][Co] as stored
  const int nw = 9 * Ci * Co;
  for (int idx = threadIdx.x; idx < nw; idx += 256) ws[idx] = w[idx];
  __syncthreads();
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (long)N * H * W) return;
  const int ci0 = blockIdx.y * CIT;
  const int ww = (int)(pix % W), h = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
  float acc[CIT];
#pragma unroll
  for (int c = 0; c < CIT; ++c) acc[c] = 0.f;
  for (int i = 0; i < 3; ++i) {
    const int hn = h + pt - i;
    if (hn < 0 || (hn & 1) || (hn >> 1) >= Ho) continue;
    for (int j = 0; j < 3; ++j) {
      const int wn = ww + pl - j;
      if (wn < 0 || (wn & 1) || (wn >> 1) >= Wo) continue;
      const float* dp = dy + (((long)n * Ho + (hn >> 1)) * Wo + (wn >> 1)) * Co;
      const float* wp = ws + ((i * 3 + j) * Ci + ci0) * Co;
      for (int co = 0; co < Co; co += 4) {
        const f32x4 dv = ld4(dp + co);
#pragma unroll
        for (int c = 0; c < CIT; ++c) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + c * Co + co);
          acc[c] += dv[0] * wv[0] + dv[1] * wv[1] + dv[2] * wv[2] + dv[3] * wv[3];
        }
      }
    }
  }
  float* xp = dx + pix * Ci + ci0;
#pragma unroll
  for (int c = 0; c < CIT; c += 4) {
    f32x4 v = {acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
    if (beta != 0.f) v += beta * ld4(xp + c);
    st4(xp + c, v);
  }
}

// weight gradient partials.  Thread = PP (tap, cin) pairs of one row group g (PP = 1: G = 256 / (9*Ci) groups share the frame's
// rows; PP = 2 for 9*Ci > 256: one group); it keeps the Co accumulators of its pairs in registers: per pixel ONE x read and
// Co/4 broadcast 16-byte dy reads feed Co FMAs.
// part[(blk*G + g)][(i*3+j)*Ci*Co + ci*Co + co] = sum over the block's frames and the group's pixels of
// x[n, ho*s - pt + i, wo*s - pl + j, ci] * dy[n, ho, wo, co]
template <int CO, int PP>
__global__ __launch_bounds__(256) void conv3x3_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                                 int N, int H, int W, int Ci, int s, int pt, int pl, int Ho, int Wo,
                                                                 int frames_per_blk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Hp = H + 2, Wp = W + 2;                    // zero halo of one pixel on every side covers pt, pl in {0, 1}
  float* ds = sm;                                       // [Ho][Wo][CO]   (first: 16-byte aligned rows)
  float* xs = sm + ((Ho * Wo * CO + 3) & ~3);           // [Hp][Wp][Ci]
  const int npair = 9 * Ci, G = PP == 1 ? 256 / npair : 1;
  const int g = PP == 1 ? threadIdx.x / npair : 0;
  const bool active = g < G;
  int xoff[PP];                                         // xs offset of (tap, ci) relative to the output pixel's window origin
  bool on[PP];
#pragma unroll
  for (int q = 0; q < PP; ++q) {
    const int pr = (PP == 1 ? threadIdx.x % npair : threadIdx.x + q * 256);
    on[q] = active && pr < npair;
    const int ci = pr % Ci, t = pr / Ci, i = t / 3, j = t % 3;
    xoff[q] = ((i + 1 - pt) * Wp + (1 - pl + j)) * Ci + ci;
  }
  float acc[PP][CO];
#pragma unroll
  for (int q = 0; q < PP; ++q)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[q][c] = 0.f;
  const int n0 = blockIdx.x * frames_per_blk, n1 = min(N, n0 + frames_per_blk);
  const int nx = Hp * Wp * Ci, nd = Ho * Wo * CO;
  for (int idx = threadIdx.x; idx < nx; idx += 256) xs[idx] = 0.f;       // the halo stays zero; frames only overwrite the interior
  const int rowf = W * Ci, rowq = rowf >> 2;                              // floats / 16-byte pieces per map row (W*Ci % 4 == 0)
  for (int n = n0; n < n1; ++n) {
    __syncthreads();
    const float* xn = x + (long)n * H * rowf;
    for (int idx = threadIdx.x; idx < H * rowq; idx += 256) {
      const int h = idx / rowq, p4 = idx % rowq;
      const f32x4 v = ld4(xn + (long)h * rowf + p4 * 4);
      float* d = xs + ((h + 1) * Wp + 1) * Ci + p4 * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    const float* dn = dy + (long)n * nd;
    for (int idx = threadIdx.x * 4; idx < nd; idx += 1024) st4(ds + idx, ld4(dn + idx));
    __syncthreads();
    if (active) {
      for (int ho = g; ho < Ho; ho += G) {
        const float* dr = ds + ho * Wo * CO;
        const int rbase = ho * s * Wp * Ci;
        for (int wo = 0; wo < Wo; ++wo) {
          float xv[PP];
#pragma unroll
          for (int q = 0; q < PP; ++q) xv[q] = on[q] ? xs[rbase + wo * s * Ci + xoff[q]] : 0.f;
#pragma unroll
          for (int c = 0; c < CO; c += 4) {
            const f32x4 dv = *reinterpret_cast<const f32x4*>(dr + wo * CO + c);
#pragma unroll
            for (int q = 0; q < PP; ++q) {
              acc[q][c] += xv[q] * dv[0]; acc[q][c + 1] += xv[q] * dv[1]; acc[q][c + 2] += xv[q] * dv[2]; acc[q][c + 3] += xv[q] * dv[3];
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < PP; ++q) {
    if (!on[q]) continue;
    const int pr = (PP == 1 ? threadIdx.x % npair : threadIdx.x + q * 256);
    float* p = part + ((long)blockIdx.x * G + g) * (npair * CO) + (long)pr * CO;
#pragma unroll
    for (int c = 0; c < CO; c += 4) st4(p + c, f32x4{acc[q][c], acc[q][c + 1], acc[q][c + 2], acc[q][c + 3]});
  }
}

}  // namespace avsr

int avsr_colsum_final_launch(const float* part, int nblk, float* out, int F, float alpha, float beta, void* stream);
// conv_mfma.hip: frame-resident MFMA kernels (AVSR_ERR_UNSUPPORTED -> the direct kernels below)
int avsr_conv3x3_mfma(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Ci, int Co, int stride, int pad_t,
                      int pad_l, int Ho, int Wo, int flip, float beta, float* stats, int* nstat, void* stream);
int avsr_conv3x3_bwd_data_s2_mfma(const float* dy, const float* w, float* dx, int N, int H, int W, int Ci, int Co, int pad_t, int pad_l, int Ho,
                                  int Wo, float beta, void* stream);
int avsr_conv3x3_bwd_weight_mfma(const float* x, const float* dy, float* dw, int N, int H, int W, int Ci, int Co, int stride, int pad_t,
                                 int pad_l, int Ho, int Wo, float beta, float* scratch, long scratch_floats, void* stream);

using namespace avsr;
#define S_(x) ((hipStream_t)(x))

static bool direct_ok(int Ci, int Co) { return Ci * Co <= 1024 && 9 * Ci <= 512 && Co % 4 == 0 && (Ci % 4 == 0 || Ci < 4); }

extern "C" int avsr_conv3x3_supported(int32_t Ci, int32_t Co, int32_t H, int32_t W) {
  return direct_ok(Ci, Co) && (Co == 4 || Co == 8 || Co == 16 || Co == 32) && (long)((H + 2) * (W + 2) * Ci + H * W * Co) * 4 <= 150 * 1024;
}

// flip = 0: forward conv (x [N,H,W,Ci] -> y [N,Ho,Wo,Co], w = TF kernel [3,3,Ci,Co], bias may be NULL).
// flip = 1: stride-1 data gradient (x = dy [N,H,W,Ci=cout], y = dx [N,H,W,Co=cin], w = the forward kernel [3,3,Co,Ci]).
extern "C" int avsr_conv3x3(const float* x, const float* w, const float* bias, float* y, int32_t N, int32_t H, int32_t W, int32_t Ci,
                            int32_t Co, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, int32_t flip, float beta,
                            void* stream) {
  if (!x || !w || !y || N <= 0 || !direct_ok(flip ? Co : Ci, flip ? Ci : Co) || Co % 4 || (flip && stride != 1)) return AVSR_ERR_ARG;
  {
    const int rc = avsr_conv3x3_mfma(x, w, bias, y, N, H, W, Ci, Co, stride, pad_t, pad_l, Ho, Wo, flip, beta, nullptr, nullptr, stream);
    if (rc != AVSR_ERR_UNSUPPORTED) return rc;
  }
  const long pix = (long)N * Ho * ((Wo + 1) / 2);       // one thread per pair of output pixels
  const size_t lds = sizeof(float) * 9 * Ci * Co;
  avsr::ProfScope ps(flip ? avsr::PROF_CONV_BWD_DATA : avsr::PROF_CONV_FWD, S_(stream), 2.0 * N * Ho * Wo * 9.0 * Ci * Co);
  if (Co % 16 == 0) hipLaunchKernelGGL((conv3x3_kernel<16>), dim3((unsigned)((pix + 255) / 256), Co / 16), dim3(256), lds, S_(stream), x, w, bias, y, N, H,
                                       W, Ci, Co, stride, pad_t, pad_l, Ho, Wo, flip, beta);
  else if (Co % 8 == 0) hipLaunchKernelGGL((conv3x3_kernel<8>), dim3((unsigned)((pix + 255) / 256), Co / 8), dim3(256), lds, S_(stream), x, w, bias, y, N, H,
                                           W, Ci, Co, stride, pad_t, pad_l, Ho, Wo, flip, beta);
  else hipLaunchKernelGGL((conv3x3_kernel<4>), dim3((unsigned)((pix + 255) / 256), Co / 4), dim3(256), lds, S_(stream), x, w, bias, y, N, H, W, Ci, Co,
                          stride, pad_t, pad_l, Ho, Wo, flip, beta);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

extern "C" int avsr_conv3x3_bwd_data_s2(const float* dy, const float* w, float* dx, int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                                        int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, float beta, void* stream) {
  if (!dy || !w || !dx || N <= 0 || !direct_ok(Ci, Co) || Ci % 4) return AVSR_ERR_ARG;
  {
    const int rc = avsr_conv3x3_bwd_data_s2_mfma(dy, w, dx, N, H, W, Ci, Co, pad_t, pad_l, Ho, Wo, beta, stream);
    if (rc != AVSR_ERR_UNSUPPORTED) return rc;
  }
  const long pix = (long)N * H * W;
  const size_t lds = sizeof(float) * 9 * Ci * Co;
  avsr::ProfScope ps(avsr::PROF_CONV_BWD_DATA, S_(stream), 2.0 * N * Ho * Wo * 9.0 * Ci * Co);
  if (Ci % 8 == 0) hipLaunchKernelGGL((conv3x3_bwd_data_s2_kernel<8>), dim3((unsigned)((pix + 255) / 256), Ci / 8), dim3(256), lds, S_(stream), dy, w, dx, N,
                                      H, W, Ci, Co, pad_t, pad_l, Ho, Wo, beta);
  else hipLaunchKernelGGL((conv3x3_bwd_data_s2_kernel<4>), dim3((unsigned)((pix + 255) / 256), Ci / 4), dim3(256), lds, S_(stream), dy, w, dx, N, H, W, Ci,
                          Co, pad_t, pad_l, Ho, Wo, beta);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

// dw (+)= weight gradient [3,3,Ci,Co] (TF layout).  scratch >= ceil(N / frames_per_block) * 9*Ci*Co floats (frames_per_block = 4).
extern "C" int avsr_conv3x3_bwd_weight(const float* x, const float* dy, float* dw, int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                                       int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, float beta, float* scratch,
                                       int64_t scratch_floats, void* stream) {
  if (!x || !dy || !dw || !scratch || N <= 0 || !avsr_conv3x3_supported(Ci, Co, H, W) || pad_t > 1 || pad_l > 1) return AVSR_ERR_ARG;
  if (Co != 4 && Co != 8 && Co != 16 && Co != 32) return AVSR_ERR_ARG;
  {
    const int rc = avsr_conv3x3_bwd_weight_mfma(x, dy, dw, N, H, W, Ci, Co, stride, pad_t, pad_l, Ho, Wo, beta, scratch, scratch_floats, stream);
    if (rc != AVSR_ERR_UNSUPPORTED) return rc;
  }
  const int PP = 9 * Ci > 256 ? 2 : 1;
  const int nout = 9 * Ci * Co, G = PP == 1 ? 256 / (9 * Ci) : 1;
  int fpb = H * W <= 128 ? 16 : 4;                      // small maps: more frames per block (fewer partial rows)
  int nblk = (N + fpb - 1) / fpb;
  if ((long)nblk * G * nout > scratch_floats) {
    nblk = (int)(scratch_floats / ((long)G * nout));
    if (nblk < 1) return AVSR_ERR_ARG;
    fpb = (N + nblk - 1) / nblk;
    nblk = (N + fpb - 1) / fpb;
  }
  const size_t lds = sizeof(float) * ((size_t)(H + 2) * (W + 2) * Ci + (size_t)((Ho * Wo * Co + 3) & ~3));
  static bool big_lds = false;                          // more than 64 KB of dynamic LDS needs the attribute (set once, outside any capture)
  if (!big_lds) {
#define BW_ATTR(CO_, PP_) \
    if (hipFuncSetAttribute((const void*)conv3x3_bwd_weight_kernel<CO_, PP_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return AVSR_ERR_HIP;
    BW_ATTR(4, 1) BW_ATTR(8, 1) BW_ATTR(16, 1) BW_ATTR(32, 1) BW_ATTR(4, 2) BW_ATTR(8, 2) BW_ATTR(16, 2) BW_ATTR(32, 2)
#undef BW_ATTR
    big_lds = true;
  }
  avsr::ProfScope ps(avsr::PROF_CONV_BWD_WEIGHT, S_(stream), 2.0 * N * Ho * Wo * 9.0 * Ci * Co);
#define BW_GO(CO_, PP_) hipLaunchKernelGGL((conv3x3_bwd_weight_kernel<CO_, PP_>), dim3(nblk), dim3(256), lds, S_(stream), x, dy, scratch, N, H, W, Ci, \
                                           stride, pad_t, pad_l, Ho, Wo, fpb)
  if (PP == 1) { if (Co == 32) BW_GO(32, 1); else if (Co == 16) BW_GO(16, 1); else if (Co == 8) BW_GO(8, 1); else BW_GO(4, 1); }
  else { if (Co == 32) BW_GO(32, 2); else if (Co == 16) BW_GO(16, 2); else if (Co == 8) BW_GO(8, 2); else BW_GO(4, 2); }
#undef BW_GO
  AVSR_CHECK_LAUNCH();
  return avsr_colsum_final_launch(scratch, nblk * G, dw, nout, 1.0f, beta, stream);
}
