// fp32 GEMM on the gfx950 f32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 157 TF peak).
//
// Serves every "hoisted" matmul of the AVSR hot path: the per-layer input projections X*Wx over all
// B*T rows (encoder.py:80/:110 -> LSTMCell kernel, x part), attention memory_layer keys
// (attention.py:26-72), the vocabulary projection, and the weight-gradient / input-gradient GEMMs
// of BPTT (tf.gradients, seq2seq.py:222).
//
//   C[M,N] = alpha * opA(A)[M,K] * opB(B)[K,N] + beta * C + bias[N]
//
// Every stored matrix uses two-level row addressing so that batch-major [B, T(+pad), F] activations
// (and time-shifted views of them) are consumed in place, without gather/transposes:
//     row r  ->  ptr + (T ? (r / T) * ldo + (r % T) * ld : r * ld)
//
// Tile: 128x128x16 per 256-thread workgroup (4 waves, each a 64x64 quadrant = 2x2 MFMA 32x32 tiles,
// 64 accumulator VGPRs).  LDS holds both operands k-major ([16][128]) so MFMA operand reads are
// conflict-free ds_read_b32 of 32 consecutive floats per half-wave.  Global loads for tile i+1 are
// issued into registers before the MFMAs of tile i (register double-buffer), LDS is double-buffered:
// one barrier per K-tile.  Split-K (grid.z) with a deterministic second-pass reduction covers the
// tall-skinny dW GEMMs (K = B*T rows, tiny M x N).
#include "common.h"
#include "avsr_hip.h"
#include "prof.h"

namespace avsr {

constexpr int BM = 128, BN = 128, BK = 16;

struct MatView {
  const float* p;
  long ld;   // inner row stride (floats)
  int T;     // rows per outer group (0 = flat)
  long ldo;  // outer stride
  __device__ __forceinline__ const float* row(int r) const {
    return T ? p + (long)(r / T) * ldo + (long)(r % T) * ld : p + (long)r * ld;
  }
};

struct GemmArgs {
  MatView A, B;
  float* C; long ldc; int Tc; long ldoc;
  const float* bias;
  const float* alpha_dev;
  int M, N, K;
  int ta, tb;           // ta: A stored [K][M]; tb: B stored [N][K]
  float alpha, beta;
  int splitk, kper;     // split-K: slices of kper (multiple of BK) along K
  float* ws;            // split-K partials [splitk][M][N]
  long sA, sB, sC;      // batch strides (floats)
  int batch;
};

// Load one operand tile into registers.  The LDS image is k-major: T[k][mn], k in [0,16), mn in [0,128).
// kcontig: stored rows are indexed by mn and k is the contiguous dim (A with ta=0, B with tb=1).
// else   : stored rows are indexed by k and mn is the contiguous dim (A with ta=1, B with tb=0).
template <bool VEC>
__device__ __forceinline__ void load_tile(const MatView& S, bool kcontig, int mn0, int k0, int MN, int K,
                                          int kend, f32x4 (&r)[2]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (kcontig) {
      const int mn = mn0 + (tid & 127);
      const int k = k0 + 4 * ((tid >> 7) + 2 * p);
      if (mn < MN) {
        const float* src = S.row(mn) + k;
        if (VEC) {
          if (k < kend) v = ld4(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < kend) v[e] = src[e];
        }
      }
    } else {
      const int k = k0 + (tid >> 5) + 8 * p;
      const int mn = mn0 + 4 * (tid & 31);
      if (k < kend) {
        const float* src = S.row(k) + mn;
        if (VEC) {
          if (mn < MN) v = ld4(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (mn + e < MN) v[e] = src[e];
        }
      }
    }
    r[p] = v;
  }
}

__device__ __forceinline__ void store_tile(float* T, bool kcontig, const f32x4 (&r)[2]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (kcontig) {
      const int mn = tid & 127;
      const int k = 4 * ((tid >> 7) + 2 * p);
#pragma unroll
      for (int e = 0; e < 4; ++e) T[(k + e) * BM + mn] = r[p][e];
    } else {
      const int k = (tid >> 5) + 8 * p;
      const int mn = 4 * (tid & 31);
      st4(&T[k * BM + mn], r[p]);
    }
  }
}

template <bool VECA, bool VECB>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][BK * BM];  // [buf][A|B][k][mn]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int z = blockIdx.z;
  const int bz = z / g.splitk, sz = z % g.splitk;
  MatView A = g.A, B = g.B;
  A.p += (long)bz * g.sA;
  B.p += (long)bz * g.sB;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = sz * g.kper;
  const int kend = min(g.K, kbeg + g.kper);
  const bool akc = (g.ta == 0), bkc = (g.tb != 0);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[2], rb[2];
  load_tile<VECA>(A, akc, m0, kbeg, g.M, g.K, kend, ra);
  load_tile<VECB>(B, bkc, n0, kbeg, g.N, g.K, kend, rb);
  store_tile(lds[0][0], akc, ra);
  store_tile(lds[0][1], bkc, rb);
  __syncthreads();

  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = (k0 + BK) < kend;
    if (more) {
      load_tile<VECA>(A, akc, m0, k0 + BK, g.M, g.K, kend, ra);
      load_tile<VECB>(B, bkc, n0, k0 + BK, g.N, g.K, kend, rb);
    }
    const float* As = lds[buf][0];
    const float* Bs = lds[buf][1];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int k = 2 * kk + (lane >> 5);
      const float a0 = As[k * BM + wm * 64 + (lane & 31)];
      const float a1 = As[k * BM + wm * 64 + 32 + (lane & 31)];
      const float b0 = Bs[k * BN + wn * 64 + (lane & 31)];
      const float b1 = Bs[k * BN + wn * 64 + 32 + (lane & 31)];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
      store_tile(lds[buf ^ 1][0], akc, ra);
      store_tile(lds[buf ^ 1][1], bkc, rb);
    }
    __syncthreads();
    buf ^= 1;
  }

  // epilogue: C/D layout of mfma 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool split = g.splitk > 1;
  const float alpha = g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
      if (col >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= g.M) continue;
        if (split) {
          g.ws[((long)z * g.M + row) * g.N + col] = acc[i][j][r];
        } else {
          float* c = g.C + (long)bz * g.sC +
                     (g.Tc ? (long)(row / g.Tc) * g.ldoc + (long)(row % g.Tc) * g.ldc : (long)row * g.ldc) + col;
          float v = alpha * acc[i][j][r];
          if (g.beta != 0.f) v += g.beta * *c;
          if (g.bias) v += g.bias[col];
          *c = v;
        }
      }
    }
}

__global__ void gemm_splitk_reduce_kernel(GemmArgs g) {
  const long total = (long)g.batch * g.M * g.N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int col = (int)(i % g.N);
    const int row = (int)((i / g.N) % g.M);
    const int bz = (int)(i / ((long)g.M * g.N));
    float s = 0.f;
    for (int z = 0; z < g.splitk; ++z) s += g.ws[((long)(bz * g.splitk + z) * g.M + row) * g.N + col];
    float* c = g.C + (long)bz * g.sC +
               (g.Tc ? (long)(row / g.Tc) * g.ldoc + (long)(row % g.Tc) * g.ldc : (long)row * g.ldc) + col;
    float v = (g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha) * s;
    if (g.beta != 0.f) v += g.beta * *c;
    if (g.bias) v += g.bias[col];
    *c = v;
  }
}

static bool vec_ok(const avsr_mat* m, bool contig_is_k, int MN, int K) {
  const int contig = contig_is_k ? K : MN;
  return (((uintptr_t)m->ptr) % 16 == 0) && (m->ld % 4 == 0) && (contig % 4 == 0) && (m->T == 0 || m->ldo % 4 == 0);
}

}  // namespace avsr

extern "C" int avsr_gemm(const avsr_gemm_desc* d, void* stream) {
  using namespace avsr;
  if (!d || d->M <= 0 || d->N <= 0 || d->K < 0 || !d->A.ptr || !d->B.ptr || !d->C.ptr) return AVSR_ERR_ARG;
  GemmArgs g;
  g.A = MatView{d->A.ptr, d->A.ld, d->A.T, d->A.ldo};
  g.B = MatView{d->B.ptr, d->B.ld, d->B.T, d->B.ldo};
  g.C = d->C.ptr; g.ldc = d->C.ld; g.Tc = d->C.T; g.ldoc = d->C.ldo;
  g.bias = d->bias;
  g.alpha_dev = d->alpha_dev;
  g.M = d->M; g.N = d->N; g.K = d->K;
  g.ta = d->trans_a; g.tb = d->trans_b;
  g.alpha = d->alpha; g.beta = d->beta;
  g.batch = d->batch > 0 ? d->batch : 1;
  g.sA = d->stride_a; g.sB = d->stride_b; g.sC = d->stride_c;
  int splitk = d->splitk > 0 ? d->splitk : 1;
  if (splitk > 1 && (!d->workspace || d->workspace_floats < (long)g.batch * splitk * g.M * g.N)) return AVSR_ERR_ARG;
  int kper = ((g.K + splitk - 1) / splitk + BK - 1) / BK * BK;
  if (kper <= 0) kper = BK;
  splitk = (g.K + kper - 1) / kper;
  if (splitk < 1) splitk = 1;
  g.splitk = splitk; g.kper = kper; g.ws = d->workspace;
  const bool va = vec_ok(&d->A, g.ta == 0, g.M, g.K);
  const bool vb = vec_ok(&d->B, g.tb != 0, g.N, g.K);
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.batch * splitk);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(PROF_GEMM, s, 2.0 * g.M * g.N * g.K * g.batch);
  if (va && vb) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, s, g);
  else if (va) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, s, g);
  else if (vb) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, s, g);
  AVSR_CHECK_LAUNCH();
  if (splitk > 1) {
    const long total = (long)g.batch * g.M * g.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, g);
    AVSR_CHECK_LAUNCH();
  }
  return AVSR_OK;
}
