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
  float* cs_out; float cs_beta; float* cs_ws;     // column sums of B (bias gradient): output, its beta, split-K partials [splitk][N]
};

// Operand loader.  One thread fetches two 16-byte pieces of each operand per K tile into registers; the LDS image is
// k-major: T[k][mn], k in [0,16), mn in [0,128).
//   KC (k contiguous in memory: A with ta=0, B with tb=1): piece p = row mn0 + (tid&127), k = k0 + 4*((tid>>7) + 2p)
//   else (mn contiguous: A with ta=1, B with tb=0)       : piece p = row k0 + (tid>>5) + 8p, mn = mn0 + 4*(tid&31)
// Loads are raw buffer loads with an out-of-range offset for pieces outside the matrix / K slice (the hardware
// returns 0): no branch around any load, so the compiler keeps exact vmcnt counts and all pieces of a tile -- and of
// the next tiles -- are in flight together.  (The first version branched per piece; the compiler then waited for
// each load before issuing the next: four serial ~0.5 us round trips per K tile.)  The two-level row address
// (r / T, r % T) is computed once and advanced incrementally.
#define G_OOB ((int)0x80000000)
template <bool KC, bool VEC>
struct OperandLoader {
  __amdgpu_buffer_rsrc_t rs;
  int off[2];        // KC: byte offset of (row, k = this thread's k phase) ; else: byte offset of this thread's mn within a row
  int q[2], r[2];    // !KC: two-level position of the k row of piece p
  bool ok;           // KC: row inside the matrix; else: mn inside the matrix
  int ldb, ldob, T;  // strides in bytes
  int tail;          // !VEC: number of valid elements handling
  int MN, mn;

  __device__ __forceinline__ void init(const MatView& S, int mn0, int k0, int MN_) {
    const int tid = threadIdx.x;
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.p), 0, 0x80000000u, 0x00020000);
    ldb = (int)S.ld * 4; ldob = (int)S.ldo * 4; T = S.T; MN = MN_;
    if (KC) {
      mn = mn0 + (tid & 127);
      ok = mn < MN;
      const int ro = T ? (mn / T) * ldob + (mn % T) * ldb : mn * ldb;
#pragma unroll
      for (int p = 0; p < 2; ++p) off[p] = ro + (k0 + 4 * ((tid >> 7) + 2 * p)) * 4;
    } else {
      mn = mn0 + 4 * (tid & 31);
      ok = mn < MN;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int k = k0 + (tid >> 5) + 8 * p;
        q[p] = T ? k / T : 0; r[p] = T ? k % T : k;
        off[p] = mn * 4;
      }
    }
  }
  // fetch the pieces of the K tile starting at k0 (k0 only feeds the bounds test) and advance to the next tile
  __device__ __forceinline__ void load(int k0, int kend, f32x4 (&out)[2]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int o, k;
      if (KC) { k = k0 + 4 * ((tid >> 7) + 2 * p); o = off[p]; off[p] += BK * 4; }
      else {
        k = k0 + (tid >> 5) + 8 * p;
        o = (T ? q[p] * ldob + r[p] * ldb : r[p] * ldb) + off[p];
        r[p] += BK;
        if (T) { while (r[p] >= T) { r[p] -= T; ++q[p]; } }
      }
      const bool in = ok && k < kend;
      typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
      if (VEC) {
        out[p] = __builtin_bit_cast(f32x4, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(rs, in ? o : G_OOB, 0, 0));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ine = in && (KC ? k + e < kend : mn + e < MN);
          out[p][e] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, ine ? o + 4 * e : G_OOB, 0, 0));
        }
      }
    }
  }
};

template <bool KC>
__device__ __forceinline__ void store_tile(float* T, const f32x4 (&r)[2]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (KC) {
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

template <bool AKC, bool BKC, bool VECA, bool VECB>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int bx, const int by, const int z) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][BK * BM];  // [buf][A|B][k][mn]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bz = z / g.splitk, sz = z % g.splitk;
  MatView A = g.A, B = g.B;
  A.p += (long)bz * g.sA;
  B.p += (long)bz * g.sB;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = sz * g.kper;
  const int kend = min(g.K, kbeg + g.kper);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Software pipeline, prefetch distance 2: while tile i is multiplied out of LDS, tile i+1 sits in registers (it is
  // written to the other LDS buffer after the MFMAs) and tile i+2 is in flight from L2/HBM.  With distance 1 the LDS
  // write waited ~1 us of load latency behind every 0.85 us of MFMAs (one workgroup ran at 2 us per K tile).
  f32x4 ra[2][2], rb[2][2];
  OperandLoader<AKC, VECA> la;
  OperandLoader<BKC, VECB> lb;
  la.init(A, m0, kbeg, g.M);
  lb.init(B, n0, kbeg, g.N);
  la.load(kbeg, kend, ra[0]);
  lb.load(kbeg, kend, rb[0]);
  la.load(kbeg + BK, kend, ra[1]);
  lb.load(kbeg + BK, kend, rb[1]);
  // column sums of B (g.cs_out; B stored [K][N] only): a thread's pieces of every K tile are the same four columns, summed as the tiles
  // are committed to LDS (out-of-range pieces are zeros); the first row tile's workgroups own the sums
  const bool do_cs = !BKC && g.cs_out != nullptr && by == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  store_tile<AKC>(lds[0][0], ra[0]);
  store_tile<BKC>(lds[0][1], rb[0]);
  if (do_cs) csum += rb[0][0] + rb[0][1];
  __syncthreads();

  auto ktile = [&](int k0, int buf, f32x4 (&ra_next)[2], f32x4 (&rb_next)[2], f32x4 (&ra_free)[2], f32x4 (&rb_free)[2]) {
    // ra_free held tile k0 (already in LDS): refill it with tile k0 + 2*BK; ra_next holds tile k0 + BK
    la.load(k0 + 2 * BK, kend, ra_free);       // past the slice: every piece is out of range, nothing is fetched
    lb.load(k0 + 2 * BK, kend, rb_free);
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
    store_tile<AKC>(lds[buf ^ 1][0], ra_next);
    store_tile<BKC>(lds[buf ^ 1][1], rb_next);
    if (do_cs) csum += rb_next[0] + rb_next[1];
    // LDS-only barrier: __syncthreads() also drains vmcnt, i.e. it would wait for the tiles still in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  for (int k0 = kbeg; k0 < kend; k0 += 2 * BK) {
    ktile(k0, 0, ra[1], rb[1], ra[0], rb[0]);
    if (k0 + BK < kend) ktile(k0 + BK, 1, ra[0], rb[0], ra[1], rb[1]);
  }

  if (!BKC && g.cs_out != nullptr) {                  // (uniform per workgroup; the LDS tiles are free: every K tile ends with a barrier)
    float* red = &lds[0][0][0];                       // [8 k rows of the staging layout][128 columns]
    if (do_cs) st4(red + (tid >> 5) * BN + 4 * (tid & 31), csum);
    __syncthreads();
    if (do_cs && tid < BN) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) v += red[r * BN + tid];
      const int col = n0 + tid;
      if (col < g.N) {
        if (g.splitk > 1) g.cs_ws[(long)z * g.N + col] = v;
        else g.cs_out[col] = g.cs_beta != 0.f ? v + g.cs_beta * g.cs_out[col] : v;
      }
    }
    __syncthreads();
  }
  // epilogue: C/D layout of mfma 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  // Per 32x32 tile: the 16 destination addresses of the lane first (the two-level row address by an exact multiply-high division for
  // row counts below 2^16), then -- beta != 0 -- its 16 old values with all loads in flight together, then the 16 stores.  (The
  // first version read, combined and stored one element at a time: 64 dependent memory round trips per lane whenever beta != 0 --
  // a 256 x 256 x 64 accumulating GEMM took 31 us for 4 us of work -- and an integer division per element on two-level outputs.)
  const bool split = g.splitk > 1;
  const float alpha = g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha;
  const unsigned mTc = (g.Tc > 1 && g.M < 65536) ? (unsigned)(((1ull << 32) / (unsigned)g.Tc) + 1ull) : 0u;
  if (split) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= g.N) continue;
        const int rbase = m0 + wm * 64 + i * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (row < g.M) g.ws[((long)z * g.M + row) * g.N + col] = acc[i][j][r];
        }
      }
    return;
  }
  // eight half-tiles (i, j, h) of eight rows each, software-pipelined: the offsets and old values of half-tile t + 1 are requested
  // BEFORE half-tile t is stored (the compiler cannot move a load above a store that may alias it, so the program order does it).
  // Raw buffer accesses with 32-bit byte offsets (an output matrix stays below 2 GB: checked by the host); rows / columns outside the
  // matrix use an out-of-range offset -- their loads return 0 and their stores are dropped.
  const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(g.C + (long)bz * g.sC, 0, 0x80000000u, 0x00020000);
  int co[2][8];
  float old[2][8];
  auto prep = [&](int t, int (&cot)[8], float (&oldt)[8]) {
    const int i = t >> 2, j = (t >> 1) & 1, h = t & 1;
    const int col = n0 + wn * 64 + j * 32 + (lane & 31);
    const int rbase = m0 + wm * 64 + i * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = h * 8 + rr;
      const int row = rbase + (r & 3) + 8 * (r >> 2);
      int o;
      if (!g.Tc) o = row * (int)g.ldc;
      else {
        const int qd = mTc ? (int)__umulhi((unsigned)row, mTc) : (g.Tc == 1 ? row : row / g.Tc);
        o = qd * (int)g.ldoc + (row - qd * g.Tc) * (int)g.ldc;
      }
      cot[rr] = (row < g.M && col < g.N) ? (o + col) * 4 : G_OOB;
    }
    if (g.beta != 0.f) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) oldt[rr] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b32(c_rs, cot[rr], 0, 0));
    }
  };
  prep(0, co[0], old[0]);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t + 1 < 8) prep(t + 1, co[(t + 1) & 1], old[(t + 1) & 1]);
    const int i = t >> 2, j = (t >> 1) & 1, h = t & 1;
    const int col = n0 + wn * 64 + j * 32 + (lane & 31);
    const float bias_v = (g.bias && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      float v = alpha * acc[i][j][h * 8 + rr];
      if (g.beta != 0.f) v += g.beta * old[t & 1][rr];
      v += bias_v;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_rs, co[t & 1][rr], 0, 0);
    }
  }
}

template <bool AKC, bool BKC, bool VECA, bool VECB>
__global__ __launch_bounds__(256, 3) void gemm_f32_kernel(GemmArgs g) {
  gemm_body<AKC, BKC, VECA, VECB>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several INDEPENDENT GEMMs of one operand-layout class in one launch (avsr_gemm_batch): the small matmuls of a train step -- state
// bridges, per-memory attention gradients, the row blocks of a cell kernel's gradient -- are a few workgroups each and cost a
// dependent launch (and a split-K reduction launch) apiece when issued one by one; side by side they fill the chip together.
// Workgroup b of the launch belongs to problem p with blk_off[p] <= b < blk_off[p + 1]; inside the problem the usual (x, y, z) grid.
#define GEMM_GROUP_MAX 8
struct GemmGroup { int n; int blk_off[GEMM_GROUP_MAX + 1]; int gx[GEMM_GROUP_MAX], gy[GEMM_GROUP_MAX]; int red_off[GEMM_GROUP_MAX + 1]; GemmArgs g[GEMM_GROUP_MAX]; };

template <bool AKC, bool BKC, bool VECA, bool VECB>
__global__ __launch_bounds__(256, 3) void gemm_f32_group_kernel(const GemmGroup G) {
  int p = 0;
#pragma unroll
  for (int k = 1; k < GEMM_GROUP_MAX; ++k) if (k < G.n && (int)blockIdx.x >= G.blk_off[k]) p = k;
  p = __builtin_amdgcn_readfirstlane(p);
  const int b = blockIdx.x - G.blk_off[p];
  const int gx = G.gx[p], gy = G.gy[p];
  const int bx = b % gx, by = (b / gx) % gy, z = b / (gx * gy);
  gemm_body<AKC, BKC, VECA, VECB>(G.g[p], bx, by, z);
}

__device__ __forceinline__ void splitk_reduce_body(const GemmArgs& g, const long first, const long stride) {
  const long total = (long)g.batch * g.M * g.N;
  if (g.cs_out) {                                     // the column sums' split-K partials (batch == 1)
    for (long col = first; col < g.N; col += stride) {
      float v = 0.f;
      int zz = 0;
      for (; zz + 8 <= g.splitk; zz += 8) {             // eight slices in flight (one at a time was a chain of ~48 memory round trips)
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = g.cs_ws[(long)(zz + u) * g.N + col];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += w[u];
      }
      for (; zz < g.splitk; ++zz) v += g.cs_ws[(long)zz * g.N + col];
      g.cs_out[col] = g.cs_beta != 0.f ? v + g.cs_beta * g.cs_out[col] : v;
    }
  }
  // four consecutive columns per thread with 16-byte loads, eight slices in flight (the K = 32000 weight gradients are reduced from
  // ~48 one-megabyte slabs each: scalar loads, four in flight, ran at 1.5 TB/s).  Every element keeps its summation order.
  const bool vec = (g.N & 3) == 0 && ((uintptr_t)g.ws & 15) == 0 && ((uintptr_t)g.C & 15) == 0 && (g.ldc & 3) == 0 && (g.sC & 3) == 0 &&
                   (g.Tc == 0 || (g.ldoc & 3) == 0) && (!g.bias || ((uintptr_t)g.bias & 15) == 0);
  if (vec) {
    const long zs = (long)g.M * g.N;
    const float al = g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha;
    for (long i = first * 4; i < total; i += stride * 4) {
      const int col = (int)(i % g.N);
      const int row = (int)((i / g.N) % g.M);
      const int bz = (int)(i / zs);
      const float* w = g.ws + ((long)bz * g.splitk * g.M + row) * g.N + col;
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      int z = 0;
      for (; z + 8 <= g.splitk; z += 8) {
        f32x4 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = ld4(w + (z + u) * zs);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u];
      }
      for (; z < g.splitk; ++z) s += ld4(w + z * zs);
      float* c = g.C + (long)bz * g.sC +
                 (g.Tc ? (long)(row / g.Tc) * g.ldoc + (long)(row % g.Tc) * g.ldc : (long)row * g.ldc) + col;
      f32x4 v = al * s;
      if (g.beta != 0.f) v += g.beta * ld4(c);
      if (g.bias) v += ld4(g.bias + col);
      st4(c, v);
    }
    return;
  }
  for (long i = first; i < total; i += stride) {
    const int col = (int)(i % g.N);
    const int row = (int)((i / g.N) % g.M);
    const int bz = (int)(i / ((long)g.M * g.N));
    // slices summed in index order (deterministic), four loads in flight
    const float* w = g.ws + ((long)bz * g.splitk * g.M + row) * g.N + col;
    const long zs = (long)g.M * g.N;
    float s = 0.f;
    int z = 0;
    for (; z + 4 <= g.splitk; z += 4) {
      const float a0 = w[z * zs], a1 = w[(z + 1) * zs], a2 = w[(z + 2) * zs], a3 = w[(z + 3) * zs];
      s = (((s + a0) + a1) + a2) + a3;
    }
    for (; z < g.splitk; ++z) s += w[z * zs];
    float* c = g.C + (long)bz * g.sC +
               (g.Tc ? (long)(row / g.Tc) * g.ldoc + (long)(row % g.Tc) * g.ldc : (long)row * g.ldc) + col;
    float v = (g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha) * s;
    if (g.beta != 0.f) v += g.beta * *c;
    if (g.bias) v += g.bias[col];
    *c = v;
  }
}
__global__ void gemm_splitk_reduce_kernel(GemmArgs g) {
  splitk_reduce_body(g, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}
__global__ void gemm_splitk_reduce_group_kernel(const GemmGroup G) {
  int p = 0;
#pragma unroll
  for (int k = 1; k < GEMM_GROUP_MAX; ++k) if (k < G.n && (int)blockIdx.x >= G.red_off[k]) p = k;
  p = __builtin_amdgcn_readfirstlane(p);
  if (G.g[p].splitk <= 1) return;
  const int nb = G.red_off[p + 1] - G.red_off[p];
  splitk_reduce_body(G.g[p], (long)(blockIdx.x - G.red_off[p]) * blockDim.x + threadIdx.x, (long)nb * blockDim.x);
}

static bool vec_ok(const avsr_mat* m, bool contig_is_k, int MN, int K) {
  const int contig = contig_is_k ? K : MN;
  return (((uintptr_t)m->ptr) % 16 == 0) && (m->ld % 4 == 0) && (contig % 4 == 0) && (m->T == 0 || m->ldo % 4 == 0);
}

}  // namespace avsr

// AVSR_GEMM_LOG=1: every launch is bracketed by events, waited for, and printed to stderr with its shapes (tools/gemm_step_log.py);
// a debugging aid -- it serialises the stream -- never set in a timed run
#include <cstdio>
#include <cstdlib>
static bool gemm_log_on() { static int v = -1; if (v < 0) { const char* e = getenv("AVSR_GEMM_LOG"); v = (e && e[0] == '1') ? 1 : 0; } return v == 1; }
struct GemmLog {
  hipEvent_t e0, e1; hipStream_t s; bool on;
  explicit GemmLog(hipStream_t st) : s(st), on(gemm_log_on()) { if (on) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, s); } }
  void done(const avsr::GemmArgs* g, int n, const char* what) {
    if (!on) return;
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    double fl = 0.0;
    for (int i = 0; i < n; ++i) fl += 2.0 * g[i].M * g[i].N * g[i].K * g[i].batch;
    fprintf(stderr, "[gemm] %s %8.1f us %6.1f TF :", what, ms * 1e3, fl / (ms * 1e-3) * 1e-12);
    for (int i = 0; i < n; ++i) fprintf(stderr, " (M=%d N=%d K=%d ta=%d tb=%d splitk=%d batch=%d beta=%g%s)", g[i].M, g[i].N, g[i].K, g[i].ta, g[i].tb,
                                         g[i].splitk, g[i].batch, g[i].beta, g[i].cs_out ? " colsum" : "");
    fprintf(stderr, "\n");
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
};

// descriptor -> kernel arguments; *cls = operand-layout class (akc, bkc, vec a, vec b as bits 3..0); AVSR_OK or an error code
static int gemm_prepare(const avsr_gemm_desc* d, avsr::GemmArgs& g, dim3* grid, int* cls) {
  using namespace avsr;
  if (!d || d->M <= 0 || d->N <= 0 || d->K < 0 || !d->A.ptr || !d->B.ptr || !d->C.ptr) return AVSR_ERR_ARG;
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
  g.cs_out = d->colsum; g.cs_beta = d->colsum_beta; g.cs_ws = nullptr;
  if (g.cs_out) {
    if (g.tb || g.batch != 1) return AVSR_ERR_UNSUPPORTED;
    if (splitk > 1) {
      if (d->workspace_floats < (long)splitk * g.M * g.N + (long)splitk * g.N) return AVSR_ERR_ARG;
      g.cs_ws = d->workspace + (long)splitk * g.M * g.N;
    }
  }
  const bool va = vec_ok(&d->A, g.ta == 0, g.M, g.K);
  const bool vb = vec_ok(&d->B, g.tb != 0, g.N, g.K);
  *grid = dim3((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.batch * splitk);
  if ((long)d->A.ld * 4 >= (1L << 31) || (long)d->B.ld * 4 >= (1L << 31)) return AVSR_ERR_UNSUPPORTED;
  {   // the epilogue addresses C with 32-bit byte offsets per batch entry
    const long rows = g.M - 1;
    const long last = (g.Tc ? (rows / g.Tc) * g.ldoc + (rows % g.Tc) * g.ldc : rows * g.ldc) + g.N;
    if (last * 4 >= (1L << 31)) return AVSR_ERR_UNSUPPORTED;
  }
  *cls = ((g.ta == 0) ? 8 : 0) | ((g.tb != 0) ? 4 : 0) | (va ? 2 : 0) | (vb ? 1 : 0);
  return AVSR_OK;
}

extern "C" int avsr_gemm_batch(const avsr_gemm_desc* descs, int32_t n, void* stream) {
  using namespace avsr;
  if (n <= 0) return AVSR_OK;
  if (!descs) return AVSR_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  static thread_local GemmArgs ga[64];
  dim3 grids[64];
  int cls[64];
  if (n > 64) return AVSR_ERR_ARG;
  for (int i = 0; i < n; ++i) {
    const int rc = gemm_prepare(&descs[i], ga[i], &grids[i], &cls[i]);
    if (rc != AVSR_OK) return rc;
  }
  bool done[64] = {};
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    static thread_local GemmGroup G;
    G.n = 0;
    int blocks = 0, rblocks = 0;
    double flops = 0.0;
    bool any_split = false;
    for (int j = i; j < n && G.n < GEMM_GROUP_MAX; ++j) {
      if (done[j] || cls[j] != cls[i]) continue;
      const int p = G.n++;
      G.g[p] = ga[j];
      G.gx[p] = grids[j].x; G.gy[p] = grids[j].y;
      G.blk_off[p] = blocks; blocks += grids[j].x * grids[j].y * grids[j].z;
      G.red_off[p] = rblocks;
      if (ga[j].splitk > 1) {
        long tot = ((long)ga[j].batch * ga[j].M * ga[j].N + 255) / 256;
        rblocks += (int)(tot > 1024 ? 1024 : tot);
        any_split = true;
      } else rblocks += 1;
      flops += 2.0 * ga[j].M * ga[j].N * ga[j].K * ga[j].batch;
      done[j] = true;
    }
    G.blk_off[G.n] = blocks; G.red_off[G.n] = rblocks;
    GemmLog gl(s);
    ProfScope ps(PROF_GEMM, s, flops);
#define GG(AK, BK_, VA, VB) hipLaunchKernelGGL((gemm_f32_group_kernel<AK, BK_, VA, VB>), dim3(blocks), dim3(256), 0, s, G)
    switch (cls[i]) {
      case 15: GG(true, true, true, true); break;   case 14: GG(true, true, true, false); break;
      case 13: GG(true, true, false, true); break;  case 12: GG(true, true, false, false); break;
      case 11: GG(true, false, true, true); break;  case 10: GG(true, false, true, false); break;
      case 9: GG(true, false, false, true); break;  case 8: GG(true, false, false, false); break;
      case 7: GG(false, true, true, true); break;   case 6: GG(false, true, true, false); break;
      case 5: GG(false, true, false, true); break;  case 4: GG(false, true, false, false); break;
      case 3: GG(false, false, true, true); break;  case 2: GG(false, false, true, false); break;
      case 1: GG(false, false, false, true); break; default: GG(false, false, false, false); break;
    }
#undef GG
    AVSR_CHECK_LAUNCH();
    if (any_split) {
      hipLaunchKernelGGL(gemm_splitk_reduce_group_kernel, dim3(rblocks), dim3(256), 0, s, G);
      AVSR_CHECK_LAUNCH();
    }
    gl.done(G.g, G.n, "group ");
  }
  return AVSR_OK;
}

extern "C" int avsr_gemm(const avsr_gemm_desc* d, void* stream) {
  using namespace avsr;
  GemmArgs g;
  dim3 grid;
  int cls = 0;
  const int rc = gemm_prepare(d, g, &grid, &cls);
  if (rc != AVSR_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  GemmLog gl(s);
  ProfScope ps(PROF_GEMM, s, 2.0 * g.M * g.N * g.K * g.batch);
#define GO(AK, BK_, VA, VB) hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, VA, VB>), grid, dim3(256), 0, s, g)
  switch (cls) {
    case 15: GO(true, true, true, true); break;   case 14: GO(true, true, true, false); break;
    case 13: GO(true, true, false, true); break;  case 12: GO(true, true, false, false); break;
    case 11: GO(true, false, true, true); break;  case 10: GO(true, false, true, false); break;
    case 9: GO(true, false, false, true); break;  case 8: GO(true, false, false, false); break;
    case 7: GO(false, true, true, true); break;   case 6: GO(false, true, true, false); break;
    case 5: GO(false, true, false, true); break;  case 4: GO(false, true, false, false); break;
    case 3: GO(false, false, true, true); break;  case 2: GO(false, false, true, false); break;
    case 1: GO(false, false, false, true); break; default: GO(false, false, false, false); break;
  }
#undef GO
  AVSR_CHECK_LAUNCH();
  if (g.splitk > 1) {
    const long total = (long)g.batch * g.M * g.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, g);
    AVSR_CHECK_LAUNCH();
  }
  gl.done(&g, 1, "single");
  return AVSR_OK;
}
