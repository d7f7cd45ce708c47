// Frame-resident MFMA convolutions of the lip-crop CNN (avsr/video.py:143-195: 3x3 SAME convolutions, stride 1 / 2, 3..64
// channels, on 36x36 .. 5x5 maps, B*T = 4800 frames per step).
//
// The maps of ONE frame are tiny (36x36x8 fp32 = 41 KB), so a workgroup stages whole frames (+ a one-pixel zero halo) in LDS and
// runs the convolution as an implicit GEMM on v_mfma_f32_16x16x4_f32 straight out of LDS:
//   rows  (M) = 16 output positions of the staged frames,
//   cols  (N) = 16 destination channels,
//   depth (K) = (tap, source channel), 16 per chunk = four 16-byte LDS reads per row; the weight fragments of a wave's column
//               tile stay in VGPRs for all frames.
// One kernel covers every data-path convolution through a tap list: out[a, b] = sum_t src[a*S + da_t, b*S + db_t] . W_t
//   forward, stride s:            S = s, taps (i - pt, j - pl)
//   data gradient, stride 1:      src = dy, taps (pt - i, pl - j), weights read transposed
//   data gradient, stride 2:      one launch per parity class (ph, pw) of the input pixels: the taps whose parity matches, OS = 2
// The weight gradient is the transposed GEMM (rows = (tap, cin), cols = cout, depth = positions) with the frame and its output
// gradient staged the same way; per-workgroup partial sums are reduced by avsr_colsum_final_launch.
// The epilogue of the data-path kernel can emit per-channel sum / sum-of-squares partials of what it wrote (batch-norm
// statistics of the producing convolution: removes two full passes over the map per batch norm).
#include "common.h"
#include "persist.h"
#include "prof.h"
#include "avsr_hip.h"

namespace avsr {

// exact n / d for n < 2^16, 0 < d < 2^16: one 32x32 -> high-32 multiply instead of the ~40-instruction integer division
// (d = 1 has no 32-bit magic: encoded as 0)
__device__ __forceinline__ int fdiv(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }
static inline unsigned fmagic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) / (unsigned)d) + 1ull); }
__device__ __forceinline__ unsigned fmagic_dev(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) / (unsigned)d) + 1ull); }

// Stage `nf4` 16-byte pieces global -> LDS with all of a thread's loads of a batch in flight before the first LDS store
// (one load, one store per loop trip left every trip exposed to the full memory latency: 40 trips per 36x36x8 frame).
template <class SrcOff, class DstOff>
__device__ __forceinline__ void stage4(const float* __restrict__ src, float* __restrict__ lds, int nf4, int tid, SrcOff so, DstOff dof) {
  for (int base = 0; base < nf4; base += 256 * 16) {
    f32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = base + u * 256 + tid;
      v[u] = idx < nf4 ? ld4(src + so(idx)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = base + u * 256 + tid;
      if (idx < nf4) st4(lds + dof(idx), v[u]);
    }
  }
}

// Row-structured staging of one [rows][rq pieces] block (16-byte pieces) into a pitched LDS image: a thread keeps its piece
// column and walks rows by pointer increments; up to 16 loads in flight before the first LDS store.
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, float* __restrict__ dst, int rows, int rq, int dst_pitch, int tid,
                                           int my_row, int my_p4, int rpp) {
  if (my_row < 0) return;
  const float* sp = src + ((long)my_row * rq + my_p4) * 4;
  float* dp = dst + my_row * dst_pitch + my_p4 * 4;
  const int sstep = rpp * rq * 4, dstep = rpp * dst_pitch;
  int r = my_row;
  while (r < rows) {
    f32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = (r + u * rpp < rows) ? ld4(sp + u * sstep) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 16; ++u) if (r + u * rpp < rows) st4(dp + u * dstep, v[u]);
    r += 16 * rpp; sp += 16 * sstep; dp += 16 * dstep;
  }
}

#define CG_MAXTAP 16
// One (wide) tap of the product's depth: source offset (da, db) from the row's base position; w[sp] = index of the kernel tap this
// source pixel meets for sub-position sp of the row, or -1 (zero weights).
struct CGTap { int da, db; short w[4]; };
static inline CGTap cgtap1(int da, int db, int widx) { CGTap t; t.da = da; t.db = db; t.w[0] = (short)widx; t.w[1] = t.w[2] = t.w[3] = -1; return t; }
struct CGArgs {
  const float* src; const float* w; const float* bias; float* dst; float* stats;
  const float* acc;                      // beta != 0: the map beta multiplies (NULL = dst itself); same shape / indexing as dst
  const float* res;                      // optional residual input, same shape / indexing as dst
  const float* bn_sc; const float* bn_sh; // non-NULL: the source is relu(src * bn_sc[c] + bn_sh[c]) applied while staging (BN-ReLU of the
                                          // consumer's loader: the normalised map is never written); halo / padding stays zero
  const float* res_sc; const float* res_sh; // same for the residual input (per destination channel)
  // Batch-norm backward, stage 1, fused into a data-gradient epilogue (bnb_x != NULL): dst is the gradient of y = relu(x*sc + sh) for the
  // pre-normalisation map x (same shape / indexing as dst).  The epilogue writes dz = dst * [x*sc + sh > 0] instead and emits, in the
  // statistics slots, the per-channel partial sums of dz and of dz*x -- what the separate two-map statistics pass of the batch-norm
  // backward computed (d beta = sum dz, d gamma = invstd * (sum dz*x - mean * sum dz)).
  const float* bnb_x; const float* bnb_sc; const float* bnb_sh;
  int N, SH, SW, Cs, CsL;
  int DH, DW, Cd;
  int OA, OB, S, OS, oh0, ow0;
  // Sub-position columns: a row of the product is a SUPER position (a, b) of nsp destination pixels, its columns are (sp, channel).
  //   nsp = 1: one pixel per row (SB = S, OSA = OSB = OS).
  //   nsp = 2 (8-channel stride-1 layers): two horizontally adjacent pixels share one row over the union of their windows (3 x 4
  //            taps): the 16 columns of a tile are all used (8 channels alone leave half of every MFMA multiplying padding).
  //   nsp = 4 (stride-2 data gradient): the four parity classes of a 2x2 destination cell in ONE launch: dy staged once, whole
  //            destination rows written instead of every other pixel per launch.
  // source base of row (a, b): (a*S, b*SB); destination pixel of (a, b, sp): (a*OSA + oh0 + sp_dh[sp], b*OSB + ow0 + sp_dw[sp]).
  int nsp, SB, OSA, OSB, lin;
  signed char sp_dh[4], sp_dw[4];
  int ntap, wmode, F;
  int dbg;                                // CONV_DEBUG builds only: bit 0 no stores, bit 1 no LDS operand reads, bit 2 no MFMAs
  float beta;
  unsigned m_opf, m_ob, m_rq, m_per, m_sw;   // division magics: positions per frame, OB, pieces per source row / per frame (Cs % 4 == 0),
                                          // or channels / floats per frame / SW (otherwise)
  CGTap tap[CG_MAXTAP];
};

// The product is computed TRANSPOSED: the weight fragments are the MFMA's A operand (rows = 16 destination columns), the staged
// activations its B operand (columns = 16 positions), so a lane of the result holds FOUR CONSECUTIVE CHANNELS of ONE position
// (D[4q + r][i] = out[position i][column 4q + r]): the epilogue is one 16-byte store per lane (and one 16-byte load per fused
// operand: residual, accumulate, batch-norm-backward map), bias / scale / shift are per-lane constants, and one address is computed
// per lane and tile instead of four.  (Round 2 computed D = X.W: four 4-byte stores per lane, 64-byte segments.)
// Tile addressing comes from two tables built in LDS once per workgroup (they are the same for every pass): rowtab[m] = LDS offset of
// the source window of product row m, dsttab[m] = byte offset of its destination cell (non-linear destinations) -- the per-tile
// divisions / multiplications (about 40 VALU + quarter-rate integer multiplies per tile) become one ds_read each, issued a tile ahead.
// MAXCH: K chunks (16 deep) held per wave
// EP: epilogue operands, bit 0 = an extra map (residual or batch-norm-backward x), bit 1 = accumulate onto the destination (beta != 0)
template <int MAXCH, bool CH4, int EP>
__global__ __launch_bounds__(256, 2) void conv_gen_kernel(const CGArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef CONV_DEBUG
  const long t_k0 = __builtin_readcyclecounter();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int Cs = A.Cs, CsL = A.CsL, Cd = A.Cd, C4 = CsL >> 2;
  const int PH = A.SH + 2, PW = A.SW + 2, fstride = PH * PW * CsL;
  const int KQ = A.ntap * C4, nch = (KQ + 3) >> 2;
  const int NC = A.nsp * Cd;                          // columns of the product: (sub-position, channel)
  const int NT = (NC + 15) >> 4;                      // 1, 2 or 4 column tiles; a wave keeps ONE
  const int nt = wave % NT, mslot = wave / NT, mstep = 4 / NT;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const bool lin = A.lin != 0;                         // destination index linear in (row, column) of the product
  const int opf = A.OA * A.OB;                         // rows (super positions) per frame
  const int pixf = A.DH * A.DW;                        // destination pixels per frame
  const int rows_all = A.F * opf, rows_pad = ((rows_all + 15) >> 4) << 4;
  int* const rowtab = reinterpret_cast<int*>(lds + A.F * fstride);
  int* const dsttab = rowtab + rows_pad;
  // The tap list, copied to LDS: the weight-fragment set-up below indexes it per LANE (a chunk's four k-quads can lie in different taps),
  // and a lane-dependent index into the kernel argument became, per chunk, a global load of the entry followed by a wait for EVERYTHING
  // outstanding -- 18 serialised memory round trips, 22 k cycles (9 us) of every 18-chunk launch (profiles/r04_conv_deep_dissection.txt).
  int* const taptab = dsttab + rows_pad;               // [CG_MAXTAP][4]: da, db, w[0] | w[1] << 16, w[2] | w[3] << 16
  const int tapword = reinterpret_cast<const int*>(A.tap)[tid & (CG_MAXTAP * 4 - 1)];      // (in flight under the zeroing loop)

  // zero the LDS once: halos (and the padded 4th channel of a 3-channel source) stay zero, frames overwrite the interior
  for (int idx = tid; idx < A.F * fstride; idx += 256) lds[idx] = 0.f;
  if (tid < CG_MAXTAP * 4) taptab[tid] = tapword;
#ifdef CONV_DEBUG
  const long t_s1 = __builtin_readcyclecounter();
#endif
  // tile tables (rows beyond the staged frames read the window of row 0 and are never written out)
  for (int m = tid; m < rows_pad; m += 256) {
    int ro = 0, dt = 4;
    if (m < rows_all) {
      const int f = fdiv(m, A.m_opf), r = m - f * opf, a = fdiv(r, A.m_ob), b = r - a * A.OB;
      ro = f * fstride + ((a * A.S + 1) * PW + (b * A.SB + 1)) * CsL;
      const int ph = a * A.OSA + A.oh0, pw = b * A.OSB + A.ow0;
      // low bits (offsets are multiples of 16 bytes): 1 = no pixel below this one (odd map heights), 2 = none to its right, 4 = no pixel at all
      dt = lin ? (m * NC) << 2
               : (ph < A.DH && pw < A.DW) ? ((((f * A.DH + ph) * A.DW + pw) * Cd) << 2) | (ph + 1 >= A.DH ? 1 : 0) | (pw + 1 >= A.DW ? 2 : 0) : 4;
    }
    rowtab[m] = ro;
    dsttab[m] = dt;
  }

  __syncthreads();                                      // (tap table visible)
#ifdef CONV_DEBUG
  const long t_s2 = __builtin_readcyclecounter();
#endif
  // weight fragments of this wave's column tile (A operand: row i of the fragment = column nt*16 + i of the product) + per-chunk LDS
  // offsets of this lane's k-quad
  const __amdgpu_buffer_rsrc_t w_rs = make_rsrc(A.w);
  f32x4 wreg[MAXCH];
  int koff[MAXCH];
  {
    const int cn = nt * 16 + i;
    const bool colok = cn < NC;
    const int sp = colok ? cn / Cd : 0, co = colok ? cn - sp * Cd : 0;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
      const int kq = 4 * c + q;
      const bool in = c < nch && kq < KQ;
      const int t = in ? kq / C4 : 0, cs4 = in ? kq - t * C4 : 0;
      const i32x4_ tp = *reinterpret_cast<const i32x4_*>(taptab + 4 * t);
      koff[c] = in ? (tp[0] * PW + tp[1]) * CsL + cs4 * 4 : 0;
      const int wpair = (sp & 2) ? tp[3] : tp[2];
      const int widx = (int)(short)((sp & 1) ? (wpair >> 16) : (wpair & 0xffff));
      // unconditional buffer loads (out-of-range offset = 0): all fragments of the wave are in flight together
      f32x4 wv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int cs = cs4 * 4 + e;
        const long wo = A.wmode ? ((long)widx * Cd + co) * Cs + cs : ((long)widx * Cs + cs) * Cd + co;
        wv[e] = ldb1(w_rs, (in && colok && widx >= 0 && cs < Cs) ? (int)(wo * 4) : P_OOB);
      }
      wreg[c] = wv;
    }
  }
#ifdef CONV_DEBUG
  const long t_s3 = __builtin_readcyclecounter();
#endif
  // result columns of this lane: cD0 .. cD0 + 3 = four consecutive channels of sub-position spD
  const int cD0 = nt * 16 + q * 4;
  const bool colok = cD0 < NC;
  const int spD = colok ? cD0 / Cd : 0, coD = colok ? cD0 - spD * Cd : 0;
  int sdh = 0, sdw = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { sdh = (k == spD) ? (int)A.sp_dh[k] : sdh; sdw = (k == spD) ? (int)A.sp_dw[k] : sdw; }
  const f32x4 bias4 = (A.bias && colok) ? ld4(A.bias + coD) : zero4;
  // ONE extra epilogue operand map (same indexing as dst) with one scale / shift pair: the residual (forward; optionally a lazily
  // normalised map) or the batch-norm-backward map x (data gradients) -- never both, so they share registers
  const bool bnb = A.bnb_x != nullptr;
  const bool rbn = !bnb && A.res_sc != nullptr;
  const float* const e_map = bnb ? A.bnb_x : A.res;
  const float* const e_scp = bnb ? A.bnb_sc : A.res_sc;
  const float* const e_shp = bnb ? A.bnb_sh : A.res_sh;
  const f32x4 esc4 = ((bnb || rbn) && colok) ? ld4(e_scp + coD) : zero4;
  const f32x4 esh4 = ((bnb || rbn) && colok) ? ld4(e_shp + coD) : zero4;
  f32x4 ssum = zero4, ssq = zero4;
  const __amdgpu_buffer_rsrc_t e_rs = make_rsrc(e_map), dst_rs = make_rsrc(A.dst), acc_rs = make_rsrc(A.acc ? A.acc : A.dst);
  // byte offset of this lane's 16 bytes inside a tile / a destination cell
  const int lane_o = lin ? cD0 * 4 : ((sdh * A.DW + sdw) * Cd + coD) * 4;

  // staging role of this thread: piece column st_p4 of rows st_row, st_row + st_rpp, ...  (threads beyond rpp*rq idle)
  const int st_rq = (A.SW * Cs) >> 2;
  const int st_rpp = st_rq > 0 ? (256 / st_rq > 0 ? 256 / st_rq : 1) : 1;
  const int st_row = (st_rq > 0 && st_rq <= 256) ? (tid < st_rpp * st_rq ? fdiv(tid, A.m_rq) : -1) : -1;
  const int st_p4 = st_row >= 0 ? tid - st_row * st_rq : 0;
  f32x4 bsc = {1.f, 1.f, 1.f, 1.f}, bsh = zero4;
  const bool bn_on = CH4 && A.bn_sc != nullptr;
  if (bn_on && st_row >= 0) { const int cb = (st_p4 * 4) % Cs; bsc = ld4(A.bn_sc + cb); bsh = ld4(A.bn_sh + cb); }
  const int rowf = A.SW * Cs;                          // floats per source row

  // ---- software pipeline over passes: the NEXT pass's frames travel memory -> registers while the current pass runs on the matrix
  // pipe; registers -> LDS between two barriers.  A thread's pieces of a pass: (frame f, row st_row + k*st_rpp, column st_p4) for
  // 4-channel-multiple sources; 16-byte runs of the contiguous frames (element-wise scatter on store) for the 3-channel crops.
  constexpr int PF = CH4 ? (MAXCH > 9 ? 10 : 12) : 4;               // (the 18-chunk instantiation has 72 registers of weights: 10 pieces)
  f32x4 pre[PF];
  constexpr bool c4 = CH4;
  const int ppf = c4 ? (A.SH + st_rpp - 1) / st_rpp : 0;            // pieces per frame per thread (row-structured)
  const unsigned m_ppf = fmagic_dev(ppf > 0 ? ppf : 1);
  const int per3 = A.SH * A.SW * Cs;                                // floats per frame (3-channel path)
  // (hidden loads, see persist.h: left to the compiler the whole prefetch is waited for BEFORE the tile loop it should overlap)
  const i32x4_ src_rs = make_rsrc_words(A.src);                     // (source maps stay below 2 GB: checked by the host)
  // Frames of this workgroup: an EVEN share [n_begin, n_end) of the N frames, walked in passes of FP <= F frames.  (Passes of F frames dealt
  // round-robin left the busiest workgroup with ceil(passes / grid) * F frames: 4800 frames of a 9x9 map, F = 4, 512 workgroups = 12 frames
  // against 9.4 on average -- the kernel ends with its slowest workgroup: profiles/r04_conv_deep_dissection.txt, max vs mean cycles.)
  // (F = 1, the 36x36 maps: single frames dealt round-robin as before -- the same maximum, and neighbouring workgroups stream
  // neighbouring frames: measured 3-4 % faster there than 512 separate ranges)
  const int fs_per = A.N / (int)gridDim.x, fs_extra = A.N - fs_per * (int)gridDim.x;
  const int fs_cnt = fs_per + ((int)blockIdx.x < fs_extra ? 1 : 0), fs_np = (fs_cnt + A.F - 1) / A.F;
  const bool fs_rr = A.F == 1;
  const int FP = fs_rr ? 1 : (fs_np > 0 ? (fs_cnt + fs_np - 1) / fs_np : A.F);
  const int n_begin = fs_rr ? (int)blockIdx.x : (int)blockIdx.x * fs_per + min((int)blockIdx.x, fs_extra);
  const int n_end = fs_rr ? A.N : n_begin + fs_cnt, n_step = fs_rr ? (int)gridDim.x : FP;
  auto fetch = [&](int n0) {
    const int fcur = min(FP, n_end - n0);
    if (c4) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int f = fdiv(u, m_ppf), k = u - f * ppf, r = st_row + k * st_rpp;
        // branch-free: pieces this thread does not have use an out-of-range offset (the load returns zeros)
        ldb4_hidden(pre[u], src_rs, (st_row >= 0 && f < fcur && r < A.SH) ? (int)((((unsigned)(n0 + f) * A.SH + r) * rowf + st_p4 * 4) * 4u) : P_OOB);
      }
    } else {
      const unsigned so = (unsigned)n0 * per3 * 4u;
      const int tot4 = (fcur * per3) >> 2;
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int idx = u * 256 + tid;
        ldb4_hidden(pre[u], src_rs, idx < tot4 ? (int)(so + idx * 16u) : P_OOB);
      }
    }
  };
  auto commit = [&](int n0) {
    const int fcur = min(FP, n_end - n0);
    vm_wait_all();
#pragma unroll
    for (int u = 0; u < PF; ++u) vm_landed(pre[u]);
    if (c4) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int f = fdiv(u, m_ppf), k = u - f * ppf, r = st_row + k * st_rpp;
        if (st_row >= 0 && f < fcur && r < A.SH) {
          f32x4 v = pre[u];
          if (bn_on) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], bsc[e], bsh[e]), 0.f);
          }
          st4(lds + f * fstride + ((r + 1) * PW + 1) * CsL + st_p4 * 4, v);
        }
      }
    } else {
      const int tot = fcur * per3, tot4 = tot >> 2;
      auto put = [&](int e, float v) {
        const int f = fdiv(e, A.m_per), r = e - f * per3, px = fdiv(r, A.m_rq), c = r - px * Cs, h = fdiv(px, A.m_sw), pw = px - h * A.SW;
        lds[f * fstride + ((h + 1) * PW + pw + 1) * CsL + c] = v;
      };
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int idx = u * 256 + tid;
        if (idx < tot4) { put(idx * 4, pre[u][0]); put(idx * 4 + 1, pre[u][1]); put(idx * 4 + 2, pre[u][2]); put(idx * 4 + 3, pre[u][3]); }
      }
      const float* sp = A.src + (long)n0 * per3;
      for (int e = tot4 * 4 + tid; e < tot; e += 256) put(e, sp[e]);
    }
  };
  int n0 = n_begin;
#ifdef CONV_DEBUG
  long t_pro = 0, t_b1 = 0, t_commit = 0, t_b2 = 0, t_comp = 0, t_mark = __builtin_readcyclecounter();
  const long t_start = t_mark;
#define CG_STAMP(acc) { const long t_now = __builtin_readcyclecounter(); acc += t_now - t_mark; t_mark = t_now; }
#else
#define CG_STAMP(acc)
#endif
  if (n0 < n_end) fetch(n0);
  CG_STAMP(t_pro)
  for (; n0 < n_end; n0 += n_step) {
    const int fcur = min(FP, n_end - n0);
    __syncthreads();                                    // previous pass has finished reading the LDS (first pass: tables written)
    CG_STAMP(t_b1)
    commit(n0);
    CG_STAMP(t_commit)
    __syncthreads();
    CG_STAMP(t_b2)
    if (n0 + n_step < n_end) fetch(n0 + n_step);      // in flight during the MFMAs below
    // ---- implicit GEMM over the staged frames ----
    const int Mtot = fcur * opf, mtiles = (Mtot + 15) >> 4;
    const unsigned pass_o = (unsigned)((long)n0 * pixf * Cd * 4);     // (destination maps stay below 2 GB: checked by the host)
    constexpr int CB = MAXCH <= 6 ? MAXCH : 3;          // K chunks per block of LDS reads (9 and 18 are multiples of 3)
    constexpr bool XT = MAXCH <= 6;                     // whole tile in one block: the NEXT tile's reads run under this tile's MFMAs
    f32x4 bufa[CB], bufb[CB];
    // table entries travel one tile (window offsets: two tiles) ahead of their use
    const int last = mtiles - 1;
    int rb_cur = rowtab[(mslot < mtiles ? mslot : 0) * 16 + i];
    int rb_nxt = rowtab[(mslot + mstep < mtiles ? mslot + mstep : (last > 0 ? last : 0)) * 16 + i];
    int dt_cur = dsttab[(mslot < mtiles ? mslot : 0) * 16 + i];
    if (XT && mslot < mtiles) {
#pragma unroll
      for (int c = 0; c < CB; ++c) bufa[c] = ld4(lds + rb_cur + koff[c]);
    }
    // one tile: `cur` holds its first block of operands (XT: the whole tile), `nxt` receives the next block / the next tile
    // epilogue of one tile from its two accumulator chains: bias, fused operands, 16-byte store, statistics
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    auto epilogue = [&](const f32x4 a0, const f32x4 a1, const int dbo, const bool ok, const f32x4 ev, const f32x4 ov) {
      // D layout: D[4q + r][i] = out[position mt*16 + i][column cD0 + r]
      f32x4 v = (a0 + a1) + bias4;
      if ((EP & 1) && !bnb) {
        if (rbn) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += fmaxf(fmaf(ev[r], esc4[r], esh4[r]), 0.f);
        } else v += ev;
      }
      if (EP & 2) v += A.beta * ov;
      if ((EP & 1) && bnb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(ev[r], esc4[r], esh4[r]) > 0.f ? v[r] : 0.f;
      }
#ifdef CONV_DEBUG
      if (!(A.dbg & 1))
#endif
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), dst_rs, dbo, 0, 0);
      // statistics are accumulated unconditionally (a few operations); only their final write is conditional
      const f32x4 vs = ok ? v : zero4;
      ssum += vs;
      ssq += ((EP & 1) && bnb) ? vs * ev : vs * vs;
    };
    // The tile bodies are branch-free (one basic block each: the instruction scheduler -- and the scheduling groups below -- only see
    // a block): table entries come from the LDS for every destination layout, lanes without an output cell get the out-of-range
    // bit OR-ed into their offset (loads return zero, the store is dropped by the buffer bounds check).
    auto dest = [&](const int mt, const int dt, bool& ok) {
      const int bad = (dt & 4) | ((dt & 1) & sdh) | (((dt >> 1) & 1) & sdw);
      ok = (mt * 16 + i < Mtot) & colok & (bad == 0);
      return (int)(pass_o + (unsigned)((dt & ~15) + lane_o)) | (ok ? 0 : P_OOB);
    };
    auto tile = [&](f32x4 (&cur)[CB], f32x4 (&nxt)[CB], const int mt) {          // MAXCH > 6: blocks of CB chunks, epilogue in place
      const int t1 = mt + mstep < mtiles ? mt + mstep : last;
      const int dt_nxt = dsttab[t1 * 16 + i];
      const float* base = lds + rb_cur;
      bool ok;
      const int dbo = dest(mt, dt_cur, ok);
      f32x4 ev = zero4, ov = zero4;
      if (EP & 1) ev = ldb4(e_rs, dbo);
      if (EP & 2) ov = ldb4(acc_rs, dbo);
      // Chunks beyond the layer's depth read offset 0 against zero weights (no branch around any read).  The reads of the next
      // block are issued BEFORE this block's MFMAs and kept there by the scheduling barrier: left to itself the compiler sank each
      // read to just ahead of its first use, one exposed LDS round trip per chunk.
      f32x4 acc0 = zero4, acc1 = zero4;
#pragma unroll
      for (int c = 0; c < CB; ++c) cur[c] = ld4(base + koff[c]);
#pragma unroll
      for (int b0 = 0; b0 < MAXCH; b0 += CB) {
#pragma unroll
        for (int c = 0; c < CB; ++c)
          if (b0 + CB + c < MAXCH) nxt[c] = ld4(base + koff[b0 + CB + c]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CB; ++c)
          if (b0 + c < MAXCH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {               // two accumulator chains, alternating: no MFMA waits for its predecessor
              if (e & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[b0 + c][e], cur[c][e], acc1, 0, 0, 0);
              else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[b0 + c][e], cur[c][e], acc0, 0, 0, 0);
              // MFMAs keep their source order (everything else may move across): left alone the scheduler issues one chain after
              // the other -- twelve dependent MFMAs at 40 cycles each instead of 32
              __builtin_amdgcn_sched_barrier(0x7F6);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CB; ++c) cur[c] = nxt[c];
      }
      epilogue(acc0, acc1, dbo, ok, ev, ov);
      rb_cur = rb_nxt;
      rb_nxt = rowtab[(mt + 2 * mstep < mtiles ? mt + 2 * mstep : last) * 16 + i];
      dt_cur = dt_nxt;
    };
    // MAXCH <= 6 (the large maps: a tile is only 4 * MAXCH MFMAs, and the ~60 other instructions of a tile -- tables, addresses, the
    // next tile's operand reads, the epilogue -- were issued by the same wave BEFORE / AFTER them, costing about as many cycles as
    // the MFMAs themselves).  Here a tile's epilogue is deferred by one tile, which makes everything in the body independent of the
    // MFMAs being issued, and the scheduling groups spread it between them.
    f32x4 pa0 = zero4, pa1 = zero4, pev = zero4, pov = zero4;
    int pdbo = P_OOB, poki = 0;                       // (the flag travels as an integer: one VGPR, no lane-mask phi around the loop)
    auto tile_x = [&](f32x4 (&cur)[CB], f32x4 (&nxt)[CB], const int mt) {
      // The issue order is written out by hand: one MFMA, one slice of the other work, a full scheduling fence.  (Asked through
      // sched_group_barrier the compiler kept most of the other work behind the last MFMA; left alone it also issues one
      // accumulator chain after the other -- dependent MFMAs at 40 cycles each instead of 32.)
      const int t1 = mt + mstep < mtiles ? mt + mstep : last, t2 = mt + 2 * mstep < mtiles ? mt + 2 * mstep : last;
      int rb_nn = 0, dt_nxt = 0, dbo = P_OOB, oki = 0;
      bool ok = false;
      f32x4 ev = zero4, ov = zero4, v = zero4, vs = zero4, acc0 = zero4, acc1 = zero4;
      constexpr int W_RD = 1, W_DEST = W_RD + CB, W_EPI = W_DEST + 2, NWORK = W_EPI + 9, NM = 4 * MAXCH;
      auto work = [&](const int k) {
        if (k == 0) { rb_nn = rowtab[t2 * 16 + i]; dt_nxt = dsttab[t1 * 16 + i]; }
        else if (k < W_DEST) {                          // the NEXT tile's operands (this tile's arrived during the previous one)
#ifdef CONV_DEBUG
          if (!(A.dbg & 2))
#endif
          nxt[k - W_RD] = ld4(lds + rb_nxt + koff[k - W_RD]);
        } else if (k == W_DEST) {
          dbo = dest(mt, dt_cur, ok);
          oki = ok ? 1 : 0;
          asm volatile("" : "+v"(dbo), "+v"(oki));       // computed HERE (otherwise sunk behind the MFMAs, to its first use in the next tile)
        }
        else if (k == W_DEST + 1) {
          if (EP & 1) ev = ldb4(e_rs, dbo);
          if (EP & 2) ov = ldb4(acc_rs, dbo);
        }
        // ---- the PREVIOUS tile's epilogue, in slices (same arithmetic and order as epilogue()) ----
        else if (k == W_EPI) v = pa0 + pa1;
        else if (k == W_EPI + 1) v += bias4;
        else if (k == W_EPI + 2) {
          if ((EP & 1) && !bnb) {
            if (rbn) { v[0] += fmaxf(fmaf(pev[0], esc4[0], esh4[0]), 0.f); v[1] += fmaxf(fmaf(pev[1], esc4[1], esh4[1]), 0.f); }
            else v += pev;
          }
        } else if (k == W_EPI + 3) {
          if ((EP & 1) && !bnb && rbn) { v[2] += fmaxf(fmaf(pev[2], esc4[2], esh4[2]), 0.f); v[3] += fmaxf(fmaf(pev[3], esc4[3], esh4[3]), 0.f); }
        } else if (k == W_EPI + 4) {
          if (EP & 2) v += A.beta * pov;
        } else if (k == W_EPI + 5) {
          if ((EP & 1) && bnb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(pev[r], esc4[r], esh4[r]) > 0.f ? v[r] : 0.f;
          }
        } else if (k == W_EPI + 6) {
#ifdef CONV_DEBUG
          if (!(A.dbg & 1))
#endif
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), dst_rs, pdbo, 0, 0);
        } else if (k == W_EPI + 7) { vs = poki != 0 ? v : zero4; ssum += vs; }
        else if (k == W_EPI + 8) ssq += ((EP & 1) && bnb) ? vs * pev : vs * vs;
      };
#pragma unroll
      for (int k = 0; k < NM; ++k) {
        const int c = k >> 2, e = k & 3;
#ifdef CONV_DEBUG
        if (!(A.dbg & 4)) {
#endif
        if (e & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[c][e], cur[c][e], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[c][e], cur[c][e], acc0, 0, 0, 0);
#ifdef CONV_DEBUG
        }
#endif
        if (k < NWORK) work(k);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int k = NM; k < NWORK; ++k) work(k);         // (shallow layers: more slices than MFMAs)
      pa0 = acc0; pa1 = acc1; pdbo = dbo; poki = oki; pev = ev; pov = ov;
      rb_cur = rb_nxt; rb_nxt = rb_nn; dt_cur = dt_nxt;
    };
    if (XT) {                                           // ping-pong between two operand buffers: no register copies between tiles
      for (int mt = mslot; mt < mtiles; mt += 2 * mstep) {
        tile_x(bufa, bufb, mt);
        if (mt + mstep < mtiles) tile_x(bufb, bufa, mt + mstep);
        else break;
      }
      epilogue(pa0, pa1, pdbo, poki != 0, pev, pov);    // the pass's last tile
    } else {
      for (int mt = mslot; mt < mtiles; mt += mstep) tile(bufa, bufb, mt);
    }
    CG_STAMP(t_comp)
  }
#ifdef CONV_DEBUG
  if ((A.dbg & 8) && A.stats && lane == 0) {           // per-wave cycle counts behind the statistics partials (probe allocates them)
    float* o = A.stats + (long)gridDim.x * 2 * Cd + ((long)blockIdx.x * 4 + wave) * 8;
    o[0] = (float)t_pro; o[1] = (float)t_b1; o[2] = (float)t_commit; o[3] = (float)t_b2; o[4] = (float)t_comp;
    o[5] = (float)(__builtin_readcyclecounter() - t_start);
    o[6] = (float)(t_start - t_k0);                      // set-up: LDS zeroing, tile tables, weight fragments
    if (A.dbg & 16) { o[0] = (float)(t_s1 - t_k0); o[1] = (float)(t_s2 - t_s1); o[2] = (float)(t_s3 - t_s2); o[3] = (float)(t_start - t_s3); }   // set-up split
  }
#endif
  if (A.stats) {
    // per-channel partials of this workgroup: sum over the 16 positions of a lane group first (lanes q*16 .. q*16+15 hold the same
    // four channels), then over the (wave, q, r) slots that carry the channel
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[r] = group16_sum(ssum[r]); ssq[r] = group16_sum(ssq[r]); }
    float* red = lds;                                   // [4 waves][4 q][2][4]
    if (i == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { red[((wave * 4 + q) * 2) * 4 + r] = ssum[r]; red[((wave * 4 + q) * 2 + 1) * 4 + r] = ssq[r]; }
    }
    __syncthreads();
    if (tid < Cd) {
      float s = 0.f, s2 = 0.f;
      for (int w = 0; w < 4; ++w)
        for (int qq = 0; qq < 4; ++qq)
          for (int r = 0; r < 4; ++r) {
            const int cD = (w % NT) * 16 + qq * 4 + r;
            if (cD < NC && cD % Cd == tid) { s += red[((w * 4 + qq) * 2) * 4 + r]; s2 += red[((w * 4 + qq) * 2 + 1) * 4 + r]; }
          }
      A.stats[(long)blockIdx.x * 2 * Cd + tid] = s;
      A.stats[(long)blockIdx.x * 2 * Cd + Cd + tid] = s2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight gradient: part[blk][(t*Ci + ci)*Co + co] = sum over the block's frames and positions of x[pos(t)][ci] * dy[pos][co]
#define WG_PAD 2          // floats of padding per LDS pixel in the weight-gradient kernel (bank spreading, see the kernel)
// workgroups per CU the weight-gradient kernel is compiled for, by accumulator tiles per wave (its register budget): the stages of its
// per-frame pipeline hide behind one another only ACROSS waves (profiles/r05_wgrad_ablation.txt), so the small forms take every wave
// the registers and the LDS allow
#define WG_WPC(tiles) ((tiles) > 16 ? 1 : ((tiles) <= 3 ? 4 : ((tiles) <= 6 ? 3 : 2)))
struct WGArgs {
  const float* x; const float* dy; float* part;
  int N, H, W, Ci, CiL, Ho, Wo, Co, S, pt, pl, F;
  int SW;                                 // source step along W per output column (0 = S): 2 for the pixel-pair form, see conv_bwd_weight_impl
  int pad;                                // floats of padding per LDS pixel (WG_PAD; 1 where three workgroups of a 36x36 frame share a CU)
  unsigned m_opf, m_wo, m_rq, m_per, m_w;
  int t0, nt, kw;                         // taps t0 .. t0+nt-1 of a kw x kw kernel: rows (t - t0, ci) of this launch's slab
  int slab, want_bias;                    // floats per workgroup partial: nt*Ci*Co (+ Co column sums of dy = the bias gradient)
  const float* bn_sc; const float* bn_sh; // BN-ReLU applied to x while staging (see CGArgs)
  int dbg;                                // CONV_DEBUG builds: bit 3 = per-wave cycle stamps behind the partial slabs
  // FOLD (round 5): dy is not stored -- it is the batch-norm backward's output gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)) =
  // k1[c]*dz + k2[c]*y + k3[c] of the convolution's OWN output y (avsr_bn_bwd_finalize's coefficient vectors fk [3*fC]), evaluated while
  // the operand is fetched: `dy` points at dz, fy at y (same layout).  For a convolution whose only gradient consumer is this kernel
  // (layer 0: its input are the lip crops) the stand-alone avsr_bn_bwd_apply pass over three maps disappears.
  // fdx != NULL: the evaluated gradient is also WRITTEN there (same layout) for the layer's data gradient, which runs after this kernel:
  // every element is fetched by exactly one lane, so the stand-alone pass is replaced by one store per operand.
  const float* fy; const float* fk; int fC; float* fdx;
};

// 8 destination channels, stride 1, linear destination (the 36x36 layers: layer 0 and residual block 0, forward and data gradient): the
// product on v_mfma_f32_4x4x1_16B_f32 with cbsz = 4.  All 16 blocks of an instruction share the A block `abid` = 4 destination channels
// at ONE k (a weight VGPR holds 4 channels x 16 k: the whole 3x3x8x8 kernel is ten registers), the B operand is one staged activation
// per lane: ONE LANE = ONE OUTPUT POSITION, 64 positions per wave tile, two instructions (channels 0-3 / 4-7) per k.  Against the
// pixel-pair 16x16x4 form: no 3x4-tap union (12 taps computed for 9: 18 instead of 24 matrix-pipe cycles per position), a lane's result is
// its position's 8 channels (32 contiguous bytes), and the per-tile address / table / epilogue instructions serve 64 positions, not 32.
template <int NTAP, int C4T, bool CH4, int EP>
__global__ __launch_bounds__(256, 2) void conv_q4_kernel(const CGArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NQ = NTAP * C4T, NA = (4 * NQ + 15) / 16, MAXCH = 6;
  const int Cs = A.Cs, CsL = A.CsL;
  constexpr int Cd = 8;
  const int PH = A.SH + 2, PW = A.SW + 2, fstride = PH * PW * CsL;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int opf = A.OA * A.OB, pixf = A.DH * A.DW;
  const int rows_all = A.F * opf, rows_pad = ((rows_all + 63) >> 6) << 6;
  int* const rowtab = reinterpret_cast<int*>(lds + A.F * fstride);
  int* const taptab = rowtab + rows_pad;               // the tap list in LDS (see conv_gen_kernel): indexed per lane below
  const int tapword = reinterpret_cast<const int*>(A.tap)[tid & (CG_MAXTAP * 4 - 1)];      // (in flight under the zeroing loop)
  for (int idx = tid; idx < A.F * fstride; idx += 256) lds[idx] = 0.f;
  if (tid < CG_MAXTAP * 4) taptab[tid] = tapword;
  // window of position m: LDS offset of its centre pixel (floats), and -- 8-channel sources -- the swizzle phase of that pixel's column
  for (int m = tid; m < rows_pad; m += 256) {
    int ro = ((PW + 1) * CsL) | (1 << 24);             // rows beyond the staged frames: the window of position 0 (never written out)
    if (m < rows_all) {
      const int f = fdiv(m, A.m_opf), r = m - f * opf, a = fdiv(r, A.m_ob), b = r - a * A.OB;
      ro = (f * fstride + ((a + 1) * PW + (b + 1)) * CsL) | ((b + 1) << 24);      // (frames stay below 64 KB: offsets below 2^14 floats)
    }
    rowtab[m] = ro;
  }
  __syncthreads();                                      // (tap table visible)
  // weights: lane (block bb, i) holds W[channel 4h + i][k = 16a + bb]; k = 4*(tap*C4T + quad) + e
  float wA[2][NA];
  {
    const int bb = lane >> 2, i = lane & 3;
    const __amdgpu_buffer_rsrc_t w_rs = make_rsrc(A.w);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int k = 16 * a + bb, kq = k >> 2, e = k & 3;
      const bool in = kq < NQ;
      const int t = in ? kq / C4T : 0, cs = (kq - t * C4T) * 4 + e;
      const int widx = in ? (int)(short)(taptab[4 * t + 2] & 0xffff) : -1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int co = 4 * h + i;
        const long wo = A.wmode ? ((long)widx * Cd + co) * Cs + cs : ((long)widx * Cs + cs) * Cd + co;
        wA[h][a] = ldb1(w_rs, (in && widx >= 0 && cs < Cs) ? (int)(wo * 4) : P_OOB);
      }
    }
  }
  // per-(tap, quad) window offsets (wave-uniform) and the tap's column shift (for the swizzle phase)
  int koff[NQ];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
#pragma unroll
    for (int c = 0; c < C4T; ++c) koff[t * C4T + c] = (A.tap[t].da * PW + A.tap[t].db) * CsL;
  }
  const f32x4 bias0 = A.bias ? ld4(A.bias) : zero4, bias1 = A.bias ? ld4(A.bias + 4) : zero4;
  const bool bnb = A.bnb_x != nullptr;
  const bool rbn = !bnb && A.res_sc != nullptr;
  const float* const e_map = bnb ? A.bnb_x : A.res;
  const float* const e_scp = bnb ? A.bnb_sc : A.res_sc;
  const float* const e_shp = bnb ? A.bnb_sh : A.res_sh;
  const f32x4 esc0 = (bnb || rbn) ? ld4(e_scp) : zero4, esc1 = (bnb || rbn) ? ld4(e_scp + 4) : zero4;
  const f32x4 esh0 = (bnb || rbn) ? ld4(e_shp) : zero4, esh1 = (bnb || rbn) ? ld4(e_shp + 4) : zero4;
  f32x4 ssum0 = zero4, ssum1 = zero4, ssq0 = zero4, ssq1 = zero4;
  const __amdgpu_buffer_rsrc_t e_rs = make_rsrc(e_map), dst_rs = make_rsrc(A.dst), acc_rs = make_rsrc(A.acc ? A.acc : A.dst);
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));

  // staging role of this thread: piece column st_p4 of rows st_row, st_row + st_rpp, ...  (threads beyond rpp*rq idle)
  const int st_rq = (A.SW * Cs) >> 2;
  const int st_rpp = st_rq > 0 ? (256 / st_rq > 0 ? 256 / st_rq : 1) : 1;
  const int st_row = (st_rq > 0 && st_rq <= 256) ? (tid < st_rpp * st_rq ? fdiv(tid, A.m_rq) : -1) : -1;
  const int st_p4 = st_row >= 0 ? tid - st_row * st_rq : 0;
  f32x4 bsc = {1.f, 1.f, 1.f, 1.f}, bsh = zero4;
  const bool bn_on = CH4 && A.bn_sc != nullptr;
  if (bn_on && st_row >= 0) { const int cb = (st_p4 * 4) % Cs; bsc = ld4(A.bn_sc + cb); bsh = ld4(A.bn_sh + cb); }
  const int rowf = A.SW * Cs;                          // floats per source row

  // ---- software pipeline over passes: the NEXT pass's frames travel memory -> registers while the current pass runs on the matrix
  // pipe; registers -> LDS between two barriers.  A thread's pieces of a pass: (frame f, row st_row + k*st_rpp, column st_p4) for
  // 4-channel-multiple sources; 16-byte runs of the contiguous frames (element-wise scatter on store) for the 3-channel crops.
  constexpr int PF = CH4 ? 12 : 4;
  f32x4 pre[PF];
  constexpr bool c4 = CH4;
  const int ppf = c4 ? (A.SH + st_rpp - 1) / st_rpp : 0;            // pieces per frame per thread (row-structured)
  const unsigned m_ppf = fmagic_dev(ppf > 0 ? ppf : 1);
  const int per3 = A.SH * A.SW * Cs;                                // floats per frame (3-channel path)
  // (hidden loads, see persist.h: left to the compiler the whole prefetch is waited for BEFORE the tile loop it should overlap)
  const i32x4_ src_rs = make_rsrc_words(A.src);                     // (source maps stay below 2 GB: checked by the host)
  // Frames of this workgroup: an EVEN share [n_begin, n_end) of the N frames, walked in passes of FP <= F frames.  (Passes of F frames dealt
  // round-robin left the busiest workgroup with ceil(passes / grid) * F frames: 4800 frames of a 9x9 map, F = 4, 512 workgroups = 12 frames
  // against 9.4 on average -- the kernel ends with its slowest workgroup: profiles/r04_conv_deep_dissection.txt, max vs mean cycles.)
  // (F = 1, the 36x36 maps: single frames dealt round-robin as before -- the same maximum, and neighbouring workgroups stream
  // neighbouring frames: measured 3-4 % faster there than 512 separate ranges)
  const int fs_per = A.N / (int)gridDim.x, fs_extra = A.N - fs_per * (int)gridDim.x;
  const int fs_cnt = fs_per + ((int)blockIdx.x < fs_extra ? 1 : 0), fs_np = (fs_cnt + A.F - 1) / A.F;
  const bool fs_rr = A.F == 1;
  const int FP = fs_rr ? 1 : (fs_np > 0 ? (fs_cnt + fs_np - 1) / fs_np : A.F);
  const int n_begin = fs_rr ? (int)blockIdx.x : (int)blockIdx.x * fs_per + min((int)blockIdx.x, fs_extra);
  const int n_end = fs_rr ? A.N : n_begin + fs_cnt, n_step = fs_rr ? (int)gridDim.x : FP;
  auto fetch = [&](int n0) {
    const int fcur = min(FP, n_end - n0);
    if (c4) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int f = fdiv(u, m_ppf), k = u - f * ppf, r = st_row + k * st_rpp;
        // branch-free: pieces this thread does not have use an out-of-range offset (the load returns zeros)
        ldb4_hidden(pre[u], src_rs, (st_row >= 0 && f < fcur && r < A.SH) ? (int)((((unsigned)(n0 + f) * A.SH + r) * rowf + st_p4 * 4) * 4u) : P_OOB);
      }
    } else {
      const unsigned so = (unsigned)n0 * per3 * 4u;
      const int tot4 = (fcur * per3) >> 2;
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int idx = u * 256 + tid;
        ldb4_hidden(pre[u], src_rs, idx < tot4 ? (int)(so + idx * 16u) : P_OOB);
      }
    }
  };
  auto commit = [&](int n0) {
    const int fcur = min(FP, n_end - n0);
    vm_wait_all();
#pragma unroll
    for (int u = 0; u < PF; ++u) vm_landed(pre[u]);
    if (c4) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int f = fdiv(u, m_ppf), k = u - f * ppf, r = st_row + k * st_rpp;
        if (st_row >= 0 && f < fcur && r < A.SH) {
          f32x4 v = pre[u];
          if (bn_on) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], bsc[e], bsh[e]), 0.f);
          }
          // (8-channel sources: the two 16-byte halves of a pixel are swapped in every other group of four pixels, so that 64 lanes
          // reading the same half of 64 consecutive pixels -- 32-byte stride -- hit every LDS bank row fully: see the tile loop)
          const int px1 = 1 + (st_p4 >> 1), qd = (st_p4 & 1) ^ (C4T == 2 ? (px1 >> 2) & 1 : 0);
          st4(lds + f * fstride + ((r + 1) * PW) * CsL + (C4T == 2 ? px1 * 8 + qd * 4 : 4 + st_p4 * 4), v);
        }
      }
    } else {
      const int tot = fcur * per3, tot4 = tot >> 2;
      auto put = [&](int e, float v) {
        const int f = fdiv(e, A.m_per), r = e - f * per3, px = fdiv(r, A.m_rq), c = r - px * Cs, h = fdiv(px, A.m_sw), pw = px - h * A.SW;
        lds[f * fstride + ((h + 1) * PW + pw + 1) * CsL + c] = v;
      };
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int idx = u * 256 + tid;
        if (idx < tot4) { put(idx * 4, pre[u][0]); put(idx * 4 + 1, pre[u][1]); put(idx * 4 + 2, pre[u][2]); put(idx * 4 + 3, pre[u][3]); }
      }
      const float* sp = A.src + (long)n0 * per3;
      for (int e = tot4 * 4 + tid; e < tot; e += 256) put(e, sp[e]);
    }
  };
  int n0 = n_begin;
#ifdef CONV_DEBUG
  long t_pro = 0, t_b1 = 0, t_commit = 0, t_b2 = 0, t_comp = 0, t_mark = __builtin_readcyclecounter();
  const long t_start = t_mark;
#define CG_STAMP(acc) { const long t_now = __builtin_readcyclecounter(); acc += t_now - t_mark; t_mark = t_now; }
#else
#define CG_STAMP(acc)
#endif
  if (n0 < n_end) fetch(n0);
  CG_STAMP(t_pro)
  for (; n0 < n_end; n0 += n_step) {
    const int fcur = min(FP, n_end - n0);
    __syncthreads();                                    // previous pass has finished reading the LDS (first pass: tables written)
    CG_STAMP(t_b1)
    commit(n0);
    CG_STAMP(t_commit)
    __syncthreads();
    CG_STAMP(t_b2)
    if (n0 + n_step < n_end) fetch(n0 + n_step);      // in flight during the MFMAs below
    // ---- the staged frames' positions, 64 per wave tile ----
    const int Mtot = fcur * opf, mtiles = (Mtot + 63) >> 6;
    const unsigned pass_o = (unsigned)((long)n0 * pixf * Cd * 4);     // (destination maps stay below 2 GB: checked by the host)
    // Operand reads run HALF A TILE ahead of their MFMAs: reads of the second half are issued before the first half's MFMAs, reads of the
    // NEXT tile's first half before the second half's MFMAs (issued chunk by chunk, each read's LDS round trip sat in front of its eight
    // MFMAs: 4x the tile's matrix-pipe time).
    constexpr int NH = (NQ + 1) / 2;                     // chunks of the first half
    f32x4 xq[NQ];
    int h0[3], h1[3];                                   // (LDS float offsets: integer arithmetic keeps the reads in the LDS address space)
    // 8-channel sources: the two halves of a pixel are swapped in every other group of four columns (see commit): per tap COLUMN
    // (db = -1, 0, +1) the offsets of half 0 / half 1, once per tile; tap t reads column t % 3 (forward) or 2 - t % 3 (flipped taps of
    // the data gradient: A.wmode, checked by the host)
    auto window = [&](const int rt) {
      const int rb = rt & 0xffffff, bcol = rt >> 24;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int o = C4T == 2 ? (((bcol + d - 1) >> 2) & 1) << 2 : 0;
        h0[d] = rb + o; h1[d] = rb + 4 - o;
      }
      if (A.wmode) { const int x0 = h0[0]; h0[0] = h0[2]; h0[2] = x0; const int x1 = h1[0]; h1[0] = h1[2]; h1[2] = x1; }
    };
#define Q4C_LD(KQ_) xq[KQ_] = ld4(lds + (((KQ_) % C4T ? h1[((KQ_) / C4T) % 3] : h0[((KQ_) / C4T) % 3]) + koff[KQ_]));
#define Q4C_E(KQ_, E_)                                                                                              \
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[0][(4 * (KQ_) + (E_)) >> 4], x4[E_], acc0, 4, (4 * (KQ_) + (E_)) & 15, 0);   \
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[1][(4 * (KQ_) + (E_)) >> 4], x4[E_], acc1, 4, (4 * (KQ_) + (E_)) & 15, 0);
#define Q4C_MFMA(KQ_) { const f32x4 x4 = xq[KQ_]; Q4C_E(KQ_, 0) Q4C_E(KQ_, 1) Q4C_E(KQ_, 2) Q4C_E(KQ_, 3) }
    int rt_cur = rowtab[(wave < mtiles ? wave : 0) * 64 + lane];
    window(rt_cur);
    if constexpr (NQ == 18) { Q4C_LD(0) Q4C_LD(1) Q4C_LD(2) Q4C_LD(3) Q4C_LD(4) Q4C_LD(5) Q4C_LD(6) Q4C_LD(7) Q4C_LD(8) } else { Q4C_LD(0) Q4C_LD(1) Q4C_LD(2) Q4C_LD(3) Q4C_LD(4) }
    for (int mt = wave; mt < mtiles; mt += 4) {
      const int rt_nxt = rowtab[(mt + 4 < mtiles ? mt + 4 : mt) * 64 + lane];
      const int m = mt * 64 + lane;
      const bool ok = m < Mtot;
      const int dbo = (int)(pass_o + (unsigned)(m * 32)) | (ok ? 0 : P_OOB);
      f32x4 ev0 = zero4, ev1 = zero4, ov0 = zero4, ov1 = zero4;
      if (EP & 1) { ev0 = ldb4(e_rs, dbo); ev1 = ldb4(e_rs, dbo | 16); }
      if (EP & 2) { ov0 = ldb4(acc_rs, dbo); ov1 = ldb4(acc_rs, dbo | 16); }
      f32x4 acc0 = zero4, acc1 = zero4;
      if constexpr (NQ == 18) { Q4C_LD(9) Q4C_LD(10) Q4C_LD(11) Q4C_LD(12) Q4C_LD(13) Q4C_LD(14) Q4C_LD(15) Q4C_LD(16) Q4C_LD(17) } else { Q4C_LD(5) Q4C_LD(6) Q4C_LD(7) Q4C_LD(8) }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NQ == 18) { Q4C_MFMA(0) Q4C_MFMA(1) Q4C_MFMA(2) Q4C_MFMA(3) Q4C_MFMA(4) Q4C_MFMA(5) Q4C_MFMA(6) Q4C_MFMA(7) Q4C_MFMA(8) } else { Q4C_MFMA(0) Q4C_MFMA(1) Q4C_MFMA(2) Q4C_MFMA(3) Q4C_MFMA(4) }
      __builtin_amdgcn_sched_barrier(0);
      window(rt_nxt);                                   // the next tile's first half (the last tile re-reads its own: unused)
      if constexpr (NQ == 18) { Q4C_LD(0) Q4C_LD(1) Q4C_LD(2) Q4C_LD(3) Q4C_LD(4) Q4C_LD(5) Q4C_LD(6) Q4C_LD(7) Q4C_LD(8) } else { Q4C_LD(0) Q4C_LD(1) Q4C_LD(2) Q4C_LD(3) Q4C_LD(4) }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NQ == 18) { Q4C_MFMA(9) Q4C_MFMA(10) Q4C_MFMA(11) Q4C_MFMA(12) Q4C_MFMA(13) Q4C_MFMA(14) Q4C_MFMA(15) Q4C_MFMA(16) Q4C_MFMA(17) } else { static_assert(NQ == 9, "3x3 taps, 4- or 8-channel sources"); Q4C_MFMA(5) Q4C_MFMA(6) Q4C_MFMA(7) Q4C_MFMA(8) }
      __builtin_amdgcn_sched_barrier(0);
      // epilogue: this lane's position, channels 0-3 / 4-7
      f32x4 v0 = acc0 + bias0, v1 = acc1 + bias1;
      if ((EP & 1) && !bnb) {
        if (rbn) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { v0[r] += fmaxf(fmaf(ev0[r], esc0[r], esh0[r]), 0.f); v1[r] += fmaxf(fmaf(ev1[r], esc1[r], esh1[r]), 0.f); }
        } else { v0 += ev0; v1 += ev1; }
      }
      if (EP & 2) { v0 += A.beta * ov0; v1 += A.beta * ov1; }
      if ((EP & 1) && bnb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v0[r] = fmaf(ev0[r], esc0[r], esh0[r]) > 0.f ? v0[r] : 0.f;
          v1[r] = fmaf(ev1[r], esc1[r], esh1[r]) > 0.f ? v1[r] : 0.f;
        }
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v0), dst_rs, dbo, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v1), dst_rs, dbo | 16, 0, 0);
      const f32x4 s0 = ok ? v0 : zero4, s1 = ok ? v1 : zero4;
      ssum0 += s0; ssum1 += s1;
      ssq0 += ((EP & 1) && bnb) ? s0 * ev0 : s0 * s0;
      ssq1 += ((EP & 1) && bnb) ? s1 * ev1 : s1 * s1;
    }
    CG_STAMP(t_comp)
  }
#undef Q4C_LD
#undef Q4C_MFMA
#undef Q4C_E
#ifdef CONV_DEBUG
  if ((A.dbg & 8) && A.stats && lane == 0) {           // per-wave cycle counts behind the statistics partials (tools/conv_dissect.py)
    float* o = A.stats + (long)gridDim.x * 2 * Cd + ((long)blockIdx.x * 4 + wave) * 8;
    o[0] = (float)t_pro; o[1] = (float)t_b1; o[2] = (float)t_commit; o[3] = (float)t_b2; o[4] = (float)t_comp;
    o[5] = (float)(__builtin_readcyclecounter() - t_start);
  }
#endif
  if (A.stats) {
    // per-channel partials of this workgroup: every lane holds all 8 channels of its positions
    __syncthreads();
    float* red = lds;                                   // [4 waves][16]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a0 = wave_sum(ssum0[r]), a1 = wave_sum(ssum1[r]), q0 = wave_sum(ssq0[r]), q1 = wave_sum(ssq1[r]);
      if (lane == 0) { red[wave * 16 + r] = a0; red[wave * 16 + 4 + r] = a1; red[wave * 16 + 8 + r] = q0; red[wave * 16 + 12 + r] = q1; }
    }
    __syncthreads();
    if (tid < 16) A.stats[(long)blockIdx.x * 2 * Cd + tid] = (red[tid] + red[16 + tid]) + (red[32 + tid] + red[48 + tid]);
  }
}

// MT: row tiles (16 rows of (tap, ci)) held per wave; NTC: column tiles; CH4: 4-channel-multiple input (row-structured staging)
// RS (row split): the four waves own DIFFERENT row tiles (wave w: rows [w*MT*16, (w+1)*MT*16)) and each walks every chunk, instead of
// all waves sharing MT row tiles and splitting the chunks: a deep layer whose (tap, channel) rows exceed one wave's accumulators then
// takes ONE launch -- its input staged once -- instead of one per tap group, and no cross-wave reduction at the end.
template <int MT, int NTC, bool CH4, bool RS = false, int FOLD = 0>
__global__ __launch_bounds__(256, WG_WPC(MT * NTC)) void conv_wgrad_kernel(const WGArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int Ci = A.Ci, CiL = A.CiL, Co = A.Co;
  // LDS pixel stride = channels + 2 floats: the A operand is read 4 bytes at a time by lanes (row = (tap, channel), q = position group);
  // with a stride of 8 / 16 / 32 / 64 floats the four position groups hit the same banks (4-5 LDS cycles per read, SQ_LDS_BANK_CONFLICT =
  // 53 % of the LDS-active cycles); + 2 floats spreads them (2 cycles per read, the minimum for 64 lanes on 32 banks).  Staging stores
  // become 8-byte pairs.
  const int CsP = CiL + A.pad;
  const int PH = A.H + 2, PW = A.W + 2, xstride = PH * PW * CsP;
  const int opf = A.Ho * A.Wo;
  float* const xs = lds;                               // [F][PH][PW][CiL]   (the output gradient is read straight from memory:
                                                       //  every value is used once per row tile, 16 lanes = 64 contiguous bytes)
  const int Mrows = A.nt * CiL;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int idx = tid; idx < A.F * xstride; idx += 256) xs[idx] = 0.f;

  // row (t, ci) of this lane in every row tile -> LDS offset of its tap / channel (rows beyond 9*CiL read offset 0: their
  // accumulators are never written out)
  int roff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = (RS ? wave * MT * 16 : 0) + mt * 16 + i;
    const bool ok = row < Mrows;
    const int tl = ok ? row / CiL : 0, ci = ok ? row - tl * CiL : 0, t = A.t0 + tl, ti = t / A.kw, tj = t - ti * A.kw;
    roff[mt] = ok ? ((ti - A.pt) * PW + (tj - A.pl)) * CsP + ci : 0;
  }
  f32x4 acc[MT][NTC];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) acc[mt][nt] = zero4;
  const int rowf = A.W * Ci;
  const int st_rq = rowf >> 2;
  const int st_rpp = (CH4 && st_rq > 0) ? (256 / st_rq > 0 ? 256 / st_rq : 1) : 1;
  const int st_row = CH4 ? (tid < st_rpp * st_rq ? fdiv(tid, A.m_rq) : -1) : -1;
  const int st_p4 = st_row >= 0 ? tid - st_row * st_rq : 0;
  const int st_pad = (CH4 && st_row >= 0) ? (st_p4 / (Ci >> 2)) * A.pad : 0;      // padding floats ahead of this piece's pixel in its LDS row
  const bool st_odd = (A.pad & 1) != 0;                                           // odd pixel stride: 4-byte staging stores
  f32x4 bsc = {1.f, 1.f, 1.f, 1.f}, bsh = zero4;
  const bool bn_on = CH4 && A.bn_sc != nullptr;
  if (bn_on && st_row >= 0) { const int cb = (st_p4 * 4) % Ci; bsc = ld4(A.bn_sc + cb); bsh = ld4(A.bn_sh + cb); }
  float bsum[NTC];
#pragma unroll
  for (int nt = 0; nt < NTC; ++nt) bsum[nt] = 0.f;
  constexpr int PF = CH4 ? 12 : 4;
  f32x4 pre[PF];
  const int ppf = CH4 ? (A.H + st_rpp - 1) / st_rpp : 0;
  const unsigned m_ppf = fmagic_dev(ppf > 0 ? ppf : 1);
  const int per3 = A.H * A.W * Ci;
  // Frames of this workgroup: an EVEN share [n_begin, n_end) of the N frames, walked in passes of FP <= F frames.  (Passes of F frames dealt
  // round-robin left the busiest workgroup with ceil(passes / grid) * F frames: 4800 frames of a 9x9 map, F = 4, 512 workgroups = 12 frames
  // against 9.4 on average -- the kernel ends with its slowest workgroup: profiles/r04_conv_deep_dissection.txt, max vs mean cycles.)
  // (F = 1, the 36x36 maps: single frames dealt round-robin as before -- the same maximum, and neighbouring workgroups stream
  // neighbouring frames: measured 3-4 % faster there than 512 separate ranges)
  const int fs_per = A.N / (int)gridDim.x, fs_extra = A.N - fs_per * (int)gridDim.x;
  const int fs_cnt = fs_per + ((int)blockIdx.x < fs_extra ? 1 : 0), fs_np = (fs_cnt + A.F - 1) / A.F;
  const bool fs_rr = A.F == 1;
  const int FP = fs_rr ? 1 : (fs_np > 0 ? (fs_cnt + fs_np - 1) / fs_np : A.F);
  const int n_begin = fs_rr ? (int)blockIdx.x : (int)blockIdx.x * fs_per + min((int)blockIdx.x, fs_extra);
  const int n_end = fs_rr ? A.N : n_begin + fs_cnt, n_step = fs_rr ? (int)gridDim.x : FP;
  auto fetch = [&](int n0) {
    const int fcur = min(FP, n_end - n0);
    if (CH4) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int f = fdiv(u, m_ppf), k = u - f * ppf, r = st_row + k * st_rpp;
        pre[u] = (st_row >= 0 && f < fcur && r < A.H) ? ld4(A.x + ((long)(n0 + f) * A.H + r) * rowf + st_p4 * 4) : zero4;
      }
    } else {
      const float* sp = A.x + (long)n0 * per3;
      const int tot4 = (fcur * per3) >> 2;
#pragma unroll
      for (int u = 0; u < PF; ++u) { const int idx = u * 256 + tid; pre[u] = idx < tot4 ? ld4(sp + idx * 4) : zero4; }
    }
  };
  auto commit = [&](int n0) {
    const int fcur = min(FP, n_end - n0);
    if (CH4) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int f = fdiv(u, m_ppf), k = u - f * ppf, r = st_row + k * st_rpp;
        if (st_row >= 0 && f < fcur && r < A.H) {
          f32x4 v = pre[u];
          if (bn_on) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], bsc[e], bsh[e]), 0.f);
          }
          float* const dpx = xs + f * xstride + ((r + 1) * PW + 1) * CsP + st_pad + st_p4 * 4;
          if (st_odd) { dpx[0] = v[0]; dpx[1] = v[1]; dpx[2] = v[2]; dpx[3] = v[3]; }
          else {
            *reinterpret_cast<float2*>(dpx) = float2{v[0], v[1]};
            *reinterpret_cast<float2*>(dpx + 2) = float2{v[2], v[3]};
          }
        }
      }
    } else {
      const int tot = fcur * per3, tot4 = tot >> 2;
      auto put = [&](int e, float v) {
        const int f = fdiv(e, A.m_per), r = e - f * per3, px = fdiv(r, A.m_rq), c = r - px * Ci, h = fdiv(px, A.m_w), pw = px - h * A.W;
        xs[f * xstride + ((h + 1) * PW + pw + 1) * CsP + c] = v;
      };
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int idx = u * 256 + tid;
        if (idx < tot4) { put(idx * 4, pre[u][0]); put(idx * 4 + 1, pre[u][1]); put(idx * 4 + 2, pre[u][2]); put(idx * 4 + 3, pre[u][3]); }
      }
      const float* sp = A.x + (long)n0 * per3;
      for (int e = tot4 * 4 + tid; e < tot; e += 256) put(e, sp[e]);
    }
  };

  // depth = output positions; a chunk = 16 positions of ONE frame (the last chunk of a frame is partial), lane quad q takes
  // positions 4q .. 4q+3 of the chunk; chunks are dealt to the waves round-robin.
  const __amdgpu_buffer_rsrc_t dy_rs = make_rsrc(A.dy);
  const __amdgpu_buffer_rsrc_t fy_rs = make_rsrc(FOLD ? A.fy : A.dy);
  const __amdgpu_buffer_rsrc_t fdx_rs = make_rsrc(FOLD == 2 ? A.fdx : A.part);
  float fk1[NTC], fk2[NTC], fk3[NTC];                    // FOLD: coefficients of this lane's column(s)
#pragma unroll
  for (int nt = 0; nt < NTC; ++nt) {
    fk1[nt] = 1.f; fk2[nt] = 0.f; fk3[nt] = 0.f;
    if (FOLD) {
      const int col = nt * 16 + i, ch = col % A.fC;
      const bool okc = col < Co;
      fk1[nt] = okc ? A.fk[ch] : 0.f; fk2[nt] = okc ? A.fk[A.fC + ch] : 0.f; fk3[nt] = okc ? A.fk[2 * A.fC + ch] : 0.f;
    }
  }
  const int cpf = (opf + 15) >> 4;                       // chunks per frame
  const unsigned m_cpf = fmagic_dev(cpf);
  int n0 = n_begin;
  // dissection builds (tools/wgrad_ablate.sh: -DWG_ABLATE=mask, compile-time so that the rest of the code is generated as shipped):
  // 1 no dy loads, 2 no LDS operand reads, 4 no MFMAs, 16 no frame staging.  Results: profiles/r05_wgrad_ablation.txt
#ifdef WG_ABLATE
#define WG_ABL(b) (((WG_ABLATE) & (b)) != 0)
#else
#define WG_ABL(b) false
#endif
#ifdef CONV_DEBUG
  long w_b1 = 0, w_commit = 0, w_b2 = 0, w_comp = 0, w_mark = __builtin_readcyclecounter();
  const long w_start = w_mark;
#define WG_STAMP(acc) { const long t_now = __builtin_readcyclecounter(); acc += t_now - w_mark; w_mark = t_now; }
#else
#define WG_STAMP(acc)
#endif
  if (n0 < n_end) fetch(n0);
  for (; n0 < n_end; n0 += n_step) {
    const int fcur = min(FP, n_end - n0);
    __syncthreads();
    WG_STAMP(w_b1)
    if (!WG_ABL(16)) commit(n0);
    WG_STAMP(w_commit)
    const int kch = fcur * cpf;
    const unsigned dyo = (unsigned)((long)n0 * opf * Co * 4);     // [fcur][opf][Co]
    float bn[NTC][4], byn[FOLD ? NTC : 1][4];
    auto load_b = [&](int kc, float (&b)[NTC][4]) {
      // unconditional buffer loads (positions beyond the frame / columns beyond Co: out-of-range offset = 0)
      const int f = fdiv(kc, m_cpf), r0 = (kc - f * cpf) * 16 + q * 4;
      const unsigned o = dyo + (unsigned)(((f * opf + r0) * Co + i) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
          const int off = (r0 + e < opf && nt * 16 + i < Co) ? (int)(o + (unsigned)((e * Co + nt * 16) * 4)) : P_OOB;
          b[nt][e] = ldb1(dy_rs, WG_ABL(1) ? P_OOB : off);
          if (FOLD) byn[nt][e] = ldb1(fy_rs, WG_ABL(1) ? P_OOB : off);
        }
    };
    constexpr int KC0 = RS ? 0 : -1, KCS = RS ? 1 : 4;                // first chunk / chunk step of a wave
    const int kc0 = KC0 < 0 ? wave : KC0;
    if (kc0 < kch) load_b(kc0, bn);
    __syncthreads();
    WG_STAMP(w_b2)
    if (n0 + n_step < n_end && !WG_ABL(16)) fetch(n0 + n_step);      // next pass's frames: in flight during the MFMAs
    for (int kc = kc0; kc < kch; kc += KCS) {
      float av[MT][4], bv[NTC][4];
      const int f = fdiv(kc, m_cpf), r0 = (kc - f * cpf) * 16 + q * 4;
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bv[nt][e] = bn[nt][e];
          // (positions beyond the frame must stay zero: the constant term would otherwise enter the sums)
          if (FOLD) bv[nt][e] = (r0 + e < opf) ? fmaf(fk1[nt], bn[nt][e], fmaf(fk2[nt], byn[nt][e], fk3[nt])) : 0.f;
          if (FOLD == 2 && (!RS || wave == 0))
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bv[nt][e]), fdx_rs,
                                                  (r0 + e < opf && nt * 16 + i < Co) ? (int)(dyo + (unsigned)(((f * opf + r0 + e) * Co + nt * 16 + i) * 4)) : P_OOB, 0, 0);
        }
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt) bsum[nt] += (bv[nt][0] + bv[nt][1]) + (bv[nt][2] + bv[nt][3]);
      if (kc + KCS < kch) load_b(kc + KCS, bn);
      int ho = fdiv(min(r0, opf - 1), A.m_wo), wo = min(r0, opf - 1) - ho * A.Wo;
      const float* xf = xs + f * xstride + (PW + 1) * CsP;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* xb = xf + (ho * A.S * PW + wo * (A.SW ? A.SW : A.S)) * CsP;
        // positions beyond the frame read a clamped (finite) LDS address: their dy operand is zero (out-of-range buffer load), so the
        // product vanishes without a select per operand
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt][e] = WG_ABL(2) ? (float)roff[mt] : xb[roff[mt]];
        if (++wo == A.Wo) { wo = 0; ++ho; }
        if (ho >= A.Ho) { ho = A.Ho - 1; }                 // (only reached by out-of-range positions: masked above)
      }
      // all LDS reads of the chunk first, then its MFMAs (left alone the compiler waits for each read just ahead of its first use)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTC; ++nt) {
            if (WG_ABL(4)) acc[mt][nt][0] += av[mt][e] * bv[nt][e];
            else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][e], bv[nt][e], acc[mt][nt], 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
    WG_STAMP(w_comp)
  }
#ifdef CONV_DEBUG
  if ((A.dbg & 8) && lane == 0) {
    float* o = A.part + (long)gridDim.x * A.slab + ((long)blockIdx.x * 4 + wave) * 8;
    o[0] = (float)w_b1; o[1] = (float)w_commit; o[2] = (float)w_b2; o[3] = (float)w_comp; o[4] = (float)(__builtin_readcyclecounter() - w_start);
  }
#endif
  // cross-wave reduction (waves hold different depth slices of the same tiles), tile by tile through a 4 KB staging area, then
  // one partial per workgroup
  float* red = lds;                                     // [4][16][16]
  const int rr = tid >> 4, cc = tid & 15;
  if (RS) {                                             // every wave writes its own rows: C[4q + r][i] of tile (mt, nt)
    __syncthreads();                                    // (the staging area is free: the bias sums below reuse it)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wave * MT * 16 + mt * 16 + q * 4 + r;
        if (row < Mrows) {
          const int t = row / CiL, ci = row - t * CiL;
          if (ci < Ci) {
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt)
              if (nt * 16 + i < Co) A.part[(long)blockIdx.x * A.slab + ((long)t * Ci + ci) * Co + nt * 16 + i] = acc[mt][nt][r];
          }
        }
      }
  } else
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 16 + q * 4 + r) * 16 + i] = acc[mt][nt][r];
      __syncthreads();
      const int row = mt * 16 + rr, co = nt * 16 + cc;
      if (row < Mrows && co < Co) {
        const int t = row / CiL, ci = row - t * CiL;
        if (ci < Ci) A.part[(long)blockIdx.x * A.slab + ((long)t * Ci + ci) * Co + co] = (red[rr * 16 + cc] + red[(16 + rr) * 16 + cc]) + (red[(32 + rr) * 16 + cc] + red[(48 + rr) * 16 + cc]);
      }
    }
  if (A.want_bias) {                                    // column sums of dy: lanes (q, wave) hold disjoint positions of column nt*16 + i
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) red[((wave * 4 + q) * NTC + nt) * 16 + i] = (RS && wave) ? 0.f : bsum[nt];   // (RS: every wave saw every chunk)
    __syncthreads();
    if (tid < NTC * 16) {
      const int nt = tid >> 4, ci = tid & 15;
      float s = 0.f;
      for (int g = 0; g < 16; ++g) s += red[(g * NTC + nt) * 16 + ci];
      if (nt * 16 + ci < Co) A.part[(long)blockIdx.x * A.slab + (long)A.nt * Ci * Co + nt * 16 + ci] = s;
    }
  }
}

static int g_conv_mfma = 1;

}  // namespace avsr

namespace avsr {
bool slab_defer_push(const float* part, long ld, int nblk, int F, float* out, float* out2, int split, int kind, int Ci, float alpha, float beta,
                     hipStream_t s);
bool slab_deferring();
}
int avsr_colsum_final_launch(const float* part, int nblk, float* out, int F, float alpha, float beta, void* stream);

using namespace avsr;
#define S_(x) ((hipStream_t)(x))

extern "C" int avsr_conv_set_mfma(int32_t on) { g_conv_mfma = on ? 1 : 0; return AVSR_OK; }

static int cg_frames(int sh, int sw, int csl, int opf, int extra_floats_per_frame = 0) {
  const long per = (long)(sh + 2) * (sw + 2) * csl + extra_floats_per_frame;
  int F = (int)((63 * 1024 / 4) / per);                 // <= 63 KB of frames (+ tables <= 64 KB): two workgroups per CU
  if (F < 1) F = 1;
  int want = (1024 + opf - 1) / opf;                    // enough positions per pass to keep the four waves in row tiles
  if (want < 1) want = 1;
  if (F > want) F = want;
  if (F > 16) F = 16;
  return F;
}

// 8 destination channels, stride 1, 3x3 taps, whole map: the 4x4x1-MFMA kernel (conv_q4_kernel) instead of the pixel-pair form
static int g_conv_q4 = -1;
static bool cg_q4_ok(const CGArgs& A, int ntaps) {
  if (g_conv_q4 < 0) { const char* e = getenv("AVSR_CONV_Q4"); g_conv_q4 = e ? (atoi(e) != 0) : 1; }
  return g_conv_q4 && A.Cd == 8 && A.S == 1 && A.OS == 1 && !A.oh0 && !A.ow0 && A.DH == A.OA && A.DW == A.OB && ntaps == 9 &&
         (A.Cs == 8 || A.Cs == 3) && A.SW + 1 < 256;
}
// the kernel derives a tap's column from its index: db = t % 3 - 1 (forward) or 1 - t % 3 (flipped, A.wmode)
static bool cg_q4_taps_ok(const CGArgs& A) {
  for (int t = 0; t < 9; ++t)
    if (A.tap[t].db != (A.wmode ? 1 - t % 3 : t % 3 - 1)) return false;
  return true;
}
static int cg_launch_q4(CGArgs& A, hipStream_t s, int kind, double flops, bool dry) {
  int Fcap = 16;
  if (A.Cs == 8) {
    const int rq = A.SW * A.Cs / 4;
    if (rq > 256 || rq < 1) return AVSR_ERR_UNSUPPORTED;
    const int rpp = 256 / rq;
    Fcap = 12 / ((A.SH + rpp - 1) / rpp);
    A.m_rq = fmagic(rq); A.m_per = fmagic(A.SH * rq); A.m_sw = fmagic(A.SW);
  } else {
    Fcap = (4 * 256 * 4) / (A.SH * A.SW * A.Cs);
    while (Fcap > 0 && (long)Fcap * A.SH * A.SW * A.Cs >= 65536) --Fcap;
    A.m_rq = fmagic(A.Cs); A.m_per = fmagic(A.SH * A.SW * A.Cs); A.m_sw = fmagic(A.SW);
  }
  if (Fcap < 1) return AVSR_ERR_UNSUPPORTED;
  if ((long)A.N * A.DH * A.DW * A.Cd * 4 >= (1L << 31) || (long)A.N * A.SH * A.SW * A.Cs * 4 >= (1L << 31)) return AVSR_ERR_UNSUPPORTED;
  A.m_opf = fmagic(A.OA * A.OB); A.m_ob = fmagic(A.OB);
  const int opf = A.OA * A.OB;
  const size_t frame_b = sizeof(float) * (size_t)(A.SH + 2) * (A.SW + 2) * A.CsL;
  auto lds_bytes = [&](int F) { return F * frame_b + 4 * ((((size_t)F * opf + 63) / 64) * 64) + sizeof(CGTap) * CG_MAXTAP; };
  // frames per pass: (rounds of the busiest workgroup) x (64-position tiles of a pass per wave), as cg_launch
  double best = -1.0;
  int bestF = 0;
  for (int F = (Fcap < 16 ? Fcap : 16); F >= 1; --F) {
    if (lds_bytes(F) > 64 * 1024 || (long)F * frame_b / 4 >= (1 << 14) * 4 || (long)F * opf >= (1 << 24)) continue;
    const long units = (A.N + F - 1) / F, grid = units < 512 ? units : 512;
    const long cnt = (A.N + grid - 1) / grid, np = (cnt + F - 1) / F, fp = (cnt + np - 1) / np;
    const long tiles = (fp * opf + 63) / 64, per_wave = (tiles + 3) / 4;
    const double cost = (double)np * (double)(per_wave * 8 + 4);
    if (best < 0.0 || cost < 0.97 * best) { best = cost; bestF = F; }
  }
  if (!bestF) return AVSR_ERR_UNSUPPORTED;
#ifdef CONV_DEBUG
  { const char* e = getenv("AVSR_CONV_DBG"); A.dbg = e ? atoi(e) : 0; }
#endif
  A.F = bestF;
  size_t lds = lds_bytes(A.F);
  if (lds < 256) lds = 256;
  int grid = (A.N + A.F - 1) / A.F;
  static int cap = 0;
  if (!cap) { const char* e = getenv("AVSR_CONV_Q4_CAP"); cap = e ? atoi(e) : 512; }     // statistics buffers hold 512 partial rows (avsr_hip.h)
  if (grid > cap) grid = cap;
  if (dry) return grid;
  ProfScope ps(kind, s, flops);
  const bool emap = A.res != nullptr || A.bnb_x != nullptr;
  if (A.res && A.bnb_x) return AVSR_ERR_ARG;
  const int ep = (emap ? 1 : 0) | (A.beta != 0.f ? 2 : 0);
  if (A.Cs == 3) {
    if (ep) return AVSR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv_q4_kernel<9, 1, false, 0>), dim3(grid), dim3(256), lds, s, A);
  } else switch (ep) {
    case 0: hipLaunchKernelGGL((conv_q4_kernel<9, 2, true, 0>), dim3(grid), dim3(256), lds, s, A); break;
    case 1: hipLaunchKernelGGL((conv_q4_kernel<9, 2, true, 1>), dim3(grid), dim3(256), lds, s, A); break;
    case 2: hipLaunchKernelGGL((conv_q4_kernel<9, 2, true, 2>), dim3(grid), dim3(256), lds, s, A); break;
    default: hipLaunchKernelGGL((conv_q4_kernel<9, 2, true, 3>), dim3(grid), dim3(256), lds, s, A); break;
  }
  if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
  return grid;
}

static int cg_launch(CGArgs& A, hipStream_t s, int kind, double flops, bool dry = false) {
  if (A.nsp == 0) {                                     // one destination pixel per row of the product
    A.nsp = 1; A.SB = A.S; A.OSA = A.OS; A.OSB = A.OS;
    A.lin = (A.OS == 1 && A.oh0 == 0 && A.ow0 == 0 && A.DH == A.OA && A.DW == A.OB) ? 1 : 0;
  }
#ifdef CONV_DEBUG
  { const char* e = getenv("AVSR_CONV_DBG"); A.dbg = e ? atoi(e) : 0; }
#endif
  if (cg_q4_ok(A, A.ntap) && A.nsp == 1 && A.lin && cg_q4_taps_ok(A)) return cg_launch_q4(A, s, kind, flops, dry);
  const int KQ = A.ntap * (A.CsL / 4), nch = (KQ + 3) / 4;
  const int NT = (A.nsp * A.Cd + 15) / 16;
  if (!(NT == 1 || NT == 2 || NT == 4) || nch > 20) return AVSR_ERR_UNSUPPORTED;
  if (A.Cs % 4 && nch > 5) return AVSR_ERR_UNSUPPORTED;
  // a pass (F frames) must fit the prefetch registers of a staging thread
  int Fcap = 16;
  if (A.Cs % 4 == 0) {
    const int rq = A.SW * A.Cs / 4;
    if (rq > 256 || rq < 1) return AVSR_ERR_UNSUPPORTED;
    const int rpp = 256 / rq, pfmax = nch > 9 ? 10 : 12;    // prefetch registers of the instantiation that takes this depth
    Fcap = pfmax / ((A.SH + rpp - 1) / rpp);
    A.m_rq = fmagic(rq); A.m_per = fmagic(A.SH * rq); A.m_sw = fmagic(A.SW);
  } else {
    Fcap = (4 * 256 * 4) / (A.SH * A.SW * A.Cs);
    while (Fcap > 0 && (long)Fcap * A.SH * A.SW * A.Cs >= 65536) --Fcap;
    A.m_rq = fmagic(A.Cs); A.m_per = fmagic(A.SH * A.SW * A.Cs); A.m_sw = fmagic(A.SW);
  }
  if (Fcap < 1) return AVSR_ERR_UNSUPPORTED;
  if ((long)A.N * A.DH * A.DW * A.Cd * 4 >= (1L << 31)) return AVSR_ERR_UNSUPPORTED;   // 32-bit byte offsets of the epilogue's buffer loads
  if ((long)A.N * A.SH * A.SW * A.Cs * 4 >= (1L << 31)) return AVSR_ERR_UNSUPPORTED;   // ... and of the staging loads
  A.m_opf = fmagic(A.OA * A.OB); A.m_ob = fmagic(A.OB);
  const int opf = A.OA * A.OB, mstep = 4 / NT;
  const size_t frame_b = sizeof(float) * (size_t)(A.SH + 2) * (A.SW + 2) * A.CsL;
  auto lds_bytes = [&](int F) {
    const size_t rows_pad = (((size_t)F * opf + 15) / 16) * 16;
    return F * frame_b + 4 * rows_pad * 2 + sizeof(CGTap) * CG_MAXTAP;
  };
  // Frames per pass (upper bound F; a workgroup splits its even share of the frames into equal passes of FP <= F): the kernel's time is
  // (passes of a workgroup) x (row tiles of a pass per wave + the pass's fixed part).  Take the feasible F with the smallest product
  // (ties: the larger F, fewer barriers).
  auto pick = [&](size_t budget, int slots) {
    double best = -1.0;
    int bestF = 0;
    for (int F = (Fcap < 16 ? Fcap : 16); F >= 1; --F) {
      if (lds_bytes(F) > budget || (long)F * opf >= 65536) continue;
      // every workgroup walks an even share of the frames in passes of at most F (the kernel's FP)
      const long units = (A.N + F - 1) / F, grid = units < slots ? units : slots;
      const long cnt = (A.N + grid - 1) / grid, np = (cnt + F - 1) / F, fp = (cnt + np - 1) / np;
      const long tiles = (fp * opf + 15) / 16, per_wave = (tiles + mstep - 1) / mstep;
      const double cost = (double)np * (double)(per_wave * 8 + 16);     // (+ a pass's fixed part: staging, barriers, pipeline fill ~ two tiles)
      if (best < 0.0 || cost < 0.97 * best) { best = cost; bestF = F; }
    }
    return bestF;
  };
  A.F = pick(64 * 1024, 512);                            // two 256-thread workgroups per CU
  if (!A.F) return AVSR_ERR_UNSUPPORTED;
  size_t lds = lds_bytes(A.F);
  if (lds < sizeof(float) * 4 * 4 * 2 * 4) lds = sizeof(float) * 4 * 4 * 2 * 4;      // the statistics reduction's staging area
  int grid = (A.N + A.F - 1) / A.F;
  if (grid > 512) grid = 512;
  if (dry) return grid;
  ProfScope ps(kind, s, flops);
  const bool emap = A.res != nullptr || A.bnb_x != nullptr;
  if (A.res && A.bnb_x) return AVSR_ERR_ARG;            // one extra epilogue map at a time
  const int ep = (emap ? 1 : 0) | (A.beta != 0.f ? 2 : 0);
  if (A.Cs % 4 && ep != 0) return AVSR_ERR_UNSUPPORTED;   // the 3-channel crops are only ever a forward source
#define CG_ONE(M_, C_, E_)                                                                                          \
  {                                                                                                                 \
    hipLaunchKernelGGL((conv_gen_kernel<M_, C_, E_>), dim3(grid), dim3(256), lds, s, A);                            \
  }
#define CG_GO(M_, C_)                                                                                              \
  switch (ep) {                                                                                                    \
    case 0: CG_ONE(M_, C_, 0) break;                                                                               \
    case 1: CG_ONE(M_, C_, 1) break;                                                                               \
    case 2: CG_ONE(M_, C_, 2) break;                                                                               \
    default: CG_ONE(M_, C_, 3) break;                                                                              \
  }
  if (A.Cs % 4) CG_ONE(5, false, 0)
  else if (nch <= 2) { CG_GO(2, true) }
  else if (nch <= 6) { CG_GO(6, true) }
  else if (nch <= 9) { CG_GO(9, true) }
  else if (nch <= 18) { CG_GO(18, true) }
  else if (nch <= 20) { CG_GO(20, true) }
  else return AVSR_ERR_UNSUPPORTED;                     // K > 288 (64-channel sources) stays on im2col + GEMM
#undef CG_GO
#undef CG_ONE
  if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
  return grid;
}

// Forward (flip = 0) / stride-1 data gradient (flip = 1) of a 3x3 convolution, same contract as avsr_conv3x3.
// stats (may be NULL): >= 1024 * 2 * Co floats; returns the number of partial rows written through *nstat.
int avsr_conv3x3_mfma(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Ci, int Co, int stride, int pad_t,
                      int pad_l, int Ho, int Wo, int flip, float beta, float* stats, int* nstat, void* stream) {
  if (!g_conv_mfma) return AVSR_ERR_UNSUPPORTED;
  if (Co % 4 || (Ci % 4 && Ci >= 4) || pad_t > 1 || pad_l > 1 || (flip && stride != 1)) return AVSR_ERR_UNSUPPORTED;
  CGArgs A = {};
  A.src = x; A.w = w; A.bias = bias; A.dst = y; A.stats = stats;
  A.N = N; A.SH = H; A.SW = W; A.Cs = Ci; A.CsL = (Ci + 3) & ~3;
  A.DH = Ho; A.DW = Wo; A.Cd = Co; A.OA = Ho; A.OB = Wo; A.S = stride; A.OS = 1; A.oh0 = 0; A.ow0 = 0;
  A.ntap = 9; A.wmode = flip; A.beta = beta;
  for (int t = 0; t < 9; ++t) {
    const int i = t / 3, j = t % 3;
    if (!flip) A.tap[t] = cgtap1(i - pad_t, j - pad_l, t);
    else A.tap[t] = cgtap1(pad_t - i, pad_l - j, t);      // dx[h, w] += dy[h + pt - i, w + pl - j] . W[i, j]^T
  }
  A.F = cg_frames(H, W, A.CsL, Ho * Wo);
  const int rc = cg_launch(A, S_(stream), flip ? PROF_CONV_BWD_DATA : PROF_CONV_FWD, 2.0 * N * Ho * Wo * 9.0 * Ci * Co);
  if (rc < 0) return rc;
  if (nstat) *nstat = rc;
  return AVSR_OK;
}

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

// stride-2 data gradient: dx [N,H,W,Ci] (+)= from dy [N,Ho,Wo,Co]; one launch per parity class of the input pixels
int avsr_conv3x3_bwd_data_s2_mfma(const float* dy, const float* w, float* dx, int N, int H, int W, int Ci, int Co, int pad_t, int pad_l, int Ho,
                                  int Wo, float beta, void* stream) {
  if (!g_conv_mfma) return AVSR_ERR_UNSUPPORTED;
  if (Co % 4 || Ci % 4 || pad_t > 1 || pad_l > 1) return AVSR_ERR_UNSUPPORTED;
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      CGArgs A = {};
      A.src = dy; A.w = w; A.bias = nullptr; A.dst = dx; A.stats = nullptr;
      A.N = N; A.SH = Ho; A.SW = Wo; A.Cs = Co; A.CsL = Co;
      A.DH = H; A.DW = W; A.Cd = Ci; A.OA = (H - ph + 1) / 2; A.OB = (W - pw + 1) / 2; A.S = 1; A.OS = 2; A.oh0 = ph; A.ow0 = pw;
      A.wmode = 1; A.beta = beta;
      int nt = 0;
      for (int i = 0; i < 3; ++i) {
        if ((ph + pad_t - i) & 1) continue;
        for (int j = 0; j < 3; ++j) {
          if ((pw + pad_l - j) & 1) continue;
          // dx[2a+ph, 2b+pw] += dy[a + (ph+pt-i)/2, b + (pw+pl-j)/2] . W[i, j]^T   (arithmetic shift: -1/2 -> floor)
          A.tap[nt++] = cgtap1((ph + pad_t - i) >> 1, (pw + pad_l - j) >> 1, i * 3 + j);
        }
      }
      A.ntap = nt;
      if (A.OA <= 0 || A.OB <= 0) continue;
      if (nt == 0) return AVSR_ERR_UNSUPPORTED;
      A.F = cg_frames(Ho, Wo, A.CsL, A.OA * A.OB);
      const int rc = cg_launch(A, S_(stream), PROF_CONV_BWD_DATA, 2.0 * N * A.OA * A.OB * nt * (double)Ci * Co);
      if (rc < 0) return rc;
    }
  return AVSR_OK;
}

// weight gradient: dw[3,3,Ci,Co] = beta*dw + sum x (x) dy; scratch >= 256 * 9*Ci*Co floats
int avsr_conv3x3_bwd_weight_mfma(const float* x, const float* dy, float* dw, int N, int H, int W, int Ci, int Co, int stride, int pad_t,
                                 int pad_l, int Ho, int Wo, float beta, float* scratch, long scratch_floats, void* stream) {
  if (!g_conv_mfma) return AVSR_ERR_UNSUPPORTED;
  if (Co % 4 || (Ci % 4 && Ci >= 4) || pad_t > 1 || pad_l > 1) return AVSR_ERR_UNSUPPORTED;
  WGArgs A = {};
  A.x = x; A.dy = dy; A.part = scratch; A.N = N; A.H = H; A.W = W; A.Ci = Ci; A.CiL = (Ci + 3) & ~3; A.Ho = Ho; A.Wo = Wo; A.Co = Co;
  A.S = stride; A.pt = pad_t; A.pl = pad_l;
  A.t0 = 0; A.nt = 9; A.kw = 3; A.slab = 9 * Ci * Co; A.want_bias = 0; A.pad = WG_PAD;
  const int MT = (9 * A.CiL + 15) / 16, NTC = (Co + 15) / 16;
  if (MT > 18 || NTC > 2 || MT * NTC > 36) return AVSR_ERR_UNSUPPORTED;
  A.F = cg_frames(H, W, A.CiL + WG_PAD, Ho * Wo);
  const int nout = 9 * Ci * Co;
  A.m_opf = fmagic(Ho * Wo); A.m_wo = fmagic(Wo); A.m_w = fmagic(W);
  if (Ci % 4 == 0) {
    const int rq = W * Ci / 4;
    if (rq > 256 || rq < 1) return AVSR_ERR_UNSUPPORTED;
    const int rpp = 256 / rq;
    while (A.F > 1 && A.F * ((H + rpp - 1) / rpp) > 12) --A.F;
    if (A.F * ((H + rpp - 1) / rpp) > 12) return AVSR_ERR_UNSUPPORTED;
    A.m_rq = fmagic(rq); A.m_per = fmagic(H * rq);
  } else {
    while (A.F > 1 && (A.F * H * W * Ci / 4 + 255) / 256 > 4) --A.F;
    if ((A.F * H * W * Ci / 4 + 255) / 256 > 4 || (long)A.F * H * W * Ci >= 65536) return AVSR_ERR_UNSUPPORTED;
    A.m_rq = fmagic(Ci); A.m_per = fmagic(H * W * Ci);
  }
  if ((long)A.F * ((Ho * Wo + 15) / 16) >= 65536 || (long)N * Ho * Wo * Co * 4 >= (1L << 31)) return AVSR_ERR_UNSUPPORTED;
  const size_t red = sizeof(float) * 4 * 256;
  size_t lds = sizeof(float) * (size_t)A.F * (size_t)(H + 2) * (W + 2) * (A.CiL + WG_PAD);
  if (lds < red) lds = red;
  if (lds > 64 * 1024) return AVSR_ERR_UNSUPPORTED;
  int wpc = (int)((150 * 1024) / (lds + 512));
  if (wpc > 2) wpc = 2;
  if (wpc < 1) wpc = 1;
  if (nout > 2048) wpc = 1;                              // large kernels: the partial slabs, not the staging, are the traffic
  int grid = (N + A.F - 1) / A.F;
  if (grid > 256 * wpc) grid = 256 * wpc;
  if ((long)grid * nout > scratch_floats) grid = (int)(scratch_floats / nout);
  if (grid < 1) return AVSR_ERR_ARG;
  hipStream_t s = S_(stream);
  {
    ProfScope ps(PROF_CONV_BWD_WEIGHT, s, 2.0 * N * Ho * Wo * 9.0 * Ci * Co);
#define WG_GO(M_, N_, C_) hipLaunchKernelGGL((conv_wgrad_kernel<M_, N_, C_>), dim3(grid), dim3(256), lds, s, A)
    if (Ci % 4) { if (MT <= 3 && NTC == 1) WG_GO(3, 1, false); else return AVSR_ERR_UNSUPPORTED; }
    else if (NTC == 1) { if (MT <= 5) WG_GO(5, 1, true); else if (MT <= 9) WG_GO(9, 1, true); else WG_GO(18, 1, true); }
    else { if (MT <= 5) WG_GO(5, 2, true); else if (MT <= 9) WG_GO(9, 2, true); else WG_GO(18, 2, true); }
#undef WG_GO
    if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
  }
  if (avsr::slab_defer_push(scratch, nout, grid, nout, dw, nullptr, 0x7fffffff, 0, 0, 1.0f, beta, S_(stream))) return AVSR_OK;
  return avsr_colsum_final_launch(scratch, grid, dw, nout, 1.0f, beta, stream);
}

// =====================================================================================================================
// Descriptor API (include/avsr_hip.h: avsr_conv_desc): k = 1 or 3, stride 1 or 2, 3..64 channels; BN-ReLU of the input applied by the
// loader; kernels deeper than one wave's register budget run as several launches over tap groups (the later ones accumulate).
int avsr_colsum_final_launch_ld(const float* part, long ld, int nblk, float* out, int F, float alpha, float beta, void* stream);
int avsr_colsum_final_launch_split(const float* part, long ld, int nblk, float* out, float* out2, int split, int F, float alpha, float beta,
                                   void* stream);

static bool cd_ok(const avsr_conv_desc* c) {
  return c && c->N > 0 && (c->k == 1 || c->k == 3) && (c->stride == 1 || c->stride == 2) && c->Co % 4 == 0 && (c->Ci % 4 == 0 || c->Ci < 4) &&
         c->Ci > 0 && c->Co > 0 && c->Co <= 64 && c->pad_t >= 0 && c->pad_t <= 1 && c->pad_l >= 0 && c->pad_l <= 1 && (c->k == 3 || (c->pad_t == 0 && c->pad_l == 0));
}

// run the tap list in groups that fit the K chunks a wave holds; bias / beta on the first group, residual / statistics on the last
static int cg_run(CGArgs A, const CGTap* taps, int ntaps, hipStream_t s, int kind, double flops_per_tap, bool dry, int* grid_out) {
  const int C4 = A.CsL / 4;
  int G = (20 * 4) / C4;                                 // (64-channel sources: 5 + 4 taps on the 20- and 18-chunk instantiations; 4 + 4 + 1
  if (G < 1) return AVSR_ERR_UNSUPPORTED;                //  before, and the one-tap launch cost as much set-up and staging as the others)
  if (G > CG_MAXTAP) G = CG_MAXTAP;
  { const int ng = (ntaps + G - 1) / G; G = (ntaps + ng - 1) / ng; }     // even groups
  const float* res = A.res; float* stats = A.stats; const float* bias = A.bias; const float beta = A.beta;
  const float* acc = A.acc; const float* bnb_x = A.bnb_x;
  int grid = 0;
  for (int t0 = 0; t0 < ntaps; t0 += G) {
    const int nt = ntaps - t0 < G ? ntaps - t0 : G;
    const bool first = t0 == 0, last = t0 + nt >= ntaps;
    CGArgs B = A;
    for (int t = 0; t < nt; ++t) B.tap[t] = taps[t0 + t];
    B.ntap = nt;
    B.bias = first ? bias : nullptr; B.beta = first ? beta : 1.f; B.acc = first ? acc : nullptr;
    B.res = last ? res : nullptr; B.stats = last ? stats : nullptr; B.bnb_x = last ? bnb_x : nullptr;
    if (!last) { B.res_sc = nullptr; B.res_sh = nullptr; }
    const int rc = cg_launch(B, s, kind, flops_per_tap * nt, dry);
    if (rc < 0) return rc;
    grid = rc;
  }
  if (grid_out) *grid_out = grid;
  return AVSR_OK;
}

// Two horizontally adjacent destination pixels per row of the product (see CGArgs): rewrites the tap list of a stride-1 layer with 8
// destination channels into the wide taps of the pair.  Returns the number of wide taps, or 0 when the layer does not qualify.
static int cg_pair_taps(CGArgs& A, const CGTap* taps, int ntaps, CGTap* wide) {
  if (A.Cd != 8 || A.S != 1 || A.OS != 1 || (A.OB & 1) || A.oh0 || A.ow0 || A.DH != A.OA || A.DW != A.OB) return 0;
  if (cg_q4_ok(A, ntaps)) return 0;                  // the 4x4x1-MFMA kernel takes these layers, one pixel per lane
  int nw = 0;
  for (int t = 0; t < ntaps; ++t)
    for (int pp = 0; pp < 2; ++pp) {
      const int da = taps[t].da, db = taps[t].db + pp;   // pixel 2b + pp reads source column 2b + pp + db
      int k = 0;
      for (; k < nw; ++k)
        if (wide[k].da == da && wide[k].db == db) break;
      if (k == nw) {
        if (nw == CG_MAXTAP) return 0;
        wide[nw] = cgtap1(da, db, -1);
        ++nw;
      }
      wide[k].w[pp] = taps[t].w[0];
    }
  A.nsp = 2; A.SB = 2; A.OSA = 1; A.OSB = 2; A.lin = 1;
  A.sp_dh[0] = A.sp_dh[1] = 0; A.sp_dw[0] = 0; A.sp_dw[1] = 1;
  A.OB /= 2;
  return nw;
}

static int conv_fwd_impl(const avsr_conv_desc* c, const float* x, const float* w, const float* bias, const float* res, const float* res_sc,
                         const float* res_sh, float* y, float* stats, int32_t* nparts, void* stream, bool dry) {
  CGArgs A = {};
  A.src = x; A.w = w; A.bias = bias; A.dst = y; A.stats = stats; A.res = res; A.bn_sc = c->bn_scale; A.bn_sh = c->bn_shift;
  A.res_sc = res ? res_sc : nullptr; A.res_sh = res ? res_sh : nullptr;
  A.N = c->N; A.SH = c->H; A.SW = c->W; A.Cs = c->Ci; A.CsL = (c->Ci + 3) & ~3;
  A.DH = c->Ho; A.DW = c->Wo; A.Cd = c->Co; A.OA = c->Ho; A.OB = c->Wo; A.S = c->stride; A.OS = 1; A.oh0 = 0; A.ow0 = 0;
  A.wmode = 0; A.beta = 0.f;
  CGTap taps[9];
  const int nt = c->k * c->k;
  for (int t = 0; t < nt; ++t) taps[t] = cgtap1(t / c->k - c->pad_t, t % c->k - c->pad_l, t);
  A.F = cg_frames(c->H, c->W, A.CsL, c->Ho * c->Wo);
  int grid = 0;
  CGTap wide[CG_MAXTAP];
  const int nw = cg_pair_taps(A, taps, nt, wide);
  const int rc = nw ? cg_run(A, wide, nw, S_(stream), PROF_CONV_FWD, 2.0 * c->N * c->Ho * c->Wo * (double)c->Ci * c->Co * nt / nw, dry, &grid)
                    : cg_run(A, taps, nt, S_(stream), PROF_CONV_FWD, 2.0 * c->N * c->Ho * c->Wo * (double)c->Ci * c->Co, dry, &grid);
  if (rc < 0) return rc;
  if (nparts) *nparts = grid;
  return AVSR_OK;
}

// acc (may be NULL): dx = beta*acc + ... instead of beta*dx.  bnb_x != NULL: batch-norm backward stage 1 in the epilogue (see CGArgs):
// needs ONE launch group that writes every destination pixel once; stats [>= grid][2*Ci] receives the partial sums, *nparts the grid.
static int conv_bwd_data_impl(const avsr_conv_desc* c, const float* dy, const float* w, float* dx, float beta, void* stream, bool dry,
                              const float* acc = nullptr, const float* bnb_x = nullptr, const float* bnb_sc = nullptr,
                              const float* bnb_sh = nullptr, float* stats = nullptr, int32_t* nparts = nullptr) {
  if (c->Ci % 4) return AVSR_ERR_UNSUPPORTED;
  const int k = c->k;
  auto fuse = [&](CGArgs& A) { A.acc = acc; A.bnb_x = bnb_x; A.bnb_sc = bnb_sc; A.bnb_sh = bnb_sh; A.stats = stats; };
  if (c->stride == 1) {
    CGArgs A = {};
    A.src = dy; A.w = w; A.dst = dx;
    A.N = c->N; A.SH = c->Ho; A.SW = c->Wo; A.Cs = c->Co; A.CsL = c->Co;
    A.DH = c->H; A.DW = c->W; A.Cd = c->Ci; A.OA = c->H; A.OB = c->W; A.S = 1; A.OS = 1;
    A.wmode = 1; A.beta = beta;
    fuse(A);
    CGTap taps[9];
    for (int t = 0; t < k * k; ++t) taps[t] = cgtap1(c->pad_t - t / k, c->pad_l - t % k, t);   // dx[h, w] += dy[h + pt - i, w + pl - j] . W[i, j]^T
    A.F = cg_frames(c->Ho, c->Wo, A.CsL, c->H * c->W);
    CGTap wide[CG_MAXTAP];
    const int nw = cg_pair_taps(A, taps, k * k, wide);
    if (nw) return cg_run(A, wide, nw, S_(stream), PROF_CONV_BWD_DATA, 2.0 * c->N * c->H * c->W * (double)c->Ci * c->Co * (k * k) / nw, dry, nparts);
    return cg_run(A, taps, k * k, S_(stream), PROF_CONV_BWD_DATA, 2.0 * c->N * c->H * c->W * (double)c->Ci * c->Co, dry, nparts);
  }
  if (k == 3 && c->Ci * 4 <= 64) {
    // all four parity classes of a 2x2 destination cell in one launch: columns (class, channel), rows = cells
    CGArgs A = {};
    A.src = dy; A.w = w; A.dst = dx;
    A.N = c->N; A.SH = c->Ho; A.SW = c->Wo; A.Cs = c->Co; A.CsL = c->Co;
    A.DH = c->H; A.DW = c->W; A.Cd = c->Ci; A.OA = (c->H + 1) / 2; A.OB = (c->W + 1) / 2; A.S = 1; A.OS = 2;
    A.wmode = 1; A.beta = beta;
    fuse(A);
    A.nsp = 4; A.SB = 1; A.OSA = 2; A.OSB = 2; A.lin = 0;
    CGTap wide[CG_MAXTAP];
    int nw = 0, ntot = 0;
    bool fits = true;
    for (int ph = 0; ph < 2 && fits; ++ph)
      for (int pw = 0; pw < 2 && fits; ++pw) {
        const int sp = ph * 2 + pw;
        A.sp_dh[sp] = (signed char)ph; A.sp_dw[sp] = (signed char)pw;
        for (int i = 0; i < 3; ++i) {
          if ((ph + c->pad_t - i) & 1) continue;
          for (int j = 0; j < 3; ++j) {
            if ((pw + c->pad_l - j) & 1) continue;
            const int da = (ph + c->pad_t - i) >> 1, db = (pw + c->pad_l - j) >> 1;      // arithmetic shift: -1/2 -> floor
            int kk = 0;
            for (; kk < nw; ++kk)
              if (wide[kk].da == da && wide[kk].db == db) break;
            if (kk == nw) {
              if (nw == CG_MAXTAP) { fits = false; break; }
              wide[nw] = cgtap1(da, db, -1);
              ++nw;
            }
            wide[kk].w[sp] = (short)(i * 3 + j);
            ++ntot;
          }
        }
      }
    if (fits && nw > 0) {
      A.F = cg_frames(c->Ho, c->Wo, A.CsL, A.OA * A.OB);
      const int rc = cg_run(A, wide, nw, S_(stream), PROF_CONV_BWD_DATA, 2.0 * c->N * A.OA * A.OB * (double)c->Ci * c->Co * ntot / nw, dry, nparts);
      if (rc != AVSR_ERR_UNSUPPORTED) return rc;
    }
  }
  // the per-class launches below write disjoint pixel classes from separate grids: no single set of partial sums, and `acc` would
  // have to be applied class by class -- the fused forms are only offered on the single-launch paths above
  if (acc || bnb_x) return AVSR_ERR_UNSUPPORTED;
  // stride 2: one launch per parity class (ph, pw) of the input pixels; a class no tap reaches receives no gradient from this
  // convolution (beta == 0 is then refused: the caller orders its contributions so that this one accumulates)
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      CGArgs A = {};
      A.src = dy; A.w = w; A.dst = dx;
      A.N = c->N; A.SH = c->Ho; A.SW = c->Wo; A.Cs = c->Co; A.CsL = c->Co;
      A.DH = c->H; A.DW = c->W; A.Cd = c->Ci; A.OA = (c->H - ph + 1) / 2; A.OB = (c->W - pw + 1) / 2; A.S = 1; A.OS = 2; A.oh0 = ph; A.ow0 = pw;
      A.wmode = 1; A.beta = beta;
      CGTap taps[9];
      int nt = 0;
      for (int i = 0; i < k; ++i) {
        if ((ph + c->pad_t - i) & 1) continue;
        for (int j = 0; j < k; ++j) {
          if ((pw + c->pad_l - j) & 1) continue;
          taps[nt++] = cgtap1((ph + c->pad_t - i) >> 1, (pw + c->pad_l - j) >> 1, i * k + j);   // arithmetic shift: -1/2 -> floor
        }
      }
      if (A.OA <= 0 || A.OB <= 0) continue;
      if (nt == 0) {
        if (beta == 0.f) return AVSR_ERR_UNSUPPORTED;
        continue;
      }
      A.F = cg_frames(c->Ho, c->Wo, A.CsL, A.OA * A.OB);
      const int rc = cg_run(A, taps, nt, S_(stream), PROF_CONV_BWD_DATA, 2.0 * c->N * A.OA * A.OB * (double)c->Ci * c->Co, dry, nullptr);
      if (rc < 0) return rc;
    }
  return AVSR_OK;
}

// Workgroups per CU and LDS pixel padding of a weight-gradient launch whose passes hold ONE frame: up to WG_WPC(tiles) workgroups where
// the frames fit the CU's 160 KB side by side -- with + 1 float of padding instead of + 2 where that is what makes the next one fit
// (36x36x8: 3 x 52 KB; the odd pixel stride costs 4-byte staging stores and measured nothing on the operand reads).  AVSR_WG_WPC caps it.
static int wg_occupancy(int tiles, int H, int W, int CiL, int* pad) {
  static int cap_env = -1;
  if (cap_env < 0) { const char* e = getenv("AVSR_WG_WPC"); cap_env = e ? atoi(e) : 4; }
  *pad = WG_PAD;
  int cap = WG_WPC(tiles);
  if (cap > cap_env) cap = cap_env;
  auto frame = [&](int p) { return sizeof(float) * (size_t)(H + 2) * (W + 2) * (CiL + p); };
  if (2 * frame(WG_PAD) <= 64 * 1024) return 2;                       // several frames per pass: as before
  for (int w = cap; w > 2; --w) {
    if (w * (frame(WG_PAD) + 512) <= 160 * 1024) return w;
    if (w * (frame(1) + 512) <= 160 * 1024) { *pad = 1; return w; }
  }
  return 2;
}

// frames per pass of the weight-gradient kernel (upper bound; even shares in equal passes as above): its time is (passes) x (frames of a pass) -- chunks never span
// frames --, so among the feasible F the one with the smallest rounds * F wins (4800 frames on 512 workgroups: F = 4 -> 3 x 4, F = 2 or 5
// -> 10); ties: the larger F
static int wg_pick_frames(int N, int Fmax, int slots) {
  double best = -1.0;
  int bestF = Fmax;
  for (int F = Fmax; F >= 1; --F) {
    const long units = (N + F - 1) / F, grid = units < slots ? units : slots;
    const long cnt = (N + grid - 1) / grid, np = (cnt + F - 1) / F, fp = (cnt + np - 1) / np;      // (the kernel's even shares)
    const double cost = (double)np * (8.0 * fp + 1.0);
    if (best < 0.0 || cost < best) { best = cost; bestF = F; }
  }
  return bestF;
}

// final reduction of the pixel-pair weight gradient (below): part [nblk][12*Ci*16 (+16)] with rows (ti, tj', ci), columns (pp, co):
// dw[ti][tj][ci][co] = sum_blk part[(ti*4 + tj)*Ci + ci][co] + part[(ti*4 + tj + 1)*Ci + ci][8 + co];  dbias[co] = sum_blk bias[co] + bias[8 + co]
__global__ __launch_bounds__(1024) void wgrad_pair_final_kernel(const float* __restrict__ part, int nblk, int slab, int Ci, float* __restrict__ dw,
                                                               float* __restrict__ dbias, float beta) {
  // 32 outputs per workgroup, 32 slices of the slab list each (one thread per (slice, output): 512 slabs = 16 dependent fp64 adds per
  // thread with four slabs' loads in flight; 8 slices of 64 slabs took 22 us per launch, three launches per step)
  __shared__ double red[32][33];
  const int fl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int nw = 9 * Ci * 8, f = blockIdx.x * 32 + fl;              // outputs: 9*Ci*8 kernel entries, then 8 bias entries
  int o0 = -1, o1 = -1;
  if (f < nw) {
    const int co = f & 7, ci = (f >> 3) % Ci, t = (f >> 3) / Ci, ti = t / 3, tj = t - ti * 3;
    o0 = ((ti * 4 + tj) * Ci + ci) * 16 + co;
    o1 = ((ti * 4 + tj + 1) * Ci + ci) * 16 + 8 + co;
  } else if (f < nw + 8 && dbias) {
    o0 = 12 * Ci * 16 + (f - nw);
    o1 = o0 + 8;
  }
  double s = 0.0;
  if (o0 >= 0) {
    int i = g;
    for (; i + 96 < nblk; i += 128) {
      const float a0 = part[(long)i * slab + o0], b0 = part[(long)i * slab + o1];
      const float a1 = part[(long)(i + 32) * slab + o0], b1 = part[(long)(i + 32) * slab + o1];
      const float a2 = part[(long)(i + 64) * slab + o0], b2 = part[(long)(i + 64) * slab + o1];
      const float a3 = part[(long)(i + 96) * slab + o0], b3 = part[(long)(i + 96) * slab + o1];
      s += (double)a0 + (double)b0;
      s += (double)a1 + (double)b1;
      s += (double)a2 + (double)b2;
      s += (double)a3 + (double)b3;
    }
    for (; i < nblk; i += 32) s += (double)part[(long)i * slab + o0] + (double)part[(long)i * slab + o1];
  }
  red[g][fl] = s;
  __syncthreads();
  if (g == 0 && o0 >= 0) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][fl];
    float* const o = f < nw ? dw + f : dbias + (f - nw);
    *o = beta != 0.f ? (float)t + beta * *o : (float)t;
  }
}

static int conv_bwd_weight_impl(const avsr_conv_desc* c, const float* x, const float* dy, float* dw, float* dbias, float beta, float* scratch,
                                long scratch_floats, void* stream, bool dry, const float* fold_y = nullptr, const float* fold_k = nullptr,
                                float* fold_dx = nullptr) {
  const int Ci = c->Ci, Co = c->Co, H = c->H, W = c->W, Ho = c->Ho, Wo = c->Wo, N = c->N, k = c->k;
  const bool fold = fold_k != nullptr;                   // dy = k1*dz + k2*y + k3 evaluated in the operand fetch (single-launch forms only);
                                                         // fold_dx: also written out for the data gradient that follows
  // Pixel-pair form for 8 destination channels (the 36x36 layers, the most expensive weight gradients): with 8 columns half of
  // every 16-column MFMA tile multiplies padding.  dy is read as [N, Ho, Wo/2, 16] (the same bytes): a column is (pixel parity pp,
  // channel), the depth index a PAIR of horizontally adjacent output pixels; the rows run over the union of the two pixels' windows
  // (3 x 4 taps, source step 2 along W): C[(ti, tj', ci)][(pp, co)] is the gradient of tap (ti, tj' - pp) where that is a tap at all.
  // 6 row tiles per pair instead of 2 x 5 per two positions; the reduction kernel above adds the two parities' valid entries.
  if (Co == 8 && c->stride == 1 && k == 3 && (Wo & 1) == 0 && Wo == W && Ho == H && !getenv("AVSR_WGRAD_NOPAIR")) {
    WGArgs A = {};
    A.x = x; A.dy = dy; A.part = scratch; A.N = N; A.H = H; A.W = W; A.Ci = Ci; A.CiL = (Ci + 3) & ~3; A.Ho = Ho; A.Wo = Wo / 2; A.Co = 16;
    A.S = 1; A.SW = 2; A.pt = c->pad_t; A.pl = c->pad_l; A.kw = 4; A.bn_sc = c->bn_scale; A.bn_sh = c->bn_shift;
    A.t0 = 0; A.nt = 12; A.want_bias = dbias ? 1 : 0;
    A.fy = fold_y; A.fk = fold_k; A.fC = 8; A.fdx = fold_dx;
#ifdef CONV_DEBUG
    { const char* e = getenv("AVSR_CONV_DBG"); A.dbg = e ? atoi(e) : 0; }
#endif
    A.slab = 12 * Ci * 16 + (A.want_bias ? 16 : 0);
    const int MT = (12 * A.CiL + 15) / 16;
    bool ok = (Ci % 4 == 0) ? MT <= 6 : (MT <= 3 && !c->bn_scale);
    const int wpc_cap = wg_occupancy(MT, H, W, A.CiL, &A.pad);
    A.F = cg_frames(H, W, A.CiL + A.pad, Ho * A.Wo);
    A.m_opf = fmagic(Ho * A.Wo); A.m_wo = fmagic(A.Wo); A.m_w = fmagic(W);
    if (Ci % 4 == 0) {
      const int rq = W * Ci / 4;
      ok = ok && rq <= 256 && rq >= 1;
      if (ok) {
        const int rpp = 256 / rq;
        while (A.F > 1 && A.F * ((H + rpp - 1) / rpp) > 12) --A.F;
        ok = A.F * ((H + rpp - 1) / rpp) <= 12;
        A.m_rq = fmagic(rq); A.m_per = fmagic(H * rq);
      }
    } else {
      while (A.F > 1 && (A.F * H * W * Ci / 4 + 255) / 256 > 4) --A.F;
      ok = ok && (A.F * H * W * Ci / 4 + 255) / 256 <= 4 && (long)A.F * H * W * Ci < 65536;
      A.m_rq = fmagic(Ci); A.m_per = fmagic(H * W * Ci);
    }
    if (ok) A.F = wg_pick_frames(N, A.F, 256 * wpc_cap);
    size_t lds = sizeof(float) * (size_t)A.F * (size_t)(H + 2) * (W + 2) * (A.CiL + A.pad);
    if (lds < sizeof(float) * 4 * 256) lds = sizeof(float) * 4 * 256;
    ok = ok && lds <= 64 * 1024 && (long)A.F * ((Ho * A.Wo + 15) / 16) < 65536 && (long)N * Ho * Wo * Co * 4 < (1L << 31);
    if (ok) {
      int wpc = (int)((160 * 1024) / (lds + 512));
      if (wpc > wpc_cap) wpc = wpc_cap;
      if (wpc < 1) wpc = 1;
      int grid = (N + A.F - 1) / A.F;
      if (grid > 256 * wpc) grid = 256 * wpc;
      if ((long)grid * A.slab > scratch_floats) grid = (int)(scratch_floats / A.slab);
      if (grid < 1) return AVSR_ERR_ARG;
      if (dry) return AVSR_OK;
      hipStream_t s = S_(stream);
      {
        ProfScope ps(PROF_CONV_BWD_WEIGHT, s, 2.0 * N * Ho * Wo * 9.0 * Ci * Co);
        if (fold && Ci % 4 && fold_dx) hipLaunchKernelGGL((conv_wgrad_kernel<3, 1, false, false, 2>), dim3(grid), dim3(256), lds, s, A);
        else if (fold && Ci % 4) hipLaunchKernelGGL((conv_wgrad_kernel<3, 1, false, false, 1>), dim3(grid), dim3(256), lds, s, A);
        else if (fold && fold_dx) hipLaunchKernelGGL((conv_wgrad_kernel<6, 1, true, false, 2>), dim3(grid), dim3(256), lds, s, A);
        else if (fold) hipLaunchKernelGGL((conv_wgrad_kernel<6, 1, true, false, 1>), dim3(grid), dim3(256), lds, s, A);
        else if (Ci % 4) hipLaunchKernelGGL((conv_wgrad_kernel<3, 1, false>), dim3(grid), dim3(256), lds, s, A);
        else hipLaunchKernelGGL((conv_wgrad_kernel<6, 1, true>), dim3(grid), dim3(256), lds, s, A);
        if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
      }
      const int nout = 9 * Ci * 8 + 8;
      if (avsr::slab_defer_push(scratch, A.slab, grid, nout, dw, dbias, 0, 1, Ci, 1.0f, beta, s)) return AVSR_OK;
      hipLaunchKernelGGL(wgrad_pair_final_kernel, dim3((nout + 31) / 32), dim3(1024), 0, s, scratch, grid, A.slab, Ci, dw, dbias, beta);
      if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
      return AVSR_OK;
    }
  }
  WGArgs A = {};
  A.x = x; A.dy = dy; A.part = scratch; A.N = N; A.H = H; A.W = W; A.Ci = Ci; A.CiL = (Ci + 3) & ~3; A.Ho = Ho; A.Wo = Wo; A.Co = Co;
  A.S = c->stride; A.pt = c->pad_t; A.pl = c->pad_l; A.kw = k; A.bn_sc = c->bn_scale; A.bn_sh = c->bn_shift;
  A.fy = fold_y; A.fk = fold_k; A.fC = Co; A.fdx = fold_dx;
  const int NTC = (Co + 15) / 16;
  if (NTC == 3) return AVSR_ERR_UNSUPPORTED;
  const int mt_max = NTC == 4 ? 9 : 18;
  int G = mt_max * 16 / A.CiL;                           // taps per launch
  if (G < 1) return AVSR_ERR_UNSUPPORTED;
  // (tiles of the form the first launch takes: the tap groups of one call share the staging layout)
  const int mt_first = ((k * k < G ? k * k : G) * A.CiL + 15) / 16;
  const int tiles_first = (mt_first <= 5 ? 5 : (mt_first <= 9 ? 9 : 18)) * (NTC == 1 ? 1 : (NTC == 2 ? 2 : 4));
  const int wpc_cap = (Ci % 4 == 0) ? wg_occupancy(tiles_first, H, W, A.CiL, &A.pad) : (A.pad = WG_PAD, 2);
  A.F = cg_frames(H, W, A.CiL + A.pad, Ho * Wo);
  A.m_opf = fmagic(Ho * Wo); A.m_wo = fmagic(Wo); A.m_w = fmagic(W);
  if (Ci % 4 == 0) {
    const int rq = W * Ci / 4;
    if (rq > 256 || rq < 1) return AVSR_ERR_UNSUPPORTED;
    const int rpp = 256 / rq;
    while (A.F > 1 && A.F * ((H + rpp - 1) / rpp) > 12) --A.F;
    if (A.F * ((H + rpp - 1) / rpp) > 12) return AVSR_ERR_UNSUPPORTED;
    A.m_rq = fmagic(rq); A.m_per = fmagic(H * rq);
  } else {
    if (c->bn_scale) return AVSR_ERR_UNSUPPORTED;
    while (A.F > 1 && (A.F * H * W * Ci / 4 + 255) / 256 > 4) --A.F;
    if ((A.F * H * W * Ci / 4 + 255) / 256 > 4 || (long)A.F * H * W * Ci >= 65536) return AVSR_ERR_UNSUPPORTED;
    A.m_rq = fmagic(Ci); A.m_per = fmagic(H * W * Ci);
  }
  while (A.F > 1 && sizeof(float) * (size_t)A.F * (size_t)(H + 2) * (W + 2) * (A.CiL + A.pad) > 64 * 1024) --A.F;
  {
    const int nt0 = k * k < G ? k * k : G;
    A.F = wg_pick_frames(N, A.F, (nt0 * Ci * Co + Co > 2048) ? 256 : 256 * wpc_cap);
  }
  if ((long)A.F * ((Ho * Wo + 15) / 16) >= 65536 || (long)N * Ho * Wo * Co * 4 >= (1L << 31)) return AVSR_ERR_UNSUPPORTED;
  const size_t red = sizeof(float) * 4 * 256;
  size_t lds = sizeof(float) * (size_t)A.F * (size_t)(H + 2) * (W + 2) * (A.CiL + A.pad);
  if (lds < red) lds = red;
  if (lds > 64 * 1024) return AVSR_ERR_UNSUPPORTED;
  int wpc = (int)((160 * 1024) / (lds + 512));
  if (wpc > wpc_cap) wpc = wpc_cap;
  if (wpc < 1) wpc = 1;
  hipStream_t s = S_(stream);
  bool bias_done = dbias == nullptr;
  // deep layers (64 destination channels, more (tap, channel) rows than one wave's accumulators hold): the row-split form, one launch
  static int rs_on = -1;
  if (rs_on < 0) { const char* e = getenv("AVSR_WGRAD_RS"); rs_on = e ? (atoi(e) != 0) : 1; }
  const int Mall = k * k * A.CiL;
  const bool rs = rs_on && NTC == 4 && Ci % 4 == 0 && G < k * k && Mall <= 4 * 9 * 16;   // (32-column layers fit one launch already: no gain measured)
  if (rs) G = k * k;
  if (fold) {
    // one launch only (a second tap group would evaluate -- and write -- the gradient again), and only the forms instantiated below
    const int MT1 = rs ? ((k * k * A.CiL + 3) / 4 + 15) / 16 : (k * k * A.CiL + 15) / 16;
    const bool okf = G >= k * k && Ci % 4 == 0 && (rs ? MT1 <= 5 : (NTC == 1 ? MT1 <= 5 : (NTC == 2 && MT1 <= 9)));
    if (!okf) return AVSR_ERR_UNSUPPORTED;
  }
  for (int t0 = 0; t0 < k * k; t0 += G) {
    A.t0 = t0; A.nt = k * k - t0 < G ? k * k - t0 : G;
    A.want_bias = bias_done ? 0 : 1;
    const int wF = A.nt * Ci * Co;
    A.slab = wF + (A.want_bias ? Co : 0);
    const int MT = rs ? ((A.nt * A.CiL + 3) / 4 + 15) / 16 : (A.nt * A.CiL + 15) / 16;      // (rs: row tiles per WAVE)
    if (Ci % 4 && (MT > 3 || NTC != 1)) return AVSR_ERR_UNSUPPORTED;
    int grid = (N + A.F - 1) / A.F;
    const int cap = 256 * (A.slab > 2048 ? 1 : wpc);      // large kernels: the partial slabs, not the staging, are the traffic
    if (grid > cap) grid = cap;
    if ((long)grid * A.slab > scratch_floats) grid = (int)(scratch_floats / A.slab);
    if (grid < 1) return AVSR_ERR_ARG;
    if (dry) continue;
    {
      ProfScope ps(PROF_CONV_BWD_WEIGHT, s, 2.0 * N * Ho * Wo * (double)A.nt * Ci * Co);
#define WG_GO(M_, N_, C_) hipLaunchKernelGGL((conv_wgrad_kernel<M_, N_, C_>), dim3(grid), dim3(256), lds, s, A)
      if (fold) {
#define WG_FOLD(M_, N_, R_) { if (fold_dx) hipLaunchKernelGGL((conv_wgrad_kernel<M_, N_, true, R_, 2>), dim3(grid), dim3(256), lds, s, A); \
                              else hipLaunchKernelGGL((conv_wgrad_kernel<M_, N_, true, R_, 1>), dim3(grid), dim3(256), lds, s, A); }
        if (rs) WG_FOLD(5, 4, true)
        else if (NTC == 1) WG_FOLD(5, 1, false)
        else WG_FOLD(9, 2, false)
#undef WG_FOLD
      } else if (rs) {
        if (MT <= 5) hipLaunchKernelGGL((conv_wgrad_kernel<5, 4, true, true>), dim3(grid), dim3(256), lds, s, A);
        else hipLaunchKernelGGL((conv_wgrad_kernel<9, 4, true, true>), dim3(grid), dim3(256), lds, s, A);
      } else if (Ci % 4) WG_GO(3, 1, false);
      else if (NTC == 1) { if (MT <= 5) WG_GO(5, 1, true); else if (MT <= 9) WG_GO(9, 1, true); else WG_GO(18, 1, true); }
      else if (NTC == 2) { if (MT <= 5) WG_GO(5, 2, true); else if (MT <= 9) WG_GO(9, 2, true); else WG_GO(18, 2, true); }
      else { if (MT <= 5) WG_GO(5, 4, true); else WG_GO(9, 4, true); }
#undef WG_GO
      if (hipGetLastError() != hipSuccess) return AVSR_ERR_HIP;
    }
    int rc;
    if (A.want_bias) {                                     // weight and bias gradients of the slab in one reduction launch
      if (avsr::slab_defer_push(scratch, A.slab, grid, A.slab, dw + (long)t0 * Ci * Co, dbias, wF, 0, 0, 1.0f, beta, s)) rc = AVSR_OK;
      else rc = avsr_colsum_final_launch_split(scratch, A.slab, grid, dw + (long)t0 * Ci * Co, dbias, wF, A.slab, 1.0f, beta, stream);
      bias_done = true;
    } else {
      if (avsr::slab_defer_push(scratch, A.slab, grid, wF, dw + (long)t0 * Ci * Co, nullptr, 0x7fffffff, 0, 0, 1.0f, beta, s)) rc = AVSR_OK;
      else rc = avsr_colsum_final_launch_ld(scratch, A.slab, grid, dw + (long)t0 * Ci * Co, wF, 1.0f, beta, stream);
    }
    if (rc != AVSR_OK) return rc;
    // (deferred reductions: the next tap group of this call must not overwrite the slabs just recorded)
    if (avsr::slab_deferring()) { scratch += (long)grid * A.slab; scratch_floats -= (long)grid * A.slab; A.part = scratch; }
  }
  return AVSR_OK;
}

extern "C" int avsr_conv_supported(const avsr_conv_desc* c) {
  if (!g_conv_mfma || !cd_ok(c)) return 0;
  if (conv_fwd_impl(c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true) != AVSR_OK) return 0;
  if (c->Ci >= 4 && conv_bwd_data_impl(c, nullptr, nullptr, nullptr, 1.f, nullptr, true) != AVSR_OK) return 0;
  if (conv_bwd_weight_impl(c, nullptr, nullptr, nullptr, nullptr, 1.f, nullptr, 1L << 40, nullptr, true) != AVSR_OK) return 0;
  return 1;
}

extern "C" int avsr_conv_fwd(const avsr_conv_desc* c, const float* x, const float* w, const float* bias, const float* res, const float* res_scale,
                             const float* res_shift, float* y, float* stats, int32_t* nparts, void* stream) {
  if (!cd_ok(c) || !x || !w || !y) return AVSR_ERR_ARG;
  if (!g_conv_mfma) return AVSR_ERR_UNSUPPORTED;
  return conv_fwd_impl(c, x, w, bias, res, res_scale, res_shift, y, stats, nparts, stream, false);
}

extern "C" int avsr_conv_bwd_data(const avsr_conv_desc* c, const float* dy, const float* w, float* dx, float beta, void* stream) {
  if (!cd_ok(c) || !dy || !w || !dx) return AVSR_ERR_ARG;
  if (!g_conv_mfma) return AVSR_ERR_UNSUPPORTED;
  return conv_bwd_data_impl(c, dy, w, dx, beta, stream, false);
}

extern "C" int avsr_conv_bwd_data_bn(const avsr_conv_desc* c, const float* dy, const float* w, float* dx, float beta, const float* acc,
                                     const float* bn_x, const float* bn_scale, const float* bn_shift, float* stats, int32_t* nparts,
                                     void* stream) {
  if (!cd_ok(c) || !dy || !w || !dx || (bn_x && (!bn_scale || !bn_shift || !stats || !nparts))) return AVSR_ERR_ARG;
  if (!g_conv_mfma) return AVSR_ERR_UNSUPPORTED;
  return conv_bwd_data_impl(c, dy, w, dx, beta, stream, false, acc, bn_x, bn_scale, bn_shift, bn_x ? stats : nullptr, nparts);
}

// can avsr_conv_bwd_data_bn run this layer's data gradient with the fused forms (accumulate source / batch-norm backward epilogue)?
extern "C" int avsr_conv_bwd_data_bn_supported(const avsr_conv_desc* c) {
  if (!g_conv_mfma || !cd_ok(c) || c->Ci < 4) return 0;
  static float dummy;
  int32_t n = 0;
  return conv_bwd_data_impl(c, nullptr, nullptr, nullptr, 1.f, nullptr, true, &dummy, &dummy, &dummy, &dummy, &dummy, &n) == AVSR_OK;
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

// Weight (+ bias) gradient with the batch-norm backward of the convolution's own output folded into the operand fetch: dy = k[0..C)*dz +
// k[C..2C)*y + k[2C..3C) (avsr_bn_bwd_finalize's vectors) -- see WGArgs; dx_out (may be NULL): the evaluated gradient is also stored there
// for the layer's data gradient.  AVSR_ERR_UNSUPPORTED unless avsr_conv_bwd_weight_bn_supported.
extern "C" int avsr_conv_bwd_weight_bn(const avsr_conv_desc* c, const float* x, const float* dz, const float* y, const float* k, float* dx_out,
                                       float* dw, float* dbias, float beta, float* scratch, int64_t scratch_floats, void* stream) {
  if (!cd_ok(c) || !x || !dz || !y || !k || !dw || !scratch) return AVSR_ERR_ARG;
  if (!g_conv_mfma) return AVSR_ERR_UNSUPPORTED;
  return conv_bwd_weight_impl(c, x, dz, dw, dbias, beta, scratch, scratch_floats, stream, false, y, k, dx_out);
}
extern "C" int avsr_conv_bwd_weight_bn_supported(const avsr_conv_desc* c) {
  if (!g_conv_mfma || !cd_ok(c)) return 0;
  static float dummy;
  return conv_bwd_weight_impl(c, nullptr, nullptr, nullptr, nullptr, 1.f, nullptr, 1L << 40, nullptr, true, &dummy, &dummy) == AVSR_OK;
}

extern "C" int avsr_conv_bwd_weight(const avsr_conv_desc* c, const float* x, const float* dy, float* dw, float* dbias, float beta, float* scratch,
                                    int64_t scratch_floats, void* stream) {
  if (!cd_ok(c) || !x || !dy || !dw || !scratch) return AVSR_ERR_ARG;
  if (!g_conv_mfma) return AVSR_ERR_UNSUPPORTED;
  return conv_bwd_weight_impl(c, x, dy, dw, dbias, beta, scratch, scratch_floats, stream, false);
}
