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
#include "conv_mfma.h"

namespace avsr {

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
  const int Cs = A.Cs, CsL = A.CsL, CsP = A.CsP, Cd = A.Cd, C4 = CsL >> 2;
  const int PH = A.SH + 2, PW = A.SW + 2, fstride = PH * PW * CsP;
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
      ro = f * fstride + ((a * A.S + 1) * PW + (b * A.SB + 1)) * CsP;
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
      koff[c] = in ? (tp[0] * PW + tp[1]) * CsP + cs4 * 4 : 0;
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
  // LDS offset of the piece inside its row: pixels are CsP floats apart (CsP == Cs: the row is contiguous, the offset is st_p4 * 4)
  const int st_px = (CH4 && CsP != Cs) ? (st_p4 * 4) / Cs : 0;
  const int st_lo = (CH4 && CsP != Cs) ? st_px * CsP + (st_p4 * 4 - st_px * Cs) : st_p4 * 4;
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
          st4(lds + f * fstride + ((r + 1) * PW + 1) * CsP + st_lo, v);
        }
      }
    } else {
      const int tot = fcur * per3, tot4 = tot >> 2;
      auto put = [&](int e, float v) {
        const int f = fdiv(e, A.m_per), r = e - f * per3, px = fdiv(r, A.m_rq), c = r - px * Cs, h = fdiv(px, A.m_sw), pw = px - h * A.SW;
        lds[f * fstride + ((h + 1) * PW + pw + 1) * CsP + c] = v;
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

int g_conv_mfma = 1;
static const int g_conv_pad = [] { const char* e = getenv("AVSR_CONV_PAD"); return e ? atoi(e) : 1; }();   // A/B switch of the padded LDS pixel stride

}  // namespace avsr

using namespace avsr;

extern "C" int avsr_conv_set_mfma(int32_t on) { g_conv_mfma = on ? 1 : 0; return AVSR_OK; }

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
  // Floats per staged pixel.  The tile loop reads its B operand with one 16-byte LDS load per lane: lane (i, q) takes channel quad
  // (4c + q) of position mt*16 + i, and a ds_read_b128 is served in four groups of 16 lanes, each conflict-free only if its lanes hit 16
  // different 16-byte slots (MI355X_MICROARCH.md, LDS table).  Positions of a row are SB * CsP floats apart: with CsP = 32 (64) that is
  // 8 (16) slots -- 4 (8) lanes per slot, and SQ_LDS_BANK_CONFLICT counted 68-84 % of the LDS-active cycles of the 16- / 32- / 64-channel
  // launches (profiles/r06_c4_lds_pmc_v1.txt).  A stride of 2 (mod 4) slots per position spreads every group over all 16 slots.
  A.CsP = A.CsL;
  if (A.Cs % 4 == 0 && g_conv_pad) {
    for (int p = 0; p < 4; ++p)
      if ((A.SB * (A.CsL / 4 + p)) % 4 == 2) { A.CsP = A.CsL + 4 * p; break; }
  }
  size_t frame_b = sizeof(float) * (size_t)(A.SH + 2) * (A.SW + 2) * A.CsP;
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
  if (!A.F && A.CsP != A.CsL) {                          // the padded frame does not fit (36 x 36 maps): unpadded
    A.CsP = A.CsL;
    frame_b = sizeof(float) * (size_t)(A.SH + 2) * (A.SW + 2) * A.CsP;
    A.F = pick(64 * 1024, 512);
  }
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


// =====================================================================================================================
// Descriptor API (include/avsr_hip.h: avsr_conv_desc): k = 1 or 3, stride 1 or 2, 3..64 channels; BN-ReLU of the input applied by the
// loader; kernels deeper than one wave's register budget run as several launches over tap groups (the later ones accumulate).

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

int conv_fwd_impl(const avsr_conv_desc* c, const float* x, const float* w, const float* bias, const float* res, const float* res_sc,
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
int conv_bwd_data_impl(const avsr_conv_desc* c, const float* dy, const float* w, float* dx, float beta, void* stream, bool dry,
                              const float* acc, const float* bnb_x, const float* bnb_sc,
                              const float* bnb_sh, float* stats, int32_t* nparts) {
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
