// Weight (+ bias) gradient of the frame-resident MFMA convolutions (conv_mfma.h; split out of conv_mfma.hip in round 6): the transposed GEMM
// (rows = (tap, cin), columns = cout, depth = positions) with the frame staged in LDS and dy streamed; per-workgroup partial slabs, reduced
// by the column-sum launches of elementwise.hip (deferred to the end of the CNN's backward pass: avsr_slab_defer_*).
#include "conv_mfma.h"
#include <cstdlib>

namespace avsr {

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

}  // namespace avsr

using namespace avsr;

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

int conv_bwd_weight_impl(const avsr_conv_desc* c, const float* x, const float* dy, float* dw, float* dbias, float beta, float* scratch,
                                long scratch_floats, void* stream, bool dry, const float* fold_y, const float* fold_k,
                                float* fold_dx) {
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
