// Shared declarations of the frame-resident MFMA convolutions (conv_mfma.hip: the data-path kernels, forward and data gradients;
// conv_wgrad.hip: the weight gradient; conv_bn.hip: the batch-norm kernels around them).  Round 6 split one 2 156-line file along these lines.
#pragma once
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
  int CsP;                                // floats per staged pixel in LDS (>= CsL: bank spreading, set by cg_launch; conv_gen_kernel only)
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


extern int g_conv_mfma;                 // avsr_conv_set_mfma: 0 sends every layer to the direct / im2col paths (tests)
bool slab_defer_push(const float* part, long ld, int nblk, int F, float* out, float* out2, int split, int kind, int Ci, float alpha, float beta,
                     hipStream_t s);
bool slab_deferring();

}  // namespace avsr

int avsr_colsum_final_launch(const float* part, int nblk, float* out, int F, float alpha, float beta, void* stream);
int avsr_colsum_final_launch_ld(const float* part, long ld, int nblk, float* out, int F, float alpha, float beta, void* stream);
int avsr_colsum_final_launch_split(const float* part, long ld, int nblk, float* out, float* out2, int split, int F, float alpha, float beta,
                                   void* stream);

#define S_(x) ((hipStream_t)(x))

static inline int cg_frames(int sh, int sw, int csl, int opf, int extra_floats_per_frame = 0) {
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
static inline bool cd_ok(const avsr_conv_desc* c) {
  return c && c->N > 0 && (c->k == 1 || c->k == 3) && (c->stride == 1 || c->stride == 2) && c->Co % 4 == 0 && (c->Ci % 4 == 0 || c->Ci < 4) &&
         c->Ci > 0 && c->Co > 0 && c->Co <= 64 && c->pad_t >= 0 && c->pad_t <= 1 && c->pad_l >= 0 && c->pad_l <= 1 && (c->k == 3 || (c->pad_t == 0 && c->pad_l == 0));
}

// the three implementations behind the descriptor API (dry = true: only decide whether the layer is covered)
int conv_fwd_impl(const avsr_conv_desc* c, const float* x, const float* w, const float* bias, const float* res, const float* res_sc,
                  const float* res_sh, float* y, float* stats, int32_t* nparts, void* stream, bool dry);
int conv_bwd_data_impl(const avsr_conv_desc* c, const float* dy, const float* w, float* dx, float beta, void* stream, bool dry,
                       const float* acc = nullptr, const float* bnb_x = nullptr, const float* bnb_sc = nullptr,
                       const float* bnb_sh = nullptr, float* stats = nullptr, int32_t* nparts = nullptr);
int conv_bwd_weight_impl(const avsr_conv_desc* c, const float* x, const float* dy, float* dw, float* dbias, float beta, float* scratch,
                         long scratch_floats, void* stream, bool dry, const float* fold_y = nullptr, const float* fold_k = nullptr,
                         float* fold_dx = nullptr);
