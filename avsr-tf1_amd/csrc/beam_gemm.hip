// Throughput-shaped dense steps of beam search (mode 3 of avsr_attn_rnn_fwd; BeamSearchDecoder at avsr/decoder_unimodal.py:248-271,
// avsr/decoder_bimodal.py:358-381: the reference's default evaluation, width 10).
//
// Under beam search a decode step carries B * K rows (640 at the benchmark batch): the LSTM cell is a [640 x 896] x [896 x 1024]
// product and the attention layers two [640 x 512] x [512 x 256] ones.  The per-step `step_kernel` (16 x 16 output tiles, built for
// the 64-row steps of training, where only latency matters) runs them as 2560 / 1280 workgroups that re-read their operands from
// L2 -- 46 + 41 us of the 145 us step (profiles/r03_decode_rates_v3.txt).  Here the same products are 64 x 64 output tiles walked
// through LDS in 128-deep K stages (coalesced row pieces, register prefetch of the next stage under the current stage's MFMAs), with the operand ROWS
// GATHERED while they are staged -- token embedding by `tok`, attention record and recurrent state by the parent hypothesis'
// row -- and the cell's gate math, clip and state update in the epilogue:
//   beam_gemm_kernel<LSTM>    z = [emb[tok[r]] | att[parent[r]] | h[parent[r]]] . W + b -> i, j, f, o -> c, h of row r
//   beam_gemm_kernel<LINEAR>  att_m[r] = [cell_out[r] | ctx_m[r]] . W_att,m            (all mechanisms in one launch)
//   beam_ctx_merge_kernel     ctx_m[r] = softmax-merge of the per-chunk partial contexts the attention kernel left (the step kernel
//                             did this inside its operand loader, once per column tile)
// f32 MFMA (v_mfma_f32_32x32x2_f32): exact fp32 products, the summation order differs from step_kernel's (K split over waves there),
// i.e. results agree to rounding; tests/test_gpu_beam.py checks both against the fp64 oracle.
#include "persist.h"
#include "avsr_hip.h"
#include "prof.h"
#include "attn.h"
#include "step.h"
#include <cstdlib>

namespace avsr {

#define BG_MAX_SRC 3
#define BG_MAX_PROB 2
#define BG_T 64            // tile rows = tile columns
#define BG_K 128           // K per stage

struct BGSrc { const float* a; long sb; const int* gather; int K, pad; };
struct BGProb {
  BGSrc src[BG_MAX_SRC];
  int nsrc, R, N, tile0, ntx, pad;
  const float* wt; long ldw;                     // weights [N][ldw], K contiguous
  float* out; long out_sb;                       // LINEAR: out[r * out_sb + n]
  const float* bias; const float* c_in; const int* parent; float* c_out; float* h_out; float* seq_out; long seq_sb;   // LSTM
};
struct BGLaunch { int nprob, ntiles; BGProb p[BG_MAX_PROB]; };

// Operand staging.  Both operands are K-contiguous in memory (gathered activation rows; weight rows [N][ldw]).  A wave-instruction
// fetches 4 rows x 256 contiguous bytes (16 lanes x 16 bytes per row: 8 full 128-byte lines; one lane per ROW, the obvious mapping,
// is 64 different lines per instruction and ran the 640 x 1024 x 896 cell product at 34 us against its 13 us of matrix-pipe time) and
// writes them to LDS as they are, row-major [row][k] with a pitch of 68 floats: conflict-free ds_write_b128 and ds_read_b128.
// MFMA operand fetch: v_mfma_f32_32x32x2 takes ONE k per lane half; which k the two halves supply is free as long as A and B agree,
// so lane half h owns k in [32 h, 32 h + 32) of a stage -- its 32 operand values are 8 ds_read_b128 of one LDS row -- and instruction
// i multiplies the pair (i, 32 + i).
#define BG_P (BG_K + 4)    // LDS row pitch in floats (16-byte aligned rows, 4 banks apart)

// Code-generation notes (each visible in the ISA, each cost a whole serialised memory round trip per stage until fixed;
// tools/beam_gemm_dissect.sh):  (1) the problem descriptor is read through CONSTANT indices into the by-value kernel argument (PF below):
// `L.p[pi]` with a run-time pi makes the compiler copy the argument into scratch memory and turn every field access into a scratch load;
// (2) no local array is indexed by the stage's source -- such an array lives in scratch too; (3) no arithmetic on a loaded operand
// before it is committed to LDS, else the compiler waits for the load right behind its issue.
template <bool LSTM>
__global__ __launch_bounds__(512) void beam_gemm_kernel(const BGLaunch L) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][BG_T * BG_P];      // [buffer][A | B][row or column][k]: 132 KB
  const bool p1 = L.nprob > 1 && (int)blockIdx.x >= L.p[1].tile0;            // uniform
#define PF(f) (p1 ? L.p[1].f : L.p[0].f)
  const int P_tile0 = PF(tile0), P_ntx = PF(ntx), P_nsrc = PF(nsrc), P_R = PF(R), P_N = PF(N);
  const float* const P_wt = PF(wt);
  const long P_ldw = PF(ldw);
  const int tile = blockIdx.x - P_tile0, tx = tile % P_ntx, ty = tile / P_ntx;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 8 waves, two per SIMD: wave = (k half of a stage, 32 x 32 quadrant); the halves meet through LDS once, after the last stage.
  const int kh = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  // ---- staging role of this thread: piece p = row / column (tid >> 4) + 32 p of the tile, k = 4 (tid & 15) of the stage ----
  // A stage never straddles two sources: every source occupies whole 128-deep stages of a VIRTUAL K axis (its tail, if its width is
  // not a multiple of 64, reads zeros on both operands), so the source of a stage is wave-uniform: one load per piece, its resource
  // and row offset picked by scalar selects; the loads of stage s + 2 stay in flight under the MFMAs of stage s + 1.
  const int srow = tid >> 5, sk = 4 * (tid & 31);
  const int K0 = P_nsrc > 0 ? PF(src[0].K) : 0, K1 = P_nsrc > 1 ? PF(src[1].K) : 0, K2 = P_nsrc > 2 ? PF(src[2].K) : 0;
  const int sb1 = (K0 + BG_K - 1) / BG_K, sb2 = sb1 + (K1 + BG_K - 1) / BG_K, nstage = sb2 + (K2 + BG_K - 1) / BG_K;
  const float* const ab0 = P_nsrc > 0 ? PF(src[0].a) : P_wt;
  const float* const ab1 = P_nsrc > 1 ? PF(src[1].a) : P_wt;
  const float* const ab2 = P_nsrc > 2 ? PF(src[2].a) : P_wt;
  const int* const ga0 = P_nsrc > 0 ? PF(src[0].gather) : nullptr;
  const int* const ga1 = P_nsrc > 1 ? PF(src[1].gather) : nullptr;
  const int* const ga2 = P_nsrc > 2 ? PF(src[2].gather) : nullptr;
  const long rs0 = P_nsrc > 0 ? PF(src[0].sb) : 0, rs1 = P_nsrc > 1 ? PF(src[1].sb) : 0, rs2 = P_nsrc > 2 ? PF(src[2].sb) : 0;
  const int mrow0 = ty * BG_T + srow, mrow1 = mrow0 + 16, mrow2 = mrow0 + 32, mrow3 = mrow0 + 48;
  // gather indices of both pieces in the three sources: one round of unconditional loads (out of range = 0)
#define GIDX(ga, mrow) __builtin_bit_cast(int, ldb1(make_rsrc((ga) ? (const void*)(ga) : (const void*)P_wt), ((ga) && (mrow) < P_R) ? (mrow) * 4 : P_OOB))
  const int g00 = GIDX(ga0, mrow0), g01 = GIDX(ga0, mrow1), g02 = GIDX(ga0, mrow2), g03 = GIDX(ga0, mrow3);
  const int g10 = GIDX(ga1, mrow0), g11 = GIDX(ga1, mrow1), g12 = GIDX(ga1, mrow2), g13 = GIDX(ga1, mrow3);
  const int g20 = GIDX(ga2, mrow0), g21 = GIDX(ga2, mrow1), g22 = GIDX(ga2, mrow2), g23 = GIDX(ga2, mrow3);
#undef GIDX
#define ROFF(on, ga, g, mrow, rsb) (((on) && (mrow) < P_R) ? (int)(((ga) ? (long)(g) : (long)(mrow)) * (rsb) * 4) : P_OOB)
  const int o00 = ROFF(P_nsrc > 0, ga0, g00, mrow0, rs0), o01 = ROFF(P_nsrc > 0, ga0, g01, mrow1, rs0), o02 = ROFF(P_nsrc > 0, ga0, g02, mrow2, rs0), o03 = ROFF(P_nsrc > 0, ga0, g03, mrow3, rs0);
  const int o10 = ROFF(P_nsrc > 1, ga1, g10, mrow0, rs1), o11 = ROFF(P_nsrc > 1, ga1, g11, mrow1, rs1), o12 = ROFF(P_nsrc > 1, ga1, g12, mrow2, rs1), o13 = ROFF(P_nsrc > 1, ga1, g13, mrow3, rs1);
  const int o20 = ROFF(P_nsrc > 2, ga2, g20, mrow0, rs2), o21 = ROFF(P_nsrc > 2, ga2, g21, mrow1, rs2), o22 = ROFF(P_nsrc > 2, ga2, g22, mrow2, rs2), o23 = ROFF(P_nsrc > 2, ga2, g23, mrow3, rs2);
#undef ROFF
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(P_wt);
  const int n0 = tx * BG_T + srow, n1 = n0 + 16, n2 = n0 + 32, n3 = n0 + 48;
  const int woff0 = n0 < P_N ? (int)((long)n0 * P_ldw * 4) : P_OOB, woff1 = n1 < P_N ? (int)((long)n1 * P_ldw * 4) : P_OOB;
  const int woff2 = n2 < P_N ? (int)((long)n2 * P_ldw * 4) : P_OOB, woff3 = n3 < P_N ? (int)((long)n3 * P_ldw * 4) : P_OOB;

  f32x4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  // operand fetch of stage ST into ra0 / ra1 / rb0 / rb1
#ifdef BG_NO_LOAD      // dissection build (tools/beam_gemm_dissect.sh): everything but the operand fetch
#define BG_LOAD1(ra, rb, rs, in, ao, wo, kl, kw) ra = f32x4{(float)(kl), 1.f, 2.f, (float)(ao)}; rb = f32x4{(float)(kw), 1.f, 2.f, (float)(wo)};
#else
#define BG_LOAD1(ra, rb, rs, in, ao, wo, kl, kw)                                                                  \
    ra = ldb4(rs, ((in) && (ao) != P_OOB) ? (ao) + (kl) * 4 : P_OOB);                                             \
    rb = ldb4(wrs, ((in) && (wo) != P_OOB) ? (wo) + (kw) : P_OOB);
#endif
#define BG_FETCH(ST) {                                                                                             \
    const int st_ = (ST);                                                                                          \
    const int s_ = __builtin_amdgcn_readfirstlane(st_ >= sb2 ? 2 : (st_ >= sb1 ? 1 : 0));                          \
    const int st0_ = s_ == 0 ? 0 : (s_ == 1 ? sb1 : sb2);                                                          \
    const int Ks_ = s_ == 0 ? K0 : (s_ == 1 ? K1 : K2);                                                            \
    const int kr_ = s_ == 0 ? 0 : (s_ == 1 ? K0 : K0 + K1);                                                        \
    const int kl_ = (st_ - st0_) * BG_K + sk;                                                                      \
    const bool in_ = kl_ < Ks_;                                                                                    \
    /* the resource is rebuilt from a pointer forced into SGPRs (a selected resource value is treated as divergent) */ \
    const float* ab_ = s_ == 0 ? ab0 : (s_ == 1 ? ab1 : ab2);                                                      \
    const unsigned long au_ = reinterpret_cast<unsigned long>(ab_);                                               \
    const unsigned long auu_ = ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(au_ >> 32)) << 32) | \
                               (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)au_);                     \
    const __amdgpu_buffer_rsrc_t rs_ = make_rsrc(reinterpret_cast<const float*>(auu_));                          \
    const int kw_ = (kr_ + kl_) * 4;                                                                               \
    const int ao0_ = s_ == 0 ? o00 : (s_ == 1 ? o10 : o20), ao1_ = s_ == 0 ? o01 : (s_ == 1 ? o11 : o21);         \
    const int ao2_ = s_ == 0 ? o02 : (s_ == 1 ? o12 : o22), ao3_ = s_ == 0 ? o03 : (s_ == 1 ? o13 : o23);         \
    BG_LOAD1(ra0, rb0, rs_, in_, ao0_, woff0, kl_, kw_) BG_LOAD1(ra1, rb1, rs_, in_, ao1_, woff1, kl_, kw_)        \
    BG_LOAD1(ra2, rb2, rs_, in_, ao2_, woff2, kl_, kw_) BG_LOAD1(ra3, rb3, rs_, in_, ao3_, woff3, kl_, kw_) }
#define BG_COMMIT(BUF) {                                                                                           \
    *reinterpret_cast<f32x4*>(&lds[BUF][0][srow * BG_P + sk]) = ra0;                                               \
    *reinterpret_cast<f32x4*>(&lds[BUF][1][srow * BG_P + sk]) = rb0;                                               \
    *reinterpret_cast<f32x4*>(&lds[BUF][0][(srow + 16) * BG_P + sk]) = ra1;                                        \
    *reinterpret_cast<f32x4*>(&lds[BUF][1][(srow + 16) * BG_P + sk]) = rb1;                                        \
    *reinterpret_cast<f32x4*>(&lds[BUF][0][(srow + 32) * BG_P + sk]) = ra2;                                        \
    *reinterpret_cast<f32x4*>(&lds[BUF][1][(srow + 32) * BG_P + sk]) = rb2;                                        \
    *reinterpret_cast<f32x4*>(&lds[BUF][0][(srow + 48) * BG_P + sk]) = ra3;                                        \
    *reinterpret_cast<f32x4*>(&lds[BUF][1][(srow + 48) * BG_P + sk]) = rb3; }

  // LSTM epilogue operands (parent row -> previous cell state; bias) requested now: two dependent round trips that would otherwise
  // sit behind the last stage
  float e_c[2] = {0.f, 0.f};
  f32x4 e_b[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if constexpr (LSTM) {
    const float* const P_bias = PF(bias); const float* const P_c_in = PF(c_in); const int* const P_parent = PF(parent);
    const int H = P_N >> 2;
    const __amdgpu_buffer_rsrc_t prs = make_rsrc(P_parent ? (const void*)P_parent : (const void*)P_wt), crs = make_rsrc(P_c_in), brs = make_rsrc(P_bias ? P_bias : P_wt);
    int pr[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = tid + 512 * it, row = ty * BG_T + (item >> 4);
      pr[it] = __builtin_bit_cast(int, ldb1(prs, (P_parent && row < P_R) ? row * 4 : P_OOB));
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = tid + 512 * it, row = ty * BG_T + (item >> 4), u = tx * 16 + (item & 15);
      const bool ok = row < P_R && u < H;
      const long prow = P_parent ? (long)pr[it] : (long)row;
      e_c[it] = ldb1(crs, ok ? (int)((prow * H + u) * 4) : P_OOB);
      e_b[it] = ldb4(brs, (ok && P_bias) ? u * 16 : P_OOB);
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  BG_FETCH(0)
  BG_COMMIT(0)
  if (nstage > 1) BG_FETCH(1)
  __syncthreads();
  // lane half h of wave-half kh owns k in [64 kh + 32 h, + 32) of a stage: 8 ds_read_b128 per operand, instruction i multiplies the
  // pair (i, 32 + i) of the wave's 64 k
  const int hk = 64 * kh + 32 * (lane >> 5);
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    const float* Ar = lds[buf][0] + (wm * 32 + (lane & 31)) * BG_P + hk;
    const float* Br = lds[buf][1] + (wn * 32 + (lane & 31)) * BG_P + hk;
    f32x4 a4[8], b4[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a4[j] = *reinterpret_cast<const f32x4*>(Ar + 4 * j);
      b4[j] = *reinterpret_cast<const f32x4*>(Br + 4 * j);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#ifdef BG_NO_MFMA      // dissection build: everything but the matrix instructions
        acc[(4 * j + e) & 15] += a4[j][e] * b4[j][e];
#else
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j][e], b4[j][e], acc, 0, 0, 0);
#endif
      }
    if (st + 1 < nstage) {
      if (buf) BG_COMMIT(0) else BG_COMMIT(1)                 // the stage fetched while the previous one was multiplied
      if (st + 2 < nstage) BG_FETCH(st + 2)
    }
    lds_barrier();
  }
#undef BG_FETCH
#undef BG_COMMIT
#undef BG_LOAD1
  // ---- the two k halves of a quadrant meet: waves 4-7 hand their accumulators to waves 0-3 through LDS ----
  {
    float* X = &lds[0][0][0] + ((wave & 3) * 64 + lane) * 17;             // 17-float stride: conflict-free scalar rows
    if (kh == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) X[r] = acc[r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += X[r];
    }
    __syncthreads();
  }

  // C layout of mfma 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  if constexpr (!LSTM) {
    float* const P_out = PF(out);
    const long P_out_sb = PF(out_sb);
    const int col = tx * BG_T + wn * 32 + (lane & 31);
    const int rbase = ty * BG_T + wm * 32 + 4 * (lane >> 5);
    if (kh == 0 && col < P_N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        if (row < P_R) P_out[(long)row * P_out_sb + col] = acc[r];
      }
    }
  } else {
    // gate pre-activations of a unit sit in four adjacent columns: through LDS into (row, unit) order
    constexpr int CS = BG_P;                                   // row stride of the staged tile (16-byte aligned rows)
    float* Cs = &lds[1][0][0];
    if (kh == 0) {
      const int col = wn * 32 + (lane & 31), rbase = wm * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) Cs[(rbase + (r & 3) + 8 * (r >> 2)) * CS + col] = acc[r];
    }
    __syncthreads();
    float* const P_c_out = PF(c_out); float* const P_h_out = PF(h_out); float* const P_seq_out = PF(seq_out);
    const long P_seq_sb = PF(seq_sb);
    const int H = P_N >> 2;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = tid + 512 * it, rl = item >> 4, ul = item & 15;
      const int row = ty * BG_T + rl, u = tx * 16 + ul;
      if (row >= P_R || u >= H) continue;
      f32x4 z = *reinterpret_cast<const f32x4*>(&Cs[rl * CS + 4 * ul]);
      z += e_b[it];
      const float cprev = e_c[it];                           // the previous state lives in the parent hypothesis' row (fetched in the prologue)
      const float gi = p_sigmoid(z[0]), gj = p_tanh(z[1]), gf = p_sigmoid(z[2] + 1.0f), go = p_sigmoid(z[3]);   // cells.py:14-18
      float c = gf * cprev + gi * gj;
      c = fminf(1.0f, fmaxf(-1.0f, c));                        // cell_clip = 1.0
      const float h = go * p_tanh(c);
      P_c_out[(long)row * H + u] = c;
      P_h_out[(long)row * H + u] = h;
      if (P_seq_out) P_seq_out[(long)row * P_seq_sb + u] = h;
    }
  }
#undef PF
}

// ctx[r] = sum_j w_j pctx[j][r] with the softmax-merge weights of SRC_SOFTMAX (step.hip): M = max_j pm_j, w_j = exp(pm_j - M) / sum_j exp(pm_j - M) pl_j
struct BCMech { const float* pm; const float* pl; const float* pctx; float* ctx; long ctx_sb; int D, nslab; };
struct BCLaunch { int nmech, R; BCMech m[AVSR_MAX_MECH]; };

__global__ __launch_bounds__(256) void beam_ctx_merge_kernel(const BCLaunch L) {
  const BCMech& M = L.m[blockIdx.y];
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);          // one wave per hypothesis row, four rows per workgroup
  if (r >= L.R) return;
  // statistics and partial contexts through unconditional buffer loads (out of range = 0): all slabs of a row in flight together
  const __amdgpu_buffer_rsrc_t pm_rs = make_rsrc(M.pm), pl_rs = make_rsrc(M.pl), pc_rs = make_rsrc(M.pctx);
  float w[STEP_MAX_SLAB], plv[STEP_MAX_SLAB];
#pragma unroll
  for (int j = 0; j < STEP_MAX_SLAB; ++j) {
    const int o = j < M.nslab ? (j * L.R + r) * 4 : P_OOB;
    w[j] = ldb1(pm_rs, o);
    plv[j] = ldb1(pl_rs, o);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < STEP_MAX_SLAB; ++j) {
    w[j] = j < M.nslab ? w[j] : -INFINITY;
    mx = fmaxf(mx, w[j]);
  }
  float ls = 0.f;
#pragma unroll
  for (int j = 0; j < STEP_MAX_SLAB; ++j) {
    const float e = (w[j] == -INFINITY) ? 0.f : expf(w[j] - mx);
    ls += e * plv[j];                                   // plv is 0 for slabs that do not exist
    w[j] = e;
  }
  const float inv = ls > 0.f ? 1.0f / ls : 0.f;
  for (int c4 = threadIdx.x & 63; 4 * c4 < M.D; c4 += 64) {
    f32x4 sv[STEP_MAX_SLAB];
#pragma unroll
    for (int j = 0; j < STEP_MAX_SLAB; ++j) sv[j] = ldb4(pc_rs, j < M.nslab ? (int)((((long)j * L.R + r) * M.D + 4 * c4) * 4) : P_OOB);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < STEP_MAX_SLAB; ++j) acc += (w[j] * inv) * sv[j];
    st4(M.ctx + (long)r * M.ctx_sb + 4 * c4, acc);
  }
}

// ---- host side (called from avsr_attn_rnn_fwd, mode 3) --------------------------------------------------------------------------
int g_beam_dense = -1;        // -1: read AVSR_BEAM_DENSE once (default on)

bool beam_dense_on() {
  if (g_beam_dense < 0) { const char* e = getenv("AVSR_BEAM_DENSE"); g_beam_dense = e ? (atoi(e) != 0) : 1; }
  return g_beam_dense != 0;
}

// the LSTM cell step of all B rows.  Returns AVSR_ERR_UNSUPPORTED where the tiled kernel does not apply (the caller then runs step_kernel).
int beam_cell_launch(const avsr_attn_rnn& d, int l, hipStream_t s) {
  const int B = d.B, H = d.H, E = d.E, L = d.L, A = d.n_mech * H, KW = E + A + H;
  if (d.cell != 0 || d.n_extra != 0 || E % 16 || H % 16 || E <= 0 || !d.embedding || !d.tok || !d.parent_rows) return AVSR_ERR_UNSUPPORTED;
  if ((long)d.V * E * 4 >= (1L << 31) || (long)B * (L + 1) * (A > H ? A : H) * 4 >= (1L << 31) || (long)4 * H * KW * 4 >= (1L << 31)) return AVSR_ERR_UNSUPPORTED;
  static thread_local BGLaunch G;
  G = BGLaunch{};
  BGProb& P = G.p[0];
  BGSrc& x = P.src[P.nsrc++];
  x.a = d.embedding; x.sb = E; x.gather = d.tok; x.K = E;
  if (A > 0) { BGSrc& a = P.src[P.nsrc++]; a.a = d.att + (long)l * A; a.sb = (long)(L + 1) * A; a.gather = d.parent_rows; a.K = A; }
  BGSrc& h = P.src[P.nsrc++];
  h.a = d.state + (long)(l & 1) * B * H; h.sb = H; h.gather = d.parent_rows; h.K = H;
  P.R = B; P.N = 4 * H; P.wt = d.wt; P.ldw = KW;
  P.ntx = (P.N + BG_T - 1) / BG_T; P.tile0 = 0;
  P.bias = d.bias; P.c_in = d.state + (long)(2 + (l & 1)) * B * H; P.parent = d.parent_rows;
  P.c_out = d.state + (long)(2 + ((l + 1) & 1)) * B * H; P.h_out = d.state + (long)((l + 1) & 1) * B * H;
  P.seq_out = d.cell_out + (long)(l + 1) * H; P.seq_sb = (long)(L + 1) * H;
  G.nprob = 1; G.ntiles = P.ntx * ((B + BG_T - 1) / BG_T);
  ProfScope ps(PROF_STEP_LSTM_FWD, s);
  hipLaunchKernelGGL(beam_gemm_kernel<true>, dim3(G.ntiles), dim3(512), 0, s, G);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

// context merge + attention layers of all mechanisms (two launches).  nc[m] = chunks of mechanism m.
int beam_attention_layer_launch(const avsr_attn_rnn& d, int l, hipStream_t s) {
  const int B = d.B, H = d.H, L = d.L, A = d.n_mech * H;
  if (d.n_mech < 1 || d.n_mech > BG_MAX_PROB || H % 16) return AVSR_ERR_UNSUPPORTED;
  static thread_local BGLaunch G;
  static thread_local BCLaunch C;
  G = BGLaunch{}; C = BCLaunch{};
  C.nmech = d.n_mech; C.R = B;
  int tiles = 0, dmax = 0;
  for (int m = 0; m < d.n_mech; ++m) {
    const avsr_attn_mech& M = d.mech[m];
    const int nc = (M.T + M.chunk - 1) / M.chunk;
    if (M.D % 16 || nc > STEP_MAX_SLAB || M.D > 1024 || !M.ctx || (long)nc * B * M.D * 4 >= (1L << 31) || (long)B * L * M.D * 4 >= (1L << 31) || (long)H * (H + M.D) * 4 >= (1L << 31)) return AVSR_ERR_UNSUPPORTED;
    BCMech& X = C.m[m];
    X.pm = M.pstat + (long)(2 * l) * nc * B; X.pl = M.pstat + (long)(2 * l + 1) * nc * B; X.pctx = M.pctx;
    X.ctx = M.ctx + (long)l * M.D; X.ctx_sb = (long)L * M.D; X.D = M.D; X.nslab = nc;
    dmax = M.D > dmax ? M.D : dmax;
    BGProb& P = G.p[m];
    BGSrc& q = P.src[P.nsrc++];
    q.a = d.cell_out + (long)(l + 1) * H; q.sb = (long)(L + 1) * H; q.K = H;
    BGSrc& c = P.src[P.nsrc++];
    c.a = X.ctx; c.sb = X.ctx_sb; c.K = M.D;
    P.R = B; P.N = H; P.wt = M.watt_t; P.ldw = H + M.D;
    P.ntx = (H + BG_T - 1) / BG_T; P.tile0 = tiles;
    tiles += P.ntx * ((B + BG_T - 1) / BG_T);
    P.out = d.att + (long)(l + 1) * A + (long)m * H; P.out_sb = (long)(L + 1) * A;
  }
  G.nprob = d.n_mech; G.ntiles = tiles;
  ProfScope ps(PROF_STEP_LINEAR, s);
  hipLaunchKernelGGL(beam_ctx_merge_kernel, dim3((B + 3) / 4, d.n_mech), dim3(256), 0, s, C);
  hipLaunchKernelGGL(beam_gemm_kernel<false>, dim3(tiles), dim3(512), 0, s, G);
  AVSR_CHECK_LAUNCH();
  return AVSR_OK;
}

}  // namespace avsr
