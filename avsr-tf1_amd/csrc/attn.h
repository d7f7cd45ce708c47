// Device descriptors for the attention kernels (internal).
#pragma once
#include "common.h"
#include "avsr_hip.h"

#define ATTN_MAX_CHUNK 128

namespace avsr {

enum AttnType { ATT_LUONG = 0, ATT_SCALED_LUONG = 1, ATT_BAHDANAU = 2, ATT_NORMED_BAHDANAU = 3 };

struct AttnMechDev {
  const float* keys;    // [B][T][H], row stride H, batch stride T*H
  const float* values;  // [B][.][D]: row (b,t) at values + b*values_sb + t*values_st
  long values_sb, values_st;
  const int* len;       // [B]
  const float* query;   // [B][H]: cell output (Luong) or processed query (Bahdanau); batch stride query_sb
  long query_sb;
  const float* g;       // scalar (scaled_luong) or null
  const float* v;       // [H] Bahdanau score vector (already g*v/|v| for the normed variant)
  const float* bq;      // [H] normed_bahdanau bias or null
  float* scores;        // [B][T] raw scores of this step (Luong: un-scaled dot), batch stride scores_sb
  long scores_sb;
  float* pm; float* pl; // [nchunk][B] chunk max / chunk exp-sum
  float* pctx;          // [nchunk][B][D] un-normalised partial contexts
  int T, D, H, type, nchunk, chunk;
  int mem_div;          // hypothesis row b attends memory row b / mem_div (beam search over un-tiled memories; 1 otherwise)
  // backward
  const float* dctx; long dctx_sb;   // [B][D]
  const float* ctx; long ctx_sb;     // [B][D] forward context of this step
  float* dscores; long dscores_sb;   // [B][T]
  float* pdq;                        // [nchunk][B][H] partial d(query) (Luong: d cell_out, Bahdanau: d processed query)
};

struct AttnLaunch {
  int nmech, B;
  int blk_off[AVSR_MAX_MECH + 1];
  AttnMechDev m[AVSR_MAX_MECH];
};

}  // namespace avsr
