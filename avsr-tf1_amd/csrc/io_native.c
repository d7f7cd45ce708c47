/* Native TFRecord SequenceExample indexer + batch filler (include/avsr_io.h).  Plain C99 + pthreads, no dependency.
 *
 * Wire format walked here (tensorflow/core/example/{example,feature}.proto):
 *   SequenceExample { Features context = 1; FeatureLists feature_lists = 2; }
 *   Features     { map<string, Feature> feature = 1; }          map entry = { key = 1; value = 2; }
 *   FeatureLists { map<string, FeatureList> feature_list = 1; }
 *   FeatureList  { repeated Feature feature = 1; }
 *   Feature      { BytesList bytes_list = 1 | FloatList float_list = 2 | Int64List int64_list = 3 }   each { repeated T value = 1 }
 * The fast layouts are the ones io_utils._fast_float_steps / _fast_small_int_steps recognise (same tests, same order), so a record is
 * either indexed here or handed back (`slow`) to that parser. */
#include "avsr_io.h"

#include <pthread.h>
#include <string.h>

typedef struct { const uint8_t* p; int64_t n; } span;

/* varint at x[i]; returns the next index or -1 when it runs off the buffer */
static int64_t rd_varint(const uint8_t* x, int64_t n, int64_t i, uint64_t* v) {
  uint64_t r = 0;
  int s = 0;
  while (i < n && s < 64) {
    const uint8_t b = x[i++];
    r |= (uint64_t)(b & 0x7F) << s;
    if (b < 0x80) { *v = r; return i; }
    s += 7;
  }
  return -1;
}

/* next field of a message: returns the next index (or -1 on malformed input); length-delimited / fixed values come back as a span */
static int64_t rd_field(const uint8_t* x, int64_t n, int64_t i, int* fn, int* wt, uint64_t* val, span* sp) {
  uint64_t key;
  i = rd_varint(x, n, i, &key);
  if (i < 0) return -1;
  *fn = (int)(key >> 3); *wt = (int)(key & 7);
  sp->p = 0; sp->n = 0; *val = 0;
  if (*wt == 0) return rd_varint(x, n, i, val);
  if (*wt == 2) {
    uint64_t ln;
    i = rd_varint(x, n, i, &ln);
    if (i < 0 || ln > (uint64_t)(n - i)) return -1;        /* unsigned: a length >= 2^63 must not turn into a negative span */
    sp->p = x + i; sp->n = (int64_t)ln;
    return i + (int64_t)ln;
  }
  if (*wt == 5) { if (n - i < 4) return -1; sp->p = x + i; sp->n = 4; return i + 4; }
  if (*wt == 1) { if (n - i < 8) return -1; sp->p = x + i; sp->n = 8; return i + 8; }
  return -1;
}

static int key_is(span k, const char* s) { const size_t l = strlen(s); return k.n == (int64_t)l && memcmp(k.p, s, l) == 0; }

/* first value of a Feature's Int64List (packed or not); 0 = found */
static int feature_first_int(span f, int64_t* out) {
  int64_t i = 0; int fn, wt; uint64_t v; span sp;
  while (i < f.n) {
    i = rd_field(f.p, f.n, i, &fn, &wt, &v, &sp);
    if (i < 0) return -1;
    if (fn != 3 || wt != 2) continue;
    int64_t j = 0;
    while (j < sp.n) {
      int fn2, wt2; uint64_t v2; span sp2;
      j = rd_field(sp.p, sp.n, j, &fn2, &wt2, &v2, &sp2);
      if (j < 0) return -1;
      if (fn2 != 1) continue;
      if (wt2 == 0) { *out = (int64_t)v2; return 0; }
      if (wt2 == 2) { uint64_t v3; if (rd_varint(sp2.p, sp2.n, 0, &v3) < 0) return -1; *out = (int64_t)v3; return 0; }
    }
    return -1;
  }
  return -1;
}

/* first value of a Feature's BytesList; 0 = found */
static int feature_first_bytes(span f, span* out) {
  int64_t i = 0; int fn, wt; uint64_t v; span sp;
  while (i < f.n) {
    i = rd_field(f.p, f.n, i, &fn, &wt, &v, &sp);
    if (i < 0) return -1;
    if (fn != 1 || wt != 2) continue;
    int64_t j = 0;
    while (j < sp.n) {
      int fn2, wt2; uint64_t v2; span sp2;
      j = rd_field(sp.p, sp.n, j, &fn2, &wt2, &v2, &sp2);
      if (j < 0) return -1;
      if (fn2 == 1 && wt2 == 2) { *out = sp2; return 0; }
    }
    return -1;
  }
  return -1;
}

/* FeatureList of equally sized packed-float steps (io_utils._fast_float_steps): 0 = recognised */
static int fast_float_steps(span x, int64_t* h_out, int64_t* stride_out, int64_t* T_out, int64_t* F_out) {
  const int64_t n = x.n;
  uint64_t l1, l2, l3;
  if (n < 8 || x.p[0] != 0x0A) return -1;
  const int64_t i = rd_varint(x.p, n, 1, &l1);
  if (i < 0 || l1 > (uint64_t)(n - i)) return -1;          /* unsigned compares: corrupt lengths >= 2^63 are rejected, not cast */
  const int64_t stride = i + (int64_t)l1;
  if (stride <= 0 || n % stride || i >= n || x.p[i] != 0x12) return -1;
  const int64_t j = rd_varint(x.p, n, i + 1, &l2);
  if (j < 0 || j >= n || l2 > (uint64_t)n || x.p[j] != 0x0A) return -1;
  const int64_t h = rd_varint(x.p, n, j + 1, &l3);
  if (h < 0 || h > stride || l3 > (uint64_t)n) return -1;
  if (l3 % 4 || h + (int64_t)l3 != stride || (int64_t)l2 != (h - j) + (int64_t)l3) return -1;
  const int64_t T = n / stride;
  for (int64_t t = 1; t < T; ++t)
    if (memcmp(x.p + t * stride, x.p, (size_t)h) != 0) return -1;
  *h_out = h; *stride_out = stride; *T_out = T; *F_out = (int64_t)l3 / 4;
  return 0;
}

/* FeatureList of single one-byte packed int64 values per step (io_utils._fast_small_int_steps): 0 = recognised */
static int fast_small_int_steps(span x, int64_t* h_out, int64_t* stride_out, int64_t* n_out) {
  const int64_t n = x.n;
  uint64_t l1, l2, l3;
  if (n < 7 || x.p[0] != 0x0A) return -1;
  const int64_t i = rd_varint(x.p, n, 1, &l1);
  if (i < 0 || l1 > (uint64_t)(n - i)) return -1;
  const int64_t stride = i + (int64_t)l1;
  if (stride <= 0 || n % stride || i >= n || x.p[i] != 0x1A) return -1;
  const int64_t j = rd_varint(x.p, n, i + 1, &l2);
  if (j < 0 || j >= n || l2 > (uint64_t)n || x.p[j] != 0x0A) return -1;
  const int64_t h = rd_varint(x.p, n, j + 1, &l3);
  if (h < 0 || h > stride || l3 != 1 || h + 1 != stride) return -1;
  const int64_t cnt = n / stride;
  for (int64_t t = 0; t < cnt; ++t) {
    if (t && memcmp(x.p + t * stride, x.p, (size_t)h) != 0) return -1;
    if (x.p[t * stride + h] >= 0x80) return -1;
  }
  *h_out = h; *stride_out = stride; *n_out = cnt;
  return 0;
}

static void index_one(const uint8_t* buf, int64_t len, avsr_io_rec* r) {
  memset(r, 0, sizeof(*r));
  r->input_length = -1; r->labels_length = -1; r->fn_off = -1;
  int64_t i = 0; int fn, wt; uint64_t v; span top;
  while (i < len) {
    i = rd_field(buf, len, i, &fn, &wt, &v, &top);
    if (i < 0) { r->slow = 1; return; }
    if (wt != 2 || (fn != 1 && fn != 2)) continue;
    int64_t j = 0;
    while (j < top.n) {                                   /* map entries */
      int f2, w2; uint64_t v2; span entry;
      j = rd_field(top.p, top.n, j, &f2, &w2, &v2, &entry);
      if (j < 0) { r->slow = 1; return; }
      if (f2 != 1 || w2 != 2) continue;
      span key = {0, 0}, val = {0, 0};
      int64_t k = 0;
      while (k < entry.n) {
        int f3, w3; uint64_t v3; span sp;
        k = rd_field(entry.p, entry.n, k, &f3, &w3, &v3, &sp);
        if (k < 0) { r->slow = 1; return; }
        if (f3 == 1 && w3 == 2) key = sp;
        else if (f3 == 2 && w3 == 2) val = sp;
      }
      if (!key.p) continue;
      if (fn == 1) {                                      /* context */
        if (key_is(key, "input_length")) { if (feature_first_int(val, &r->input_length)) { r->slow = 1; return; } }
        else if (key_is(key, "labels_length")) { if (feature_first_int(val, &r->labels_length)) { r->slow = 1; return; } }
        else if (key_is(key, "filename")) {
          span b;
          if (feature_first_bytes(val, &b)) { r->slow = 1; return; }
          r->fn_off = b.p - buf; r->fn_len = b.n;
        }
      } else {                                            /* feature lists */
        int64_t h, st, T, F;
        if (key_is(key, "inputs")) {
          if (val.n == 0) continue;
          if (fast_float_steps(val, &h, &st, &T, &F)) { r->slow = 1; return; }
          r->in_off = (val.p - buf) + h; r->in_stride = st; r->in_T = T; r->in_F = F;
        } else if (key_is(key, "aus")) {
          if (val.n == 0) continue;
          if (fast_float_steps(val, &h, &st, &T, &F)) { r->slow = 1; return; }
          r->aus_off = (val.p - buf) + h; r->aus_stride = st; r->aus_T = T; r->aus_F = F;
        } else if (key_is(key, "labels")) {
          if (val.n == 0) continue;
          if (fast_small_int_steps(val, &h, &st, &T)) { r->slow = 1; return; }
          r->lab_off = (val.p - buf) + h; r->lab_stride = st; r->lab_n = T;
        }
      }
    }
  }
}

/* ---- a few worker threads per call (created and joined here: calls are per chunk of records / per batch) ---- */
typedef struct job {
  int kind, lo, hi;
  const uint8_t* const* bufs; const int64_t* lens; avsr_io_rec* out;
  const int64_t* off; const int64_t* stride; const int64_t* steps; int64_t step_floats, Tmax, row_floats; float* dst;
} job;

static void* run_job(void* arg) {
  const job* j = (const job*)arg;
  if (j->kind == 0) {
    for (int b = j->lo; b < j->hi; ++b) index_one(j->bufs[b], j->lens[b], j->out + b);
  } else {
    const int64_t sb = j->step_floats * 4;
    for (int b = j->lo; b < j->hi; ++b) {
      const uint8_t* src = j->bufs[b] + j->off[b];
      uint8_t* d = (uint8_t*)(j->dst + (int64_t)b * j->Tmax * j->row_floats);
      const int64_t all = j->Tmax * j->row_floats * 4;
      /* never write past the utterance's slot, never read past its payload (lens may be NULL: callers that validated already) */
      int64_t steps = j->steps[b] < 0 ? 0 : j->steps[b];
      if (sb > 0 && steps * sb > all) steps = all / sb;
      if (j->lens && sb > 0 && steps > 0) {
        const int64_t st = j->stride[b] > 0 ? j->stride[b] : sb;
        const int64_t avail = j->lens[b] - j->off[b];
        if (j->off[b] < 0 || avail < sb) steps = 0;
        else if ((steps - 1) * st + sb > avail) steps = (avail - sb) / st + 1;
      }
      if (j->stride[b] == sb) memcpy(d, src, (size_t)(sb * steps));                  /* values back to back */
      else for (int64_t t = 0; t < steps; ++t) memcpy(d + t * sb, src + t * j->stride[b], (size_t)sb);
      const int64_t used = sb * steps;                                               /* zero padding behind the utterance */
      if (all > used) memset(d + used, 0, (size_t)(all - used));
    }
  }
  return 0;
}

static void run_parallel(job* proto, int n, int nthreads) {
  if (nthreads > 16) nthreads = 16;
  if (nthreads < 2 || n < 2 * nthreads) { proto->lo = 0; proto->hi = n; run_job(proto); return; }
  job jobs[16]; pthread_t th[16]; int started[16];
  for (int t = 0; t < nthreads; ++t) {
    jobs[t] = *proto; jobs[t].lo = (int)((int64_t)n * t / nthreads); jobs[t].hi = (int)((int64_t)n * (t + 1) / nthreads);
    started[t] = t > 0 && pthread_create(&th[t], 0, run_job, &jobs[t]) == 0;
  }
  run_job(&jobs[0]);
  for (int t = 1; t < nthreads; ++t) { if (started[t]) pthread_join(th[t], 0); else run_job(&jobs[t]); }
}

int avsr_io_abi_version(void) { return 2; }

int avsr_io_index(int32_t n, const uint8_t* const* bufs, const int64_t* lens, avsr_io_rec* out, int32_t nthreads) {
  if (n <= 0) return 0;
  job j; memset(&j, 0, sizeof(j));
  j.kind = 0; j.bufs = bufs; j.lens = lens; j.out = out;
  run_parallel(&j, n, nthreads);
  return 0;
}

int avsr_io_fill_f32(int32_t n, const uint8_t* const* bufs, const int64_t* lens, const int64_t* off, const int64_t* stride,
                     const int64_t* steps, int64_t step_floats, float* dst, int64_t Tmax, int64_t row_floats, int32_t nthreads) {
  if (n <= 0) return 0;
  job j; memset(&j, 0, sizeof(j));
  j.kind = 1; j.bufs = bufs; j.lens = lens; j.off = off; j.stride = stride; j.steps = steps; j.step_floats = step_floats; j.dst = dst; j.Tmax = Tmax;
  j.row_floats = row_floats;
  run_parallel(&j, n, nthreads);
  return 0;
}

int avsr_io_fill_labels(int32_t n, const uint8_t* const* bufs, const int64_t* off, const int64_t* stride, const int64_t* cnt,
                        int32_t eos, int32_t* dst, int64_t Lmax) {
  for (int b = 0; b < n; ++b) {
    const uint8_t* src = bufs[b] + off[b];
    int32_t* d = dst + (int64_t)b * Lmax;
    for (int64_t k = 0; k < cnt[b]; ++k) d[k] = (int32_t)src[k * stride[b]];
    d[cnt[b]] = eos;
  }
  return 0;
}
