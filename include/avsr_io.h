/* avsr_io.h -- native TFRecord SequenceExample indexer + batch filler (host side, plain C, pthreads; libavsr_io.so).
 *
 * Replaces, for the record layouts avsr/dataset_writer.py produces, what the reference does inside tf.data's C++ parse ops
 * (avsr/io_utils.py:19-86: tf.parse_single_sequence_example of the data / label records, :113-127 padded_batch).  The python
 * pipeline (avsr_tf1_amd/io_utils.py) keeps the semantics - zip by position, EOS append, filter, shuffle buffer, group_by_window -
 * over light-weight record indices; the per-utterance interpreter work (protobuf field walking, array views, padding copies) moves
 * here.  Records whose layout is not the constant-stride one are flagged `slow` and parsed by the python parser: same results.
 */
#ifndef AVSR_IO_H
#define AVSR_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* One record (all offsets are byte offsets into the record payload; -1 / 0 when absent). */
typedef struct avsr_io_rec {
  int64_t slow;                         /* 1: layout not recognised -> use the generic parser for this record */
  int64_t input_length;                 /* context "input_length" (data records), -1 if absent */
  int64_t labels_length;                /* context "labels_length" (label records), -1 if absent */
  int64_t fn_off, fn_len;               /* context "filename" bytes */
  int64_t in_off, in_stride, in_T, in_F;        /* feature list "inputs": T steps of F packed floats, first value at in_off */
  int64_t aus_off, aus_stride, aus_T, aus_F;    /* feature list "aus" (Action Units), same form; aus_T = 0 when absent or empty */
  int64_t lab_off, lab_stride, lab_n;           /* feature list "labels": lab_n one-byte values (< 128) at lab_off + j*lab_stride */
} avsr_io_rec;

int avsr_io_abi_version(void);

/* Index n records (payload pointers + lengths) with up to nthreads threads.  Returns 0. */
int avsr_io_index(int32_t n, const uint8_t* const* bufs, const int64_t* lens, avsr_io_rec* out, int32_t nthreads);

/* dst[b, t, :] = the row_floats floats of step t of record b for t < T[b], zeros behind (dst need not be initialised).
 * dst is [n, Tmax, row_floats] float32.  Steps are F floats each and row_floats * T[b] == F * steps (a step may hold a whole row,
 * as every layout of the writer does).  The copy of record b is clamped to its slot (Tmax * row_floats floats) and, when `lens`
 * (payload byte lengths, may be NULL) is given, to its payload: an inconsistent index row can neither overrun dst nor read past
 * the record.  (ABI version 2: `lens` added.) */
int avsr_io_fill_f32(int32_t n, const uint8_t* const* bufs, const int64_t* lens, const int64_t* off, const int64_t* stride,
                     const int64_t* steps, int64_t step_floats, float* dst, int64_t Tmax, int64_t row_floats, int32_t nthreads);

/* dst[b, j] = label j of record b for j < cnt[b], dst[b, cnt[b]] = eos; dst is [n, Lmax] int32, zero on entry. */
int avsr_io_fill_labels(int32_t n, const uint8_t* const* bufs, const int64_t* off, const int64_t* stride, const int64_t* cnt,
                        int32_t eos, int32_t* dst, int64_t Lmax);

#ifdef __cplusplus
}
#endif
#endif
