// Optional per-launch HIP-event timing of the engine's kernels (used by bench.py for the roofline
// numbers).  Disabled by default: a ProfScope is then two predictable branches.
#pragma once
#include <hip/hip_runtime.h>

namespace avsr {
enum ProfKind { PROF_GEMM = 0, PROF_STEP_LSTM_FWD, PROF_STEP_LSTM_BWD, PROF_STEP_LINEAR, PROF_ATTN_FWD, PROF_ATTN_BWD,
                PROF_RNN_PERSIST_FWD, PROF_RNN_PERSIST_BWD, PROF_DEC_PERSIST_FWD, PROF_DEC_PERSIST_BWD, PROF_CONV_FWD, PROF_CONV_BWD_DATA, PROF_CONV_BWD_WEIGHT,
                PROF_ALIGN_PERSIST_FWD, PROF_ALIGN_PERSIST_BWD,   // the AV-Align attentive encoder layer through the fused persistent kernels
                PROF_NKIND };
void prof_record(int kind, hipStream_t s, bool begin, double work = 0.0);
extern bool g_prof_enabled;
struct ProfScope {
  int kind; hipStream_t s;
  // work = algorithmic FLOPs of the launch where the caller knows them (GEMM, persistent RNN kernels)
  ProfScope(int k, hipStream_t st, double work = 0.0) : kind(k), s(st) { if (g_prof_enabled) prof_record(kind, s, true, work); }
  ~ProfScope() { if (g_prof_enabled) prof_record(kind, s, false); }
};
}  // namespace avsr
