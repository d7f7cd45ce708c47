// ABI self-description: lets the ctypes binding verify struct layouts before any launch.
#include <string.h>
#include "avsr_hip.h"

extern "C" int avsr_abi_version(void) { return 2; }

extern "C" int64_t avsr_sizeof(const char* name) {
  if (!name) return -1;
#define SZ(T) if (!strcmp(name, #T)) return (int64_t)sizeof(T);
  SZ(avsr_mat) SZ(avsr_gemm_desc) SZ(avsr_conv_desc) SZ(avsr_rnn_layer) SZ(avsr_rnn_stack) SZ(avsr_colsum_job)
#ifdef AVSR_HAVE_ATTN
  SZ(avsr_attn_mech) SZ(avsr_attn_rnn) SZ(avsr_transpose_job) SZ(avsr_dec_layer)
#endif
#undef SZ
  return -1;
}

// ---- per-kernel event profiler ---------------------------------------------------------------------
#include <vector>
#include "prof.h"
namespace avsr {
bool g_prof_enabled = false;
static std::vector<hipEvent_t> g_ev;      // pairs: [2i] start, [2i+1] stop
static std::vector<int> g_kind;
static size_t g_used = 0;
static double g_work[PROF_NKIND];
void prof_record(int kind, hipStream_t s, bool begin, double work) {
  if (begin) {
    if (g_used >= g_kind.size()) { g_prof_enabled = false; return; }
    g_kind[g_used] = kind;
    g_work[kind] += work;
    hipEventRecord(g_ev[2 * g_used], s);
  } else {
    hipEventRecord(g_ev[2 * g_used + 1], s);
    ++g_used;
  }
}
}  // namespace avsr

extern "C" int avsr_prof_begin(int32_t max_launches) {
  using namespace avsr;
  if (max_launches <= 0) return AVSR_ERR_ARG;
  while (g_ev.size() < (size_t)2 * max_launches) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return AVSR_ERR_HIP;
    g_ev.push_back(e);
  }
  g_kind.assign(max_launches, 0);
  g_used = 0;
  for (int k = 0; k < PROF_NKIND; ++k) g_work[k] = 0.0;
  g_prof_enabled = true;
  return AVSR_OK;
}

// out_count[k], out_ms[k], out_flops[k] for k < AVSR_PROF_NKIND (gemm, lstm_fwd step, lstm_bwd step, dense step, attn fwd,
// attn bwd, persistent rnn fwd, persistent rnn bwd); out_flops (may be NULL) = algorithmic FLOPs where the launcher knows them
extern "C" int avsr_prof_end(int32_t* out_count, float* out_ms, double* out_flops) {
  using namespace avsr;
  g_prof_enabled = false;
  if (!out_count || !out_ms) return AVSR_ERR_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return AVSR_ERR_HIP;
  for (int k = 0; k < PROF_NKIND; ++k) { out_count[k] = 0; out_ms[k] = 0.f; if (out_flops) out_flops[k] = g_work[k]; }
  for (size_t i = 0; i < g_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_ev[2 * i], g_ev[2 * i + 1]) != hipSuccess) return AVSR_ERR_HIP;
    out_count[g_kind[i]] += 1;
    out_ms[g_kind[i]] += ms;
  }
  return AVSR_OK;
}
