// ABI self-description: lets the ctypes binding verify struct layouts before any launch.
#include <string.h>
#include "avsr_hip.h"

extern "C" int avsr_abi_version(void) { return 1; }

extern "C" int64_t avsr_sizeof(const char* name) {
  if (!name) return -1;
#define SZ(T) if (!strcmp(name, #T)) return (int64_t)sizeof(T);
  SZ(avsr_mat) SZ(avsr_gemm_desc) SZ(avsr_rnn_layer) SZ(avsr_rnn_stack)
#ifdef AVSR_HAVE_ATTN
  SZ(avsr_attn_mech) SZ(avsr_attn_rnn) SZ(avsr_transpose_job)
#endif
#undef SZ
  return -1;
}
