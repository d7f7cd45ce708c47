// Device-side descriptors for the per-timestep "step" kernels (internal; the C-ABI is avsr_hip.h).
#pragma once
#include "common.h"

#define STEP_MAX_SRC 5
#define STEP_MAX_TASKS 8
#define STEP_MAX_SLAB 16

namespace avsr {

enum StepMode {
  EP_LINEAR = 0,     // out = act(z + bias + add)
  EP_LSTM_FWD = 1,   // z = 4 gate pre-activations per unit (gate-interleaved columns)
  EP_LSTM_BWD = 2,   // z = dL/dh for 16 units; emits d(gate pre-activations)
  EP_GRU_GATES = 3,  // z = [r,u] pre-activations (unit-interleaved: col = 2*unit + {0:r,1:u}); writes r*h and u
  EP_GRU_CAND = 4,   // z = candidate pre-activation; h' = u*h + (1-u)*tanh(z)
  EP_GRU_BWD_CAND = 5,
  EP_GRU_BWD_GATES = 6,
};

enum SrcKind { SRC_PLAIN = 0, SRC_SLABSUM = 1, SRC_SOFTMAX = 2, SRC_OWNROW = 3 };   // SRC_OWNROW: plain, never row-gathered

// One K-segment of the concatenated A operand and the matching weight rows.
//   A row b        : a + (gather ? gather[b] : b) * sb
//   weight for col n: w + n * ldw   (K contiguous -- "NT" form, both operands float4-loadable)
struct StepSrc {
  const float* a;
  const float* w;
  long sb;
  long ldw;
  int K;
  int kind;
};

struct StepTask {
  StepSrc src[STEP_MAX_SRC];
  int nsrc, B, N, mode;
  int t, T, reverse, act;          // time step of this launch; act: 0 none, 1 tanh, 2 sigmoid
  const int* len;                  // [B] valid lengths (null = all valid)
  const float* bias;               // [N]
  const int* gather;               // row gather for src[0] (embedding lookup)
  const int* gather2;              // row gather for every other source and for the previous-state reads (beam search parents)
  // softmax-partial / slab combine (applies to the src whose kind != SRC_PLAIN)
  const float* pm; const float* pl;  // [nslab][B] chunk max / chunk sum
  long slab_stride; int nslab; int pad0;
  float* ctx_save; long ctx_sb;      // optional: combined slab source written back (by column-tile 0)
  // epilogue I/O (meaning per mode; see step.hip)
  float* p0; float* p1; float* p2; float* p3; float* p4; float* p5; float* p6; float* p7; float* p8;
  float* p9; float* p10; float* p11;   // dropout-time extra outputs: state-dropped h sequence / second dense output,
                                       // consumer-input (x~) rolling buffer, x~ sequence
  long s0, s1, s2, s3, s4, s5;
  // DropoutWrapper (cells.py:46-54): stateless masks keyed by (*seed, stream, index); seed == null -> off
  const int32_t* seed;
  float k_st, k_out, k_in;             // keep prob: own state h, own output, the consumer's input mask
  uint32_t r_st, r_out, r_in;          // RNG stream ids
  int in_W, in_coff;                   // input-mask index = (b*T + tau) * in_W + in_coff + column
  int pad1;
};

struct StepLaunch {
  int ntask;
  StepTask task[STEP_MAX_TASKS];
};

}  // namespace avsr
