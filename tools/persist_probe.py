"""Phase timing of the persistent encoder kernel (build with AVSR_HIPCC_FLAGS=-DPERSIST_TIMING).
Prints, per (stack, layer) task, the mean shader-clock ticks per step that workgroup 0 of the task spent in:
wait | operand loads | mfma+lds | epilogue issue | store drain | arrival atomic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AVSR_PERSISTENT_RNN", "3")
sys.argv = [sys.argv[0], "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-graph"] + sys.argv[1:]
import bench  # noqa: E402

bench.main()
from avsr_tf1_amd import ops  # noqa: E402

h = ops._persist_sync[:256].cpu().numpy()
print("err", h[0])
for name, base in (("fwd", 16), ("bwd", 80)):
    for ti in range(8):
        r = h[base + ti * 8:base + ti * 8 + 6]
        if r.any():
            print(name, "task", ti, ("ticks/step: wait %d loads %d mfma %d epi %d drain %d publish %d  total %d" if name == "fwd" else "ticks/step: wait %d partials %d cell %d product+stores %d publish %d (-) %d  total %d") % (*r, r.sum()))
