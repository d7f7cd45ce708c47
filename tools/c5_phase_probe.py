"""Per-phase shader-clock ticks of the AV-Align attentive layer (dec_persist*.hip MODE 0, 16-row groups) inside a c5 train step.
Build with AVSR_HIPCC_FLAGS="-DDP_TIMING -DPERSIST_TIMING"; prints workgroup 0's mean ticks per audio frame, forward and BPTT, next to
the persistent encoder kernels' per-step ticks (tools/persist_probe.py) of the same step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--workload", "c5", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-graph"] + sys.argv[1:]
import bench  # noqa: E402

bench.main()
from avsr_tf1_amd import ops  # noqa: E402

h = ops._persist_sync[:256].cpu().numpy()
print("err", h[0])
F = ["P1 loads+mfma", "P1 epilogue", "publish0", "wait0", "P2 attention", "publish1", "wait1", "P3 attlayer", "publish2", "wait2", "P4 sample",
     "loop", "p2:scores", "p2:softmax", "p2:ctx", "p3:loads+wgt", "p3:mfma", "p4:logits", "p4:sample"]
Bk = ["wait2", "A loads+mfma", "A epilogue+stage2", "publish0", "wait0", "B dctx", "B dq", "publish1", "wait1", "C cell", "publish2", "B dalpha"]
f, b = h[176:176 + len(F)], h[200:200 + len(Bk)]
print("attentive layer forward, ticks per audio frame:", ", ".join("%s %d" % x for x in zip(F, f)), "| sum of phases 0-11:", int(f[:12].sum()))
print("attentive layer BPTT, ticks per audio frame:", ", ".join("%s %d" % x for x in zip(Bk, b)), "| sum of phases 0-10:", int(b[:11].sum()))
for name, base in (("fwd", 16), ("bwd", 80)):
    for ti in range(8):
        r = h[base + ti * 8:base + ti * 8 + 6]
        if r.any():
            print("encoder", name, "task", ti, "ticks/step", list(map(int, r)), "total", int(r.sum()))
