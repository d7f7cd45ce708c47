"""Time of the weight-gradient kernel alone on one layer (final slab reductions deferred); the library is rebuilt per ablation by
tools/wgrad_ablate.sh.  usage: python tools/wgrad_ablate.py label H,Ci,Co ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avsr_tf1_amd import ops                     # noqa: E402
import tools.conv_bench as cb                    # noqa: E402

label = sys.argv[1]
out = []
for spec in sys.argv[2:]:
    H, Ci, Co = [int(v) for v in spec.split(",")]
    N, k, s = 4800, 3, 1
    x = torch.randn(N, H, H, Ci, device="cuda")
    dy = torch.randn(N, H, H, Co, device="cuda")
    dw, db = torch.zeros(k, k, Ci, Co, device="cuda"), torch.zeros(Co, device="cuda")
    scratch = torch.zeros(1 << 24, device="cuda")
    d = ops.conv_desc(N, H, H, Ci, Co, k, s, 1, 1, H, H)
    ops.slab_defer_begin()
    t = cb.timeit(lambda: ops.conv_bwd_weight(d, x, dy, dw, db, scratch))
    ops.slab_defer_end()
    torch.cuda.synchronize()
    out.append("%s %7.1f us" % (spec, t))
print("%-34s %s" % (label, "   ".join(out)))
