"""Dissection probe of the data-path convolution kernels (profiles/r03_conv_dissection_v1.txt, r04_conv_deep_dissection.txt).  Needs a
library built with -DCONV_DEBUG (tools/conv_deep_dissect.sh rebuilds csrc/conv_mfma.hip on the GPU box); AVSR_CONV_DBG = bit0 no stores |
bit1 no LDS operand reads | bit2 no MFMAs | bit3 per-wave cycle stamps (set-up / prologue / barrier / commit / barrier / tile phase),
printed as mean / max / min over the waves.  AVSR_DISSECT_LAYER="H,Ci,Co,k,s[,bwd]" picks the layer (default: the 36x36x8 -> 8 forward
layer); layers deeper than one wave's weight registers run as several launches (tap groups): the stamps are those of the LAST one."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avsr_tf1_amd import ops
import tools.conv_bench as cb
spec = os.environ.get("AVSR_DISSECT_LAYER", "36,8,8,3,1").split(",")
H, Ci, Co, k, s = [int(v) for v in spec[:5]]
bwd = len(spec) > 5 and spec[5] == "bwd"
N = 4800
Ho = (H + s - 1) // s
pt = max((Ho - 1) * s + k - H, 0) // 2
x = torch.randn(N, H, H, Ci, device="cuda"); w = torch.randn(k, k, Ci, Co, device="cuda") * 0.1; b = torch.randn(Co, device="cuda")
y = torch.zeros(N, Ho, Ho, Co, device="cuda"); stats = torch.zeros(512 * 2 * max(Ci, Co) + 512 * 4 * 8, device="cuda")
bnv = (torch.rand(Ci, device="cuda") + 0.5, torch.randn(Ci, device="cuda"))
d = ops.conv_desc(N, H, H, Ci, Co, k, s, pt, pt, Ho, Ho, bn=None if bwd else bnv)
if bwd:
    dy = torch.randn(N, Ho, Ho, Co, device="cuda"); dx = torch.zeros(N, H, H, Ci, device="cuda")
    run = lambda: ops.conv_bwd_data(d, dy, w, dx)
else:
    run = lambda: ops.conv_fwd(d, x, w, b, y, None, None, stats)
    grid = run()
print(spec, os.environ.get("AVSR_CONV_DBG"), "%.1f us" % cb.timeit(run))
if int(os.environ.get("AVSR_CONV_DBG", "0")) & 8 and not bwd:
    t = stats[grid * 2 * Co:][:grid * 32].view(grid, 4, 8).cpu()
    print("workgroups:", grid)
    t = t[t[:, :, 5] > 0]
    names = ["prologue", "barrier1", "commit", "barrier2", "compute", "total", "setup"]
    if int(os.environ["AVSR_CONV_DBG"]) & 16:
        names = ["lds zero", "tables", "weights", "constants", "compute", "total", "setup"]
    print("waves with stamps:", t.shape[0])
    print("mean cycles per wave:", {n: int(t[:, k].mean()) for k, n in enumerate(names)})
    print("max  cycles per wave:", {n: int(t[:, k].max()) for k, n in enumerate(names)})
    print("min  cycles per wave:", {n: int(t[:, k].min()) for k, n in enumerate(names)})
