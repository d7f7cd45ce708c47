"""Cycle stamps of the pixel-pair weight-gradient kernel on the 36x36x8 layer (library built with AVSR_HIPCC_FLAGS=-DCONV_DEBUG,\nAVSR_CONV_DBG=8): profiles/r03_conv_dissection_v1.txt (e)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avsr_tf1_amd import ops
import tools.conv_bench as cb
N, H, Ci, Co, k, s = 4800, 36, 8, 8, 3, 1
x = torch.randn(N, H, H, Ci, device="cuda"); dy = torch.randn(N, H, H, Co, device="cuda")
dw, db = torch.zeros(k, k, Ci, Co, device="cuda"), torch.zeros(Co, device="cuda")
scratch = torch.zeros(1 << 24, device="cuda")
d = ops.conv_desc(N, H, H, Ci, Co, k, s, 1, 1, H, H)
print("%.1f us" % cb.timeit(lambda: ops.conv_bwd_weight(d, x, dy, dw, db, scratch)))
torch.cuda.synchronize()
slab = 12 * Ci * 16 + 16
for grid in (512, 256):
    t = scratch[grid * slab: grid * slab + grid * 4 * 8].view(grid, 4, 8).cpu()
    if t[:, :, 4].min() > 0:
        names = ["barrier1", "commit", "barrier2", "compute", "total"]
        print("grid", grid, "mean cycles per wave:", {n: int(t[:, :, i].mean()) for i, n in enumerate(names)})
        break
