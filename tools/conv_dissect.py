"""Dissection probe of conv_gen_kernel on the 36x36x8 -> 8 forward layer (profiles/r03_conv_dissection_v1.txt).  Needs a library built
with AVSR_HIPCC_FLAGS=-DCONV_DEBUG; AVSR_CONV_DBG = bit0 no stores | bit1 no LDS operand reads | bit2 no MFMAs | bit3 per-wave cycle
stamps (prologue / barrier / commit / barrier / tile phase), printed as mean / max / min over the waves."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avsr_tf1_amd import ops
import tools.conv_bench as cb
N, H, Ci, Co, k, s = 4800, 36, 8, 8, 3, 1
x = torch.randn(N, H, H, Ci, device="cuda"); w = torch.randn(k, k, Ci, Co, device="cuda") * 0.1; b = torch.randn(Co, device="cuda")
y = torch.zeros(N, H, H, Co, device="cuda"); stats = torch.zeros(512 * 2 * Co + 512 * 4 * 8, device="cuda")
bnv = (torch.rand(Ci, device="cuda") + 0.5, torch.randn(Ci, device="cuda"))
d = ops.conv_desc(N, H, H, Ci, Co, k, s, 1, 1, H, H, bn=bnv)
print(os.environ.get("AVSR_CONV_PC"), os.environ.get("AVSR_CONV_DBG"), "%.1f us" % cb.timeit(lambda: ops.conv_fwd(d, x, w, b, y, None, None, stats)))
if int(os.environ.get("AVSR_CONV_DBG", "0")) & 8:
    t = stats[512 * 2 * Co:].view(512, 4, 8).cpu()
    names = ["prologue", "barrier1", "commit", "barrier2", "compute", "total"]
    print("mean cycles per wave:", {n: int(t[:, :, k].mean()) for k, n in enumerate(names)})
    print("max  cycles per wave:", {n: int(t[:, :, k].max()) for k, n in enumerate(names)})
    print("min  cycles per wave:", {n: int(t[:, :, k].min()) for k, n in enumerate(names)})
