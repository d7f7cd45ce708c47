"""Per-layer timing of the lip-CNN convolutions at the benchmark size (N = 64*75 = 4800 frames) through the descriptor API of
csrc/conv_mfma.hip: forward (with the fused BN-ReLU loader / residual / statistics where the network uses them), data gradient,
weight gradient (+ bias gradient).  Prints the time next to the time the layer's compulsory HBM bytes take at 8 TB/s and the
time its padded MFMA work takes at 157.3 TF."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avsr_tf1_amd import ops   # noqa: E402

#          name                 H   Ci  Co  k  s  bn     res
LAYERS = [("layer0 3->8", 36, 3, 8, 3, 1, False, False), ("rb0 c1 8->8", 36, 8, 8, 3, 1, True, False), ("rb0 c2 8->8 +res", 36, 8, 8, 3, 1, True, True),
          ("rb1 sc 1x1/2 8->16", 36, 8, 16, 1, 2, False, False), ("rb1 c1 /2 8->16", 36, 8, 16, 3, 2, True, False), ("rb1 c2 16->16 +res", 18, 16, 16, 3, 1, True, True),
          ("rb2 sc 1x1/2 16->32", 18, 16, 32, 1, 2, False, False), ("rb2 c1 /2 16->32", 18, 16, 32, 3, 2, True, False), ("rb2 c2 32->32 +res", 9, 32, 32, 3, 1, True, True),
          ("rb3 sc 1x1/2 32->64", 9, 32, 64, 1, 2, False, False), ("rb3 c1 /2 32->64", 9, 32, 64, 3, 2, True, False), ("rb3 c2 64->64 +res", 5, 64, 64, 3, 1, True, True)]


def same(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    N = int(os.environ.get("N", 4800))
    scratch = torch.empty(1 << 24, device="cuda")
    tot = [0.0, 0.0, 0.0]
    r16 = lambda v: -(-v // 16) * 16
    for name, H, Ci, Co, k, s, bn, res in LAYERS:
        Ho, pt = same(H, k, s)
        x = torch.randn(N, H, H, Ci, device="cuda")
        w = torch.randn(k, k, Ci, Co, device="cuda") * 0.1
        b = torch.randn(Co, device="cuda")
        y = torch.zeros(N, Ho, Ho, Co, device="cuda")
        r = torch.randn(N, Ho, Ho, Co, device="cuda") if res else None
        dy = torch.randn(N, Ho, Ho, Co, device="cuda")
        dx = torch.zeros(N, H, H, Ci, device="cuda")
        dw, db = torch.zeros(k, k, Ci, Co, device="cuda"), torch.zeros(Co, device="cuda")
        stats = torch.zeros(1024 * 2 * Co, device="cuda")
        bnv = (torch.rand(Ci, device="cuda") + 0.5, torch.randn(Ci, device="cuda")) if bn else None
        d = ops.conv_desc(N, H, H, Ci, Co, k, s, pt if k == 3 else 0, pt if k == 3 else 0, Ho, Ho, bn=bnv)
        assert ops.conv_supported(d), name
        t_f = timeit(lambda: ops.conv_fwd(d, x, w, b, y, r, None, stats))
        t_w = timeit(lambda: ops.conv_bwd_weight(d, x, dy, dw, db, scratch))
        t_d = timeit(lambda: ops.conv_bwd_data(d, dy, w, dx, beta=1.0 if (k == 1 and s == 2) else 0.0)) if Ci % 4 == 0 else 0.0
        xb, yb = 4.0 * N * H * H * Ci, 4.0 * N * Ho * Ho * Co
        hbm_f = (xb + yb * (2 if res else 1)) / 8e6                 # us at 8 TB/s
        hbm_d = (xb * (2 if (k == 1 and s == 2) else 1) + yb) / 8e6
        hbm_w = (xb + yb) / 8e6
        mfma = 2.0 * N * Ho * Ho * r16(k * k * ((Ci + 3) // 4 * 4)) * r16(Co) / 157.3e6
        tot[0] += t_f; tot[1] += t_d; tot[2] += t_w
        print(f"{name:22s} fwd {t_f:6.1f} us (hbm {hbm_f:5.1f}, mfma {mfma:5.1f})  bwd-data {t_d:6.1f} us (hbm {hbm_d:5.1f})  bwd-weight {t_w:6.1f} us (hbm {hbm_w:5.1f})")
    print(f"totals: fwd {tot[0]:.0f} us, bwd-data {tot[1]:.0f} us, bwd-weight {tot[2]:.0f} us")


if __name__ == "__main__":
    main()
