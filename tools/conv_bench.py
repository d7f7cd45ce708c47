"""Per-layer timing of the lip-CNN convolutions at the benchmark size (N = 64*75 = 4800 frames): forward, data gradient, weight
gradient through the C ABI (the frame-resident MFMA kernels of csrc/conv_mfma.hip unless AVSR_CONV_MFMA=0)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avsr_tf1_amd import ops, _lib   # noqa: E402

LAYERS = [("layer0 3->8 36x36", 36, 3, 8, 1), ("res0 8->8 36x36", 36, 8, 8, 1), ("b1c1 8->16 s2", 36, 8, 16, 2), ("b1c2 16->16 18x18", 18, 16, 16, 1),
          ("b2c1 16->32 s2", 18, 16, 32, 2), ("b2c2 32->32 9x9", 9, 32, 32, 1)]


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
    if os.environ.get("AVSR_CONV_MFMA") == "0":
        _lib.load().avsr_conv_set_mfma(0)
    scratch = torch.empty(1 << 24, device="cuda")
    tot = [0.0, 0.0, 0.0]
    for name, H, Ci, Co, s in LAYERS:
        Ho, pt = same(H, 3, s)
        x = torch.randn(N, H, H, Ci, device="cuda")
        w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.1
        b = torch.randn(Co, device="cuda")
        y = torch.zeros(N, Ho, Ho, Co, device="cuda")
        dy = torch.randn(N, Ho, Ho, Co, device="cuda")
        dx = torch.zeros(N, H, H, Ci, device="cuda")
        dw = torch.zeros(3, 3, Ci, Co, device="cuda")
        fl = 2.0 * N * Ho * Ho * 9 * Ci * Co
        t_f = timeit(lambda: ops.conv3x3(x, w, b, y, N, H, H, Ci, Co, s, pt, pt, Ho, Ho))
        t_w = timeit(lambda: ops.conv3x3_bwd_weight(x, dy, dw, N, H, H, Ci, Co, s, pt, pt, Ho, Ho, scratch))
        if Ci % 4 == 0:
            if s == 1:
                t_d = timeit(lambda: ops.conv3x3(dy, w, None, dx, N, Ho, Ho, Co, Ci, 1, 1, 1, H, H, flip=1))
            else:
                t_d = timeit(lambda: ops.conv3x3_bwd_data_s2(dy, w, dx, N, H, H, Ci, Co, pt, pt, Ho, Ho))
        else:
            t_d = 0.0
        tot[0] += t_f; tot[1] += t_d; tot[2] += t_w
        print(f"{name:22s} fwd {t_f:7.1f} us ({fl / t_f / 1e6:6.1f} TF)  bwd-data {t_d:7.1f} us  bwd-weight {t_w:7.1f} us ({fl / t_w / 1e6:6.1f} TF)")
    print(f"totals: fwd {tot[0]:.0f} us, bwd-data {tot[1]:.0f} us, bwd-weight {tot[2]:.0f} us")


if __name__ == "__main__":
    main()
