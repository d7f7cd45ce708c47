"""Time avsr_gemm on the shapes of the c4 train step (HIP events, 20 reps after 3 warm-ups).
python tools/gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsr_tf1_amd import ops  # noqa: E402


def bench(M, N, K, ta, tb, sk, reps=20):
    A = torch.randn((K, M) if ta else (M, K), device="cuda")
    B = torch.randn((N, K) if tb else (K, N), device="cuda")
    Cm = torch.zeros(M, N, device="cuda")
    ws = torch.empty(max(4, sk * M * N), device="cuda")
    f = lambda: ops.gemm(ops.mat(A, A.shape[1]), ops.mat(B, B.shape[1]), ops.mat(Cm, N), M, N, K, trans_a=ta, trans_b=tb,
                         splitk=sk, workspace=ws)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * M * N * K / us / 1e6


if __name__ == "__main__":
    shapes = [("enc dW h-part  ", 256, 1024, 32000, 1, 0), ("enc dW x(80)   ", 80, 1024, 32000, 1, 0),
              ("hoisted x.Wx   ", 32000, 1024, 80, 0, 0), ("dx = dG.W0^T   ", 32000, 80, 1024, 0, 1),
              ("keys = mem.Wk  ", 32000, 256, 256, 0, 0), ("video dx       ", 4800, 128, 1024, 0, 1),
              ("small 64x256   ", 64, 256, 512, 0, 0), ("dec 2560x128   ", 2560, 31, 256, 0, 0),
              ("dec dW 896x1024", 896, 1024, 2560, 1, 0)]
    for name, M, N, K, ta, tb in shapes:
        line = name
        for sk in (1, 4, 8, 16, 32, 48, 64, 96):
            if sk > 1 and K // sk < 64:
                continue
            us, tf = bench(M, N, K, ta, tb, sk)
            line += " | sk%-2d %7.1fus %5.1fTF" % (sk, us, tf)
        print(line)
