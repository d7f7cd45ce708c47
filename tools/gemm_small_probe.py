"""avsr_gemm at the shapes of a beam-search decode step (640 hypothesis rows): is the LDS-tiled GEMM faster than the step kernel there?
(No: 640x1024x640 runs at 27 TF = 30 us against 46 us for the step kernel including its gather and gates -- DESIGN.md section 3.)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avsr_tf1_amd import ops
dev = "cuda"
ws = torch.empty(48 << 20, device=dev); ops.set_gemm_workspace(ws)
def t(M, N, K, tb=1):
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev); Cm = torch.zeros(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    f = lambda: ops.gemm(ops.mat(A, K), ops.mat(B, K if tb else N), ops.mat(Cm, N), M, N, K, trans_b=bool(tb), bias=bias)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    print("M=%d N=%d K=%d tb=%d  %.1f us  %.1f TF" % (M, N, K, tb, us, 2.0 * M * N * K / us * 1e-6))
for tb in (1, 0):
    t(640, 1024, 640, tb); t(640, 1024, 896, tb); t(640, 256, 512, tb); t(640, 512, 768, tb); t(640, 32, 512, tb)
