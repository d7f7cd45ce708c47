"""Time avsr_gemm on free-standing shapes: python tools/gemm_shapes.py "M,N,K,ta,tb[,splitk]" ...   (fp32 MFMA peak 157.3 TF)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avsr_tf1_amd import ops                                        # noqa: E402


def main():
    ws = torch.empty(64 << 20, device="cuda")
    ops.set_gemm_workspace(ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for spec in sys.argv[1:]:
        v = [int(x) for x in spec.split(",")]
        M, N, K, ta, tb = v[:5]
        sk = v[5] if len(v) > 5 else None
        A = torch.randn((K, M) if ta else (M, K), device="cuda")
        B = torch.randn((N, K) if tb else (K, N), device="cuda")
        Cm = torch.zeros(M, N, device="cuda")
        a, b, c = ops.mat(A, A.shape[1]), ops.mat(B, B.shape[1]), ops.mat(Cm, N)
        for rep in range(2):
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                ops.gemm(a, b, c, M, N, K, trans_a=ta, trans_b=tb, splitk=sk, workspace=ws)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        print("M=%-6d N=%-6d K=%-6d ta=%d tb=%d splitk=%-5s %8.1f us %7.1f TF" % (M, N, K, ta, tb, sk, us, 2.0 * M * N * K / us * 1e-6))
        if sk is None and ta == 0 and tb == 0 and M * N * K <= 1 << 31:
            ref = A.double() @ B.double()
            print("      max |err| vs fp64: %.3g" % float((Cm.double() - ref).abs().max()))


if __name__ == "__main__":
    main()
