"""GEMM inventory of one c4 training step (shapes recorded by wrapping ops.gemm) and the time of every distinct shape, on the GPU.
Usage: python tools/gemm_bench.py [--video-frontend features|resnet_cnn]"""
import os
import sys
import collections

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                       # noqa: E402
from avsr_tf1_amd import ops                                        # noqa: E402
from avsr_tf1_amd.model import Seq2SeqModel                          # noqa: E402


def main():
    fe = "resnet_cnn" if "resnet_cnn" in sys.argv else "features"
    from avsr_tf1_amd.config import ModelConfig
    from avsr_tf1_amd.model import Batch
    wl = bench.WORKLOADS["c5" if "c5" in sys.argv else "c4"]
    cfg = ModelConfig(audio_feat=bench.FA, video_feat=bench.FV, video_processing=fe, use_dropout=True, sampling_probability=0.1, **wl["cfg"])
    batch = Batch.from_numpy(bench.NS(bench.synth(cfg, wl["B"], 0)))
    m = Seq2SeqModel(cfg, seed=1)
    m.train_step(batch)
    torch.cuda.synchronize()
    calls = []
    real = ops.gemm

    launches = collections.OrderedDict()      # (grouped launch id or call index, operand class) -> [workgroups, GEMMs]

    def spy(A, B, Cm, M, N, K, trans_a=False, trans_b=False, alpha=1.0, beta=0.0, bias=None, batch=1, strides=(0, 0, 0), splitk=None,
            workspace=None, alpha_dev=None, **kw):
        calls.append((int(M), int(N), int(K), int(bool(trans_a)), int(bool(trans_b)), int(batch), float(beta), splitk,
                      (A, B, Cm, bias, strides, workspace, alpha_dev, alpha)))
        sk = splitk if splitk else (ops.auto_splitk(M, N, K, batch) if (workspace is not None or ops._gemm_ws is not None) else 1)
        wgs = ((M + 127) // 128) * ((N + 127) // 128) * batch * max(1, int(sk))
        gid = id(ops._gemm_group) if ops._gemm_group is not None else -len(calls)
        e = launches.setdefault((gid, len([k for k in launches if k[0] == gid and False]), int(bool(trans_a)), int(bool(trans_b))), [0, 0, []])
        e[0] += wgs; e[1] += 1; e[2].append("%dx%dx%d/%d" % (M, N, K, sk))
        return real(A, B, Cm, M, N, K, trans_a, trans_b, alpha, beta, bias, batch, strides, splitk, workspace, alpha_dev, **kw)

    ops.gemm = spy
    m.train_step(batch)
    torch.cuda.synchronize()
    ops.gemm = real
    if "launches" in sys.argv:                 # workgroups per launch (768 slots = 3 per CU): rounds of the chip a launch takes
        for (gid, _z, ta, tb), (wgs, n, names) in launches.items():
            print("ta=%d tb=%d  GEMMs %d  workgroups %5d = %.2f rounds   %s" % (ta, tb, n, wgs, wgs / 768.0, " ".join(names)))
        return
    groups = collections.OrderedDict()
    for c in calls:
        groups.setdefault(c[:6], []).append(c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot, totf = 0.0, 0.0
    rows = []
    for key, cs in groups.items():
        M, N, K, ta, tb, bt = key
        c = cs[0]
        A, B, Cm, bias, strides, workspace, alpha_dev, alpha = c[8]
        # beta = 1 keeps every destination's meaning irrelevant here; timing only
        for rep in range(2):
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                real(A, B, Cm, M, N, K, ta, tb, alpha, 1.0, bias, bt, strides, c[7], workspace, alpha_dev)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        fl = 2.0 * M * N * K * bt
        rows.append((us * len(cs), len(cs), M, N, K, ta, tb, bt, us, fl / us * 1e-6))
        tot += us * len(cs)
        totf += fl * len(cs)
    rows.sort(reverse=True)
    for r in rows:
        print("%8.1f us total  x%-3d  M=%-6d N=%-5d K=%-6d ta=%d tb=%d batch=%-3d  %7.1f us  %6.1f TF" % r)
    print("python-level GEMMs per step: %d, %.3f ms, %.1f GFLOP, %.1f TF average" % (len(calls), tot * 1e-3, totf * 1e-9, totf / tot * 1e-6))


if __name__ == "__main__":
    main()
