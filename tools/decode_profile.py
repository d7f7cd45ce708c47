"""Greedy / beam-search decode of the benchmark batch (c4), for rocprofv3 --kernel-trace: python tools/decode_profile.py [beam|greedy] [reps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                       # noqa: E402
from avsr_tf1_amd.config import ModelConfig                          # noqa: E402
from avsr_tf1_amd.model import Batch, Seq2SeqModel                   # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "beam"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    wl = bench.WORKLOADS["c4"]
    cfg = ModelConfig(audio_feat=bench.FA, video_feat=bench.FV, video_processing="resnet_cnn", use_dropout=True, sampling_probability=0.1, **wl["cfg"])
    batch = Batch.from_numpy(bench.NS(bench.synth(cfg, wl["B"], 0)))
    m = Seq2SeqModel(cfg, seed=2001)
    fn = (lambda: m.beam_search_decode(batch, beam_width=10, max_steps=bench.LDEC)) if what == "beam" else (lambda: m.greedy_decode(batch, max_steps=bench.LDEC))
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%s: %.3f ms per batch of %d = %.0f utt/s" % (what, 1e3 * dt, wl["B"], wl["B"] / dt))


if __name__ == "__main__":
    main()
