"""PCIe-inclusive rate of the c4 train step: every step receives a fresh HOST batch (numpy, as the TFRecord pipeline produces it) and
copies it to the device before the step (synchronous pageable copies, as AVSR._to_batch does).  python tools/pcie_inclusive.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from avsr_tf1_amd.config import ModelConfig  # noqa: E402
from avsr_tf1_amd.model import Batch, Seq2SeqModel  # noqa: E402
from avsr_tf1_amd.parallel import DataParallelTrainer  # noqa: E402

for front in ("features", "resnet_cnn"):
    wl = bench.WORKLOADS["c4"]
    cfg = ModelConfig(audio_feat=bench.FA, video_feat=bench.FV, video_processing=front, use_dropout=True, sampling_probability=0.1, **wl["cfg"])
    host = bench.NS(bench.synth(cfg, 64, 0))
    nbytes = sum(getattr(host, k).nbytes for k in ("audio", "audio_len", "video", "video_len", "aus", "labels", "labels_len") if getattr(host, k, None) is not None)
    m = Seq2SeqModel(cfg, seed=2001)
    t = DataParallelTrainer(m, None, use_graph=False)
    for _ in range(3):
        t.train_step(Batch.from_numpy(host))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        t.train_step(Batch.from_numpy(host))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    dev = Batch.from_numpy(host)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        t.train_step(dev)
    torch.cuda.synchronize()
    dr = (time.perf_counter() - t0) / 10
    print("%-10s host batch %.1f MB: %.2f ms/step PCIe-inclusive (%.0f utt/s) vs %.2f ms resident (%.0f utt/s)" %
          (front, nbytes / 1e6, 1e3 * dt, 64 / dt, 1e3 * dr, 64 / dr), flush=True)
    del t, m
    torch.cuda.empty_cache()
