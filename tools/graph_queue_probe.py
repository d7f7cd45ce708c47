"""Probe of the hipGraph queueing hazard: N train steps queued back to back through the captured graphs,
optionally with eager work between the launches.  python tools/graph_queue_probe.py [features|resnet_cnn] [graph|eager]
env: SYNC_EVERY=k host sync every k steps; INTERLEAVE=tiny|big_other|big_same_memcpy|fresh_batch; NO_STAGE_COPY=1"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
from avsr_tf1_amd import ops
from avsr_tf1_amd.config import ModelConfig
from avsr_tf1_amd.model import Batch, Seq2SeqModel
from avsr_tf1_amd.parallel import DataParallelTrainer
wl = bench.WORKLOADS["c4"]
stoch = dict(use_dropout=True, sampling_probability=0.1)
front = sys.argv[1] if len(sys.argv) > 1 else "resnet_cnn"
use_graph = (sys.argv[2] != "eager") if len(sys.argv) > 2 else True
cfg2 = ModelConfig(audio_feat=bench.FA, video_feat=bench.FV, video_processing=front, **wl["cfg"], **stoch)
m2 = Seq2SeqModel(cfg2, seed=2001)
t2 = DataParallelTrainer(m2, None, use_graph=use_graph)
b2 = Batch.from_numpy(bench.NS(bench.synth(cfg2, 64, 0)))
for i in range(3):
    t2.train_step(b2)
    torch.cuda.synchronize()
print("warm", float(m2.loss.item()), ops.rnn_persistent_error(), flush=True)
orig_video = b2.video
if os.environ.get("NO_STAGE_COPY") and t2._static:
    b2 = list(t2._static.values())[0]
for rep in range(6):
    t0 = time.perf_counter()
    se = int(os.environ.get("SYNC_EVERY", "0"))
    for i in range(10):
        mode_x = os.environ.get("INTERLEAVE", "")
        if mode_x == "tiny":
            m2.scratch[:64].zero_()
        elif mode_x == "big_other":
            if not hasattr(t2, "_dummy"):
                t2._dummy = torch.empty_like(b2.video)
            t2._copy_into(t2._dummy, list(t2._static.values())[0].video)
        elif mode_x == "big_same_memcpy":
            st_ = list(t2._static.values())[0]
            st_.video.copy_(orig_video)
        if mode_x == "fresh_batch":
            if not hasattr(t2, "_alt"):
                t2._alt = Batch(*[None if getattr(b2, n) is None else getattr(b2, n).clone() for n in ("audio", "audio_len", "video", "video_len", "aus", "labels", "labels_len")])
            t2.train_step(t2._alt)
        else:
            t2.train_step(b2)
        if se and (i + 1) % se == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("rep", rep, "%.1f ms/step" % (1e2 * (time.perf_counter() - t0)), float(m2.loss.item()), "err", ops.rnn_persistent_error(), t2.mode, flush=True)
