"""Every GEMM launch of one eager c4 train step with its shapes and time (AVSR_GEMM_LOG=1 brackets each launch with events and waits:
the stream is serialised, so the times are stand-alone launch times).  python tools/gemm_step_log.py [workload]"""
import os
import sys

os.environ["AVSR_GEMM_LOG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                         # noqa: E402
import bench                                                         # noqa: E402
from avsr_tf1_amd.config import ModelConfig                          # noqa: E402
from avsr_tf1_amd.model import Batch, Seq2SeqModel                   # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "c4"
wl = bench.WORKLOADS[w]
cfg = ModelConfig(audio_feat=bench.FA, video_feat=bench.FV, video_processing="resnet_cnn", use_dropout=True, sampling_probability=0.1, **wl["cfg"])
m = Seq2SeqModel(cfg, seed=2001)
b = Batch.from_numpy(bench.NS(bench.synth(cfg, wl["B"], 0)))
os.environ["AVSR_GEMM_LOG"] = "1"
for i in range(3):
    sys.stderr.write("==== step %d ====\n" % i)
    m.train_step(b)
    torch.cuda.synchronize()
