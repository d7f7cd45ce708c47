"""Average time per launch class over N eager steps of a workload (HIP events around every engine launch, avsr_prof_*): the A/B tool for
kernel experiments.   python tools/kind_times.py [workload=c4] [steps=10] [reps=3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                         # noqa: E402
import bench                                                         # noqa: E402
from avsr_tf1_amd import ops                                         # noqa: E402
from avsr_tf1_amd.config import ModelConfig                          # noqa: E402
from avsr_tf1_amd.model import Batch, Seq2SeqModel                   # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
wl = bench.WORKLOADS[w]
cfg = ModelConfig(audio_feat=bench.FA, video_feat=bench.FV, video_processing="resnet_cnn", use_dropout=True, sampling_probability=0.1, **wl["cfg"])
m = Seq2SeqModel(cfg, seed=2001)
b = Batch.from_numpy(bench.NS(bench.synth(cfg, wl["B"], 0)))
for _ in range(3):
    m.train_step(b)
torch.cuda.synchronize()
for r in range(reps):
    ops.prof_begin(1 << 16)
    torch.cuda._sleep(int(0.3 * 2.0e9))
    for _ in range(steps):
        m.train_step(b)
    prof = ops.prof_end()
    print("rep %d: " % r + "  ".join("%s %.1f" % (k, 1e3 * ms / steps) for k, (cnt, ms, fl) in sorted(prof.items()) if cnt) + "   (us per step)", flush=True)
print("persistent wait expired:", bool(ops.rnn_persistent_error()), " loss %.5f" % float(m.loss.item()))
