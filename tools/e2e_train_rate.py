"""End-to-end rate of AVSR.train on a synthetic TFRecord dataset (audio-only 3x256 uni-LSTM LAS, 300-500 frame utterances): TFRecord
parsing, bucketing, host-to-device copies, per-shape workspaces, eager launches, logging.  python tools/e2e_train_rate.py [n_utt]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avsr_tf1_amd as avsr  # noqa: E402
from avsr_tf1_amd import io_utils as IO  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d = tempfile.mkdtemp()
os.chdir(d)
unit_file = os.path.join(d, "character_list")
open(unit_file, "w").write("\n".join(list("' abcdefghijklmnopqrstuvwxyz")) + "\n")
rng = np.random.default_rng(0)
a, l = os.path.join(d, "a.tfrecord"), os.path.join(d, "l.tfrecord")
with IO.TFRecordFileWriter(a) as fa, IO.TFRecordFileWriter(l) as fl:
    for i in range(n):
        T = int(rng.integers(300, 501))
        fa.write(IO.make_feature_example("u%d" % i, rng.standard_normal((T, 80)).astype(np.float32)))
        fl.write(IO.make_label_example("u%d" % i, rng.integers(1, 28, int(rng.integers(20, 41))).tolist(), "character"))
exp = avsr.AVSR(unit="character", unit_file=unit_file, audio_processing="features", audio_train_record=a, audio_test_record=a,
                labels_train_record=l, labels_test_record=l, batch_size=(64, 64), encoder_units_per_layer=((256,), (256, 256, 256)),
                decoder_units_per_layer=(256,), embedding_size=128, decoding_algorithm="greedy")
import io, contextlib  # noqa: E402
for ep in range(3):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        exp.train(logfile="logs/e2e", num_epochs=2)
    dt = time.perf_counter() - t0
    print("epoch %d: %d utterances in %.2f s = %.0f utt/s" % (ep, n, dt, n / dt), flush=True)
