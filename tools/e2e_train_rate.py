"""End-to-end rate of AVSR.train on a synthetic TFRecord dataset (audio-only 3x256 uni-LSTM LAS, 300-500 frame utterances; "c4": the benchmark workload from 36x36x3 lip crops + audio): TFRecord
parsing, bucketing, host-to-device copies, per-shape workspaces, eager launches, logging.  python tools/e2e_train_rate.py [n_utt] [c4 [ragged]]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avsr_tf1_amd as avsr  # noqa: E402
from avsr_tf1_amd import io_utils as IO  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ragged = "ragged" in sys.argv    # c4: utterance lengths 450-500 / 20-40 labels instead of the benchmark's fixed 500 / 75 / 40
c4 = "c4" in sys.argv            # the benchmark workload from lip crops: 36x36x3 frames + Action Units + audio, AV dual-attention, B = 64
d = tempfile.mkdtemp()
os.chdir(d)
unit_file = os.path.join(d, "character_list")
open(unit_file, "w").write("\n".join(list("' abcdefghijklmnopqrstuvwxyz")) + "\n")
rng = np.random.default_rng(0)
a, l, v = os.path.join(d, "a.tfrecord"), os.path.join(d, "l.tfrecord"), os.path.join(d, "v.tfrecord")
with IO.TFRecordFileWriter(a) as fa, IO.TFRecordFileWriter(l) as fl, IO.TFRecordFileWriter(v) as fv:
    for i in range(n):
        T = (int(rng.integers(450, 501)) if ragged else 500) if c4 else int(rng.integers(300, 501))
        fa.write(IO.make_feature_example("u%d" % i, rng.standard_normal((T, 80)).astype(np.float32)))
        fl.write(IO.make_label_example("u%d" % i, rng.integers(1, 28, 39 if (c4 and not ragged) else int(rng.integers(20, 41))).tolist(), "character"))
        if c4:
            Tv = T * 75 // 500
            fv.write(IO.make_video_example("u%d" % i, rng.uniform(-1, 1, (Tv, 36, 36, 3)).astype(np.float32), aus=rng.uniform(0, 3, (Tv, 2))))
if c4:
    exp = avsr.AVSR(unit="character", unit_file=unit_file, video_processing="resnet_cnn", video_train_record=v, video_test_record=v,
                    audio_processing="features", audio_train_record=a, audio_test_record=a, labels_train_record=l, labels_test_record=l,
                    batch_size=(64, 64), architecture="bimodal", regress_aus=True, encoder_units_per_layer=((256,), (256, 256, 256)),
                    decoder_units_per_layer=(256,), embedding_size=128, decoding_algorithm="greedy")
else:
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
