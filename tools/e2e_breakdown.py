"""Where does AVSR.train's time per step go on the c4 workload from lip-crop TFRecords?  Replays the loop of avsr.py:train with a host timer
around each phase (no extra device syncs): wait for the uploaded batch | start the next upload | trainer.train_step (host side) |
loss.item() (= waiting for the GPU) | print.   python tools/e2e_breakdown.py [n_utt=1024]"""
import contextlib
import io
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                           # noqa: E402
import avsr_tf1_amd as avsr            # noqa: E402
from avsr_tf1_amd import io_utils as IO  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
d = tempfile.mkdtemp()
os.chdir(d)
unit_file = os.path.join(d, "character_list")
open(unit_file, "w").write("\n".join(list("' abcdefghijklmnopqrstuvwxyz")) + "\n")
rng = np.random.default_rng(0)
a, l, v = (os.path.join(d, x) for x in ("a.tfrecord", "l.tfrecord", "v.tfrecord"))
with IO.TFRecordFileWriter(a) as fa, IO.TFRecordFileWriter(l) as fl, IO.TFRecordFileWriter(v) as fv:
    for i in range(n):
        fa.write(IO.make_feature_example("u%d" % i, rng.standard_normal((500, 80)).astype(np.float32)))
        fl.write(IO.make_label_example("u%d" % i, rng.integers(1, 28, 39).tolist(), "character"))
        fv.write(IO.make_video_example("u%d" % i, rng.uniform(-1, 1, (75, 36, 36, 3)).astype(np.float32), aus=rng.uniform(0, 3, (75, 2))))
exp = avsr.AVSR(unit="character", unit_file=unit_file, video_processing="resnet_cnn", video_train_record=v, video_test_record=v,
                audio_processing="features", audio_train_record=a, audio_test_record=a, labels_train_record=l, labels_test_record=l,
                batch_size=(64, 64), architecture="bimodal", regress_aus=True, encoder_units_per_layer=((256,), (256, 256, 256)),
                decoder_units_per_layer=(256,), embedding_size=128, decoding_algorithm="greedy")
with contextlib.redirect_stdout(io.StringIO()):
    exp.train(logfile="logs/e2e", num_epochs=3)         # two warm epochs: workspaces, graphs
for rep in range(int(os.environ.get("REPS", "2"))):
    threaded = False        # (round 6 tried the uploads on the prefetch thread: no gain, see profiles/r06_experiments.txt call D)
    it = exp._iterator('train')
    it.reuse_buffers = True
    copy_stream = torch.cuda.Stream()
    T = dict(wait=0.0, upload=0.0, step=0.0, item=0.0, next=0.0)
    def upload(bd):
        if bd is None:
            return None
        with torch.cuda.stream(copy_stream):
            batch, _ = exp._to_batch(bd)
            ev = torch.cuda.Event(); ev.record(copy_stream)
        return batch, ev
    it = iter(exp._prefetched(it, depth=2))
    fetch = (lambda: next(it, None)) if threaded else (lambda: upload(next(it, None)))
    nxt = fetch()
    steps = 0
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    while nxt is not None:
        batch, ev = nxt
        t0 = time.perf_counter()
        cur = torch.cuda.current_stream(); cur.wait_event(ev)
        for t_ in vars(batch).values():
            if torch.is_tensor(t_):
                t_.record_stream(cur)
        t1 = time.perf_counter()
        t2 = t1
        nxt = fetch()
        t3 = time.perf_counter()
        loss, gnorm = exp._trainer.train_step(batch)
        t4 = time.perf_counter()
        _ = float(loss.item()), float(gnorm.item())
        t5 = time.perf_counter()
        T["wait"] += t1 - t0; T["next"] += t2 - t1; T["upload"] += t3 - t2; T["step"] += t4 - t3; T["item"] += t5 - t4
        steps += 1
    dt = time.perf_counter() - t_all
    print("rep %d: %d steps, %.2f ms per step (%.0f utt/s); host ms per step: " % (rep, steps, 1e3 * dt / steps, 64 * steps / dt) +
          "  ".join("%s %.2f" % (k, 1e3 * x / steps) for k, x in T.items()) + "   launch mode: %s" % exp._trainer.mode, flush=True)
