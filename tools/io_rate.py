"""Host-side rate of the TFRecord input pipeline (read + index/parse + shuffle + bucket + pad): the native indexer / batch filler
(libavsr_io.so) against the python parser.  python tools/io_rate.py [n_utt] [video] [reuse]   (video: 75 x 36x36x3 lip crops + audio)"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsr_tf1_amd import io_utils as IO  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    video = len(sys.argv) > 2
    d = tempfile.mkdtemp()
    unit_file = os.path.join(d, "character_list")
    open(unit_file, "w").write("\n".join(list("' abcdefghijklmnopqrstuvwxyz")) + "\n")
    ud = IO.create_unit_dict(unit_file)
    rng = np.random.default_rng(0)
    a, l, v = os.path.join(d, "a.tfrecord"), os.path.join(d, "l.tfrecord"), os.path.join(d, "v.tfrecord")
    with IO.TFRecordFileWriter(a) as fa, IO.TFRecordFileWriter(l) as fl, IO.TFRecordFileWriter(v) as fv:
        for i in range(n):
            T = int(rng.integers(300, 501))
            fa.write(IO.make_feature_example("u%d" % i, rng.standard_normal((T, 80)).astype(np.float32)))
            fl.write(IO.make_label_example("u%d" % i, rng.integers(1, 28, int(rng.integers(20, 41))).tolist(), "character"))
            if video:
                Tv = T * 75 // 500
                fv.write(IO.make_video_example("u%d" % i, rng.standard_normal((Tv, 36, 36, 3)).astype(np.float32)))
    reuse = "reuse" in sys.argv                      # batch buffers from the pipeline's ring (what AVSR.train asks for)
    for native in (False, True):
        if video:
            pipe = IO.make_iterator_from_two_records(v, a, l, batch_size=64, unit_dict=ud, shuffle=True, bucket_width=45, seed=0)
        else:
            pipe = IO.make_iterator_from_one_record(a, l, ud, batch_size=64, shuffle=True, bucket_width=45, seed=0)
        if not native:
            pipe.native = None
        pipe.reuse_buffers = reuse and native
        for rep in range(2):
            t0 = time.perf_counter()
            cnt = sum(b.labels.shape[0] for b in pipe)
            dt = time.perf_counter() - t0
        print("%s parser%s: %d utterances in %.2f s = %.0f utt/s" % ("native" if native else "python", " + buffer ring" if pipe.reuse_buffers else "", cnt, dt, cnt / dt), flush=True)
