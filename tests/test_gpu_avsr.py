"""End-to-end smoke of the reference's Python surface on the engine: AVSR(...).train / evaluate / run_experiment on a tiny
synthetic TFRecord dataset (written with our own writer), checkpoints, log + prediction files."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dataset(tmp, n=12, feat=8):
    from avsr_tf1_amd import io_utils as IO
    rng = np.random.default_rng(0)
    unit_file = os.path.join(tmp, "character_list")
    open(unit_file, "w").write("\n".join(list("' abcdefghijklmnopqrstuvwxyz")) + "\n")
    paths = {k: os.path.join(tmp, k + ".tfrecord") for k in ("audio", "video", "labels")}
    with IO.TFRecordFileWriter(paths["audio"]) as fa, IO.TFRecordFileWriter(paths["video"]) as fv, IO.TFRecordFileWriter(paths["labels"]) as fl:
        for i in range(n):
            L = int(rng.integers(2, 6))
            lab = rng.integers(3, 10, size=L)
            T = 6 * L + int(rng.integers(0, 4))
            x = rng.standard_normal((T, feat)).astype(np.float32) * 0.1
            for j, c in enumerate(lab):                          # make the task learnable: label-dependent bumps
                x[6 * j:6 * j + 6, int(c) % feat] += 2.0
            fa.write(IO.make_feature_example("utt%02d" % i, x))
            fv.write(IO.make_feature_example("utt%02d" % i, x[::3, :4].copy()))
            fl.write(IO.make_label_example("utt%02d" % i, lab.tolist(), "character"))
    return unit_file, paths


def test_avsr_train_evaluate_and_resume(tmp_path, monkeypatch):
    import avsr_tf1_amd as avsr
    monkeypatch.chdir(tmp_path)
    unit_file, p = _dataset(str(tmp_path))
    kw = dict(unit="character", unit_file=unit_file, audio_processing="features", audio_train_record=p["audio"],
              audio_test_record=p["audio"], labels_train_record=p["labels"], labels_test_record=p["labels"], batch_size=(4, 4),
              encoder_units_per_layer=((32,), (32, 32)), decoder_units_per_layer=(32,), embedding_size=16, decoding_algorithm="greedy",
              warmup_steps=0, learning_rate=0.01, shuffle_seed=0)      # fixed shuffle order: the assertions below see one trajectory
    exp = avsr.AVSR(**kw)
    exp.train(logfile="logs/smoke", num_epochs=11)               # 10 epochs -> checkpoint + evaluation at epoch 10
    assert os.path.exists("checkpoints/smoke/checkpoint.ckp-10.npz")
    assert os.path.exists("predictions/smoke/predicted_epoch_10.mlf")
    log = open("logs/smoke").read()
    assert log.count("Average batch_loss") == 10 and "character:" in log and "word:" in log
    losses = [float(l.split()[-1]) for l in log.splitlines() if l.startswith("Average")]
    assert losses[-1] < losses[0]
    err = exp.evaluate("checkpoints/smoke/checkpoint.ckp-10", epoch=10)
    # (an under-trained model may never emit EOS: 150 decoded symbols against 2-5 reference ones, so only sanity-bound the rate)
    assert set(err) == {"character", "word"} and np.isfinite(err["character"]) and err["character"] >= 0.0
    # a fresh object resumes from the newest checkpoint and continues the epoch count from its file name
    exp2 = avsr.AVSR(**kw)
    exp2.train(logfile="logs/smoke", num_epochs=2, try_restore_latest_checkpoint=True)
    assert "Average batch_loss as epoch 11" in open("logs/smoke").read()
    assert int(exp2._model.step.item()) > int(30)                # optimiser step restored and advanced


def test_run_experiment_bimodal(tmp_path, monkeypatch):
    import avsr_tf1_amd as avsr
    monkeypatch.chdir(tmp_path)
    unit_file, p = _dataset(str(tmp_path))
    os.makedirs("logs", exist_ok=True)
    avsr.run_experiment(video_train_record=p["video"], video_test_record=p["video"], labels_train_record=p["labels"],
                        labels_test_record=p["labels"], audio_train_records=(p["audio"],), audio_test_records=(p["audio"],),
                        unit="character", unit_list_file=unit_file, iterations=((2, 1),), learning_rates=((0.01, 0.001),),
                        logfile="exp_av", architecture="bimodal", video_processing="features", audio_processing="features",
                        batch_size=(4, 4), encoder_units_per_layer=((32,), (32, 32)), decoder_units_per_layer=(32,), embedding_size=16,
                        decoding_algorithm="beam_search", beam_width=4)
    log = open("logs/exp_av").read()
    assert log.count("Average batch_loss") == 3 and "=====" in log


def test_avsr_visual_only_from_lip_crops(tmp_path, monkeypatch):
    """run_video.py-style experiment from raw frames: video_processing='resnet_cnn' (avsr/video.py:143-195)."""
    import avsr_tf1_amd as avsr
    from avsr_tf1_amd import io_utils as IO
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(1)
    unit_file = os.path.join(str(tmp_path), "character_list")
    open(unit_file, "w").write("\n".join(list("' abcdefghijklmnopqrstuvwxyz")) + "\n")
    vrec, lrec = os.path.join(str(tmp_path), "video.tfrecord"), os.path.join(str(tmp_path), "labels.tfrecord")
    with IO.TFRecordFileWriter(vrec) as fv, IO.TFRecordFileWriter(lrec) as fl:
        for i in range(8):
            L = int(rng.integers(2, 5))
            lab = rng.integers(3, 10, size=L)
            T = 3 * L + int(rng.integers(0, 3))
            frames = rng.standard_normal((T, 36, 36, 3)).astype(np.float32) * 0.1
            for j, c in enumerate(lab):
                frames[3 * j:3 * j + 3, int(c) * 3:int(c) * 3 + 3, :, :] += 1.0        # label-dependent stripe
            fv.write(IO.make_video_example("utt%02d" % i, frames))
            fl.write(IO.make_label_example("utt%02d" % i, lab.tolist(), "character"))
    exp = avsr.AVSR(unit="character", unit_file=unit_file, video_processing="resnet_cnn", video_train_record=vrec, video_test_record=vrec,
                    labels_train_record=lrec, labels_test_record=lrec, batch_size=(4, 4), encoder_units_per_layer=((32,), (32,)),
                    decoder_units_per_layer=(32,), embedding_size=16, decoding_algorithm="greedy", cnn_filters=(8, 8, 16, 16),
                    cnn_dense_units=32, warmup_steps=0, learning_rate=0.01, shuffle_seed=0)
    exp.train(logfile="logs/video", num_epochs=4)
    log = open("logs/video").read()
    losses = [float(l.split()[-1]) for l in log.splitlines() if l.startswith("Average")]
    assert len(losses) == 3 and np.isfinite(losses).all() and losses[-1] < losses[0]
    exp.save("checkpoints/video/checkpoint.ckp-3")
    w = np.load("checkpoints/video/checkpoint.ckp-3.npz")
    assert "params:video/cnn/flatten/kernel" in w.files and w["params:video/cnn/flatten/kernel"].shape == (5, 5, 16, 32)


def test_avsr_writes_attention_alignment_images(tmp_path, monkeypatch):
    """write_attention_alignment=True (avsr/avsr.py:354-366, :404-436): <file>.png + <file>_av.png for AV-Align,
    <file>_video.png + <file>_audio.png for the bimodal decoder."""
    import avsr_tf1_amd as avsr
    monkeypatch.chdir(tmp_path)
    unit_file, p = _dataset(str(tmp_path), n=5)
    common = dict(unit="character", unit_file=unit_file, video_processing="features", video_train_record=p["video"], video_test_record=p["video"],
                  audio_processing="features", audio_train_record=p["audio"], audio_test_record=p["audio"], labels_train_record=p["labels"],
                  labels_test_record=p["labels"], batch_size=(4, 4), encoder_units_per_layer=((32,), (32, 32)), decoder_units_per_layer=(32,),
                  embedding_size=16, decoding_algorithm="greedy", write_attention_alignment=True, warmup_steps=0)
    for arch, suffixes in (("av_align", (".png", "_av.png")), ("bimodal", ("_video.png", "_audio.png"))):
        exp = avsr.AVSR(architecture=arch, **common)
        exp.train(logfile="logs/al_" + arch, num_epochs=2)
        exp.save("checkpoints/al_%s/checkpoint.ckp-1" % arch)
        exp.evaluate("checkpoints/al_%s/checkpoint.ckp-1" % arch, epoch=1, alignments_outdir="alignments/" + arch)
        for i in range(5):
            for suf in suffixes:
                f = "alignments/%s/utt%02d%s" % (arch, i, suf)
                assert os.path.exists(f) and open(f, "rb").read(8) == b"\x89PNG\r\n\x1a\n", f
    with pytest.raises(NotImplementedError):
        avsr.AVSR(architecture="bimodal", **dict(common, decoding_algorithm="beam_search"))
