"""The REAL engine under the REAL DataParallelTrainer with two ranks: both processes share the box's single GPU and talk over
gloo (RCCL refuses two ranks on one device), so everything except the transport of the collectives is what runs on an 8-GPU
node: per-rank shards, all-reduced loss normaliser, sync batch-norm phases, summed gradients, captured graphs replayed around the
collectives.  After four steps every rank must hold the parameters of ONE engine trained on the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = dict(architecture="bimodal", encoder_type="unidirectional", video_units=(32,), audio_units=(32, 32), decoder_units=(32,),
            embedding_size=16, video_feat=12, audio_feat=20, regress_aus=True, use_dropout=False, warmup_steps=0)
# widths the kernels do not take natively: the engine pads to multiples of 4, the batches keep the reference's widths
CASE_ODD = dict(CASE, video_units=(26,), audio_units=(22, 26), decoder_units=(26,), embedding_size=0, video_feat=13, audio_feat=39)
STEPS = 4


def _setup(odd=False):
    import dataclasses
    from avsr_tf1_amd.config import ModelConfig
    from oracle import avsr_oracle as O
    ocfg = O.OracleConfig(**(CASE_ODD if odd else CASE))
    mcfg = ModelConfig(**{f.name: getattr(ocfg, f.name) for f in dataclasses.fields(ModelConfig) if hasattr(ocfg, f.name)})
    W = O.init_params(ocfg, seed=5)
    full = O.synthetic_batch(ocfg, B=6, T_a=17, T_v=7, L=6, ragged=True)
    return O, mcfg, W, full


def _shard(O, b, lo, hi):
    return O.Batch(**{k: (None if getattr(b, k) is None else np.ascontiguousarray(getattr(b, k)[lo:hi]))
                      for k in ("audio", "audio_len", "video", "video_len", "aus", "labels", "labels_len")})


def _worker(rank, world, port, out_dir, use_graph, overlap, odd, persistent=False, drain="1", steps=STEPS):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["AVSR_DP_DRAIN"] = drain
    # persistent=False: two processes on ONE GPU do not both claim the chip.  persistent=True: they do -- each rank's shard (<= 8
    # utterances) is one row group, i.e. BOTH ranks' persistent encoder / fused decoder kernels want the workgroup slots of XCD 0 at
    # the same time.  The dispatcher may run them side by side, one after the other, or interleave them so that a bounded wait
    # expires: then the flag is MAX-reduced, both ranks redo the pass through the per-step launches, and the result must not change.
    os.environ["AVSR_PERSISTENT_RNN"] = "3" if persistent else "0"
    os.environ["AVSR_DP_OVERLAP"] = "1" if overlap else "0"
    import torch.distributed as dist
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    from avsr_tf1_amd.parallel import DataParallelTrainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O, mcfg, W, full = _setup(odd)
    cut = [0, 2, 6]                                   # unequal shards
    model = Seq2SeqModel(mcfg, weights=W)
    trainer = DataParallelTrainer(model, dist, use_graph=use_graph)
    batch = Batch.from_numpy(_shard(O, full, cut[rank], cut[rank + 1]))
    for _ in range(steps):
        trainer.train_step(batch)
    torch.cuda.synchronize()
    assert (trainer._bucket is not None) == bool(overlap)
    assert trainer.drain_around_collectives == (drain != "0")
    from avsr_tf1_amd import ops
    assert not ops.rnn_persistent_error()            # every flagged pass was redone and the flag cleared (check_every_step)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), mode=np.array(trainer.mode), sync_bn=np.array(trainer.sync_bn),
             persistent=np.array(bool(model.persistent_rnn)), **model.export_tf_weights("params"))
    dist.destroy_process_group()


@pytest.mark.parametrize("use_graph,overlap", [(False, False), (True, False), (True, True)])
def test_two_ranks_with_persistent_kernels_contending_for_one_gpu(tmp_path, use_graph, overlap, monkeypatch):
    """Both ranks run the persistent encoder kernels and the fused decoder on the SAME GPU (see _worker).  Replicas must stay
    bit-identical and equal one engine on the whole batch, whether or not a pass was flagged and redone on the way."""
    import torch.multiprocessing as mp
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), use_graph, overlap, False, True), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert bool(r0["persistent"]) == bool(r1["persistent"])          # the ranks switched paths together (MAX-reduced flag) or not at all
    print("mode", str(r0["mode"]), "| persistent kernels still on:", bool(r0["persistent"]))
    monkeypatch.setenv("AVSR_PERSISTENT_RNN", "3")
    O, mcfg, W, full = _setup(False)
    model = Seq2SeqModel(mcfg, weights=W)
    batch = Batch.from_numpy(full)
    for _ in range(STEPS):
        model.train_step(batch)
    torch.cuda.synchronize()
    ref = model.export_tf_weights("params")
    for k, v in ref.items():
        assert np.array_equal(r0[k], r1[k]), k
        assert np.abs(r0[k] - v).max() < 5e-6 + 1e-4 * np.abs(v).max() * 0.01, (k, np.abs(r0[k] - v).max())


@pytest.mark.parametrize("use_graph,overlap,odd", [(False, False, False), (True, False, False), (False, True, False), (True, True, False),
                                                   (True, True, True)])
def test_two_ranks_equal_one_engine_on_the_whole_batch(tmp_path, use_graph, overlap, odd, monkeypatch):
    """overlap: the decoder's gradient block is all-reduced on a side stream between the two halves of the backward pass
    (AVSR_DP_OVERLAP=1; with graphs the pass is captured as two graphs)."""
    import torch.multiprocessing as mp
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), use_graph, overlap, odd), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # graphs are replayed around the collectives when asked for (exact against eager launches once the captured
    # graphs held no memset / memcpy nodes)
    assert bool(r0["sync_bn"]) and str(r0["mode"]).startswith("hipgraph" if use_graph else "eager")
    monkeypatch.setenv("AVSR_PERSISTENT_RNN", "0")      # same launch path as the two workers (restored after the test)
    O, mcfg, W, full = _setup(odd)
    model = Seq2SeqModel(mcfg, weights=W)
    batch = Batch.from_numpy(full)
    for _ in range(STEPS):
        model.train_step(batch)
    torch.cuda.synchronize()
    ref = model.export_tf_weights("params")
    for k, v in ref.items():
        assert np.array_equal(r0[k], r1[k]), k                       # replicas stay bit-identical
        assert np.abs(r0[k] - v).max() < 5e-6 + 1e-4 * np.abs(v).max() * 0.01, (k, np.abs(r0[k] - v).max())


@pytest.mark.parametrize("overlap", [False, True])
def test_two_ranks_without_the_stream_drains_around_the_collectives(tmp_path, overlap, monkeypatch):
    """AVSR_DP_DRAIN=0: graph replays and the collectives queued back to back, no host wait in between (the default drains the stream on
    both sides of every collective -- two host round trips per step -- as belt and braces after the round-1 memset / memcpy-node finding,
    DESIGN.md section 5).  Twelve queued steps with graphs (and the overlapped decoder bucket) must leave both replicas bit-identical
    and equal to one engine on the whole batch.  Transport here is gloo (device-to-host copy, host reduction, copy back between the
    graph launches -- the very pattern that misbehaved with memcpy nodes INSIDE the graphs); RCCL's kernels cannot be run with two
    ranks on a one-GPU box, so the default stays 1 until a multi-GPU run has been seen."""
    import torch.multiprocessing as mp
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), True, overlap, False, False, "0", 12), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert str(r0["mode"]).startswith("hipgraph")
    monkeypatch.setenv("AVSR_PERSISTENT_RNN", "0")
    O, mcfg, W, full = _setup(False)
    model = Seq2SeqModel(mcfg, weights=W)
    batch = Batch.from_numpy(full)
    for _ in range(12):
        model.train_step(batch)
    torch.cuda.synchronize()
    ref = model.export_tf_weights("params")
    for k, v in ref.items():
        assert np.array_equal(r0[k], r1[k]), k
        assert np.abs(r0[k] - v).max() < 2e-5 + 3e-4 * np.abs(v).max(), (k, np.abs(r0[k] - v).max())


# ------------------------------------------------------------------------------------------------
# lip-crop front-end under data parallelism.  The batch norms INSIDE the CNN (and the input batch norm of the CNN-fed stream) normalise
# with PER-RANK statistics -- the documented deviation bench.py states in `config.dp_batch_norm` (synchronising them would put ~20
# small collectives inside the step).  What must hold under the deviation, and is tested here:
#   * replicas stay bit-identical, INCLUDING the moving statistics (each rank sees different frames, so the moving averages are
#     averaged over the ranks in the gradient all-reduce's tail);
#   * when the ranks' shards hold the same utterances the per-rank statistics ARE the global ones, and two ranks must then reproduce
#     one engine on the whole (duplicated) batch: the whole CNN path (gradient sums, loss normaliser, L2, update) minus the deviation.
CASE_CNN = dict(architecture="bimodal", encoder_type="unidirectional", video_units=(32,), audio_units=(32, 32), decoder_units=(32,),
                embedding_size=16, audio_feat=20, regress_aus=True, use_dropout=False, warmup_steps=0, video_processing="resnet_cnn",
                cnn_filters=(8, 8, 16, 16), cnn_dense_units=16, video_feat=16)


# the reference's own filter widths (avsr/avsr.py:33): the 32 -> 64 stride-2 layer's data gradient is four launches, so its batch norm takes
# the stand-alone stage 1 (avsr_bn_bwd_stage1) under sync_cnn_bn -- the path the full-size benchmark shape runs
CASE_CNN_WIDE = dict(CASE_CNN, cnn_filters=(8, 16, 32, 64))


def _setup_cnn(duplicate, case=None):
    import dataclasses
    from avsr_tf1_amd.config import ModelConfig
    from oracle import avsr_oracle as O
    ocfg = O.OracleConfig(**(case or CASE_CNN))
    mcfg = ModelConfig(**{f.name: getattr(ocfg, f.name) for f in dataclasses.fields(ModelConfig) if hasattr(ocfg, f.name)})
    W = O.init_params(ocfg, seed=7)
    if duplicate:
        half = O.synthetic_batch(ocfg, B=2, T_a=15, T_v=5, L=5, ragged=True)
        full = O.Batch(**{k: np.concatenate([getattr(half, k)] * 2) for k in ("audio", "audio_len", "video", "video_len", "aus", "labels", "labels_len")})
    else:
        full = O.synthetic_batch(ocfg, B=4, T_a=15, T_v=5, L=5, ragged=True)
    return O, mcfg, W, full


def _worker_cnn(rank, world, port, out_dir, duplicate):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["AVSR_PERSISTENT_RNN"] = "0"
    import torch.distributed as dist
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    from avsr_tf1_amd.parallel import DataParallelTrainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O, mcfg, W, full = _setup_cnn(duplicate)
    model = Seq2SeqModel(mcfg, weights=W)
    trainer = DataParallelTrainer(model, dist, use_graph=True)
    batch = Batch.from_numpy(_shard(O, full, 2 * rank, 2 * rank + 2))
    for _ in range(3):
        trainer.train_step(batch)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), mode=np.array(trainer.mode), **model.export_tf_weights("params"))
    dist.destroy_process_group()


@pytest.mark.parametrize("duplicate", [True, False])
def test_two_ranks_with_the_lip_cnn(tmp_path, duplicate, monkeypatch):
    import torch.multiprocessing as mp
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_cnn, args=(2, port, str(tmp_path), duplicate), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert str(r0["mode"]).startswith("hipgraph")
    names = [k for k in r0.files if k != "mode"]
    assert any(k.endswith("moving_variance") and "cnn" in k for k in names)
    for k in names:
        assert np.array_equal(r0[k], r1[k]), k                       # replicas stay bit-identical, moving statistics included
    if not duplicate:
        return
    monkeypatch.setenv("AVSR_PERSISTENT_RNN", "0")
    O, mcfg, W, full = _setup_cnn(True)
    model = Seq2SeqModel(mcfg, weights=W)
    batch = Batch.from_numpy(full)
    for _ in range(3):
        model.train_step(batch)
    torch.cuda.synchronize()
    ref = model.export_tf_weights("params")
    for k, v in ref.items():
        if "/cnn/" in k and k.endswith("/bias") and "flatten" not in k:
            continue    # a conv bias ahead of a batch norm has a zero gradient: Adam turns its rounding noise into +-lr steps
        # (moving variances take the Bessel-corrected batch variance: n / (n - 1) with n the rank's rows against the whole batch's)
        tol = 2e-3 if k.endswith("moving_variance") else 1e-4
        # (absolute part: Adam turns the rounding noise of a near-zero gradient component into a visible fraction of lr = 1e-3)
        assert np.abs(r0[k] - v).max() <= 2e-5 + tol * np.abs(v).max(), (k, np.abs(r0[k] - v).max(), np.abs(v).max())


# ------------------------------------------------------------------------------------------------
# data parallelism through the drop-in surface: AVSR(...).train under torch.distributed (two ranks on the box's one GPU, gloo)
# on TFRecords must leave the parameters one rank leaves after training on the whole data (bucket first, then split by rank).
def _avsr_kwargs(tmp):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_avsr import _dataset
    unit_file = os.path.join(tmp, "character_list")
    p = {k: os.path.join(tmp, k + ".tfrecord") for k in ("audio", "video", "labels")}
    if not os.path.exists(unit_file):                                  # written once by the parent; the workers only read
        unit_file, p = _dataset(tmp, n=14)
    return dict(unit="character", unit_file=unit_file, audio_processing="features", audio_train_record=p["audio"],
                audio_test_record=p["audio"], labels_train_record=p["labels"], labels_test_record=p["labels"], batch_size=(4, 4),
                encoder_units_per_layer=((32,), (32, 32)), decoder_units_per_layer=(32,), embedding_size=16, decoding_algorithm="greedy",
                warmup_steps=0, learning_rate=0.01, shuffle_seed=3, use_dropout=False, sampling_probability_outputs=0.0)


def _avsr_worker(rank, world, port, tmp):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["AVSR_PERSISTENT_RNN"] = "0"
    os.chdir(tmp)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import avsr_tf1_amd as avsr
    exp = avsr.AVSR(**_avsr_kwargs(os.path.join(tmp, "data")))
    assert exp._world == 2 and exp._trainer.collective
    exp.train(logfile="logs/dp", num_epochs=3)
    torch.cuda.synchronize()
    np.savez(os.path.join(tmp, "dp_rank%d.npz" % rank), **exp._model.export_tf_weights("params"))
    dist.destroy_process_group()


def test_avsr_train_two_ranks_equals_one_rank_on_the_whole_data(tmp_path, monkeypatch):
    import torch.multiprocessing as mp
    tmp = str(tmp_path)
    os.makedirs(os.path.join(tmp, "data"))
    kw = _avsr_kwargs(os.path.join(tmp, "data"))                       # writes the records once, before the workers read them
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_avsr_worker, args=(2, port, tmp), nprocs=2, join=True)
    r0, r1 = np.load(os.path.join(tmp, "dp_rank0.npz")), np.load(os.path.join(tmp, "dp_rank1.npz"))
    assert os.path.exists(os.path.join(tmp, "logs", "dp")) and open(os.path.join(tmp, "logs", "dp")).read().count("Average batch_loss") == 2
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("AVSR_PERSISTENT_RNN", "0")
    import avsr_tf1_amd as avsr
    one = avsr.AVSR(**kw)
    one.train(logfile="logs/single", num_epochs=3)
    ref = one._model.export_tf_weights("params")
    for k, v in ref.items():
        assert np.array_equal(r0[k], r1[k]), k                       # replicas stay bit-identical
        assert np.abs(r0[k] - v).max() < 2e-5 + 2e-3 * np.abs(v).max(), (k, np.abs(r0[k] - v).max(), np.abs(v).max())


@pytest.mark.parametrize("use_graph", [False, True])
def test_a_redone_pass_moves_the_batch_norm_averages_once(use_graph, monkeypatch):
    """A pass whose persistent kernels flagged is repeated by the trainer (parallel.py).  Every batch norm sits upstream of those kernels,
    so the flagged pass has already blended this step's statistics into the moving averages; the repeat must not blend them again
    (found as a 1-in-6 failure of the two-rank contention test above: moving_mean off by (1 - 0.99^5) / (1 - 0.99^4)).  Forced here:
    check_persistent reports a flag on the second step (without switching anything off), so the repeat is the same computation and the
    result must equal the undisturbed run bit for bit."""
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    from avsr_tf1_amd.parallel import DataParallelTrainer
    monkeypatch.setenv("AVSR_PERSISTENT_RNN", "0")
    O, mcfg, W, full = _setup(False)
    batch = Batch.from_numpy(full)
    outs = []
    for force in (False, True):
        model = Seq2SeqModel(mcfg, weights=W)
        trainer = DataParallelTrainer(model, None, use_graph=use_graph)
        calls = {"n": 0}
        if force:
            def flagged(disable=True, force=False, _c=calls):
                _c["n"] += 1
                return _c["n"] == 2
            monkeypatch.setattr(model, "check_persistent", flagged)
        for _ in range(STEPS):
            trainer.train_step(batch)
        torch.cuda.synchronize()
        if force:
            assert calls["n"] >= 2
        outs.append(model.export_tf_weights("params"))
    for k, v in outs[0].items():
        assert np.array_equal(outs[1][k], v), k


# ------------------------------------------------------------------------------------------------
# Opt-in: DataParallelTrainer(sync_cnn_bn=True) -- the seven batch norms inside the lip CNN and the input batch norm of the CNN-fed stream
# normalise with the statistics of the GLOBAL batch (16 small all-reduces inside the step, eager launches).  Two ranks holding DIFFERENT
# utterances (unequal shards) must then reproduce ONE engine on the whole batch from lip crops -- parameters AND moving statistics --
# which the default (per-rank statistics) cannot (VERDICT r4 missing #2).
def _worker_cnn_sync(rank, world, port, out_dir, sync=True):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["AVSR_PERSISTENT_RNN"] = "0"
    import torch.distributed as dist
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    from avsr_tf1_amd.parallel import DataParallelTrainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O, mcfg, W, full = _setup_cnn(False, CASE_CNN_WIDE)
    model = Seq2SeqModel(mcfg, weights=W)
    trainer = DataParallelTrainer(model, dist, use_graph=True, sync_cnn_bn=sync)
    from avsr_tf1_amd import ops
    calls, stage1 = [0], ops.bn_bwd_stage1

    def counted(*a, **k):
        calls[0] += 1
        return stage1(*a, **k)
    ops.bn_bwd_stage1 = counted
    cut = [0, 1, 4]                                   # unequal shards: 1 and 3 utterances
    batch = Batch.from_numpy(_shard(O, full, cut[rank], cut[rank + 1]))
    losses = []
    for _ in range(3):
        loss, _g = trainer.train_step(batch)
        losses.append(float(loss.item()))
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "%srank%d.npz" % ("" if sync else "nosync_", rank)), mode=np.array(trainer.mode), step_losses=np.array(losses),
             stage1_calls=np.array(calls[0]), **model.export_tf_weights("params"))
    dist.destroy_process_group()


def test_two_ranks_with_synchronised_cnn_batch_norms_equal_one_engine(tmp_path, monkeypatch):
    import torch.multiprocessing as mp
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_cnn_sync, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert "sync_cnn_bn" in str(r0["mode"])
    assert int(r0["stage1_calls"]) == 3                             # the 32 -> 64 stride-2 layer's batch norm, once per step
    names = [k for k in r0.files if k not in ("mode", "step_losses", "stage1_calls")]
    for k in names:
        assert np.array_equal(r0[k], r1[k]), k                       # replicas bit-identical, moving statistics included
    assert np.array_equal(r0["step_losses"], r1["step_losses"])
    monkeypatch.setenv("AVSR_PERSISTENT_RNN", "0")
    O, mcfg, W, full = _setup_cnn(False, CASE_CNN_WIDE)
    model = Seq2SeqModel(mcfg, weights=W)
    batch = Batch.from_numpy(full)
    ref_losses = []
    for _ in range(3):
        loss, _g = model.train_step(batch)
        ref_losses.append(float(loss.item()))
    torch.cuda.synchronize()
    ref = model.export_tf_weights("params")
    # the loss of every step (the third is computed from twice-updated weights and moving statistics)
    assert np.abs(r0["step_losses"] - np.array(ref_losses)).max() < 2e-5 * max(1.0, max(ref_losses)), (r0["step_losses"], ref_losses)
    moved = 0
    for k, v in ref.items():
        if "/cnn/" in k and k.endswith("/bias") and "flatten" not in k:
            continue    # a conv bias ahead of a batch norm has a zero gradient: Adam turns its rounding noise into +-lr steps
        d, tol = np.abs(r0[k] - v), 2e-5 + 1e-4 * np.abs(v).max()
        if "moving_" in k:
            # global statistics: the moving averages agree as closely as everything else (per-rank statistics needed 2e-3 on the
            # variances, and only for duplicated shards)
            assert d.max() <= tol, (k, d.max(), np.abs(v).max())
            moved += int("/cnn/" in k)
        else:
            # Adam normalises every component: a gradient component at rounding-noise level (deep layers over 20 frames) may take its
            # three +-lr steps differently under the two summation orders, so a few entries (<= 2 %) may sit up to 3 * lr = 3e-3 apart;
            # all others must agree to the tolerance
            assert d.max() <= 4e-3 and float((d > tol).mean()) <= 2e-2, (k, d.max(), float((d > tol).mean()))
    assert moved >= 14                                                # seven batch norms x (moving_mean, moving_variance)
    # ... and the check discriminates: the same two shards WITHOUT the option (per-rank statistics) miss the single engine's moving
    # variances by far more than the tolerance above
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_cnn_sync, args=(2, port, str(tmp_path), False), nprocs=2, join=True)
    n0 = np.load(tmp_path / "nosync_rank0.npz")
    worst = max(np.abs(n0[k] - v).max() / (2e-5 + 1e-4 * np.abs(v).max()) for k, v in ref.items() if "/cnn/" in k and k.endswith("moving_variance"))
    assert worst > 3.0, worst
