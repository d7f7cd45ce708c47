"""Seeded random walks over the option space: every valid combination of the reference's model options must give the same train
step and the same greedy ids on the HIP engine as on the CPU oracle.  Catches interactions the hand-picked cases miss
(e.g. highway + input Dense + instance norm + bidirectional + dropout + multi-layer decoder + label smoothing)."""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ATT = ["luong", "scaled_luong", "bahdanau", "normed_bahdanau"]


def _sample(rng):
    from oracle import avsr_oracle as O
    for _ in range(200):
        arch = rng.choice(["unimodal", "unimodal", "bimodal", "av_align"])
        cell = rng.choice(["lstm", "lstm", "gru"])
        u = int(rng.choice([64, 128, 256])) if os.environ.get("AVSR_FUZZ_BIG") else int(rng.choice([16, 32]))
        na, nv = int(rng.integers(1, 4)), int(rng.integers(1, 3))
        kw = dict(architecture=arch, cell_type=cell, encoder_type=rng.choice(["unidirectional", "bidirectional"]),
                  audio_units=(u,) * na, video_units=(u,) * nv if (arch != "unimodal" or rng.random() < 0.3) else None,
                  decoder_units=(u,) * int(rng.choice([1, 1, 2, 3])), embedding_size=int(rng.choice([8, 16])),
                  attention_type=((rng.choice(ATT),), (rng.choice(ATT),)), video_feat=12, audio_feat=20,
                  batch_normalisation=bool(rng.random() < 0.7), instance_normalisation=bool(rng.random() < 0.3),
                  input_dense_layers=(int(rng.choice([16, 24])),) if rng.random() < 0.3 else (0,),
                  residual_encoder=bool(rng.random() < 0.25), highway_encoder=bool(rng.random() < 0.25),
                  encoder_weight_sharing=bool(rng.random() < 0.2), enable_attention=bool(rng.random() < 0.9),
                  regress_aus=bool(rng.random() < 0.4), use_dropout=bool(rng.random() < 0.5),
                  sampling_probability=float(rng.choice([0.0, 0.0, 0.3])),
                  loss_fun=rng.choice([None, None, None, "focal_loss", "mc_loss"]),
                  label_smoothing=float(rng.choice([0.0, 0.0, 0.1])), optimiser=rng.choice(["Adam", "Adam", "Nadam", "AdamW", "Momentum"]),
                  lr_decay_steps=int(rng.choice([0, 0, 7])), warmup_steps=int(rng.choice([0, 5, 750])),
                  clip_gradients=bool(rng.random() < 0.8), recurrent_l2=rng.choice([None, 1e-4]))
        if cell == "gru":                        # (the bimodal decoder needs LSTM state tuples, in the reference too)
            if arch == "bimodal":
                kw["architecture"] = arch = "unimodal"
        if kw["video_units"] is None and arch == "unimodal" and rng.random() < 0.2:
            kw["audio_units"], kw["video_units"] = None, (u,) * nv
        if kw["video_units"] is None:
            kw["regress_aus"] = False
        if kw["loss_fun"] is not None:
            kw["label_smoothing"] = 0.0
        try:
            cfg = O.OracleConfig(**kw)
            cfg.validate()
            return cfg
        except Exception:
            continue
    raise RuntimeError("no valid configuration sampled")


# 137 / 244 / 256 / 292: GRU decoders whose lower beams change parents -- the seeds that exposed the beam-search gather of r*h
@pytest.mark.parametrize("seed", sorted(set(range(int(os.environ.get("AVSR_FUZZ_N", "160")))) | {137, 244, 256, 292}))
def test_random_configuration(seed):
    rng = np.random.default_rng(1000 + seed)
    _check(_sample(rng), rng, seed)


@pytest.mark.parametrize("seed", range(int(os.environ.get("AVSR_FUZZ_VOCAB_N", "32"))))
def test_random_configuration_other_vocabularies(seed):
    """The same walk with the vocabularies the reference ships besides characters (viseme V = 15, phoneme V = 41; avsr/misc/*_list) and
    the largest the fused decode takes (V = 64): logits split, sampler, sequence loss, one-hot table and (V > 32) the 64-symbol rows of
    the fused kernel under random model options.  (A separate generator picks the vocabulary: the configurations of the seeds above
    stay what they were.)"""
    rng = np.random.default_rng(1000 + seed)
    ocfg = _sample(rng)
    V = int(np.random.default_rng(7000 + seed).choice([15, 41, 41, 64]))
    ocfg = dataclasses.replace(ocfg, vocab_size=V, eos_id=V - 2, go_id=V - 1)
    ocfg.validate()
    _check(ocfg, rng, 7000 + seed)


def _unpadded(ocfg, rng):
    """The same configuration at widths the kernels do not take natively (the engine pads them to multiples of 4 inside;
    ModelConfig.engine()): odd / 4k+2 unit, feature, embedding and Dense sizes, sometimes one-hot decoder inputs.  Dropout off: its
    masks are hashed over the padded index space, so only the deterministic graph is comparable with the oracle."""
    d = int(rng.choice([1, 2, 3, 5, 6, 7]))
    cut = lambda t: None if t is None else tuple(u - d for u in t)
    dense = ocfg.input_dense_layers if ocfg.input_dense_layers[0] <= 0 else tuple(u - int(rng.choice([1, 2, 3])) for u in ocfg.input_dense_layers)
    return dataclasses.replace(ocfg, audio_units=cut(ocfg.audio_units), video_units=cut(ocfg.video_units), decoder_units=cut(ocfg.decoder_units),
                               embedding_size=int(rng.choice([0, 5, 7, 10, 13])), video_feat=int(rng.choice([5, 11, 13])),
                               audio_feat=int(rng.choice([19, 21, 39])), input_dense_layers=dense, use_dropout=False)


@pytest.mark.parametrize("seed", range(int(os.environ.get("AVSR_FUZZ_ODD_N", "64"))))
def test_random_configuration_at_unpadded_sizes(seed):
    rng = np.random.default_rng(9000 + seed)
    ocfg = _unpadded(_sample(rng), rng)
    ocfg.validate()
    _check(ocfg, rng, seed)


def _check(ocfg, rng, seed):
    from avsr_tf1_amd.config import ModelConfig
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    from oracle import avsr_oracle as O
    mcfg = ModelConfig(**{f.name: getattr(ocfg, f.name) for f in dataclasses.fields(ModelConfig) if hasattr(ocfg, f.name)})
    mcfg.validate()
    W = O.init_params(ocfg, seed=seed)
    for k in W:
        if k.endswith(("bias", "/b", "beta")):
            W[k] = (rng.standard_normal(W[k].shape) * 0.1).astype(np.float32)
    B, Ta, Tv, L = int(rng.integers(1, 7)), int(rng.integers(3, 24)), int(rng.integers(2, 10)), int(rng.integers(2, 8))
    if os.environ.get("AVSR_FUZZ_BIG"):              # sizes at which the persistent encoder kernels (and their 64-row slices) run
        B, Ta, Tv = int(rng.integers(8, 71)), int(rng.integers(20, 60)), int(rng.integers(8, 20))
    batch = O.synthetic_batch(ocfg, B=B, T_a=Ta, T_v=Tv, L=L, ragged=True)
    r1 = O.train_step(W, None, ocfg, batch)
    r2 = O.train_step(r1["params"], r1["opt"], ocfg, batch)
    model = Seq2SeqModel(mcfg, weights=W)
    db = Batch.from_numpy(batch)
    logits = model.forward_train(db)
    model.backward()
    model.apply_update()
    torch.cuda.synchronize()
    desc = repr(ocfg)
    assert np.abs(logits.cpu().numpy() - r1["logits"]).max() < 1e-4, desc
    assert abs(float(model.loss.item()) - r1["loss"]) < 1e-4, desc
    if ocfg.clip_gradients:
        assert abs(float(model.gnorm.item()) - r1["global_norm"]) < 1e-4 * max(1.0, r1["global_norm"]), desc
    grads = model.export_tf_weights("grads")
    for k, g in r1["grads"].items():
        scale = max(1e-3, np.abs(g).max())
        # (+3e-6 absolute: some gradients are mathematically zero -- e.g. a batch-norm beta in front of an instance norm -- and then
        # consist of rounding noise only: where the fp64 oracle itself reports < 1e-9 the engine's fp32 sum of cancelling terms gets
        # 1e-5; seed 4094 of the 9 000-seed run had 3.3e-6 on such a `video/bn/beta`, profiles/r06_fuzz_final.txt)
        assert np.abs(grads[k] - g).max() < 2e-4 * scale + (1e-5 if np.abs(g).max() < 1e-9 else 3e-6), (k, desc)
    loss2, _ = model.train_step(db)
    torch.cuda.synchronize()
    assert abs(float(loss2.item()) - r2["loss"]) < 3e-4, desc
    ids_ref, lg_ref = O.greedy_decode(r2["params"], ocfg, batch, max_steps=6, return_logits=True)
    ids = model.greedy_decode(db, max_steps=6).cpu().numpy()
    if ids.shape != ids_ref.shape or not (ids == ids_ref).all():
        # the only acceptable difference: an exact argmax tie in the oracle (top-2 logits within 1e-5) at the first step that differs
        T = min(ids.shape[1], ids_ref.shape[1])
        bad = np.nonzero((ids[:, :T] != ids_ref[:, :T]).any(axis=0))[0]
        assert len(bad), desc
        t0 = int(bad[0])
        for b in np.nonzero(ids[:, t0] != ids_ref[:, t0])[0]:
            top2 = np.sort(lg_ref[b, t0])[-2:]
            assert top2[1] - top2[0] < 1e-5, (desc, b, t0, top2)
        return
    # beam search on the same weights with EOS made reachable (all kept beams compared, not only the best)
    W2 = {k: v.copy() for k, v in r2["params"].items()}
    W2["dec/out/bias"][ocfg.eos_id] += 1.0
    K = int(rng.integers(1, 5))
    res = O.beam_search_decode(W2, ocfg, batch, beam_width=K, max_steps=7, return_all=True)
    ref, score = res[0], res[1]
    m2 = Seq2SeqModel(mcfg, weights=W2)
    out = m2.beam_search_decode(db, beam_width=K, max_steps=7, check_every=3, return_all=True).cpu().numpy()
    assert out.shape == ref.shape, desc
    assert (out[:, :, 0] == ref[:, :, 0]).all(), desc                         # the returned (best) hypothesis
    if (out != ref).any():
        # lower beams may follow another branch where two candidates score within fp32 noise of each other (random weights give
        # near-uniform distributions).  The fp64 oracle then FOLLOWS the engine's search (tests/test_gpu_beam.py::_follow_check): at every
        # step the engine's j-th selection must score within 2e-5 of the oracle's j-th best from the same state, the selections must be
        # distinct, and the kept beams must be what the oracle arrives at along that branch.  (Until round 6 this compared the FINAL
        # scores' gaps with 5e-3 -- a tie at an early step can end far wider than that: seed 187 of the large-size run.)
        X = m2._beam_ws[2]
        Bq, T = out.shape[0], out.shape[1]
        sid = X["sid"].cpu().numpy().reshape(-1, Bq, K)[:T]
        pid = X["pid"].cpu().numpy().reshape(-1, Bq, K)[:T]
        ref2, _lp, _ln, tr = O.beam_search_decode(W2, ocfg, batch, beam_width=K, max_steps=7, return_trace=True, follow=(sid, pid))
        assert not tr["follow_short"] and tr["step_ids"].shape[0] == T, desc
        assert tr["follow_distinct"].all(), desc
        assert float(tr["follow_dev"].max()) < 2e-5, (desc, float(tr["follow_dev"].max()), np.argwhere(tr["follow_dev"] >= 2e-5)[:4])
        assert (out == ref2).all(), desc


@pytest.mark.parametrize("seed", range(int(os.environ.get("AVSR_FUZZ_DP_N", "32"))))
def test_random_configuration_two_shards(seed):
    """Data-parallel algebra under random options: two engine instances hold unequal shards of a batch, the test plays the collectives
    (loss normalisers, sync batch-norm phases, gradient sum) exactly as DataParallelTrainer issues them; the summed gradient and the
    summed loss must be those of one engine on the whole batch."""
    import dataclasses as dc
    from avsr_tf1_amd.config import ModelConfig
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    from oracle import avsr_oracle as O
    rng = np.random.default_rng(5000 + seed)
    ocfg = dc.replace(_sample(rng), use_dropout=False, sampling_probability=0.0)
    mcfg = ModelConfig(**{f.name: getattr(ocfg, f.name) for f in dc.fields(ModelConfig) if hasattr(ocfg, f.name)})
    W = O.init_params(ocfg, seed=seed)
    B = int(rng.integers(3, 9))
    batch = O.synthetic_batch(ocfg, B=B, T_a=int(rng.integers(5, 24)), T_v=int(rng.integers(3, 10)), L=int(rng.integers(2, 8)), ragged=True)
    cut = int(rng.integers(1, B))
    names = ("audio", "audio_len", "video", "video_len", "aus", "labels", "labels_len")

    def shard(lo, hi):
        return Batch.from_numpy(O.Batch(**{k: (None if getattr(batch, k) is None else np.ascontiguousarray(getattr(batch, k)[lo:hi])) for k in names}))
    whole = Seq2SeqModel(mcfg, weights=W)
    whole.forward_train(Batch.from_numpy(batch))
    whole.backward()
    models, shards = [Seq2SeqModel(mcfg, weights=W) for _ in range(2)], [shard(0, cut), shard(cut, B)]
    sync = [m.bn_sync_enable() is not None for m in models][0]
    norm = sum(torch.cat([m.local_loss_denominator(b), m.local_au_count(b)]) for m, b in zip(models, shards))
    if sync:
        tot = sum(m.bn_sync_sums(b).clone() for m, b in zip(models, shards))
        for m in models:
            m.bn_sync["sum"].copy_(tot)
        tot = sum(m.bn_sync_squares(b).clone() for m, b in zip(models, shards))
        for m in models:
            m.bn_sync["sq"].copy_(tot)
    for m, b in zip(models, shards):
        m.dp_norm[:2].copy_(norm)
        m.au_scale, m.au_external = 1.0, True
        m.forward_train(b, compute_denom=False)
        m.backward()
    torch.cuda.synchronize()
    g, gw = models[0].grads + models[1].grads, whole.grads
    desc = repr(ocfg)
    assert float((g - gw).abs().max()) < 3e-5 * max(1.0, float(gw.abs().max())), (float((g - gw).abs().max()), desc)
    assert abs(float(models[0].loss.item() + models[1].loss.item()) - float(whole.loss.item())) < 2e-4, desc


# ------------------------------------------------------------------------------------------------
# Lip-CNN geometries the benchmark does not visit: other crop sizes (odd, non-square, one channel), other filter ladders (one block, five
# blocks, 4-channel layers that fall off the pixel-pair / 4x4x1 forms), other dense widths -- every convolution form's shape dispatch
# (frames per pass, tap groups, workgroups per CU, stride-2 parity classes) against the oracle's plain im2col arithmetic.
CNN_HW = [(36, 36, 3), (24, 24, 3), (20, 28, 3), (12, 12, 3), (36, 36, 1), (17, 17, 3), (9, 13, 3), (30, 18, 3)]
CNN_FILTERS = [(8, 16, 32, 64), (8, 8, 16, 16), (4, 8), (8,), (16, 32), (4, 4, 4, 8, 8), (12, 20), (8, 8)]


@pytest.mark.parametrize("seed", range(int(os.environ.get("AVSR_FUZZ_CNN_N", "48"))))
def test_random_lip_cnn_geometry(seed):
    from test_gpu_model import make
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    rng = np.random.default_rng(9000 + seed)
    hw = CNN_HW[int(rng.integers(len(CNN_HW)))]
    filters = CNN_FILTERS[int(rng.integers(len(CNN_FILTERS)))]
    while min(hw[0], hw[1]) < 2 ** (len(filters) - 1):          # every strided block must leave at least one pixel
        filters = filters[:-1]
    dense = int(rng.choice([8, 16, 32]))
    case = "c4_bimodal_cnn" if rng.integers(2) else "c3_video_cnn_bi"
    B, Tv = int(rng.integers(2, 5)), int(rng.integers(3, 7))
    O, ocfg, mcfg, W, batch = make(case, B=B, Ta=4 * Tv, Tv=Tv, L=5, video_hw=hw, cnn_filters=filters, cnn_dense_units=dense, video_feat=dense,
                                   use_dropout=bool(rng.integers(2)))
    tag = (seed, case, hw, filters, dense, B, Tv)
    ref = O.train_step(W, None, ocfg, batch)
    model = Seq2SeqModel(mcfg, weights=W)
    db = Batch.from_numpy(batch)
    logits = model.forward_train(db)
    model.backward()
    model.apply_update()
    torch.cuda.synchronize()
    assert np.abs(logits.cpu().numpy() - ref["logits"]).max() < 1e-4, tag
    assert abs(float(model.loss.item()) - ref["loss"]) < 1e-4, tag
    grads = model.export_tf_weights("grads")
    bad = [] if abs(float(model.gnorm.item()) - ref["global_norm"]) < 1e-4 * max(1.0, ref["global_norm"]) else ["global_norm"]
    for k, g in ref["grads"].items():
        scale = max(1e-3, np.abs(g).max())
        if not np.abs(grads[k] - g).max() < 2e-4 * scale + 1e-6:
            bad.append(k)
    if bad and ref["relu_margin"] < 2e-6:
        # A ReLU input within fp32 rounding of zero (seed 255 of the 400-seed run of round 6: 5.3e-7): the forward values agree, the
        # gradient is discontinuous there and the fp32 engine and the fp64 oracle sit on different sides -- exactly one element's
        # contribution apart (tools/relu_kink_probe.py).  Not a comparison this test can make; anything else is a failure.
        pytest.skip("ReLU kink: |input| %.2g at the smallest; gradients differ by that element's contribution (%s)" % (ref["relu_margin"], bad[:3]))
    assert not bad, (tag, bad, ref["relu_margin"])
    ids_ref = O.greedy_decode(ref["params"], ocfg, batch, max_steps=6)
    assert (model.greedy_decode(db, max_steps=6).cpu().numpy() == ids_ref).all(), tag
