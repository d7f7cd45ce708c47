"""CPU tests of the oracle itself (no GPU): golden regression, gradient checks, independent cross-checks and
the structural invariants SURVEY.md 8(c) lists.  The oracle is test infrastructure -- "parity unpinned"."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import avsr_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(path):
    z = np.load(path)
    cfg = O.OracleConfig(**{k: (tuple(tuple(x) if isinstance(x, list) else x for x in v) if isinstance(v, list) else v)
                            for k, v in json.loads(str(z["cfg_json"])).items()})
    W = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    b = O.Batch(**{k[3:]: z[k] for k in z.files if k.startswith("in:")})
    out = {k[4:]: z[k] for k in z.files if k.startswith("out:")}
    return cfg, W, b, out


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "oracle_restatement_*.npz"))))
def test_oracle_matches_committed_fixture(path):
    cfg, W, b, out = load_fixture(path)
    r = O.train_step(W, None, cfg, b)
    assert abs(r["loss"] - float(out["loss"])) < 1e-9
    assert abs(r["global_norm"] - float(out["global_norm"])) < 1e-9
    assert np.abs(r["logits"] - out["logits"]).max() < 1e-6
    assert (O.greedy_decode(W, cfg, b, max_steps=8) == out["greedy_ids"]).all()


def test_lstm_cell_against_torch_lstmcell():
    """TF gate order i,j,f,o with forget_bias 1.0 == torch's i,f,g,o after permutation (independent implementation)."""
    rng = np.random.default_rng(0)
    B, I, H = 5, 7, 6
    W = rng.standard_normal((I + H, 4 * H)) * 0.4
    b = rng.standard_normal(4 * H) * 0.2
    x, c, h = rng.standard_normal((B, I)), rng.standard_normal((B, H)) * 0.3, rng.standard_normal((B, H)) * 0.3
    c2, h2 = O.lstm_cell(*[torch.tensor(a) for a in (x, c, h, W, b)])
    cell = torch.nn.LSTMCell(I, H).double()
    i_, j_, f_, o_ = [W[:, k * H:(k + 1) * H] for k in range(4)]
    bi, bj, bf, bo = [b[k * H:(k + 1) * H] for k in range(4)]
    Wt = np.concatenate([i_, f_, j_, o_], axis=1)            # torch order i, f, g, o
    bt = np.concatenate([bi, bf + 1.0, bj, bo])
    with torch.no_grad():
        cell.weight_ih.copy_(torch.tensor(Wt[:I].T)); cell.weight_hh.copy_(torch.tensor(Wt[I:].T))
        cell.bias_ih.copy_(torch.tensor(bt)); cell.bias_hh.zero_()
        ht, ct = cell(torch.tensor(x), (torch.tensor(h), torch.tensor(c)))
    ct_clip = torch.clamp(ct, -1, 1)
    assert np.abs(c2.numpy() - ct_clip.numpy()).max() < 1e-12
    unclipped = (ct.abs() < 1).numpy()
    assert np.abs((h2 - ht).numpy()[unclipped]).max() < 1e-12


def _small(arch="bimodal", **kw):
    base = dict(architecture=arch, video_units=(8,), audio_units=(8, 8), decoder_units=(8,), embedding_size=4,
                video_feat=4, audio_feat=8, regress_aus=True)
    if arch == "unimodal":
        base["video_units"], base["regress_aus"] = None, False
    base.update(kw)
    cfg = O.OracleConfig(**base)
    return cfg, O.init_params(cfg, seed=5), O.synthetic_batch(cfg, B=3, T_a=7, T_v=4, L=4, ragged=True)


@pytest.mark.parametrize("arch,att,extra", [("bimodal", "scaled_luong", {}), ("av_align", "scaled_luong", {}),
                                            ("unimodal", "normed_bahdanau", {}),
                                            ("bimodal", "scaled_luong", dict(input_dense_layers=(6, 5))),    # encoder.py:148-171
                                            ("unimodal", "scaled_luong", dict(decoder_units=(8, 8))),        # MultiRNNCell decoder
                                            ("unimodal", "bahdanau", dict(highway_encoder=True, audio_units=(8, 8, 8))),
                                            ("bimodal", "scaled_luong", dict(instance_normalisation=True, residual_encoder=True,
                                                                             audio_units=(8, 8, 8))),
                                            ("bimodal", "bahdanau", dict(decoder_units=(8, 8, 8), encoder_weight_sharing=True,
                                                                         audio_units=(8, 8, 8)))])
def test_gradients_by_finite_differences(arch, att, extra):
    cfg, W, b = _small(arch, attention_type=((att,), (att,)), **extra)
    r = O.train_step(W, None, cfg, b)

    def loss_of(Wn):
        P = O.to_torch(Wn)
        logits, m = O.forward_train(P, cfg, b)
        return float(O.loss_fn(P, cfg, b, logits, m)[0])

    rng = np.random.default_rng(1)
    names = [k for k in O.trainable_names(W)]
    picked = [names[i] for i in rng.choice(len(names), size=min(8, len(names)), replace=False)]
    for k in picked + [n for n in names if ("/dense" in n or "/in/" in n or "carry" in n or n.startswith("dec/l")) and n not in picked]:
        idx = tuple(int(rng.integers(0, s)) for s in W[k].shape)
        eps = 1e-5
        Wp = {n: v.astype(np.float64).copy() for n, v in W.items()}
        Wm = {n: v.astype(np.float64).copy() for n, v in W.items()}
        Wp[k][idx] += eps
        Wm[k][idx] -= eps
        fd = (loss_of(Wp) - loss_of(Wm)) / (2 * eps)
        an = float(r["grads"][k][idx])
        assert abs(fd - an) < 1e-6 + 1e-4 * abs(an), (k, idx, fd, an)


@pytest.mark.parametrize("base,flags", [
    (dict(arch="unimodal", encoder_type="bidirectional", audio_units=(8, 8, 8)), dict(residual_encoder=True)),
    (dict(arch="bimodal", encoder_type="bidirectional", audio_units=(8, 8, 8), video_units=(8, 8)), dict(highway_encoder=True)),
    (dict(arch="unimodal", encoder_type="bidirectional", audio_units=(8, 8, 8)), dict(encoder_weight_sharing=True)),
    (dict(arch="av_align", audio_units=(8, 8, 8), video_units=(8,)), dict(residual_encoder=True, encoder_weight_sharing=True)),
    (dict(arch="av_align", audio_units=(8, 8, 8), video_units=(8,)), dict(highway_encoder=True))])
def test_wrapper_flags_are_inert_where_the_reference_ignores_them(base, flags):
    """residual / highway / weight sharing reach build_rnn_layers only from the unidirectional Seq2SeqEncoder branch
    (encoder.py:67-78); bidirectional stacks (encoder.py:92-108) and the AV-Align audio stack (encoder.py:225-233) never see them:
    same variables, same loss, same gradients with and without the flags."""
    base = dict(base)
    arch = base.pop("arch")
    cfg0, W0, b = _small(arch, **base)
    cfg1, W1, _ = _small(arch, **base, **flags)
    assert sorted(W0) == sorted(W1) and all(np.array_equal(W0[k], W1[k]) for k in W0)
    r0, r1 = O.train_step(W0, None, cfg0, b), O.train_step(W1, None, cfg1, b)
    assert r0["loss"] == r1["loss"]
    assert all(np.array_equal(r0["grads"][k], r1["grads"][k]) for k in r0["grads"])


def test_wrapper_flags_apply_to_unidirectional_stacks():
    cfg0, W0, b = _small("bimodal", audio_units=(8, 8, 8), video_units=(8, 8))
    for flags, extra_vars in ((dict(residual_encoder=True), 0), (dict(highway_encoder=True), 2 * (2 + 1)),
                              (dict(encoder_weight_sharing=True), -2)):
        cfg1, W1, _ = _small("bimodal", audio_units=(8, 8, 8), video_units=(8, 8), **flags)
        assert len(W1) - len(W0) == extra_vars, (flags, sorted(set(W1) ^ set(W0)))
        if extra_vars == 0:
            assert O.train_step(W0, None, cfg0, b)["loss"] != O.train_step(W1, None, cfg1, b)["loss"]


def test_input_batch_norm_moving_variance_is_the_biased_one():
    """Rank-3 [B,T,F] input: TF 1.13 drops fused=True (encoder.py:44-50) -> moving variance from tf.nn.moments' biased variance;
    the rank-4 CNN maps keep the fused kernel and its Bessel-corrected update (video.py:8-12)."""
    x = torch.randn(3, 5, 4, dtype=torch.float64)
    P = {"p/gamma": torch.ones(4, dtype=torch.float64), "p/beta": torch.zeros(4, dtype=torch.float64),
         "p/moving_mean": torch.zeros(4, dtype=torch.float64), "p/moving_variance": torch.ones(4, dtype=torch.float64)}
    var = x.reshape(-1, 4).var(dim=0, unbiased=False)
    up = {}
    O.batch_norm(x, P, "p", True, up)
    assert torch.allclose(up["p/moving_variance"], 0.99 + 0.01 * var)
    up = {}
    O.batch_norm(x, P, "p", True, up, fused=True)
    assert torch.allclose(up["p/moving_variance"], 0.99 + 0.01 * var * 15 / 14)


def test_masking_invariants():
    cfg, W, b = _small("bimodal")
    outs = O.encoder_outputs(W, cfg, b, training=False)
    for s, lens in (("video", b.video_len), ("audio", b.audio_len)):
        mem, (c, h) = outs[s]
        for i, n in enumerate(lens):
            assert np.all(mem[i, n:] == 0.0)                    # zero outputs past sequence_length
            assert np.allclose(mem[i, n - 1], h[i])             # state carried = output at last valid step
    P = O.to_torch(W)
    m = O._Model(P, cfg, b, False, torch.float64)
    for mech in m.mechs:
        al, _ = mech(torch.zeros(3, cfg.decoder_units[0], dtype=torch.float64))
        assert np.allclose(al.sum(-1).numpy(), 1.0)
        assert np.all(al.numpy()[~mech.mask.numpy()] == 0.0)    # exactly zero past the memory length


def test_greedy_zero_after_eos_and_stops():
    cfg, W, b = _small("unimodal")
    W = {k: v.copy() for k, v in W.items()}
    W["dec/out/bias"][cfg.eos_id] += 2.5                         # EOS soon, but not necessarily at step 0
    ids = O.greedy_decode(W, cfg, b, max_steps=20)
    assert ids.shape[1] < 20
    for row in ids:
        pos = np.where(row == cfg.eos_id)[0]
        if len(pos):
            assert np.all(row[pos[0] + 1:] == 0)
    assert (ids[:, -1] == cfg.eos_id).any()                      # the loop ended when the LAST utterance emitted EOS


def test_adam_and_warmup_formula():
    cfg, W, b = _small("unimodal", warmup_steps=3)
    assert abs(O.lr_at(cfg, 0) - cfg.learning_rate / 3) < 1e-12 and O.lr_at(cfg, 5) == cfg.learning_rate
    r = O.train_step(W, None, cfg, b)
    k = "dec/out/bias"
    g = r["grads"][k] * min(1.0, cfg.max_gradient_norm / r["global_norm"])
    m, v = 0.1 * g, 0.001 * g * g
    lr_t = O.lr_at(cfg, 0) * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.abs(r["params"][k] - (W[k] - lr_t * m / (np.sqrt(v) + 1e-8))).max() < 1e-7


def test_cosine_restarts_schedule():
    """lr_decay=('cosine_restarts', 10): periods 10, 20, 40 ... (t_mul=2), full lr at every restart, half-way = lr/2."""
    cfg, _, _ = _small("unimodal", warmup_steps=0, lr_decay_steps=10)
    lr = cfg.learning_rate
    for step, want in ((0, lr), (5, lr / 2), (10, lr), (20, lr / 2), (30, lr), (50, lr / 2), (70, lr)):
        assert abs(O.lr_at(cfg, step) - want) < 1e-12, (step, O.lr_at(cfg, step), want)
    assert 0 < O.lr_at(cfg, 9) < 0.03 * lr
    cfg2, _, _ = _small("unimodal", warmup_steps=4, lr_decay_steps=10)
    assert abs(O.lr_at(cfg2, 1) - O.lr_at(cfg, 1) * 0.5) < 1e-12         # warm-up multiplies the decayed rate


@pytest.mark.parametrize("arch,cell,w", [("unimodal", "lstm", 0.0), ("unimodal", "gru", 0.6), ("bimodal", "lstm", 0.5), ("av_align", "lstm", 0.0)])
def test_beam_search_with_a_beam_that_never_prunes_finds_the_exhaustive_optimum(arch, cell, w):
    """Pins the search logic of the beam-search restatement ('low confidence') to an independent ground truth: with a
    beam wide enough to hold every live prefix, the best hypothesis must be the arg-max over ALL output sequences of the
    length-normalised log-probability, computed here by plain recursion over prefixes with the model's own step function
    (finished-beam masking, score accumulation, parent gathering, gather_tree and the EOS padding all have to be right for that)."""
    V, eos, go, steps = 5, 3, 4, 3
    kw = dict(architecture=arch, cell_type=cell, encoder_type="unidirectional", audio_units=(8,), video_units=(8,) if arch != "unimodal" else None,
              decoder_units=(8,), embedding_size=4, vocab_size=V, eos_id=eos, go_id=go, audio_feat=4, video_feat=4,
              attention_type=(("luong",), ("scaled_luong",)))
    cfg = O.OracleConfig(**kw)
    cfg.validate()
    rng = np.random.default_rng(11)
    W = O.init_params(cfg, seed=4)
    for k in W:                                             # sharpen the output distribution so that the optimum is not a near-tie
        if k.startswith("dec/out"):
            W[k] = (rng.standard_normal(W[k].shape) * 1.5).astype(np.float32)
    batch = O.synthetic_batch(cfg, B=3, T_a=6, T_v=4, L=3, ragged=True)
    K = V ** (steps - 1)
    ids, scores, lens, _gap = O.beam_search_decode(W, cfg, batch, beam_width=K, length_penalty_weight=w, max_steps=steps, return_all=True)

    P = O.to_torch(W, torch.float64)
    m = O._Model(P, cfg, batch, False, torch.float64)
    emb = O._embedding(P, cfg)
    B = m.B
    for b in range(B):
        sel = lambda s: tuple(sel(x) for x in s) if isinstance(s, (tuple, list)) else s[b:b + 1]
        best = [(-np.inf, None)]

        def rec(prefix, logp, state, att, tok, t):
            with torch.no_grad():
                # the model's step runs the whole batch; row b is what we follow
                x = emb[torch.full((B,), tok, dtype=torch.int64)]
                out, ns, natt, _ = m.step(x, state, att, t)
                lp = torch.log_softmax(m.logits(out), dim=-1)[b].numpy()
            for v in range(V):
                seq, tot = prefix + [v], logp + lp[v]
                n_words = sum(1 for s in seq if s != eos)
                if v == eos or t + 1 == steps:
                    score = tot / (((5.0 + n_words) / 6.0) ** w)
                    if score > best[0][0]:
                        best[0] = (score, seq)
                else:
                    rec(seq, tot, ns, natt, v, t + 1)

        with torch.no_grad():
            rec([], 0.0, m.init_state, torch.zeros(B, m.att_dim, dtype=torch.float64), go, 0)
        score, seq = best[0]
        want = seq + [eos] * (ids.shape[1] - len(seq))
        assert list(ids[b, :, 0]) == want[:ids.shape[1]], (b, list(ids[b, :, 0]), want, score)


def test_beam_search_follow_mode_checks_every_step_of_another_search():
    """`follow` (the test aid tests/test_gpu_beam.py uses on the engine's selections): following the oracle's own search reports zero
    deviation and identical selections at every step and reproduces its result; a search that swaps two beams of a step (what fp32
    rounding does at a near-tie) deviates by exactly their score gap and is followed onto ITS branch; a search that picks a
    candidate from outside the top K deviates by far more than any rounding."""
    V, eos, go = 7, 5, 6
    cfg = O.OracleConfig(architecture="bimodal", audio_units=(8,), video_units=(8,), decoder_units=(8,), embedding_size=4, vocab_size=V,
                         eos_id=eos, go_id=go, audio_feat=4, video_feat=4)
    cfg.validate()
    W = O.init_params(cfg, seed=7)
    rng = np.random.default_rng(3)
    W["dec/out/kernel"] = (rng.standard_normal(W["dec/out/kernel"].shape) * 2).astype(np.float32)
    batch = O.synthetic_batch(cfg, B=3, T_a=6, T_v=4, L=3, ragged=True)
    K, steps = 4, 6
    ids, lp, ln, tr = O.beam_search_decode(W, cfg, batch, beam_width=K, max_steps=steps, return_trace=True)
    ids2, lp2, ln2, tr2 = O.beam_search_decode(W, cfg, batch, beam_width=K, max_steps=steps, return_trace=True,
                                               follow=(tr["step_ids"], tr["parent_ids"]))
    assert (ids2 == ids).all() and np.array_equal(lp2, lp) and (ln2 == ln).all()
    assert tr2["follow_same"].all() and tr2["follow_distinct"].all() and not tr2["follow_short"] and tr2["follow_dev"].max() == 0.0
    # swap beams 1 and 2 of utterance 0 at step 1: deviation = their score gap, and the continuation follows the swapped order
    sid, pid = tr["step_ids"].copy(), tr["parent_ids"].copy()
    sid[1, 0, [1, 2]], pid[1, 0, [1, 2]] = sid[1, 0, [2, 1]], pid[1, 0, [2, 1]]
    _i, _l, _n, tr3 = O.beam_search_decode(W, cfg, batch, beam_width=K, max_steps=2, return_trace=True, follow=(sid[:2], pid[:2]))
    assert not tr3["follow_same"][1, 0] and tr3["follow_same"][1, 1:].all() and tr3["follow_same"][0].all()
    assert 0.0 < tr3["follow_dev"][1, 0] < 10.0 and tr3["follow_distinct"].all()
    assert (tr3["step_ids"][1, 0] == sid[1, 0]).all() and (tr3["parent_ids"][1, 0] == pid[1, 0]).all()
    # a duplicate candidate is flagged
    sid4, pid4 = tr["step_ids"].copy(), tr["parent_ids"].copy()
    sid4[0, 2, 1], pid4[0, 2, 1] = sid4[0, 2, 0], pid4[0, 2, 0]
    _i, _l, _n, tr4 = O.beam_search_decode(W, cfg, batch, beam_width=K, max_steps=1, return_trace=True, follow=(sid4[:1], pid4[:1]))
    assert not tr4["follow_distinct"][0, 2] and tr4["follow_distinct"][0, :2].all()
