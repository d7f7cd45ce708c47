"""The reference's DEFAULT evaluation path at its default width: beam search of width 10 (avsr/avsr.py:58-59; BeamSearchDecoder at
decoder_unimodal.py:248-271, decoder_bimodal.py:358-381; length penalty 0.6 unimodal / 0.5 bimodal) and the vocabularies the
reference ships besides characters -- `phoneme` (V = 41) and `viseme` (V = 15), avsr/misc/*_list through io_utils.py:354-370.

* widths K in {1, 4, 5, 9, 10, 16}: K > 4 reaches the second trip of the per-wave query loop and K > 8 the second 8-query block
  (and its reused LDS partial buffer) of attn_fwd_beam_kernel; K = 16 > V = 15 selects a -inf candidate at step 0;
* the K-hypotheses-per-workgroup attention kernel against the general per-hypothesis kernel: bit-identical ids AND scores;
* D = 768 memories (the G == 1 store path of attn_fwd_beam_kernel with idle threads);
* width 10 at the benchmark widths and lengths (256 units, T_a = 500, T_v = 75).

Every step's (word, parent) selections of all K beams are compared with the fp64 oracle up to the first step at which the ORACLE
saw two distinct candidate scores within 2e-5 among its best K + 1 (fp32 rounding may order those either way); utterances without
such a step must agree in all kept beams, lengths and accumulated log-probabilities (_beam_check).  Round 5: the comparison no
longer stops there -- the oracle then FOLLOWS the engine's search (_follow_check): at every step the engine's selections must lie
within the oracle's tied set, and the final beams / lengths / log-probabilities of every utterance must be what the oracle computes
along the engine's branch; at the benchmark size >= 90 % of all (step, utterance) selections must be identical outright.
"vs CPU restatement of TF-1.13.1; TF parity unpinned"."""
import dataclasses

import numpy as np
import pytest
import torch

from test_gpu_model import make

pytestmark = pytest.mark.gpu

VOCABS = {"viseme": dict(vocab_size=15, eos_id=13, go_id=14), "character": dict(vocab_size=31, eos_id=29, go_id=30),
          "phoneme": dict(vocab_size=41, eos_id=39, go_id=40)}


def _trained(O, ocfg, W, batch, eos_bias=1.0, scale=10.0, seed=5):
    """One train step off the initial point, then a SHARP output layer (kernel x scale, unit-variance bias): with the near-uniform
    distributions of random weights almost every step has two candidates within fp32 noise of each other and nothing could be
    compared strictly."""
    r = O.train_step(W, None, ocfg, batch)
    W2 = {k: v.copy() for k, v in r["params"].items()}
    rng = np.random.default_rng(seed)
    W2["dec/out/kernel"] = (W2["dec/out/kernel"] * scale).astype(np.float32)
    W2["dec/out/bias"] = rng.standard_normal(W2["dec/out/bias"].shape).astype(np.float32)
    W2["dec/out/bias"][ocfg.eos_id] += eos_bias                # EOS reachable within a few steps, not immediately
    return W2


TIE = 2e-5       # candidate scores (log-probability / length penalty) closer than this may be ordered either way by fp32 arithmetic


def _beam_check(O, ocfg, mcfg, W2, batch, K, max_steps, what, check_every=3, min_strict=0.7, min_same=0.8):
    """Engine vs oracle, step by step: the per-step (word, parent) selections of every utterance must be identical up to the first
    step at which the ORACLE saw two distinct scores within TIE among its best K + 1 candidates (from there on fp32 rounding may
    legitimately follow another branch); utterances without such a step must agree in everything: all kept beams after gather_tree,
    lengths, and accumulated log-probabilities to 1e-4.  Returns the fraction of (utterance, step) pairs compared strictly."""
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    ref, lp, ln, tr = O.beam_search_decode(W2, ocfg, batch, beam_width=K, max_steps=max_steps, return_trace=True)
    model = Seq2SeqModel(mcfg, weights=W2)
    out = model.beam_search_decode(Batch.from_numpy(batch), beam_width=K, max_steps=max_steps, check_every=check_every, return_all=True).cpu().numpy()
    assert not model.check_persistent()
    X = model._beam_ws[2]
    B = ref.shape[0]
    T = tr["step_ids"].shape[0]
    sid = X["sid"].cpu().numpy().reshape(-1, B, K)
    pid = X["pid"].cpu().numpy().reshape(-1, B, K)
    near = tr["gaps"] < TIE                                       # [T, B]
    strict = 0
    clean = []
    for b in range(B):
        hit = np.nonzero(near[:, b])[0]
        t_ok = int(hit[0]) if len(hit) else T
        strict += t_ok
        assert (sid[:t_ok, b] == tr["step_ids"][:t_ok, b]).all(), (what, "word ids", b, t_ok)
        assert (pid[:t_ok, b] == tr["parent_ids"][:t_ok, b]).all(), (what, "parents", b, t_ok)
        if not len(hit):
            clean.append(b)
    if clean:
        assert out.shape == ref.shape, (what, out.shape, ref.shape)
        assert (out[clean] == ref[clean]).all(), what
        par = out.shape[1] & 1                                    # state after the last step lives at this parity
        assert (X["ln"][par].cpu().numpy().reshape(B, K)[clean] == ln[clean]).all(), what
        glp = X["logp"][par].cpu().numpy().reshape(B, K)[clean]
        fin = np.isfinite(lp[clean])
        assert (np.isfinite(glp) == fin).all() and np.abs(glp[fin] - lp[clean][fin]).max() < 1e-4, what
    frac = strict / float(B * T)
    assert frac >= min_strict, (what, frac)
    _follow_check(O, ocfg, W2, batch, K, max_steps, what, out, sid, pid, X, min_same)
    return frac


def _follow_check(O, ocfg, W2, batch, K, max_steps, what, out, sid, pid, X, min_same):
    """EVERY step of the engine's search, not only those before its first near-tie: the fp64 oracle FOLLOWS the
    engine's selections (oracle beam_search_decode(follow=...)).  At every (step, utterance) it scores all K * V candidates from the
    state the engine's own branch leads to; the engine's j-th selection must score within TIE (relative to max(1, |score|)) of the
    oracle's j-th best -- i.e. be the oracle's choice or one of the candidates tied with it --, the K selections must be distinct,
    and the search then continues from the ENGINE's selection.  Both searches must stop at the same step, and at the end the
    engine's kept beams, lengths and accumulated log-probabilities of ALL utterances must equal what the oracle computed along that
    branch.  Returns (and bounds from below) the fraction of (step, utterance) pairs where the selections were identical."""
    T = out.shape[1]
    B = out.shape[0]
    ref, lp, ln, tr = O.beam_search_decode(W2, ocfg, batch, beam_width=K, max_steps=max_steps, return_trace=True, follow=(sid[:T], pid[:T]))
    assert not tr["follow_short"] and tr["step_ids"].shape[0] == T, (what, "the searches stop at different steps", T, tr["step_ids"].shape[0])
    assert tr["follow_distinct"].all(), (what, "duplicate candidates", np.argwhere(~tr["follow_distinct"])[:4])
    worst = float(tr["follow_dev"].max())
    assert worst < TIE, (what, "a selection outside the oracle's tied set", worst, np.argwhere(tr["follow_dev"] >= TIE)[:4])
    assert (out == ref).all(), what
    par = T & 1
    assert (X["ln"][par].cpu().numpy().reshape(B, K) == ln).all(), what
    glp = X["logp"][par].cpu().numpy().reshape(B, K)
    fin = np.isfinite(lp)
    assert (np.isfinite(glp) == fin).all() and np.abs(glp[fin] - lp[fin]).max() < 1e-4 * max(1.0, np.abs(lp[fin]).max()), what
    same = float(tr["follow_same"].mean())
    assert same >= min_same, (what, "identical selections", same)
    return same


@pytest.mark.parametrize("case", ["c1_audio_uni_luong", "c4_bimodal_uni", "c2_audio_bi_bahdanau", "c5_av_align"])
@pytest.mark.parametrize("K", [1, 4, 5, 9, 10, 16])
def test_beam_width_parity(case, K):
    O, ocfg, mcfg, W, batch = make(case, B=5, Ta=70 if K >= 9 else 21, Tv=9)     # T_a = 70: two 64-frame chunks per utterance
    for eos_bias in ((0.0, 1.0, 3.0) if K >= 9 else (0.0, 3.0)):   # 3.0: every beam finishes early (the all-finished stop)
        _beam_check(O, ocfg, mcfg, _trained(O, ocfg, W, batch, eos_bias), batch, K, 14, (case, K, eos_bias))


@pytest.mark.parametrize("setting", [0, 2])
@pytest.mark.parametrize("case", ["c4_bimodal_uni", "c2_audio_bi_bahdanau"])
def test_beam_parity_with_the_general_kernels(case, setting):
    """avsr_attn_rnn_set_beam_kernel(0 / 2): the small-tile step kernel for the dense steps (what GRU / multi-layer decoders still
    take) with the general / the beam-shaped attention kernel -- the same oracle check as the default path above."""
    from avsr_tf1_amd import ops
    O, ocfg, mcfg, W, batch = make(case, B=5, Ta=70, Tv=9)
    try:
        ops.attn_rnn_set_beam_kernel(setting)
        _beam_check(O, ocfg, mcfg, _trained(O, ocfg, W, batch, 1.0), batch, 10, 14, (case, setting))
    finally:
        ops.attn_rnn_set_beam_kernel(1)


@pytest.mark.parametrize("unit", ["viseme", "phoneme"])
@pytest.mark.parametrize("case", ["c1_audio_uni_luong", "c4_bimodal_uni", "c2_audio_bi_bahdanau", "c5_av_align", "gru_audio_uni"])
def test_vocabulary_sizes_train_greedy_beam(case, unit):
    """vocab_size 15 / 41: the logits split, sample kernel, sequence loss, embedding table (and at V > 32 the path taken) change."""
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    O, ocfg, mcfg, W, batch = make(case, B=6, use_dropout=True, sampling_probability=0.25, **VOCABS[unit])
    assert batch.labels.max() == ocfg.eos_id and batch.labels.max() < ocfg.vocab_size
    r1 = O.train_step(W, None, ocfg, batch)
    r2 = O.train_step(r1["params"], r1["opt"], ocfg, batch)
    model = Seq2SeqModel(mcfg, weights=W)
    db = Batch.from_numpy(batch)
    logits = model.forward_train(db)
    torch.cuda.synchronize()
    consumed = np.arange(batch.labels.shape[1])[None, :] < batch.labels_len[:, None]      # scheduled-sampling draws over V classes
    assert (model._cur[0]["dec"]["fed"].cpu().numpy()[consumed] == r1["fed_tokens"][consumed]).all()
    model.backward()
    model.apply_update()
    torch.cuda.synchronize()
    assert np.abs(logits.cpu().numpy() - r1["logits"]).max() < 1e-4
    assert abs(float(model.loss.item()) - r1["loss"]) < 1e-4
    assert abs(float(model.gnorm.item()) - r1["global_norm"]) < 1e-4 * max(1.0, r1["global_norm"])
    grads = model.export_tf_weights("grads")
    for k, g in r1["grads"].items():
        scale = max(1e-3, np.abs(g).max())
        assert np.abs(grads[k] - g).max() < 2e-4 * scale + 1e-6, k
    loss2, _ = model.train_step(db)
    torch.cuda.synchronize()
    assert abs(float(loss2.item()) - r2["loss"]) < 3e-4
    assert not model.check_persistent()
    # evaluation graph on the twice-updated weights
    P2 = r2["params"]
    ids_ref, lg_ref = O.greedy_decode(P2, ocfg, batch, max_steps=9, return_logits=True)
    m2 = Seq2SeqModel(mcfg, weights=P2)
    ids = m2.greedy_decode(db, max_steps=9).cpu().numpy()
    assert ids.shape == ids_ref.shape and (ids == ids_ref).all()
    for K in (4, 10, 16):                                       # K = 16 > V = 15: a -inf candidate is selected at step 0
        _beam_check(O, ocfg, mcfg, _trained(O, ocfg, P2, batch), batch, K, 10, (case, unit, K), check_every=4)


@pytest.mark.parametrize("unit", ["viseme", "phoneme"])
def test_vocabulary_sizes_full_width_fused_block(unit):
    """The benchmark block (bimodal, 256 units, B = 64) with the other vocabularies: the fused persistent decode kernels where they
    take the shape, the per-step launches where they decline (both must match the oracle)."""
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    over = dict(video_units=(256,), audio_units=(256, 256), decoder_units=(256,), embedding_size=128, video_feat=128, audio_feat=80,
                use_dropout=True, sampling_probability=0.1, **VOCABS[unit])
    O, ocfg, mcfg, W, batch = make("c4_bimodal_uni", B=64, Ta=26, Tv=9, L=6, **over)
    ref = O.train_step(W, None, ocfg, batch)
    model = Seq2SeqModel(mcfg, weights=W)
    db = Batch.from_numpy(batch)
    logits = model.forward_train(db)
    model.backward()
    model.apply_update()
    torch.cuda.synchronize()
    assert not model.check_persistent()
    assert np.abs(logits.cpu().numpy() - ref["logits"]).max() < 1e-4
    assert abs(float(model.loss.item()) - ref["loss"]) < 1e-4
    assert abs(float(model.gnorm.item()) - ref["global_norm"]) < 1e-4 * max(1.0, ref["global_norm"])
    grads = model.export_tf_weights("grads")
    for k, g in ref["grads"].items():
        scale = max(1e-3, np.abs(g).max())
        assert np.abs(grads[k] - g).max() < 2e-4 * scale + 1e-6, k
    ids_ref = O.greedy_decode(ref["params"], ocfg, batch, max_steps=6)
    assert (model.greedy_decode(db, max_steps=6).cpu().numpy() == ids_ref).all()


@pytest.mark.parametrize("case,K", [("c1_audio_uni_luong", 10), ("c4_bimodal_uni", 10), ("c4_bimodal_uni", 16), ("c5_av_align", 5),
                                    ("audio_uni3_luong", 9)])
def test_beam_attention_kernel_equals_general_kernel(case, K):
    """attn_fwd_beam_kernel (one workgroup per (utterance, chunk) serving the K hypotheses) against attn_fwd_kernel on tiled queries:
    same arithmetic and summation order per hypothesis => the ids of ALL beams, their accumulated log-probabilities, lengths and every
    step's logits are bit-identical."""
    from avsr_tf1_amd import ops
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    O, ocfg, mcfg, W, batch = make(case, B=6, Ta=150, Tv=70)
    W2 = _trained(O, ocfg, W, batch)
    db = Batch.from_numpy(batch)
    res = []
    try:
        for on in (2, 0):                                        # 2: beam attention kernel, dense steps as in 0
            ops.attn_rnn_set_beam_kernel(on)
            m = Seq2SeqModel(mcfg, weights=W2)
            out = m.beam_search_decode(db, beam_width=K, max_steps=12, check_every=5, return_all=True)
            D, T = m._last_beam
            X = m._beam_ws[2]
            torch.cuda.synchronize()
            res.append((out.cpu().numpy().copy(), X["logp"].cpu().numpy().copy(), X["ln"].cpu().numpy().copy(),
                        D["logits"][:, :T].cpu().numpy().copy(), T))
    finally:
        ops.attn_rnn_set_beam_kernel(1)
    a, b = res
    assert a[4] == b[4]
    assert (a[0] == b[0]).all()
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])           # scores: bit for bit (-inf == -inf)
    assert np.array_equal(a[3], b[3])


def test_beam_search_over_768_wide_memories():
    """D = 768 (a 384-unit bidirectional encoder): 192 float4 columns -> one row group, 64 idle threads in attn_fwd_beam_kernel's
    context phase (they used to store zeros over the real partial contexts: ADVICE r3)."""
    from avsr_tf1_amd import ops
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    O, ocfg, mcfg, W, batch = make("c1_audio_uni_luong", B=3, Ta=80, encoder_type="bidirectional", audio_units=(384,))
    W2 = _trained(O, ocfg, W, batch)
    db = Batch.from_numpy(batch)
    outs = []
    try:
        for on in (2, 0):
            ops.attn_rnn_set_beam_kernel(on)
            for _rep in range(3):                                # the race was nondeterministic
                outs.append(Seq2SeqModel(mcfg, weights=W2).beam_search_decode(db, beam_width=10, max_steps=10, check_every=4, return_all=True).cpu().numpy())
            _beam_check(O, ocfg, mcfg, W2, batch, 10, 10, ("D768", on))
    finally:
        ops.attn_rnn_set_beam_kernel(1)
    for o in outs:
        assert (o == outs[0]).all()


FULL = dict(video_units=(256,), audio_units=(256, 256, 256), decoder_units=(256,), embedding_size=128, video_feat=128, audio_feat=80)


@pytest.mark.parametrize("case,over", [("c4_bimodal_uni", FULL),
                                       ("audio_uni3_luong", dict(audio_units=(256, 256, 256), decoder_units=(256,), embedding_size=128, audio_feat=80,
                                                                 attention_type=(("scaled_luong",), ("scaled_luong",))))])
def test_beam_width_10_at_benchmark_widths_and_lengths(case, over):
    """Width 10 (the reference default and the only width bench.py times) at 256 units, T_a = 500, T_v = 75, up to 40 steps:
    bimodal with length penalty 0.5, unimodal with 0.6; all ten kept beams of both utterances compared."""
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    O, ocfg, mcfg, W, batch = make(case, B=2, Ta=500, Tv=75, L=40, ragged=True, **over)
    frac = _beam_check(O, ocfg, mcfg, _trained(O, ocfg, W, batch), batch, 10, 40, case, check_every=8, min_strict=0.5, min_same=0.9)
    print("compared strictly up to the first near-tie:", frac, "of the (utterance, step) pairs; every step checked along the engine's branch")


@pytest.mark.parametrize("K", [10, 16])
@pytest.mark.parametrize("unit", ["viseme", "character", "phoneme"])
def test_output_layer_inside_the_beam_step(unit, K):
    """At 256 units the dense beam path runs the decoder's Dense(vocab) inside beam_step_kernel<NCT> (csrc/attn_rnn.hip: two 16-column
    MFMA tiles for V <= 32, four for V <= 64; O = 512 attention outputs split over the four waves).  Oracle parity step by step, and
    the logits it leaves behind against those of the separate projection launch (setting 2: general step kernels), to rounding."""
    from avsr_tf1_amd import ops
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    over = dict(video_units=(256,), audio_units=(256, 256), decoder_units=(256,), embedding_size=128, video_feat=128, audio_feat=80,
                **VOCABS[unit])
    O, ocfg, mcfg, W, batch = make("c4_bimodal_uni", B=3, Ta=60, Tv=20, L=12, ragged=True, **over)
    W2 = _trained(O, ocfg, W, batch)
    frac = _beam_check(O, ocfg, mcfg, W2, batch, K, 12, (unit, K), check_every=4, min_strict=0.5)
    print("strictly compared fraction of (utterance, step) pairs:", frac)
    db = Batch.from_numpy(batch)
    first = []
    try:
        for setting in (1, 2):
            ops.attn_rnn_set_beam_kernel(setting)
            m = Seq2SeqModel(mcfg, weights=W2)
            m.beam_search_decode(db, beam_width=K, max_steps=12, check_every=4, return_all=True)
            torch.cuda.synchronize()
            first.append(m._beam_ws[1]["dec"]["logits"][:, 0].cpu().numpy().copy())     # step 0: identical inputs on both paths
    finally:
        ops.attn_rnn_set_beam_kernel(1)
    assert np.isfinite(first[0]).all() and np.abs(first[0] - first[1]).max() < 1e-4 * max(1.0, np.abs(first[1]).max())


@pytest.mark.parametrize("output_layer_inside", [False, True])
def test_hip_beam_step_replays_the_reference_trace(output_layer_inside):
    """TensorFlow's OWN BeamSearchDecoder output -- the reference's sample search avsr/visualise/00025.html (19 steps x 10 beams:
    scores, predicted ids, parent ids; tests/golden/reference_beam_trace_00025.json) -- replayed through the HIP beam step
    (avsr_beam_search_step = the selection kernel of avsr_attn_rnn_fwd mode 3): fed with logits whose log-softmax carries the trace's step
    log-probabilities (tests/beam_trace.py), the kernel must keep TensorFlow's symbols and parents in TensorFlow's order at every step,
    reach TensorFlow's printed scores to their third decimal, continue finished beams with EOS whatever their logits say, re-score them
    once with the longer length, and report zero unfinished beams exactly at step 19.  This part of the path is pinned to TF, not to the
    restatement (the same replay through the oracle: tests/test_beam_trace.py).
    output_layer_inside: the form the default evaluation path launches (beam_step_kernel<2>: logits = x . Wout^T + b computed by the step's
    own MFMA tiles) -- the table goes in as the output kernel's first K input columns, the rows' inputs are unit vectors."""
    import beam_trace as bt
    from avsr_tf1_amd import ops
    tr = bt.load()
    rec = bt.reconstruct(tr)
    K, V, eos, T = tr["beam_width"], tr["V"], tr["eos"], len(tr["steps"])
    U, L = 3, T + 2                                         # three copies of the utterance (one workgroup each), two steps past the end
    dev = torch.device("cuda:0")
    i32 = dict(dtype=torch.int32, device=dev)
    logp = torch.full((2, U * K), -float("inf"), device=dev)
    logp[0].view(U, K)[:, 0] = 0.0
    fin, length = torch.zeros(2, U * K, **i32), torch.zeros(2, U * K, **i32)
    tok, prow = torch.zeros(U * K, **i32), torch.zeros(U * K, **i32)
    step_ids, parent_ids, nun = torch.zeros(L, U * K, **i32), torch.zeros(L, U * K, **i32), torch.zeros(L, **i32)
    worst = 0.0
    for t in range(L):
        if t < T:
            lg = np.tile(bt.logits_for_step(tr, rec, t)[None], (U, 1, 1))
            lg += np.arange(U)[:, None, None] * 1.5        # log_softmax is shift-invariant: the copies must not differ
        else:
            lg = np.random.default_rng(t).normal(0.0, 3.0, (U, K, V))
        logits = torch.as_tensor(lg.reshape(U * K, V), dtype=torch.float32).to(dev)
        a, b = t & 1, (t + 1) & 1
        extra = {}
        if output_layer_inside:
            O = 256
            bias = np.random.default_rng(100 + t).normal(0.0, 1.0, V)
            wout_t = np.zeros((V, O), np.float32)
            wout_t[:, :K] = (lg[0] - bias[None]).T          # copy 0's table; the copies' shifts ride on input column K
            wout_t[:, K] = 1.0
            x = np.zeros((U * K, O), np.float32)
            x[np.arange(U * K), np.arange(U * K) % K] = 1.0
            x[:, K] = 1.5 * (np.arange(U * K) // K)
            extra = dict(x=torch.as_tensor(x).to(dev), x_stride=O, O=O, wout_t=torch.as_tensor(wout_t).to(dev),
                         bout=torch.as_tensor(bias, dtype=torch.float32).to(dev))
            want, logits = logits, torch.full_like(logits, float("nan"))
        ops.beam_search_step(logits, U, K, V, t, eos, bt.W, logp[a], fin[a], length[a], logp[b], fin[b], length[b], tok, prow,
                             step_ids, parent_ids, nun, **extra)
        if output_layer_inside and t < T:
            assert float((logits - want).abs().max()) < 1e-5          # the step wrote the logits it computed
        torch.cuda.synchronize()
        ids_t, par_t = step_ids[t].view(U, K).cpu().numpy(), parent_ids[t].view(U, K).cpu().numpy()
        if t >= T:                                          # the search is over: state handed through, EOS recorded
            assert (ids_t == eos).all() and (par_t == np.arange(K)[None]).all() and int(nun[t]) == 0
            assert torch.equal(logp[b], logp[a]) and torch.equal(length[b], length[a]) and torch.equal(fin[b], fin[a])
            continue
        st = tr["steps"][t]
        for u in range(U):
            assert ids_t[u].tolist() == st["ids"] and par_t[u].tolist() == st["parents"], (t, u, ids_t[u], st["ids"], par_t[u], st["parents"])
        assert (tok.view(U, K).cpu().numpy() == ids_t).all()
        assert (prow.view(U, K).cpu().numpy() == par_t + K * np.arange(U)[:, None]).all()
        # BeamSearchDecoderOutput.scores = accumulated log-probability / penalty(the length the continuation was scored with)
        lp_t, len_t, fin_t = (x[b].view(U, K).cpu().numpy() for x in (logp, length, fin))
        first_eos = np.array([[n == "EOS" and rec[t][k]["step_lp"] is not None for k, n in enumerate(st["names"])]] * U)
        score = lp_t / bt.penalty(len_t - first_eos.astype(np.int64))
        worst = max(worst, float(np.abs(score - np.asarray(st["score"])[None]).max()))
        assert (len_t == np.asarray([c["length"] for c in rec[t]])[None]).all() and (fin_t == np.asarray([c["fin"] for c in rec[t]])[None]).all()
        assert int(nun[t]) == U * sum(not c["fin"] for c in rec[t])
    assert worst < 2.5e-3, worst                            # three printed decimals (see tests/test_beam_trace.py)
    assert int(nun[T - 1]) == 0 and int(nun[T - 2]) > 0     # dynamic_decode stops after step 19, as TensorFlow's output did
    out = torch.zeros(U, T, K, **i32)
    final = (T + 2) & 1
    ops.beam_gather_tree(step_ids, parent_ids, length[final], out, U, K, T, eos)
    torch.cuda.synchronize()
    names = {v: k for k, v in tr["vocab"].items()}
    for u in range(U):
        best = [names[int(i)] for i in out[u, :, 0].cpu()]
        assert "".join(n for n in best if n != "EOS") == "and the next day" and best[16:] == ["EOS"] * 3
