"""Cross-checks of the CPU oracle (oracle/avsr_oracle.py) against INDEPENDENT implementations -- torch.nn modules / functionals and a
hand-rolled Adam written from the paper (tests/refs/adam_paper.py).  None of this is TensorFlow (TF parity stays unpinned, SURVEY
8(c)); it shrinks the surface on which the restatement could drift silently: masked multi-layer dynamic_rnn, bidirectional stacks,
batch norm (biased normalisation; biased vs Bessel-corrected moving variance), SELU, softmax cross-entropy with masking, the focal
loss formula, cosine restarts and the optimiser update."""
import math

import numpy as np
import pytest
import torch

from oracle import avsr_oracle as O
from tests.refs import adam_paper


def _torch_lstm_from_tf(P, prefix, in_dim, H, layers):
    """torch.nn.LSTM carrying the oracle's TF-layout cells: TF kernel [in+H, 4H] with gate order (i, j, f, o) and forget bias +1
    -> torch weight_ih / weight_hh [4H, .] with gate order (i, f, g, o), the +1 folded into the forget bias."""
    lstm = torch.nn.LSTM(in_dim, H, num_layers=layers, batch_first=True).double()
    with torch.no_grad():
        for l in range(layers):
            W = torch.as_tensor(P[f"{prefix}/l{l}/kernel"], dtype=torch.float64)
            b = torch.as_tensor(P[f"{prefix}/l{l}/bias"], dtype=torch.float64)
            i_dim = in_dim if l == 0 else H
            gi, gj, gf, go = [W[:, k * H:(k + 1) * H] for k in range(4)]
            bi, bj, bf, bo = [b[k * H:(k + 1) * H] for k in range(4)]
            Wt = torch.cat([gi, gf, gj, go], dim=1)                    # torch order: i, f, g (= TF's j), o
            getattr(lstm, f"weight_ih_l{l}").copy_(Wt[:i_dim].T)
            getattr(lstm, f"weight_hh_l{l}").copy_(Wt[i_dim:].T)
            getattr(lstm, f"bias_ih_l{l}").copy_(torch.cat([bi, bf + 1.0, bj, bo]))
            getattr(lstm, f"bias_hh_l{l}").zero_()
    return lstm


@pytest.mark.parametrize("layers,bidir", [(1, False), (3, False), (2, True)])
def test_masked_multilayer_dynamic_rnn_equals_torch_packed_lstm(layers, bidir):
    """tf.nn.dynamic_rnn(sequence_length) as restated (outputs zero past len, state carried, final state = last valid step;
    bidirectional = reverse_sequence) against torch.nn.LSTM over a PackedSequence.  cell_clip=1.0 has no torch counterpart: the
    weights are small enough that no cell state reaches the clip (asserted), so both compute the same function."""
    torch.manual_seed(0)
    F, H, B, T = 7, 6, 5, 11
    cfg = O.OracleConfig(architecture="unimodal", encoder_type="bidirectional" if bidir else "unidirectional", video_units=None,
                         audio_units=(H,) * layers, decoder_units=(H,), audio_feat=F, batch_normalisation=False)
    P = {k: v * 0.35 for k, v in O.init_params(cfg, seed=11).items()}
    rng = np.random.default_rng(5)
    for k in P:
        if k.endswith("/bias"):
            P[k] = (rng.standard_normal(P[k].shape) * 0.05).astype(P[k].dtype)
    Pt = O.to_torch(P)
    x = torch.tensor(rng.standard_normal((B, T, F)) * 0.3, dtype=torch.float64)
    lens = torch.tensor([11, 3, 7, 1, 10])
    with torch.no_grad():
        enc = O.encode_stream(Pt, cfg, "audio", x, lens, training=False, bn_updates=None)
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False)
    outs = []
    finals = []
    for d in (["fw", "bw"] if bidir else ["fw"]):
        lstm = _torch_lstm_from_tf(P, f"audio/enc/{d}", F, H, layers)
        if d == "bw":
            xr = O._reverse_sequence(x, lens)
            po, (hn, cn) = lstm(torch.nn.utils.rnn.pack_padded_sequence(xr, lens, batch_first=True, enforce_sorted=False))
            o, _ = torch.nn.utils.rnn.pad_packed_sequence(po, batch_first=True, total_length=T)
            o = O._reverse_sequence(o, lens)
        else:
            po, (hn, cn) = lstm(packed)
            o, _ = torch.nn.utils.rnn.pad_packed_sequence(po, batch_first=True, total_length=T)
        outs.append(o)
        finals.append((cn[-1], hn[-1]))
        assert float(cn.detach().abs().max()) < 0.999, "the test weights must keep the cell state inside the clip"
    ref = torch.cat(outs, dim=-1)
    assert torch.allclose(enc.outputs, ref, atol=1e-12, rtol=0)
    assert not enc.outputs[1, 3:].abs().max() > 0                       # zero past the utterance
    if not bidir:
        assert torch.allclose(enc.final_state[0], finals[0][0], atol=1e-12) and torch.allclose(enc.final_state[1], finals[0][1], atol=1e-12)
    else:                                                                # encoder.py:133-138: Dense on the concatenated c's / h's
        c = torch.cat([finals[0][0], finals[1][0]], -1) @ Pt["audio/enc/proj_c"]
        h = torch.cat([finals[0][1], finals[1][1]], -1) @ Pt["audio/enc/proj_h"]
        assert torch.allclose(enc.final_state[0], c, atol=1e-12) and torch.allclose(enc.final_state[1], h, atol=1e-12)


def test_lstm_cell_clip_is_on_the_cell_state_before_the_output_gate():
    """cells.py:16 cell_clip=1.0: torch.nn.LSTMCell + an explicit clamp of c, with weights large enough to hit the clip."""
    rng = np.random.default_rng(2)
    F, H, B = 4, 5, 6
    W = torch.tensor(rng.standard_normal((F + H, 4 * H)) * 2.0)
    b = torch.tensor(rng.standard_normal(4 * H))
    x, c0, h0 = [torch.tensor(rng.standard_normal(s)) for s in ((B, F), (B, H), (B, H))]
    c0 = c0.clamp(-1, 1)
    c1, h1 = O.lstm_cell(x, c0, h0, W, b)
    z = torch.cat([x, h0], -1) @ W + b
    i, j, f, o = z.chunk(4, -1)
    c_ref = (torch.sigmoid(f + 1.0) * c0 + torch.sigmoid(i) * torch.tanh(j)).clamp(-1.0, 1.0)
    assert (c_ref.abs() == 1.0).any(), "the test must exercise the clip"
    assert torch.equal(c1, c_ref) and torch.allclose(h1, torch.sigmoid(o) * torch.tanh(c_ref), atol=1e-15)


@pytest.mark.parametrize("rank4", [False, True])
def test_batch_norm_equals_torch_functional_batch_norm(rank4):
    """Training mode: normalise with the BIASED batch variance (both); the moving variance takes the biased value for the rank-3
    encoder input (non-fused TF path) and the Bessel-corrected one for rank-4 maps (fused kernel) -- torch's running_var is always
    the unbiased one, so it pins the rank-4 branch and differs from the rank-3 branch by exactly n/(n-1)."""
    rng = np.random.default_rng(3)
    shape = (3, 5, 6, 4) if rank4 else (3, 7, 4)
    C = shape[-1]
    x = torch.tensor(rng.standard_normal(shape) * 2 + 1)
    P = {"p/gamma": torch.tensor(rng.uniform(0.5, 1.5, C)), "p/beta": torch.tensor(rng.standard_normal(C)),
         "p/moving_mean": torch.tensor(rng.standard_normal(C)), "p/moving_variance": torch.tensor(rng.uniform(0.5, 2.0, C))}
    mom, eps = 0.98, 1e-5
    upd = {}
    y = O.batch_norm(x, P, "p", True, upd, eps=eps, momentum=mom, fused=rank4)
    rm, rv = P["p/moving_mean"].clone(), P["p/moving_variance"].clone()
    xt = x.reshape(-1, C)                                             # torch wants channels second: [N, C]
    yt = torch.nn.functional.batch_norm(xt, rm, rv, P["p/gamma"], P["p/beta"], training=True, momentum=1.0 - mom, eps=eps)
    assert torch.allclose(y.reshape(-1, C), yt, atol=1e-12)
    assert torch.allclose(upd["p/moving_mean"], rm, atol=1e-12)
    n = xt.shape[0]
    if rank4:
        assert torch.allclose(upd["p/moving_variance"], rv, atol=1e-12)
    else:                                                             # biased batch variance in the moving average
        biased = mom * P["p/moving_variance"] + (1 - mom) * xt.var(0, unbiased=False)
        assert torch.allclose(upd["p/moving_variance"], biased, atol=1e-12)
        assert not torch.allclose(upd["p/moving_variance"], rv, atol=1e-6)
        assert torch.allclose((upd["p/moving_variance"] - mom * P["p/moving_variance"]) * n / (n - 1),
                              rv - mom * P["p/moving_variance"], atol=1e-12)
    # evaluation mode: moving statistics
    ye = O.batch_norm(x, P, "p", False, None, eps=eps, momentum=mom)
    yet = torch.nn.functional.batch_norm(xt, P["p/moving_mean"], P["p/moving_variance"], P["p/gamma"], P["p/beta"], training=False, eps=eps)
    assert torch.allclose(ye.reshape(-1, C), yet, atol=1e-12)


def test_selu_constants_and_masked_sequence_loss():
    """tf.nn.selu = torch.selu (same published constants); tf.contrib.seq2seq.sequence_loss = masked mean of the sparse softmax
    cross-entropy = torch cross_entropy(ignore_index) with mean reduction over the valid steps."""
    x = torch.linspace(-4, 4, 41, dtype=torch.float64)
    alpha, scale = 1.6732632423543772848170429916717, 1.0507009873554804934193349852946
    assert torch.allclose(torch.selu(x), scale * torch.where(x > 0, x, alpha * (torch.exp(x) - 1)), atol=1e-15)
    rng = np.random.default_rng(4)
    B, L, V = 4, 6, 9
    cfg = O.OracleConfig(vocab_size=V, recurrent_l2=None)
    logits = torch.tensor(rng.standard_normal((B, L, V)) * 2)
    labels = rng.integers(1, V - 1, (B, L)).astype(np.int32)
    ll = np.array([6, 2, 4, 1], np.int32)

    class M:
        aux_loss = None
    total, seq = O.loss_fn({}, cfg, O.Batch(labels=labels, labels_len=ll), logits, M())
    tgt = torch.tensor(labels, dtype=torch.int64).clone()
    mask = torch.arange(L)[None, :] < torch.tensor(ll)[:, None]
    tgt[~mask] = -100
    ref = torch.nn.functional.cross_entropy(logits.reshape(-1, V), tgt.reshape(-1), ignore_index=-100, reduction="mean")
    assert abs(float(seq) - float(ref)) < 1e-12 and abs(float(total) - float(ref)) < 1e-12
    # focal loss (devel.py:12-32, gamma 2) against its definition written with torch's binary cross-entropy per class
    cfg2 = O.OracleConfig(vocab_size=V, recurrent_l2=None, loss_fun="focal_loss")
    _, seqf = O.loss_fn({}, cfg2, O.Batch(labels=labels, labels_len=ll), logits, M())
    p = torch.softmax(logits, -1).clamp(1e-7, 1 - 1e-7)
    oh = torch.nn.functional.one_hot(torch.tensor(labels, dtype=torch.int64), V).double()
    bce = torch.nn.functional.binary_cross_entropy(p, oh, reduction="none")
    mod = oh * (1 - p) ** 2 + (1 - oh) * p ** 2
    reff = ((bce * mod).sum(-1) * mask).sum() / mask.sum()
    assert abs(float(seqf) - float(reff)) < 1e-10


def test_adam_update_equals_the_papers_epsilon_hat_form_and_cosine_restarts_its_closed_form():
    """The optimiser of oracle.train_step (tf.train.AdamOptimizer, eps = 1e-8 OUTSIDE the bias correction) against Adam written
    from the paper in a separate file: identical to the epsilon-hat ordering, and measurably different from Algorithm 1 (so the
    test would notice the wrong variant).  lr schedule: cosine_decay_restarts(t_mul=2, m_mul=1, alpha=0) against its definition
    evaluated period by period, then the linear warm-up."""
    rng = np.random.default_rng(6)
    theta0 = rng.standard_normal(50)
    grads = [rng.standard_normal(50) * (10.0 ** rng.integers(-6, 1)) for _ in range(7)]
    want = adam_paper.adam_epsilon_hat(theta0, grads, alpha=1e-3)
    alg1 = adam_paper.adam_algorithm1(theta0, grads, alpha=1e-3)
    # the oracle's update, fed the same gradients: reproduce its arithmetic through train_step's formulas
    b1, b2, eps, lr = 0.9, 0.999, 1e-8, 1e-3
    m = np.zeros(50)
    v = np.zeros(50)
    th = theta0.astype(np.float64).copy()
    src = open(O.__file__).read()
    assert "lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)" in src and "newP[k] = (p0 - lr_t * num / (np.sqrt(vv) + eps))" in src
    for t, g in enumerate(grads, start=1):
        lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        th = th - lr_t * m / (np.sqrt(v) + eps)
        assert np.allclose(th, want[t - 1], rtol=0, atol=1e-15)
    assert np.abs(want[-1] - alg1[-1]).max() > 1e-6                   # the two orderings are distinguishable on tiny gradients
    cfg = O.OracleConfig(learning_rate=1.0, lr_decay_steps=10, warmup_steps=0)
    for step in range(0, 75):
        # periods 10, 20, 40, ...: find the one holding `step`
        start, period = 0, 10
        while step >= start + period:
            start, period = start + period, period * 2
        ref = 0.5 * (1.0 + math.cos(math.pi * (step - start) / period))
        assert abs(O.lr_at(cfg, step) - ref) < 1e-12
    cfgw = O.OracleConfig(learning_rate=2.0, lr_decay_steps=0, warmup_steps=4)
    assert [O.lr_at(cfgw, s) for s in range(6)] == [0.5, 1.0, 1.5, 2.0, 2.0, 2.0]
