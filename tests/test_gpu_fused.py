"""The fused persistent decode kernels (csrc/dec_persist.hip forward, csrc/dec_persist_bwd.hip BPTT: avsr/decoder_bimodal.py:241-275, avsr/decoder_unimodal.py:320-350 and the
AV-Align attentive layer avsr/encoder.py:265-290 as ONE launch per call) against the per-step launch path of the same engine and
against the CPU oracle ("vs CPU restatement; TF-1.13.1 parity unpinned").  Tolerances: records / logits / gradients 2e-4 of the
tensor's largest entry between the two engine paths (different summation orders), fed tokens and greedy ids bit-exact."""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    # (config overrides, B, T_a, T_v, L)
    "bimodal_teacher_forcing": (dict(architecture="bimodal", video_units=(32,), audio_units=(32, 32), regress_aus=True), 5, 19, 8, 6),
    "bimodal_dropout_sampling_2groups": (dict(architecture="bimodal", video_units=(32,), audio_units=(32, 32), use_dropout=True,
                                              sampling_probability=0.3), 11, 37, 9, 7),
    "unimodal_luong_h48": (dict(architecture="unimodal", video_units=None, audio_units=(48,), decoder_units=(48,), embedding_size=32,
                                attention_type=(("luong",), ("luong",)), sampling_probability=0.2), 9, 70, 0, 9),
    "av_align": (dict(architecture="av_align", video_units=(32,), audio_units=(32, 32)), 6, 23, 9, 5),
    "c4_width_64_utterances": (dict(architecture="bimodal", video_units=(256,), audio_units=(256,), decoder_units=(256,), embedding_size=128,
                                    video_feat=128, audio_feat=80, use_dropout=True, sampling_probability=0.1, regress_aus=True), 64, 60, 20, 10),
    "long_memory_quarters_of_125": (dict(architecture="unimodal", video_units=None, audio_units=(64,), decoder_units=(64,), embedding_size=16),
                                    3, 500, 0, 5),
    # memories wider than 256 (bidirectional encoders): attended through their projection values . W_ctx (model.py `proj`), which is what
    # makes these blocks fit the fused kernels; train step, records and greedy ids must still equal the oracle's plain formulation
    "wide_bi_memory_projected": (dict(architecture="unimodal", encoder_type="bidirectional", video_units=None, audio_units=(160,),
                                      decoder_units=(64,), embedding_size=16, use_dropout=True, sampling_probability=0.2), 10, 45, 0, 7),
    "wide_bi_memories_bimodal_projected": (dict(architecture="bimodal", encoder_type="bidirectional", video_units=(144,), audio_units=(160,),
                                                decoder_units=(48,), embedding_size=16, regress_aus=True), 6, 33, 11, 6),
    # Bahdanau family (attention.py:25-42; output_attention False: logits from the cell output): fused forward with the processed-query
    # phase, fused BPTT pulling d pq through the query layer
    "unimodal_bahdanau": (dict(architecture="unimodal", video_units=None, audio_units=(32, 32), attention_type=(("bahdanau",), ("bahdanau",)),
                               sampling_probability=0.25), 9, 41, 0, 8),
    "unimodal_normed_bahdanau_dropout_bi": (dict(architecture="unimodal", encoder_type="bidirectional", video_units=None, audio_units=(32,),
                                                 decoder_units=(64,), attention_type=(("normed_bahdanau",), ("normed_bahdanau",)),
                                                 use_dropout=True, sampling_probability=0.2), 11, 70, 0, 6),
    "c2_width_bahdanau": (dict(architecture="unimodal", encoder_type="bidirectional", video_units=None, audio_units=(128,), decoder_units=(256,),
                               embedding_size=128, audio_feat=80, attention_type=(("bahdanau",), ("bahdanau",)), use_dropout=True,
                               sampling_probability=0.1), 64, 130, 0, 8),
    # vocabularies other than characters (avsr/misc/phoneme_list: V = 41 -> the 64-symbol rows of dec_persist_kernel<.., V64>;
    # viseme_list: V = 15), Luong and Bahdanau families, one and two memories, scheduled sampling over V classes
    "phoneme_bimodal_sampling": (dict(architecture="bimodal", video_units=(32,), audio_units=(32, 32), use_dropout=True,
                                      sampling_probability=0.4, vocab_size=41, eos_id=39, go_id=40), 11, 37, 9, 7),
    "phoneme_bahdanau": (dict(architecture="unimodal", video_units=None, audio_units=(32, 32), attention_type=(("bahdanau",), ("bahdanau",)),
                              sampling_probability=0.3, vocab_size=41, eos_id=39, go_id=40), 9, 41, 0, 8),
    "phoneme_c4_width_64_utterances": (dict(architecture="bimodal", video_units=(256,), audio_units=(256,), decoder_units=(256,), embedding_size=128,
                                            video_feat=128, audio_feat=80, use_dropout=True, sampling_probability=0.1, vocab_size=41, eos_id=39,
                                            go_id=40), 64, 60, 20, 10),
    # the benchmark block's memories (T_a = 500, T_v = 75 at 256 units: quarters of 125 + 19 frames fill LDS to its last 2 KB) with the
    # phoneme vocabulary: declined until round 5 (the 64-wide output-kernel rows needed 2 KB more), now round4(V)-wide rows + the scores
    # in the input-row buffer
    "phoneme_c4_full_memories": (dict(architecture="bimodal", video_units=(256,), audio_units=(256,), decoder_units=(256,), embedding_size=128,
                                      video_feat=128, audio_feat=80, use_dropout=True, sampling_probability=0.1, vocab_size=41, eos_id=39,
                                      go_id=40), 2, 500, 75, 5),
    "vocab64_unimodal": (dict(architecture="unimodal", video_units=None, audio_units=(48,), decoder_units=(48,), embedding_size=32,
                              attention_type=(("luong",), ("luong",)), sampling_probability=0.3, vocab_size=64, eos_id=62, go_id=63), 9, 70, 0, 9),
    "viseme_bimodal_sampling": (dict(architecture="bimodal", video_units=(32,), audio_units=(32, 32), use_dropout=True,
                                     sampling_probability=0.4, vocab_size=15, eos_id=13, go_id=14), 11, 37, 9, 7),
}


def _setup(name):
    from avsr_tf1_amd.config import ModelConfig
    from oracle import avsr_oracle as O
    over, B, Ta, Tv, L = CASES[name]
    kw = dict(decoder_units=(32,), embedding_size=16, video_feat=12, audio_feat=20, encoder_type="unidirectional")
    kw.update(over)
    ocfg = O.OracleConfig(**kw)
    mcfg = ModelConfig(**{f.name: getattr(ocfg, f.name) for f in dataclasses.fields(ModelConfig) if hasattr(ocfg, f.name)})
    W = O.init_params(ocfg, seed=11)
    batch = O.synthetic_batch(ocfg, B=B, T_a=max(Ta, 1), T_v=max(Tv, 1), L=L, ragged=True)
    return O, ocfg, mcfg, W, batch


def _run(mcfg, W, batch, fused, greedy_steps):
    from avsr_tf1_amd import ops
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    ops.attn_rnn_set_fused(fused)
    try:
        m = Seq2SeqModel(mcfg, weights=W)
        db = Batch.from_numpy(batch)
        out = {}
        m.forward_train(db)
        D = m._cur[0]["dec"]
        out["eligible"] = ops.attn_rnn_fused_eligible(D["desc"])
        out["logits"], out["fed"] = D["logits"].clone(), D["fed"].clone()
        for i, mm in enumerate(D["mems"]):
            out["ctx%d" % i] = mm["ctx"].clone()
        m.backward()
        out["grads"] = m.grads.clone()
        m.apply_update()
        out["loss"], out["gnorm"] = m.loss.clone(), m.gnorm.clone()
        out["ids"] = m.greedy_decode(db, max_steps=greedy_steps).clone()
        torch.cuda.synchronize()
        assert not ops.rnn_persistent_error()
        return out
    finally:
        ops.attn_rnn_set_fused(True)


@pytest.mark.parametrize("name", list(CASES))
def test_fused_decode_equals_per_step_launches_and_oracle(name):
    O, ocfg, mcfg, W, batch = _setup(name)
    L = CASES[name][4]
    a = _run(mcfg, W, batch, True, L + 3)
    b = _run(mcfg, W, batch, False, L + 3)
    assert a["eligible"], "the fused persistent kernel declined a configuration it is built for"
    assert (a["fed"] == b["fed"]).all() and a["ids"].shape == b["ids"].shape and (a["ids"] == b["ids"]).all()
    for k in a:
        if k in ("eligible", "fed", "ids"):
            continue
        x, y = a[k].double(), b[k].double()
        assert float((x - y).abs().max()) <= 2e-4 * max(1e-3, float(y.abs().max())), k
    ref = O.train_step(W, None, ocfg, batch)
    assert np.abs(a["logits"].cpu().numpy() - ref["logits"]).max() < 1e-4
    assert abs(float(a["loss"].item()) - ref["loss"]) < 1e-4
    consumed = np.arange(batch.labels.shape[1])[None, :] < batch.labels_len[:, None]      # draws behind finished rows feed frozen steps
    assert (a["fed"].cpu().numpy()[consumed] == ref["fed_tokens"][consumed]).all()
    ids_ref = O.greedy_decode(ref["params"], ocfg, batch, max_steps=L + 3)
    assert (a["ids"].cpu().numpy() == ids_ref).all()


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("name", ["bimodal_dropout_sampling_2groups", "av_align", "c4_width_64_utterances", "long_memory_quarters_of_125",
                                  "unimodal_normed_bahdanau_dropout_bi", "c2_width_bahdanau", "wide_bi_memory_projected"])
def test_fused_forward_only_and_backward_only(name, mode):
    """avsr_attn_rnn_set_fused(2): fused forward, per-step BPTT; (3): per-step forward, fused BPTT (csrc/dec_persist_bwd.hip) -- the
    two kernels only share the record layouts, so each must also work on the other path's records."""
    O, ocfg, mcfg, W, batch = _setup(name)
    L = CASES[name][4]
    a = _run(mcfg, W, batch, mode, L + 3)
    b = _run(mcfg, W, batch, 0, L + 3)
    assert a["eligible"]
    assert (a["fed"] == b["fed"]).all() and (a["ids"] == b["ids"]).all()
    for k in ("logits", "grads", "loss", "gnorm"):
        x, y = a[k].double(), b[k].double()
        assert float((x - y).abs().max()) <= 2e-4 * max(1e-3, float(y.abs().max())), k
    if mode == 3:       # same forward path: the gradients differ only by the backward kernels' summation order
        x, y = a["grads"].double(), b["grads"].double()
        assert float((x - y).abs().max()) <= 2e-5 * max(1e-3, float(y.abs().max()))


def test_benchmark_decoders_take_the_fused_path():
    """c4 (dual attention, B=64, T_a=500, T_v=75) and c5 (decoder over the audio memory + the attentive layer over the video memory,
    B=128) at the benchmark shapes: both blocks are run by the fused kernel, in all three modes (teacher forcing, scheduled sampling
    with dropout, greedy)."""
    from avsr_tf1_amd import ops
    from avsr_tf1_amd.config import ModelConfig
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    rng = np.random.default_rng(0)
    for arch, B, kw in (("bimodal", 64, dict(use_dropout=True, sampling_probability=0.1)), ("bimodal", 64, {}), ("av_align", 128, dict(use_dropout=True))):
        cfg = ModelConfig(architecture=arch, video_units=(256,), audio_units=(256, 256, 256), video_feat=128, audio_feat=80, **kw)
        m = Seq2SeqModel(cfg, seed=1)
        f = lambda *s: torch.tensor(rng.standard_normal(s), dtype=torch.float32).cuda()
        lab = torch.tensor(rng.integers(1, 28, (B, 40)), dtype=torch.int32).cuda()
        b = Batch(audio=f(B, 500, 80), audio_len=torch.full((B,), 500, dtype=torch.int32).cuda(), video=f(B, 75, 128),
                  video_len=torch.full((B,), 75, dtype=torch.int32).cuda(), labels=lab, labels_len=torch.full((B,), 40, dtype=torch.int32).cuda())
        m.forward_train(b)
        ws = m._cur[0]
        assert ops.attn_rnn_fused_eligible(ws["dec"]["desc"])
        if arch == "av_align":
            assert ops.attn_rnn_fused_eligible(ws["enc"]["audio"]["blk"]["desc"])
        ids = m.greedy_decode(b, max_steps=16)
        torch.cuda.synchronize()
        assert ids.shape[0] == B and not ops.rnn_persistent_error()


def test_a_flagged_pass_is_redone_through_the_per_step_launches():
    """The safety net under every persistent kernel (encoders, fused decoder forward and BPTT): a bounded device-side wait that
    expires raises the sticky error word, the pass is invalid, and the trainer / the decode wrappers redo it with per-step launches.
    Here the word is raised by hand before the step, so every persistent kernel of the pass bails out of its waits and leaves garbage."""
    import warnings
    from avsr_tf1_amd import ops
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    from avsr_tf1_amd.parallel import DataParallelTrainer
    O, ocfg, mcfg, W, batch = _setup("bimodal_dropout_sampling_2groups")
    L = CASES["bimodal_dropout_sampling_2groups"][4]
    ref = _run(mcfg, W, batch, 0, L + 3)                                 # per-step launches throughout
    ops.attn_rnn_set_fused(1)
    m = Seq2SeqModel(mcfg, weights=W)
    assert m.persistent_rnn and m.fused_decode
    db = Batch.from_numpy(batch)
    t = DataParallelTrainer(m, None, use_graph=False, check_every_step=True)
    ops._persist_sync[:1].fill_(1)                                        # "a wait expired"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        loss, gnorm = t.train_step(db)
    torch.cuda.synchronize()
    assert not m.persistent_rnn and not m.fused_decode and not ops.rnn_persistent_error()
    assert abs(float(loss.item()) - float(ref["loss"].item())) <= 2e-4 * max(1.0, abs(float(ref["loss"].item())))
    assert abs(float(gnorm.item()) - float(ref["gnorm"].item())) <= 2e-4 * max(1.0, abs(float(ref["gnorm"].item())))
    # decode wrapper: same story on a fresh engine
    m2 = Seq2SeqModel(mcfg, weights=W)
    m2.forward_train(db); m2.backward(); m2.apply_update()               # same parameters as `ref` had when it decoded
    torch.cuda.synchronize()
    assert not ops.rnn_persistent_error()
    ops._persist_sync[:1].fill_(1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ids = m2.greedy_decode(db, max_steps=L + 3)
    torch.cuda.synchronize()
    assert (ids == ref["ids"]).all() and not ops.rnn_persistent_error()
