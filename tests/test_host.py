"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, ctypes struct layouts match the
header, weight-layout conversion, parameter inventory, config validation and the CER/WER metric (pinned against
the real reference's avsr/utils.py via tests/golden/reference_cer_wer.json)."""
import ctypes
import dataclasses
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_header_symbol():
    from avsr_tf1_amd import _lib
    lib = _lib.load()                              # builds with hipcc (cross-compiles gfx950 without a GPU) if needed
    assert lib.avsr_abi_version() == 2
    hdr = open(os.path.join(ROOT, "include", "avsr_hip.h")).read()
    declared = sorted(set(re.findall(r"^\s*(?:int|int64_t)\s+(avsr_\w+)\s*\(", hdr, flags=re.M)))
    assert len(declared) >= 20
    for sym in declared:
        assert hasattr(lib, sym), "header declares %s but the library does not export it" % sym
        assert sym in _lib.EXPORTS, "%s missing from the ctypes binding" % sym
    # ... and nothing else: the library is built with -fvisibility=hidden, the header's declarations are its whole dynamic symbol table
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.lib_path()]).decode()
    syms = [l.split() for l in out.splitlines() if len(l.split()) == 3]
    exported = sorted(n for _a, k, n in syms if k == "T")          # strong functions (kernel handles are data objects, "D")
    assert not [n for _a, k, n in syms if k == "W" and "avsr" in n]   # weak ones: libstdc++ template instances only
    assert exported == declared, (sorted(set(exported) - set(declared)), sorted(set(declared) - set(exported)))
    assert sorted(_lib.EXPORTS) == declared


def test_beam_search_step_rejects_bad_arguments_before_any_launch():
    """avsr_beam_search_step (include/avsr_hip.h): argument errors are decided on the host -- no GPU needed, nothing is dereferenced."""
    from avsr_tf1_amd import _lib
    lib = _lib.load()
    p = 64                                          # any non-NULL address: the checks below return before a launch could read it

    def call(logits=p, n_utt=2, K=10, V=31, step=0, eos=29, x=None, x_stride=0, O=0, wout=None, bout=None, state=p):
        return lib.avsr_beam_search_step(logits, n_utt, K, V, step, eos, 0.6, state, p, p, p, p, p, p, p, p, p, p, x, x_stride, O, wout, bout, None)

    assert call(logits=None) == -1 and call(state=None) == -1          # AVSR_ERR_ARG
    assert call(n_utt=0) == -1 and call(K=0) == -1 and call(step=-1) == -1 and call(eos=31) == -1
    assert call(K=40, V=31) == -3                                       # AVSR_ERR_UNSUPPORTED: beam_width * V > 1024
    assert call(x=p, x_stride=256, O=256) == -1                         # output layer inside the step: kernel and bias are required
    assert call(x=p, x_stride=256, O=200, wout=p, bout=p) == -3         # ... and O % 256 == 0
    assert call(x=p, x_stride=256, O=256, wout=p, bout=p, K=20, V=31) == -3    # ... beam_width <= 16


def test_io_helper_builds_loads_and_exports_every_header_symbol():
    """include/avsr_io.h / libavsr_io.so: the native record indexer + batch filler of the input pipeline (plain C99)."""
    import subprocess
    import tempfile
    from avsr_tf1_amd import _io_native as N
    lib = N.load()
    assert lib is not None and lib.avsr_io_abi_version() == 2
    hdr = open(os.path.join(ROOT, "include", "avsr_io.h")).read()
    declared = sorted(set(re.findall(r"^\s*int\s+(avsr_io_\w+)\s*\(", hdr, flags=re.M)))
    assert declared == ["avsr_io_abi_version", "avsr_io_fill_f32", "avsr_io_fill_labels", "avsr_io_index"]
    for sym in declared:
        assert hasattr(lib, sym)
    with tempfile.TemporaryDirectory() as d:       # the record struct has the 16 int64 fields the binding indexes
        src = os.path.join(d, "p.c")
        open(src, "w").write('#include "avsr_io.h"\nint main(void){return sizeof(avsr_io_rec) == 16 * 8 ? 0 : 1;}\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", src + ".exe"])
        assert subprocess.call([src + ".exe"]) == 0
    assert N.NFIELD == 16 and len(N.F) == 16


def test_ctypes_structs_match_c_layout():
    from avsr_tf1_amd import _lib
    lib = _lib.load()
    for name, st in _lib._STRUCTS.items():
        assert lib.avsr_sizeof(name.encode()) == ctypes.sizeof(st), name
    assert lib.avsr_sizeof(b"no_such_struct") == -1


def test_header_is_plain_c():
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "p.c")
        open(src, "w").write('#include "avsr_hip.h"\nint main(void){return (int)sizeof(avsr_attn_rnn) == 0;}\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", src, "-o", src + ".o"])


def test_lstm_layout_roundtrip_and_semantics():
    from avsr_tf1_amd import params as PR
    rng = np.random.default_rng(0)
    K, H = 6, 4
    W = rng.standard_normal((K, 4 * H)).astype(np.float32)
    b = rng.standard_normal(4 * H).astype(np.float32)
    We, be = PR.lstm_kernel_to_engine(W), PR.lstm_bias_to_engine(b)
    for g in range(4):
        for u in range(H):
            assert np.all(We[:, u * 4 + g] == W[:, g * H + u]) and be[u * 4 + g] == b[g * H + u]
    assert np.array_equal(PR.lstm_kernel_from_engine(We), W) and np.array_equal(PR.lstm_bias_from_engine(be), b)


CONFIGS = [
    dict(architecture="unimodal", video_units=None, audio_units=(8, 8)),
    dict(architecture="unimodal", encoder_type="bidirectional", video_units=(8,), audio_units=None, regress_aus=True,
         attention_type=(("normed_bahdanau",), ("normed_bahdanau",))),
    dict(architecture="bimodal", video_units=(8,), audio_units=(8, 8), regress_aus=True),
    dict(architecture="bimodal", encoder_type="bidirectional", video_units=(8,), audio_units=(8,), attention_type=(("bahdanau",), ("luong",))),
    dict(architecture="av_align", video_units=(8,), audio_units=(8, 8)),
    dict(architecture="av_align", video_units=(8,), audio_units=(8,), attention_type=(("bahdanau",), ("scaled_luong",))),
    dict(architecture="unimodal", video_units=None, audio_units=(8, 8), cell_type="gru"),
    dict(architecture="unimodal", encoder_type="bidirectional", video_units=None, audio_units=(8,), cell_type="gru"),
    dict(architecture="av_align", video_units=(8,), audio_units=(8, 8), cell_type="gru"),
]


@pytest.mark.parametrize("kw", CONFIGS)
def test_inventory_matches_oracle_variables(kw):
    """The engine's variable set / shapes are exactly the oracle's (so weights are interchangeable by name)."""
    from avsr_tf1_amd import params as PR
    from avsr_tf1_amd.config import ModelConfig
    from oracle import avsr_oracle as O
    base = dict(decoder_units=(8,), embedding_size=4, video_feat=4, audio_feat=8)
    ocfg = O.OracleConfig(**base, **kw)
    mcfg = ModelConfig(**{f.name: getattr(ocfg, f.name) for f in dataclasses.fields(ModelConfig) if hasattr(ocfg, f.name)})
    W = O.init_params(ocfg)
    inv = PR.inventory(mcfg)
    assert set(inv) == set(W)
    for k, (shape, _kind, _init) in inv.items():
        assert tuple(W[k].shape) == tuple(shape), k
    assert sorted(k for k in inv if PR.is_l2(k)) == sorted(O.l2_names(W, ocfg))
    init = PR.initialise(mcfg, seed=1)
    assert all(init[k].shape == tuple(inv[k][0]) and init[k].dtype == np.float32 for k in inv)
    lstm = [k for k in inv if inv[k][1] == "lstm_kernel"]
    assert all(np.abs(init[k]).max() <= 2.0 * np.sqrt(1.0 / init[k].shape[0]) / 0.8796 + 1e-6 for k in lstm)   # truncated at 2 sigma
    assert mcfg.output_attention() == ocfg.output_attention() and mcfg.decoder_memories() == ocfg.decoder_memories()


ODD_CONFIGS = [
    dict(architecture="unimodal", video_units=None, audio_units=(7, 5), decoder_units=(5,), embedding_size=3, audio_feat=13),
    dict(architecture="bimodal", encoder_type="bidirectional", video_units=(6,), audio_units=(7, 9), decoder_units=(10,), embedding_size=5,
         video_feat=5, audio_feat=11, regress_aus=True, attention_type=(("normed_bahdanau",), ("bahdanau",))),
    dict(architecture="av_align", video_units=(6,), audio_units=(7, 9), decoder_units=(9,), embedding_size=6, video_feat=5, audio_feat=11,
         input_dense_layers=(7,)),
    dict(architecture="unimodal", encoder_type="bidirectional", video_units=None, audio_units=(7,), decoder_units=(6,), embedding_size=5,
         audio_feat=9, cell_type="gru", instance_normalisation=True, attention_type=(("scaled_luong",), ("scaled_luong",))),
    dict(architecture="unimodal", video_units=None, audio_units=(6, 6, 6), decoder_units=(6, 6), embedding_size=0, audio_feat=9,
         highway_encoder=True),
]


@pytest.mark.parametrize("kw", ODD_CONFIGS)
def test_padding_to_engine_widths_computes_the_same_model(kw):
    """`ModelConfig.engine()` + `params.embed`: the 4-padded model with zero padding entries IS the unpadded model.  Shown here with an
    independent implementation - the CPU oracle run on the padded configuration / weights / inputs gives the same logits and loss,
    gradients that crop to the unpadded ones, and exactly-zero gradients in every padding entry."""
    from avsr_tf1_amd import params as PR
    from avsr_tf1_amd.config import ModelConfig
    from oracle import avsr_oracle as O
    ocfg = O.OracleConfig(**kw)
    mcfg = ModelConfig(**{f.name: getattr(ocfg, f.name) for f in dataclasses.fields(ModelConfig) if hasattr(ocfg, f.name)})
    mcfg.validate()
    ecfg = mcfg.engine()
    assert all(d % 4 == 0 for d in [ecfg.embedding_size, *ecfg.decoder_units, *(ecfg.audio_units or ()), *(ecfg.video_units or ()),
                                    ecfg.audio_feat, ecfg.video_feat])
    ocfg_e = dataclasses.replace(ocfg, **{f.name: getattr(ecfg, f.name) for f in dataclasses.fields(ModelConfig)
                                          if hasattr(ocfg, f.name) and f.name != "embedding_size"},
                                 embedding_size=ecfg.embedding_size)
    W = O.init_params(ocfg, seed=3)
    rng = np.random.default_rng(1)
    for k in W:
        if k.endswith(("bias", "/b", "beta")):
            W[k] = (rng.standard_normal(W[k].shape) * 0.1).astype(np.float32)
    st, se = PR.segments(mcfg), PR.segments(ecfg)
    We = {k: PR.embed(st[k], se[k], W[k]) for k in W}
    inv_e = PR.inventory(ecfg)
    assert all(We[k].shape == tuple(inv_e[k][0]) for k in We)
    assert all(np.array_equal(PR.extract(st[k], se[k], We[k]), W[k]) for k in W)
    if mcfg.one_hot():                      # the padded oracle needs the table as a (here: trainable) variable
        We["dec/embedding"] = np.eye(mcfg.vocab_size, ecfg.embedding_size, dtype=np.float32)
    batch = O.synthetic_batch(ocfg, B=3, T_a=9, T_v=5, L=4, ragged=True)
    pad = lambda x, F: None if x is None else np.concatenate([x, np.zeros(x.shape[:2] + (F - x.shape[2],), x.dtype)], axis=2)
    batch_e = dataclasses.replace(batch, audio=pad(batch.audio, ecfg.audio_feat), video=pad(batch.video, ecfg.video_feat))
    r, re_ = O.train_step(W, None, ocfg, batch), O.train_step(We, None, ocfg_e, batch_e)
    assert np.abs(r["logits"] - re_["logits"]).max() < 1e-6 and abs(r["loss"] - re_["loss"]) < 1e-6
    assert abs(r["global_norm"] - re_["global_norm"]) < 1e-6 or mcfg.one_hot()
    for k, g in r["grads"].items():
        ge = re_["grads"][k]
        assert np.abs(PR.extract(st[k], se[k], ge) - g).max() < 1e-6 * max(1.0, np.abs(g).max()), k
        assert np.array_equal(PR.embed(st[k], se[k], PR.extract(st[k], se[k], ge)), ge), k       # padding entries: exactly zero gradients


def test_config_validation_errors_follow_the_reference():
    from avsr_tf1_amd.config import ModelConfig
    with pytest.raises(Exception, match="Unknown architecture"):
        ModelConfig(architecture="trimodal").validate()
    with pytest.raises(Exception, match="Allowed encoder types"):
        ModelConfig(encoder_type="sideways").validate()
    with pytest.raises(Exception, match="cell type not supported"):
        ModelConfig(cell_type="nas").validate()
    with pytest.raises(Exception, match="unknown attention mechanism"):
        ModelConfig(attention_type=(("luong",), ("dot",))).validate()
    with pytest.raises(ValueError):
        ModelConfig(architecture="av_align", encoder_type="bidirectional", video_units=(8,)).validate()
    with pytest.raises(ValueError):
        ModelConfig(architecture="bimodal", video_units=(256,), cell_type="gru").validate()   # decoder_bimodal.py:130-142
    ModelConfig(cell_type="gru").validate()
    ModelConfig(architecture="bimodal", video_units=(256,)).validate()
    # the non-default options built this round, and the combinations the reference's graph construction would reject
    with pytest.raises(ValueError, match="Unknown loss function"):
        ModelConfig(loss_fun="hinge").validate()                                          # seq2seq.py:163
    with pytest.raises(Exception, match="Unsupported optimiser"):
        ModelConfig(optimiser="SGD").validate()                                           # seq2seq.py:218
    with pytest.raises(ValueError, match="residual_encoder needs equal layer widths"):
        ModelConfig(residual_encoder=True, audio_units=(128, 256)).validate()
    # the reference hands residual / highway / weight sharing to build_rnn_layers only from the unidirectional Seq2SeqEncoder branch
    # (encoder.py:67-78); bidirectional stacks (:92-108) and the AV-Align audio stack (:225-233) ignore them -> accepted, inert
    for inert in (dict(architecture="av_align", video_units=(64,), audio_units=(64, 128), residual_encoder=True),
                  dict(architecture="av_align", video_units=(64,), audio_units=(64, 128, 64), encoder_weight_sharing=True),
                  dict(encoder_type="bidirectional", audio_units=(128, 256), residual_encoder=True),
                  dict(encoder_type="bidirectional", audio_units=(128, 256, 64), highway_encoder=True, encoder_weight_sharing=True),
                  dict(encoder_type="bidirectional", audio_units=(128, 256), cell_type="gru", highway_encoder=True)):
        c = ModelConfig(**inert)
        c.validate()
        assert not any(c.highway(s) or c.residual(s) or c.shared_layer(s, 2) != 2 for s in ("audio",))
    c = ModelConfig(architecture="av_align", video_units=(64, 64, 64), audio_units=(64, 128), residual_encoder=True, encoder_weight_sharing=True)
    c.validate()
    assert c.residual("video") and not c.residual("audio") and c.shared_layer("video", 2) == 1 and c.shared_layer("audio", 2) == 2
    with pytest.raises(ValueError, match="encoder_weight_sharing"):
        ModelConfig(encoder_weight_sharing=True, audio_units=(128, 256, 256)).validate()
    with pytest.raises(NotImplementedError):
        ModelConfig(decoder_units=(256, 128)).validate()                                   # multi-layer decoders: equal widths only
    ModelConfig(decoder_units=(256, 256), cell_type="gru").validate()                      # GRU multi-layer decoders: built in round 6
    with pytest.raises(ValueError, match="no encoders"):
        ModelConfig(architecture="lm").validate()                                          # default audio_units is set
    for ok in (dict(architecture="lm", video_units=None, audio_units=None), dict(decoder_units=(256, 256, 256)),
               dict(residual_encoder=True), dict(encoder_weight_sharing=True), dict(instance_normalisation=True),
               dict(optimiser="AdamW"), dict(loss_fun="focal_loss"), dict(label_smoothing=0.1), dict(lr_decay_steps=1000),
               dict(input_dense_layers=(128, 64))):
        ModelConfig(**ok).validate()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "avsr-tf1_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "import oracle" not in src and "from oracle" not in src, f


def test_model_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from avsr_tf1_amd.config import ModelConfig
    from avsr_tf1_amd.model import Seq2SeqModel
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Seq2SeqModel(ModelConfig())


def test_cer_wer_against_reference_outputs():
    from avsr_tf1_amd import utils
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_cer_wer.json")))
    for c in g["levenshtein"]:
        assert utils.levenshtein(c["a"], c["b"]) == c["d"]
    cer, per = utils.compute_wer(g["predictions"], g["truth"], split_words=False)
    assert abs(cer - g["cer"]) < 1e-12 and all(abs(per[k] - v) < 1e-12 for k, v in g["cer_per_file"].items())
    wer, per = utils.compute_wer(g["predictions"], g["truth"], split_words=True)
    assert abs(wer - g["wer"]) < 1e-12 and all(abs(per[k] - v) < 1e-12 for k, v in g["wer_per_file"].items())


def test_alignment_image_and_png_writer(tmp_path):
    """tf.summary.image rendering of 1 - alpha (per-image max -> 255, truncation) and a PNG any decoder can read back."""
    import struct
    import zlib
    from avsr_tf1_amd.utils import alignment_image, write_png_gray
    rng = np.random.default_rng(3)
    alpha = rng.random((5, 7)).astype(np.float32)
    alpha /= alpha.sum(-1, keepdims=True)
    alpha[3:] = 0.0                                            # steps after EOS
    img = alignment_image(alpha)
    assert img.shape == (7, 5) and img.dtype == np.uint8 and img.max() == 255 and (img[:, 3:] == 255).all()
    want = np.floor((1 - alpha.T) * (255.0 / (1 - alpha.T).max())).astype(np.uint8)
    assert np.abs(img.astype(int) - want.astype(int)).max() <= 1
    assert (alignment_image(np.ones((2, 1), np.float32)) == 0).all()      # an all-zero image stays zero (max < 1e-6)
    f = tmp_path / "a.png"
    write_png_gray(str(f), img)
    raw = f.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(raw):
        n, tag = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xFFFFFFFF
        chunks[tag] = data
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (w, h, depth, ctype) == (5, 7, 8, 0)
    rows = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), np.uint8).reshape(h, w + 1)
    assert (rows[:, 0] == 0).all() and (rows[:, 1:] == img).all()


def test_integration_doc_lists_every_entry_point():
    """INTEGRATION.md section 4 maps each C entry point declared in include/avsr_hip.h to the reference call site it replaces."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "avsr_hip.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = set(re.findall(r"^(?:int|int32_t|int64_t|const char\*|void)\s+(avsr_[a-z0-9_]+)\(", hdr, flags=re.M))
    assert len(names) > 40
    missing = sorted(n for n in names if not re.search(r"\b%s\b" % n, doc))
    assert not missing, missing
