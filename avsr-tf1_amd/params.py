"""Parameter inventory, initialisers and TF-layout <-> engine-layout conversion (host side, numpy).

Variable set and shapes follow the reference graph (cells.py:14-18, encoder.py:124-141/:174-176,
attention.py:26-72 mechanisms + AttentionWrapper attention_layer, decoder_unimodal.py:80-91/:112,
decoder_bimodal.py:482).  TF stores `LSTMCell.kernel` as [in+H, 4H] with gate blocks i, j, f, o;
the engine interleaves gates per unit (column u*4+g) so a 16-column MFMA tile holds 4 whole units."""
import math
from collections import OrderedDict

import numpy as np

from .config import BAHDANAU_TYPES, ModelConfig

NON_TRAINABLE = ("moving_mean", "moving_variance")


def lstm_kernel_to_engine(W):
    K, H4 = W.shape
    H = H4 // 4
    return np.ascontiguousarray(W.reshape(K, 4, H).transpose(0, 2, 1).reshape(K, H4))


def lstm_kernel_from_engine(W):
    K, H4 = W.shape
    H = H4 // 4
    return np.ascontiguousarray(W.reshape(K, H, 4).transpose(0, 2, 1).reshape(K, H4))


def lstm_bias_to_engine(b):
    H = b.shape[0] // 4
    return np.ascontiguousarray(b.reshape(4, H).T.reshape(4 * H))


def lstm_bias_from_engine(b):
    H = b.shape[0] // 4
    return np.ascontiguousarray(b.reshape(H, 4).T.reshape(4 * H))


def _interleave(a, G):
    """columns g*H + u -> u*G + g (last axis)"""
    H = a.shape[-1] // G
    return np.ascontiguousarray(a.reshape(a.shape[:-1] + (G, H)).swapaxes(-1, -2).reshape(a.shape))


def _deinterleave(a, G):
    H = a.shape[-1] // G
    return np.ascontiguousarray(a.reshape(a.shape[:-1] + (H, G)).swapaxes(-1, -2).reshape(a.shape))


def to_engine(kind, a):
    if kind in ("lstm_kernel", "lstm_bias"):
        return _interleave(a, 4)
    if kind in ("gru_gates_kernel", "gru_gates_bias"):      # GRUCell gate kernel: blocks r, u -> per unit (r, u)
        return _interleave(a, 2)
    return np.ascontiguousarray(a)


def from_engine(kind, a):
    if kind in ("lstm_kernel", "lstm_bias"):
        return _deinterleave(a, 4)
    if kind in ("gru_gates_kernel", "gru_gates_bias"):
        return _deinterleave(a, 2)
    return np.ascontiguousarray(a)


def _layout(cfg: ModelConfig):
    """OrderedDict name -> (segments, kind, init).  `segments` lists, per axis, the widths of the logical parts the axis is a
    concatenation of (a cell kernel's rows: [input parts..., recurrent part]; its columns: one part per gate block).  The
    reference's shape of an axis is the sum of its parts; the engine's (`cfg.engine()`) pads every part separately, which is
    why the parts are kept."""
    inv = OrderedDict()
    dec = cfg.decoder_units[0]
    depth = lambda stream: [cfg.units(stream)[-1]] * (2 if cfg.encoder_type == "bidirectional" else 1)   # [fw | bw] memories
    for stream in cfg.streams():
        feat, units = cfg.feat(stream), cfg.units(stream)
        if cfg.batch_normalisation:
            inv[f"{stream}/bn/gamma"] = ([[feat]], "plain", "ones")
            inv[f"{stream}/bn/beta"] = ([[feat]], "plain", "zeros")
            inv[f"{stream}/bn/moving_mean"] = ([[feat]], "plain", "zeros")
            inv[f"{stream}/bn/moving_variance"] = ([[feat]], "plain", "ones")
        if cfg.instance_normalisation:                        # encoder.py:51-55
            inv[f"{stream}/in/gamma"] = ([[feat]], "plain", "ones")
            inv[f"{stream}/in/beta"] = ([[feat]], "plain", "zeros")
        attentive = cfg.architecture == "av_align" and stream == "audio"
        if cfg.input_dense_layers[0] > 0:                     # encoder.py:148-171: Dense(units, selu, use_bias=False)
            w_in = feat
            for k, u in enumerate(cfg.input_dense_layers):
                inv[f"{stream}/dense{k}/kernel"] = ([[w_in], [u]], "plain", "vs")
                w_in = u
        for d in cfg.directions():
            i = cfg.layer0_in(stream)
            for l, u in enumerate(units):
                extra = [units[-1]] if (attentive and l == len(units) - 1) else []
                if cfg.shared_layer(stream, l) == l:                  # encoder_weight_sharing: layers >= 2 own no variables
                    _cell(inv, cfg, f"{stream}/enc/{d}/l{l}", [i] + extra, u)
                if cfg.highway(stream) and l > 0:   # HighwayWrapper carry gate over the layer's input (cells.py:89-90)
                    inv[f"{stream}/enc/{d}/l{l}/carry_w"] = ([[i], [i]], "plain", "glorot")
                    inv[f"{stream}/enc/{d}/l{l}/carry_b"] = ([[i]], "plain", "ones")
                i = u
        if attentive:
            _attention(inv, "audio/enc/att0", cfg.attention_type[0][0], depth("video"), units[-1])
        if cfg.encoder_type == "bidirectional":
            if cfg.cell_type == "gru":                        # encoder.py:128-131
                inv[f"{stream}/enc/proj"] = ([[units[-1]] * 2, [dec]], "plain", "glorot")
            else:                                             # encoder.py:132-138
                inv[f"{stream}/enc/proj_c"] = ([[units[-1]] * 2, [dec]], "plain", "glorot")
                inv[f"{stream}/enc/proj_h"] = ([[units[-1]] * 2, [dec]], "plain", "glorot")
        if stream == "video" and cfg.regress_aus:
            inv["video/au/kernel"] = ([depth("video"), [2]], "plain", "glorot")
            inv["video/au/bias"] = ([[2]], "plain", "zeros")
    if cfg.video_units is not None and cfg.video_processing == "resnet_cnn":
        from .cnn import param_shapes
        init_of = {"conv_kernel": "conv_vs", "bias": "zeros", "gamma": "ones", "beta": "zeros", "moving_mean": "zeros", "moving_variance": "ones"}
        for name, shape, role in param_shapes(cfg.video_hw, cfg.cnn_filters, cfg.cnn_dense_units):
            inv["video/cnn/" + name] = ([[n] for n in shape], "plain", init_of[role])
    V, E = cfg.vocab_size, cfg.emb_width()
    if not cfg.one_hot():                                     # decoder_unimodal.py:76-77: one-hot inputs own no variable
        inv["dec/embedding"] = ([[V], [E]], "plain", "emb")
    mems = cfg.decoder_memories()
    _cell(inv, cfg, "dec/l0", [E] + [dec] * len(mems), dec)
    for j in range(1, len(cfg.decoder_units)):                # MultiRNNCell: layer j consumes layer j-1's output
        _cell(inv, cfg, "dec/l%d" % j, [cfg.decoder_units[j - 1]], cfg.decoder_units[j])
    for i, (stream, t) in enumerate(mems):
        _attention(inv, f"dec/att{i}", t, depth(stream), dec)
    O = [dec] * len(mems) if cfg.output_attention() else [dec]
    inv["dec/out/kernel"] = ([O, [V]], "plain", "glorot")
    inv["dec/out/bias"] = ([[V]], "plain", "zeros")
    if cfg.architecture == "bimodal":
        inv["dec/state_proj"] = ([[dec, dec], [dec]], "plain", "glorot")
    return inv


def inventory(cfg: ModelConfig):
    """OrderedDict name -> (shape, kind, init) for every variable of the model, in a fixed order.
    kind: lstm_kernel | lstm_bias | plain;  init: vs (variance scaling) | glorot | zeros | ones | emb | const:x"""
    return OrderedDict((n, (tuple(sum(ax) for ax in segs), kind, init)) for n, (segs, kind, init) in _layout(cfg).items())


def segments(cfg: ModelConfig):
    return OrderedDict((n, segs) for n, (segs, _k, _i) in _layout(cfg).items())


def _place(a, segs_from, segs_to, out_shape):
    """Copy every (part x part x ...) block of `a` (parts `segs_from` per axis) to the position the same block has in an array
    whose axes are the parts `segs_to`; the rest of the result is zero.  Padding when segs_to >= segs_from, cropping otherwise."""
    out = np.zeros(out_shape, a.dtype)
    def starts(parts):
        o, r = 0, []
        for p in parts:
            r.append(o)
            o += p
        return r
    ax = [list(zip(starts(f), starts(t), [min(x, y) for x, y in zip(f, t)])) for f, t in zip(segs_from, segs_to)]
    def rec(k, src, dst):
        if k == len(ax):
            out[tuple(dst)] = a[tuple(src)]
            return
        for so, do, n in ax[k]:
            rec(k + 1, src + [slice(so, so + n)], dst + [slice(do, do + n)])
    rec(0, [], [])
    return out


def embed(seg_tf, seg_e, a):
    """Reference-shaped array (TF layout) -> the engine configuration's shape, padding entries zero (still TF gate order)."""
    if seg_tf == seg_e:
        return a
    return _place(a, seg_tf, seg_e, tuple(sum(ax) for ax in seg_e))


def extract(seg_tf, seg_e, a):
    """Inverse of `embed`: drop the padding entries."""
    if seg_tf == seg_e:
        return a
    return _place(a, seg_e, seg_tf, tuple(sum(ax) for ax in seg_tf))


def _cell(inv, cfg, prefix, in_parts, u):
    rows = list(in_parts) + [u]
    if cfg.cell_type == "lstm":                               # cells.py:14-18
        inv[prefix + "/kernel"] = ([rows, [u] * 4], "lstm_kernel", "vs")
        inv[prefix + "/bias"] = ([[u] * 4], "lstm_bias", "zeros")
    else:                                                     # cells.py:25-29: kernel AND bias variance-scaling initialised
        inv[prefix + "/gates_kernel"] = ([rows, [u] * 2], "gru_gates_kernel", "vs")
        inv[prefix + "/gates_bias"] = ([[u] * 2], "gru_gates_bias", "vs")
        inv[prefix + "/cand_kernel"] = ([rows, [u]], "plain", "vs")
        inv[prefix + "/cand_bias"] = ([[u]], "plain", "vs")


def _attention(inv, prefix, att_type, depth, units):
    inv[prefix + "/memory_kernel"] = ([depth, [units]], "plain", "glorot")
    if att_type == "scaled_luong":
        inv[prefix + "/g"] = ([[1]], "plain", "const:1.0")
    if att_type in BAHDANAU_TYPES:
        inv[prefix + "/query_kernel"] = ([[units], [units]], "plain", "glorot")
        inv[prefix + "/v"] = ([[units]], "plain", "glorot_vec")
        if att_type == "normed_bahdanau":
            inv[prefix + "/g"] = ([[1]], "plain", "const:%r" % math.sqrt(1.0 / units))
            inv[prefix + "/b"] = ([[units]], "plain", "zeros")
    inv[prefix + "/layer_kernel"] = ([[units] + depth, [units]], "plain", "glorot")


def is_cnn_l2(name):
    """conv2d kernel_regularizer l2(0.001) of the lip CNN (video.py:26; summed at seq2seq.py:180-184)."""
    return name.startswith("video/cnn/") and name.endswith("/kernel")


def is_dense_l2(name):
    """input Dense layers' l2(0.0001) (encoder.py:164): summed into the loss only on the CNN branch of seq2seq.py:180-184."""
    return "/dense" in name and name.endswith("/kernel")


def is_l2(name):
    """seq2seq.py:283-290: variables whose name contains 'lstm_' and not 'bias' = the RNN cell kernels."""
    return name.endswith(("/kernel", "/gates_kernel", "/cand_kernel")) and \
        ("/enc/fw/" in name or "/enc/bw/" in name or name.startswith("dec/l"))


def initialise(cfg: ModelConfig, seed=0):
    """Reference initialisers (TF layout): variance-scaling truncated normal for LSTM kernels (cells.py:17),
    glorot-uniform Dense kernels, zero biases, uniform +-1.732/V embedding (decoder_unimodal.py:80-83)."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, (shape, _kind, init) in inventory(cfg).items():
        if init == "vs":
            std = math.sqrt(1.0 / shape[0]) / 0.87962566103423978
            x = rng.standard_normal(shape)
            bad = np.abs(x) > 2.0
            while bad.any():
                x[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(x) > 2.0
            a = x * std
        elif init == "conv_vs":                 # variance_scaling(scale=2.0, fan_in) on [kh, kw, cin, cout] (video.py:24)
            std = math.sqrt(2.0 / (shape[0] * shape[1] * shape[2])) / 0.87962566103423978
            x = rng.standard_normal(shape)
            bad = np.abs(x) > 2.0
            while bad.any():
                x[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(x) > 2.0
            a = x * std
        elif init == "glorot":
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            a = rng.uniform(-lim, lim, shape)
        elif init == "glorot_vec":              # TF-1.13 _compute_fans on a rank-1 shape: fan_in = fan_out = n  (attention_v, attention.py:25-42)
            lim = math.sqrt(6.0 / (shape[0] + shape[0]))
            a = rng.uniform(-lim, lim, shape)
        elif init == "emb":
            lim = 1.732 / shape[0]
            a = rng.uniform(-lim, lim, shape)
        elif init == "ones":
            a = np.ones(shape)
        elif init == "zeros":
            a = np.zeros(shape)
        elif init.startswith("const:"):
            a = np.full(shape, float(init[6:]))
        else:
            raise ValueError(init)
        out[name] = a.astype(np.float32)
    return out
