"""CPU oracle for the AVSR seq2seq hot path -- TEST INFRASTRUCTURE ONLY.

**Parity unpinned.**  This file is a CPU restatement (torch-CPU tensors, autograd for
BPTT) of the TensorFlow-1.13.1 semantics that the reference's hot path selects.  The
reference (georgesterpu/avsr-tf1) is pure Python driving `tensorflow==1.13.1`; it cannot
be imported here (no TensorFlow), has no tests and no golden vectors (SURVEY.md section 0,
F1-F3).  The *wiring* below follows the reference files line by line (cited per
function); the *arithmetic* follows the published TF r1.13 algorithms
(`rnn_cell_impl.LSTMCell/GRUCell`, `rnn.dynamic_rnn`,
`contrib.seq2seq.{LuongAttention,BahdanauAttention,AttentionWrapper,BasicDecoder,
GreedyEmbeddingHelper,TrainingHelper,dynamic_decode,sequence_loss}`, `layers.batch_normalization`,
`train.AdamOptimizer`, `clip_by_global_norm`) as summarised in SURVEY.md Appendix A.
Nothing here has been checked against a running TensorFlow.  Two pieces ARE pinned to outputs of the reference: the beam-search
bookkeeping (`beam_candidates` / `beam_advance`: finished-beam masking, length penalty and its exponent, which length a step is
scored with, top-k order, end of the search) replays TensorFlow's own BeamSearchDecoder output for the reference's sample utterance
avsr/visualise/00025.html (tests/test_beam_trace.py; three printed decimals, 190 nodes), and CER / WER equal the reference's own
avsr/utils.py functions executed here (tests/golden/reference_cer_wer.json).  Everything else stays unpinned.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module.  The product (`avsr_tf1_amd`) never does.

Weight naming / layout ("TF layout"): every kernel is `[in, out]` exactly as TF stores it
(`LSTMCell.kernel [in+H, 4H]`, gate blocks ordered i, j, f, o).  The HIP engine uses a
different internal layout; `avsr_tf1_amd.params` converts.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor

LUONG_TYPES = ("luong", "scaled_luong")
BAHDANAU_TYPES = ("bahdanau", "normed_bahdanau")


# ----------------------------------------------------------------------------------------
# configuration (mirrors the hparams the hot path reads: avsr/avsr.py:150-198)
# ----------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    architecture: str = "unimodal"            # unimodal | bimodal | av_align   (seq2seq.py:47,55,96,114)
    encoder_type: str = "unidirectional"      # unidirectional | bidirectional (encoder.py:67,90)
    cell_type: str = "lstm"                   # lstm | gru                       (cells.py:13,24)
    video_units: Optional[Tuple[int, ...]] = None   # encoder_units_per_layer[0]; None = no video stream
    audio_units: Optional[Tuple[int, ...]] = (256, 256, 256)  # encoder_units_per_layer[1]
    decoder_units: Tuple[int, ...] = (256,)
    attention_type: Tuple[Tuple[str, ...], Tuple[str, ...]] = (("scaled_luong",), ("scaled_luong",))
    enable_attention: bool = True
    embedding_size: int = 128
    vocab_size: int = 31                      # len(unit_dict) - 1 (decoder_unimodal.py:49)
    go_id: int = 30
    eos_id: int = 29
    video_feat: int = 128
    audio_feat: int = 80
    batch_normalisation: bool = True          # avsr.py:36
    regress_aus: bool = False                 # encoder.py:34, :173
    au_loss_weight: float = 10.0              # seq2seq.py:189
    recurrent_l2: Optional[float] = 1e-4      # avsr.py:44
    clip_gradients: bool = True
    max_gradient_norm: float = 1.0
    learning_rate: float = 1e-3
    warmup_steps: int = 750                   # seq2seq.py:275
    encoder_weight_sharing: bool = False      # avsr.py:49, cells.py:77
    instance_normalisation: bool = False      # avsr.py:37, encoder.py:51-55: contrib.layers.instance_norm after the batch norm
    residual_encoder: bool = False            # avsr.py:42, cells.py:91-92: ResidualWrapper on encoder layers > 0
    highway_encoder: bool = False             # avsr.py:41, cells.py:89-90: HighwayWrapper on encoder layers > 0 (wins over residual)
    optimiser: str = "Adam"                   # Adam | Nadam | AdamW | Momentum (seq2seq.py:195-218)
    weight_decay: float = 1e-4                # AdamW only (avsr.py:45)
    loss_fun: Optional[str] = None            # None | 'focal_loss' | 'mc_loss'  (seq2seq.py:147-163, devel.py)
    label_smoothing: float = 0.0              # avsr.py:57; > 0 switches to tf.losses.softmax_cross_entropy (seq2seq.py:151-155)
    lr_decay_steps: int = 0                   # lr_decay=('cosine_restarts', N), seq2seq.py:266-270; 0 = constant
    max_label_length: int = 150               # avsr.py:157
    # stochastic train-time features (off for parity fixtures; SURVEY section 7 "hard parts")
    use_dropout: bool = False
    video_dropout: Tuple[float, float, float] = (0.9, 0.9, 0.9)   # keep probs (in, state, out)
    audio_dropout: Tuple[float, float, float] = (0.9, 0.9, 0.9)
    decoder_dropout: Tuple[float, float, float] = (0.9, 0.9, 0.9)
    sampling_probability: float = 0.0
    # visual front-end (avsr.py:24, :33-34; video.py): "features" = the record already holds cnn_dense_units-d vectors,
    # "resnet_cnn" = lip crops [B, T, H, W, C] through video.resnet_cnn
    video_processing: str = "features"
    cnn_filters: Tuple[int, ...] = (8, 16, 32, 64)
    cnn_dense_units: int = 128
    video_hw: Tuple[int, int, int] = (36, 36, 3)
    input_dense_layers: Tuple[int, ...] = (0,)                    # avsr.py:38; encoder.py:148-171 (SELU Dense stack, no bias)

    def layer0_in(self, stream: str) -> int:
        """Width of the first encoder layer's input: the last input Dense layer if any, else the feature size."""
        if self.input_dense_layers[0] > 0:
            return self.input_dense_layers[-1]
        return self.video_feat if stream == "video" else self.audio_feat

    def streams(self) -> List[str]:
        s = []
        if self.video_units is not None:
            s.append("video")
        if self.audio_units is not None:
            s.append("audio")
        return s

    def directions(self) -> List[str]:
        return ["fw", "bw"] if self.encoder_type == "bidirectional" else ["fw"]

    def memory_depth(self, stream: str) -> int:
        units = self.video_units if stream == "video" else self.audio_units
        mult = 2 if self.encoder_type == "bidirectional" else 1
        return units[-1] * mult

    def decoder_memories(self) -> List[Tuple[str, str]]:
        """[(stream, attention_type)] in AttentionWrapper order.

        bimodal: video mechanisms first, then audio (decoder_bimodal.py:184-223).
        unimodal / av_align: audio if present else video, types = attention_type[1]
        (seq2seq.py:96-112, decoder_unimodal.py:322)."""
        if not self.enable_attention or self.architecture == "lm":
            return []
        if self.architecture == "bimodal":
            out = []
            if self.video_units is not None:
                out += [("video", t) for t in self.attention_type[0]]
            if self.audio_units is not None:
                out += [("audio", t) for t in self.attention_type[1]]
            return out
        stream = "audio" if self.audio_units is not None else "video"
        return [(stream, t) for t in self.attention_type[1]]

    def output_attention(self) -> bool:
        """output_attention flag = that of the LAST mechanism created
        (attention.py:108-117; decoder_bimodal.py:396-441 side effect)."""
        mems = self.decoder_memories()
        if not mems:
            return False
        return mems[-1][1] in LUONG_TYPES

    def wrapped(self, stream: str) -> bool:
        """Do residual_encoder / highway_encoder / encoder_weight_sharing reach this stream's stack?  Only the unidirectional
        branch of Seq2SeqEncoder hands them to build_rnn_layers (encoder.py:67-78); the bidirectional branch builds _fw_cells /
        _bw_cells without them (encoder.py:92-108) and so does AttentiveEncoder for the AV-Align audio stack (encoder.py:225-233):
        there the flags are silently inert."""
        return self.encoder_type == "unidirectional" and not (self.architecture == "av_align" and stream == "audio")

    def highway(self, stream: str) -> bool:
        return self.highway_encoder and self.wrapped(stream)

    def residual(self, stream: str) -> bool:
        return self.residual_encoder and not self.highway_encoder and self.wrapped(stream)      # cells.py:89-92: highway wins

    def shared_layer(self, stream: str, l: int) -> int:
        """cells.py:77: `layer > 1 and weight_sharing` reuses cell_list[-1] -> layers >= 2 are layer 1's cell object."""
        return 1 if (self.encoder_weight_sharing and self.wrapped(stream) and l > 1) else l

    def validate(self):
        if self.architecture == "lm":                                     # avsr.LM (lm.py:275-471): labels only
            if self.video_units is not None or self.audio_units is not None:
                raise ValueError("the language model has no encoders")
        elif self.architecture not in ("unimodal", "bimodal", "av_align"):
            raise Exception("Unknown architecture")                       # seq2seq.py:66
        if self.encoder_type not in ("unidirectional", "bidirectional"):
            raise Exception("Allowed encoder types: `unidirectional`, `bidirectional`")  # encoder.py:146
        if self.cell_type not in ("lstm", "gru"):
            raise Exception("cell type not supported: {}".format(self.cell_type))  # cells.py:44
        for types in self.attention_type:
            for t in types:
                if t not in LUONG_TYPES + BAHDANAU_TYPES:
                    raise Exception("unknown attention mechanism")        # attention.py:86
        if self.architecture == "bimodal" and self.cell_type != "lstm":
            raise ValueError("bimodal decoder requires LSTM state tuples")  # decoder_bimodal.py:130-142
        if self.architecture == "av_align":
            if self.encoder_type != "unidirectional":
                raise ValueError("AttentiveEncoder implements only unidirectional")  # encoder.py:229
            if self.video_units is None or self.audio_units is None:
                raise ValueError("av_align needs both streams")
        for st in self.streams():
            u = self.video_units if st == "video" else self.audio_units
            if self.highway(st) or self.residual(st):
                if len(set(u)) != 1:
                    raise ValueError("residual_encoder needs equal layer widths")
                if self.highway(st) and self.encoder_weight_sharing:
                    raise NotImplementedError("highway_encoder with encoder_weight_sharing")
            if self.encoder_weight_sharing and self.wrapped(st):
                if len(u) > 2 and (len(set(u[1:])) != 1 or u[0] != u[1]):
                    raise ValueError("encoder_weight_sharing needs equal layer sizes: layers >= 2 reuse layer 1's kernel")
        if len(set(self.decoder_units)) != 1:
            raise NotImplementedError("multi-layer decoders: equal layer widths only")


# ----------------------------------------------------------------------------------------
# initialisers
# ----------------------------------------------------------------------------------------
def _variance_scaling(rng: np.random.Generator, shape) -> np.ndarray:
    """tf.variance_scaling_initializer(): scale 1, fan_in, truncated normal (cells.py:17)."""
    fan_in = shape[0] if len(shape) > 1 else shape[0]
    std = math.sqrt(1.0 / max(1.0, fan_in)) / 0.87962566103423978
    x = rng.standard_normal(size=shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float32)


def _glorot_uniform(rng: np.random.Generator, shape) -> np.ndarray:
    fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (shape[0], shape[0])
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _cell_params(rng, cfg: OracleConfig, prefix: str, in_dim: int, units: int, P: Dict[str, np.ndarray]):
    if cfg.cell_type == "lstm":
        P[prefix + "/kernel"] = _variance_scaling(rng, (in_dim + units, 4 * units))
        P[prefix + "/bias"] = np.zeros((4 * units,), np.float32)
    else:  # gru: kernel AND bias use variance scaling (cells.py:26-27)
        P[prefix + "/gates_kernel"] = _variance_scaling(rng, (in_dim + units, 2 * units))
        P[prefix + "/gates_bias"] = _variance_scaling(rng, (2 * units,))
        P[prefix + "/cand_kernel"] = _variance_scaling(rng, (in_dim + units, units))
        P[prefix + "/cand_bias"] = _variance_scaling(rng, (units,))


def _attention_params(rng, prefix: str, att_type: str, depth: int, units: int, P):
    P[prefix + "/memory_kernel"] = _glorot_uniform(rng, (depth, units))       # memory_layer, no bias
    if att_type == "scaled_luong":
        P[prefix + "/g"] = np.ones((1,), np.float32)                          # attention_g init 1.0
    if att_type in BAHDANAU_TYPES:
        P[prefix + "/query_kernel"] = _glorot_uniform(rng, (units, units))    # query_layer, no bias
        P[prefix + "/v"] = _glorot_uniform(rng, (units,))                     # attention_v [num_units]: rank 1 -> fan_in = fan_out = units
        if att_type == "normed_bahdanau":
            P[prefix + "/g"] = np.full((1,), math.sqrt(1.0 / units), np.float32)
            P[prefix + "/b"] = np.zeros((units,), np.float32)
    P[prefix + "/layer_kernel"] = _glorot_uniform(rng, (units + depth, units))  # attention_layer, no bias


def cnn_layout(cfg: OracleConfig):
    """resnet_cnn (video.py:143-195) as a list of ops over NHWC maps; SAME padding as TF computes it
    (total = max((ceil(in/s) - 1) * s + k - in, 0), the odd pixel goes to the bottom / right).
    Returns (ops, out_hw) with ops = [(kind, name, dict)]."""
    H, W, C = cfg.video_hw
    f = cfg.cnn_filters
    ops = [("conv", "layer0", dict(k=3, s=1, cin=C, cout=f[0], src="in", dst="a0")),
           ("bnrelu", "layer0_bn", dict(c=f[0], src="a0", dst="b0")),
           # res_block_0: skip_bn, identity shortcut
           ("conv", "res_block_0_conv1", dict(k=3, s=1, cin=f[0], cout=f[0], src="b0", dst="r0a")),
           ("bnrelu", "res_block_0_second_bn", dict(c=f[0], src="r0a", dst="r0b")),
           ("conv", "res_block_0_conv2", dict(k=3, s=1, cin=f[0], cout=f[0], src="r0b", dst="r0c")),
           ("add", "res_block_0", dict(a="r0c", b="b0", dst="x0"))]
    prev, cin = "x0", f[0]
    for i, c in enumerate(f[1:], start=1):
        n = "res_block_%d" % i
        ops += [("bnrelu", n + "_first_bn", dict(c=cin, src=prev, dst=n + "_p")),
                ("conv", n + "_shortcut", dict(k=1, s=2, cin=cin, cout=c, src=prev, dst=n + "_s")),
                ("conv", n + "_conv1", dict(k=3, s=2, cin=cin, cout=c, src=n + "_p", dst=n + "_a")),
                ("bnrelu", n + "_second_bn", dict(c=c, src=n + "_a", dst=n + "_b")),
                ("conv", n + "_conv2", dict(k=3, s=1, cin=c, cout=c, src=n + "_b", dst=n + "_c")),
                ("add", n, dict(a=n + "_c", b=n + "_s", dst="x%d" % i))]
        prev, cin = "x%d" % i, c
        H, W = (H + 1) // 2, (W + 1) // 2
    ops.append(("flatten", "flatten", dict(kh=H, kw=W, cin=cin, cout=cfg.cnn_dense_units, src=prev, dst="out")))
    return ops, (H, W)


def _conv_init(rng, shape):
    """tf.variance_scaling_initializer(scale=2.0, mode='fan_in') on a [kh, kw, cin, cout] kernel (video.py:24)."""
    fan_in = shape[0] * shape[1] * shape[2]
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    x = rng.standard_normal(size=shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float32)


def _cnn_params(rng, cfg: OracleConfig, P):
    for kind, name, a in cnn_layout(cfg)[0]:
        if kind == "conv":
            P[f"video/cnn/{name}/kernel"] = _conv_init(rng, (a["k"], a["k"], a["cin"], a["cout"]))
            P[f"video/cnn/{name}/bias"] = np.zeros((a["cout"],), np.float32)
        elif kind == "flatten":
            P[f"video/cnn/{name}/kernel"] = _conv_init(rng, (a["kh"], a["kw"], a["cin"], a["cout"]))
            P[f"video/cnn/{name}/bias"] = np.zeros((a["cout"],), np.float32)
        elif kind == "bnrelu":
            P[f"video/cnn/{name}/gamma"] = np.ones((a["c"],), np.float32)
            P[f"video/cnn/{name}/beta"] = np.zeros((a["c"],), np.float32)
            P[f"video/cnn/{name}/moving_mean"] = np.zeros((a["c"],), np.float32)
            P[f"video/cnn/{name}/moving_variance"] = np.ones((a["c"],), np.float32)


def _same_pad(n, k, s):
    out = (n + s - 1) // s
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


# Test diagnostic: the smallest |ReLU input| the last cnn_forward saw.  The gradient is discontinuous at a ReLU input of exactly zero; an
# fp32 implementation and this fp64 restatement can land on different sides of it when an input lies within fp32 rounding (~1e-6) of
# zero -- forward values agree to 1e-7, one element's gradient contribution is there or not (found by the 4 000-seed fuzz run of round 6,
# profiles/r06_fuzz_large.txt).  Tests that compare CNN gradients read it (train_step's "relu_margin") to tell a kink from a bug.
RELU_MARGIN = [float("inf")]


def _note_relu(z: Tensor) -> Tensor:
    if z.numel():
        RELU_MARGIN[0] = min(RELU_MARGIN[0], float(z.detach().abs().min()))
    return z


def cnn_forward(P, cfg: OracleConfig, frames: Tensor, training: bool, updates: Optional[dict]) -> Tensor:
    """video.resnet_cnn on [N, H, W, C] frames -> [N, cnn_dense_units] (video.py:143-195, cnn_layers :224-233).
    Every frame goes through, zero padding frames included (the reference reshapes [B*T, H, W, C] without a mask)."""
    F = torch.nn.functional
    maps = {"in": frames.permute(0, 3, 1, 2)}                      # NCHW for torch
    for kind, name, a in cnn_layout(cfg)[0]:
        pre = f"video/cnn/{name}"
        if kind == "conv":
            x = maps[a["src"]]
            pt, pb = _same_pad(x.shape[2], a["k"], a["s"])
            pl, pr = _same_pad(x.shape[3], a["k"], a["s"])
            w = P[pre + "/kernel"].permute(3, 2, 0, 1)             # HWIO -> OIHW
            maps[a["dst"]] = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, P[pre + "/bias"], stride=a["s"])
        elif kind == "bnrelu":
            x = maps[a["src"]].permute(0, 2, 3, 1)                 # channels last for the statistics
            y = batch_norm(x, P, pre, training, updates, eps=1e-5, momentum=0.98, fused=True)   # rank 4: the fused kernel
            maps[a["dst"]] = torch.relu(_note_relu(y)).permute(0, 3, 1, 2)
        elif kind == "add":
            maps[a["dst"]] = maps[a["a"]] + maps[a["b"]]
        elif kind == "flatten":
            x = maps[a["src"]]
            w = P[pre + "/kernel"].permute(3, 2, 0, 1)
            maps[a["dst"]] = torch.relu(_note_relu(F.conv2d(x, w, P[pre + "/bias"]))).reshape(x.shape[0], -1)
    return maps["out"]


def cnn_l2_names(P) -> List[str]:
    """conv2d kernel_regularizer l2(0.001) (video.py:26), summed into the loss at seq2seq.py:180-184."""
    return [k for k in P if k.startswith("video/cnn/") and k.endswith("/kernel")]


def init_params(cfg: OracleConfig, seed: int = 2001) -> Dict[str, np.ndarray]:
    """All trainable variables + BN moving stats, TF layout, reference initialisers."""
    cfg.validate()
    rng = np.random.default_rng(seed)
    P: Dict[str, np.ndarray] = {}
    dec_units = cfg.decoder_units[0]
    for stream in cfg.streams():
        feat = cfg.video_feat if stream == "video" else cfg.audio_feat
        units = cfg.video_units if stream == "video" else cfg.audio_units
        if cfg.batch_normalisation:
            P[f"{stream}/bn/gamma"] = np.ones((feat,), np.float32)
            P[f"{stream}/bn/beta"] = np.zeros((feat,), np.float32)
            P[f"{stream}/bn/moving_mean"] = np.zeros((feat,), np.float32)
            P[f"{stream}/bn/moving_variance"] = np.ones((feat,), np.float32)
        attentive = cfg.architecture == "av_align" and stream == "audio"
        if cfg.instance_normalisation:
            P[f"{stream}/in/gamma"] = np.ones((feat,), np.float32)
            P[f"{stream}/in/beta"] = np.zeros((feat,), np.float32)
        if cfg.input_dense_layers[0] > 0:                        # Dense(units, selu, use_bias=False), variance-scaling init
            w_in = feat
            for i, u in enumerate(cfg.input_dense_layers):
                P[f"{stream}/dense{i}/kernel"] = _variance_scaling(rng, (w_in, u))
                w_in = u
        for d in cfg.directions():
            in_dim = cfg.layer0_in(stream)
            for l, u in enumerate(units):
                extra = units[-1] if (attentive and l == len(units) - 1) else 0   # + attention feedback
                if cfg.shared_layer(stream, l) == l:                               # shared layers own no variables
                    _cell_params(rng, cfg, f"{stream}/enc/{d}/l{l}", in_dim + extra, u, P)
                if cfg.highway(stream) and l > 0:                                  # HighwayWrapper's carry gate over the layer input
                    P[f"{stream}/enc/{d}/l{l}/carry_w"] = _glorot_uniform(rng, (in_dim, in_dim))
                    P[f"{stream}/enc/{d}/l{l}/carry_b"] = np.ones((in_dim,), np.float32)
                in_dim = u
        if attentive:
            _attention_params(rng, "audio/enc/att0", cfg.attention_type[0][0],
                              cfg.memory_depth("video"), units[-1], P)
        if cfg.encoder_type == "bidirectional":
            # final-state projection of the LAST layer (encoder.py:124-141); lower layers'
            # projections exist in the reference graph but never reach a 1-layer decoder.
            if cfg.cell_type == "lstm":
                P[f"{stream}/enc/proj_c"] = _glorot_uniform(rng, (2 * units[-1], dec_units))
                P[f"{stream}/enc/proj_h"] = _glorot_uniform(rng, (2 * units[-1], dec_units))
            else:
                P[f"{stream}/enc/proj"] = _glorot_uniform(rng, (2 * units[-1], dec_units))
        if stream == "video" and cfg.regress_aus:
            P["video/au/kernel"] = _glorot_uniform(rng, (cfg.memory_depth("video"), 2))
            P["video/au/bias"] = np.zeros((2,), np.float32)
    # decoder
    V, E = cfg.vocab_size, cfg.embedding_size
    lim = 1.732 / V
    if E > 0:
        P["dec/embedding"] = rng.uniform(-lim, lim, size=(V, E)).astype(np.float32)
    else:                      # decoder_unimodal.py:76-77: non-positive embedding_size -> tf.eye(vocab_size), a constant, no variable
        E = V
    mems = cfg.decoder_memories()
    att_total = dec_units * len(mems)
    _cell_params(rng, cfg, "dec/l0", E + att_total, dec_units, P)
    for j in range(1, len(cfg.decoder_units)):                  # MultiRNNCell: layer j consumes layer j-1's output
        _cell_params(rng, cfg, f"dec/l{j}", cfg.decoder_units[j - 1], cfg.decoder_units[j], P)
    for i, (stream, t) in enumerate(mems):
        _attention_params(rng, f"dec/att{i}", t, cfg.memory_depth(stream), dec_units, P)
    out_dim = att_total if cfg.output_attention() else dec_units
    P["dec/out/kernel"] = _glorot_uniform(rng, (out_dim, V))
    P["dec/out/bias"] = np.zeros((V,), np.float32)
    if cfg.architecture == "bimodal":
        P["dec/state_proj"] = _glorot_uniform(rng, (2 * dec_units, dec_units))  # decoder_bimodal.py:482
    if cfg.video_units is not None and cfg.video_processing == "resnet_cnn":
        _cnn_params(np.random.default_rng(seed + 77), cfg, P)     # own stream: the other tensors keep their values
    return P


NON_TRAINABLE = ("moving_mean", "moving_variance")


def _embedding(P, cfg):
    """decoder_unimodal.py:72-91: the embedding variable, or the constant one-hot matrix when embedding_size <= 0."""
    if "dec/embedding" in P:
        return P["dec/embedding"]
    ref = P["dec/out/kernel"]
    return torch.eye(cfg.vocab_size, dtype=ref.dtype)


def trainable_names(P: Dict[str, np.ndarray]) -> List[str]:
    return [k for k in P if not k.endswith(NON_TRAINABLE)]


def l2_names(P, cfg: OracleConfig) -> List[str]:
    """seq2seq.py:283-290: trainable vars whose name contains 'lstm_'/'gru_' and not 'bias'.
    In TF those are exactly the RNN cell kernels (encoders and decoder)."""
    out = []
    for k in trainable_names(P):
        if "bias" in k:
            continue
        if k.endswith("/kernel") and ("/enc/fw/" in k or "/enc/bw/" in k or k.startswith("dec/l")):
            out.append(k)
        if k.endswith(("/gates_kernel", "/cand_kernel")):
            out.append(k)
    return out


# ----------------------------------------------------------------------------------------
# batches (io_utils.BatchedData: io_utils.py:8-18)
# ----------------------------------------------------------------------------------------
@dataclass
class Batch:
    audio: Optional[np.ndarray] = None          # [B, T_a, F_a] float32
    audio_len: Optional[np.ndarray] = None      # [B] int32
    video: Optional[np.ndarray] = None          # [B, T_v, F_v]
    video_len: Optional[np.ndarray] = None
    aus: Optional[np.ndarray] = None            # [B, T_v, 2]
    labels: Optional[np.ndarray] = None         # [B, L] int32, EOS appended, 0 padded
    labels_len: Optional[np.ndarray] = None     # [B] int32 (includes EOS)


def synthetic_batch(cfg: OracleConfig, B: int, T_a: int = 500, T_v: int = 75, L: int = 40,
                    ragged: bool = False, seed: int = 1000) -> Batch:
    """SURVEY section 8(d) synthetic inputs (fixed seeds)."""
    b = Batch()
    r = lambda k: np.random.default_rng(seed + k)
    if cfg.audio_units is not None:
        b.audio = r(1).standard_normal((B, T_a, cfg.audio_feat)).astype(np.float32)
        b.audio_len = (r(2).integers(T_a // 2, T_a + 1, size=B) if ragged else np.full(B, T_a)).astype(np.int32)
        b.audio *= (np.arange(T_a)[None, :, None] < b.audio_len[:, None, None])   # padded_batch zero pads
    if cfg.video_units is not None:
        vshape = tuple(cfg.video_hw) if cfg.video_processing == "resnet_cnn" else (cfg.video_feat,)
        b.video = r(3).standard_normal((B, T_v) + vshape).astype(np.float32)
        b.video_len = (r(4).integers((T_v + 1) // 2, T_v + 1, size=B) if ragged else np.full(B, T_v)).astype(np.int32)
        b.video *= (np.arange(T_v)[None, :] < b.video_len[:, None]).reshape((B, T_v) + (1,) * len(vshape))
        b.aus = r(5).uniform(0.0, 3.0, size=(B, T_v, 2)).astype(np.float32)
    lab = r(6).integers(1, cfg.eos_id, size=(B, L)).astype(np.int32)
    ll = (r(7).integers(max(1, L // 2), L + 1, size=B) if ragged else np.full(B, L)).astype(np.int32)
    if ragged:
        ll[0] = L                       # padded_batch pads to the batch max, so max == L
    for i in range(B):
        lab[i, ll[i] - 1] = cfg.eos_id
        lab[i, ll[i]:] = 0
    b.labels, b.labels_len = lab, ll
    return b


# ----------------------------------------------------------------------------------------
# stateless counter RNG shared bit-for-bit with the HIP kernels (dropout / scheduled sampling)
# ----------------------------------------------------------------------------------------
def hash_u32(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """lowbias32 mix of (idx ^ key); identical integer arithmetic in csrc/common.h."""
    with np.errstate(over="ignore"):
        key = np.uint32((seed * 0x9E3779B9 + stream * 0x85EBCA6B) & 0xFFFFFFFF)
        x = idx.astype(np.uint32) ^ key
        x ^= x >> np.uint32(16)
        x = x * np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x = x * np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def uniform01(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    return (hash_u32(seed, stream, idx) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


# ----------------------------------------------------------------------------------------
# cells
# ----------------------------------------------------------------------------------------
def lstm_cell(x: Tensor, c: Tensor, h: Tensor, W: Tensor, b: Tensor):
    """tf LSTMCell(use_peepholes=False, cell_clip=1.0, forget_bias=1.0) -- cells.py:14-18."""
    z = torch.cat([x, h], dim=-1) @ W + b
    i, j, f, o = z.chunk(4, dim=-1)
    c_new = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
    c_new = torch.clamp(c_new, -1.0, 1.0)
    h_new = torch.sigmoid(o) * torch.tanh(c_new)
    return c_new, h_new


def gru_cell(x: Tensor, h: Tensor, Wg: Tensor, bg: Tensor, Wc: Tensor, bc: Tensor):
    """tf GRUCell -- cells.py:25-29."""
    ru = torch.sigmoid(torch.cat([x, h], dim=-1) @ Wg + bg)
    r, u = ru.chunk(2, dim=-1)
    cand = torch.tanh(torch.cat([x, r * h], dim=-1) @ Wc + bc)
    return u * h + (1.0 - u) * cand


# stream ids of the stateless RNG (shared with the HIP engine): cell id * 4 + kind
DROP_IN, DROP_STATE, DROP_OUT = 0, 1, 2
CELL_ID_DECODER = 40
STREAM_SS_SELECT, STREAM_SS_SAMPLE = 1000, 1001


def encoder_cell_id(stream: str, direction: str, layer: int) -> int:
    return 1 + ((0 if stream == "video" else 1) * 2 + (0 if direction == "fw" else 1)) * 8 + layer


class _Cell:
    """One RNN layer's weights + step function; state is (c, h) for LSTM, (h,) for GRU.
    With `drop` set it behaves like tf.contrib.rnn.DropoutWrapper(cell, input/state/output keep prob,
    variational_recurrent=False) (cells.py:46-54): fresh masks every step, h-state dropped, c-state not."""

    def __init__(self, P: Dict[str, Tensor], prefix: str, cell_type: str, units: int):
        self.t, self.units = cell_type, units
        self.drop = None
        if cell_type == "lstm":
            self.W, self.b = P[prefix + "/kernel"], P[prefix + "/bias"]
        else:
            self.Wg, self.bg = P[prefix + "/gates_kernel"], P[prefix + "/gates_bias"]
            self.Wc, self.bc = P[prefix + "/cand_kernel"], P[prefix + "/cand_bias"]

    def set_dropout(self, keep, cid, T, seed, lens=None, reverse=False):
        self.drop = dict(keep=tuple(float(k) for k in keep), cid=cid, T=T, seed=seed, lens=lens, reverse=reverse)

    def zero_state(self, B, dtype):
        z = torch.zeros(B, self.units, dtype=dtype)
        return (z, z) if self.t == "lstm" else (z,)

    def _mask(self, kind, width, t, B, dtype):
        d = self.drop
        keep = d["keep"][kind]
        if keep >= 1.0:
            return None
        if d["reverse"]:
            tau = torch.clamp(d["lens"] - 1 - t, min=0)
        else:
            tau = torch.full((B,), t, dtype=torch.int64)
        idx = ((torch.arange(B) * d["T"] + tau)[:, None] * width + torch.arange(width)[None, :]).numpy().astype(np.uint32)
        u = uniform01(d["seed"], d["cid"] * 4 + kind, idx)
        return torch.tensor((u < np.float32(keep)).astype(np.float64) / keep, dtype=dtype)

    def __call__(self, x, state, t=0):
        if self.drop is not None:
            m = self._mask(DROP_IN, x.shape[1], t, x.shape[0], x.dtype)
            if m is not None:
                x = x * m
        if self.t == "lstm":
            c, h = lstm_cell(x, state[0], state[1], self.W, self.b)
            out, ns = h, (c, h)
        else:
            h = gru_cell(x, state[0], self.Wg, self.bg, self.Wc, self.bc)
            out, ns = h, (h,)
        if self.drop is not None:
            ms = self._mask(DROP_STATE, self.units, t, x.shape[0], x.dtype)
            mo = self._mask(DROP_OUT, self.units, t, x.shape[0], x.dtype)
            if ms is not None:
                ns = ns[:-1] + (ns[-1] * ms,)
            if mo is not None:
                out = out * mo
        return out, ns


class _MultiCell:
    """MultiRNNCell of _Cell layers (cells.py:96-100) behind the single-cell interface; the state is the FLAT tuple
    (c0, h0, c1, h1, ...).  Each layer's DropoutWrapper draws from its own RNG stream family (cell id + layer)."""

    def __init__(self, cells):
        self.cells = cells
        self.t, self.units = cells[0].t, cells[-1].units
        self.ns = 2 if self.t == "lstm" else 1

    def set_dropout(self, keep, cid, T, seed, lens=None, reverse=False):
        for j, c in enumerate(self.cells):
            c.set_dropout(keep, cid + j, T, seed, lens, reverse)

    def zero_state(self, B, dtype):
        out = ()
        for c in self.cells:
            out += c.zero_state(B, dtype)
        return out

    def __call__(self, x, state, t=0):
        new = ()
        for j, c in enumerate(self.cells):
            x, ns = c(x, tuple(state[j * self.ns:(j + 1) * self.ns]), t)
            new += ns
        return x, new


def _select(mask: Tensor, new, old):
    return tuple(torch.where(mask, n, o) for n, o in zip(new, old))


# ----------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------
class _Mechanism:
    """contrib.seq2seq Luong/Bahdanau attention over one memory (attention.py:25-72)."""

    def __init__(self, P, prefix: str, att_type: str, memory: Tensor, memory_len: Tensor):
        B, T, D = memory.shape
        self.type = att_type
        mask = (torch.arange(T)[None, :] < memory_len[:, None])
        self.mask = mask
        self.values = memory * mask[:, :, None].to(memory.dtype)          # _prepare_memory
        self.keys = self.values @ P[prefix + "/memory_kernel"]            # memory_layer, once per batch
        self.layer = P[prefix + "/layer_kernel"]
        self.g = P.get(prefix + "/g")
        if att_type in BAHDANAU_TYPES:
            self.Wq, self.v = P[prefix + "/query_kernel"], P[prefix + "/v"]
            self.b = P.get(prefix + "/b")

    def __call__(self, query: Tensor):
        if self.type in LUONG_TYPES:
            score = torch.einsum("bth,bh->bt", self.keys, query)
            if self.type == "scaled_luong":
                score = score * self.g
        else:
            pq = query @ self.Wq
            if self.type == "normed_bahdanau":
                v = self.g * self.v / torch.sqrt(torch.sum(self.v * self.v))
                score = torch.sum(v * torch.tanh(self.keys + pq[:, None, :] + self.b), dim=-1)
            else:
                score = torch.sum(self.v * torch.tanh(self.keys + pq[:, None, :]), dim=-1)
        score = torch.where(self.mask, score, torch.full_like(score, -float("inf")))
        align = torch.softmax(score, dim=-1)
        ctx = torch.einsum("bt,btd->bd", align, self.values)
        return align, ctx


def attention_wrapper_step(cell: _Cell, mechs: List[_Mechanism], output_attention: bool,
                           x: Tensor, cell_state, attention: Tensor, t: int = 0):
    """contrib.seq2seq.AttentionWrapper.call (attention.py:173-181; decoder_bimodal.py:241-248)."""
    cell_out, new_state = cell(torch.cat([x, attention], dim=-1), cell_state, t)
    atts, aligns = [], []
    for m in mechs:
        al, ctx = m(cell_out)
        atts.append(torch.cat([cell_out, ctx], dim=-1) @ m.layer)
        aligns.append(al)
    new_att = torch.cat(atts, dim=-1)
    out = new_att if output_attention else cell_out
    return out, new_state, new_att, aligns


# ----------------------------------------------------------------------------------------
# encoders
# ----------------------------------------------------------------------------------------
def batch_norm(x: Tensor, P, prefix: str, training: bool, updates: Optional[dict], eps: float = 1e-3, momentum: float = 0.99,
               stats=None, fused: bool = False):
    """tf.layers.batch_normalization(axis=-1, fused=True), momentum .99 eps 1e-3 (encoder.py:44-50); the CNN front-end
    uses momentum .98 eps 1e-5 (video.py:8-11).  Statistics over all rows INCLUDING zero padding."""
    if training:
        flat = x.reshape(-1, x.shape[-1])
        if stats is not None:                                  # data-parallel shard: (mean, biased var, rows) of the GLOBAL batch
            mean, var, n = stats
        else:
            mean = flat.mean(dim=0)
            var = ((flat - mean) ** 2).mean(dim=0)
            n = flat.shape[0]
        if updates is not None:
            # TF 1.13 keras BatchNormalization.build: `fused=True` survives only for rank-4 inputs (the CNN maps, video.py:8-12);
            # the rank-3 [B,T,F] encoder input (encoder.py:44-50) silently falls back to the non-fused path, whose moving variance
            # takes the BIASED batch variance of tf.nn.moments.  The fused kernel feeds the Bessel-corrected one.
            unbiased = var.detach() * (n / max(1, n - 1)) if fused else var.detach()
            updates[prefix + "/moving_mean"] = momentum * P[prefix + "/moving_mean"] + (1 - momentum) * mean.detach()
            updates[prefix + "/moving_variance"] = momentum * P[prefix + "/moving_variance"] + (1 - momentum) * unbiased
    else:
        mean, var = P[prefix + "/moving_mean"], P[prefix + "/moving_variance"]
    return (x - mean) * torch.rsqrt(var + eps) * P[prefix + "/gamma"] + P[prefix + "/beta"]


def _reverse_sequence(x: Tensor, lens: Tensor) -> Tensor:
    B, T = x.shape[0], x.shape[1]
    t = torch.arange(T)[None, :]
    idx = torch.where(t < lens[:, None], lens[:, None] - 1 - t, t)
    return torch.gather(x, 1, idx[:, :, None].expand_as(x))


def dynamic_rnn(step_fn, zero_state, x: Tensor, lens: Tensor):
    """tf.nn.dynamic_rnn with sequence_length: zero output / state copy-through past len.
    step_fn(x_t, state, t) -> (out_t, state)."""
    B, T = x.shape[0], x.shape[1]
    state = zero_state
    outs = []
    for t in range(T):
        o, ns = step_fn(x[:, t], state, t)
        valid = (t < lens)[:, None]
        outs.append(torch.where(valid, o, torch.zeros_like(o)))
        state = _map_state(lambda n, s: torch.where(valid, n, s), ns, state)
    return torch.stack(outs, dim=1), state


def _map_state(fn, new, old):
    """Apply fn(new_leaf, old_leaf) over a nested tuple of tensors."""
    if isinstance(new, Tensor):
        return fn(new, old)
    return tuple(_map_state(fn, n, o) for n, o in zip(new, old))


def _wrap_output(wrap, P, prefix: str, l: int, x: Tensor, y: Tensor) -> Tensor:
    """cells.py:89-92 for layer l > 0: HighwayWrapper (tf.contrib.rnn.HighwayWrapper defaults: coupled gates, carry bias init 1.0:
    carry = sigmoid(x W_c + b_c); out = x * carry + y * (1 - carry)) takes precedence over ResidualWrapper (out = y + x).
    x is the layer's RAW input, y the (dropout-wrapped) cell's output.  (rnn_cell.py of TF r1.13, recalled.)"""
    if l == 0 or wrap is None:
        return y
    if wrap == "highway":
        carry = torch.sigmoid(x @ P[f"{prefix}/l{l}/carry_w"] + P[f"{prefix}/l{l}/carry_b"])
        return x * carry + y * (1.0 - carry)
    if wrap == "residual":
        return y + x
    return y


def _stack_step(cells: List[_Cell], wrap=None, P=None, prefix: str = ""):
    """MultiRNNCell step; wrap in (None, "highway", "residual") = the wrapper cells.py:89-92 puts around every cell but the first
    (unidirectional Seq2SeqEncoder stacks only: OracleConfig.wrapped)."""
    def step(x, states, t=0):
        new_states = []
        for l, (c, s) in enumerate(zip(cells, states)):
            y, ns = c(x, s, t)
            x = _wrap_output(wrap, P, prefix, l, x, y)
            new_states.append(ns)
        return x, tuple(new_states)
    return step


@dataclass
class EncoderOut:
    outputs: Tensor            # [B, T, D]
    final_state: tuple         # last layer's state: (c, h) or (h,)
    alignments: Optional[Tensor] = None


def _make_cells(P, cfg: OracleConfig, stream: str, direction: str, units, training: bool, seed: int, T: int, lens: Tensor):
    # encoder_weight_sharing (cells.py:77): `layer > 1` reuses the previous cell object -> layers >= 2 share layer 1's variables
    cells = [_Cell(P, f"{stream}/enc/{direction}/l{cfg.shared_layer(stream, l)}", cfg.cell_type, u) for l, u in enumerate(units)]
    if cfg.use_dropout and training:                                  # cells.py:46: only in the train graph
        keep = cfg.video_dropout if stream == "video" else cfg.audio_dropout
        for l, c in enumerate(cells):
            c.set_dropout(keep, encoder_cell_id(stream, direction, l), T, seed, lens, direction == "bw")
    return cells


def encode_stream(P, cfg: OracleConfig, stream: str, x: Tensor, lens: Tensor, training: bool,
                  bn_updates: Optional[dict], attended: Optional[Tuple[Tensor, Tensor]] = None, seed: int = 0,
                  bn_stats=None) -> EncoderOut:
    """Seq2SeqEncoder / AttentiveEncoder (encoder.py:14-196, :199-335)."""
    units = cfg.video_units if stream == "video" else cfg.audio_units
    B, dtype = x.shape[0], x.dtype
    T = x.shape[1]
    if cfg.batch_normalisation:
        x = batch_norm(x, P, f"{stream}/bn", training, bn_updates, stats=bn_stats)
    if cfg.instance_normalisation:
        # tf.contrib.layers.instance_norm(inputs) on [B,T,F] (encoder.py:51-55): moments over the time axis per (utterance, feature),
        # padding included; center + scale per feature, epsilon 1e-6 (contrib/layers/python/layers/normalization.py, recalled)
        mu = x.mean(dim=1, keepdim=True)
        var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
        x = (x - mu) * torch.rsqrt(var + 1e-6) * P[f"{stream}/in/gamma"] + P[f"{stream}/in/beta"]
    if cfg.input_dense_layers[0] > 0:                            # encoder.py:148-171, after _init_data (BN), before the RNN
        for i in range(len(cfg.input_dense_layers)):
            x = torch.selu(x @ P[f"{stream}/dense{i}/kernel"])
    if attended is not None:                                   # av_align: top layer attention-wrapped
        cells = _make_cells(P, cfg, stream, "fw", units, training, seed, T, lens)
        mech = _Mechanism(P, "audio/enc/att0", cfg.attention_type[0][0], attended[0], attended[1])
        out_att = cfg.attention_type[0][0] in LUONG_TYPES
        aligns = []

        def step(x_t, state, t=0):
            lower, (top_state, att) = state
            new_lower = []
            for l, (c, s) in enumerate(zip(cells[:-1], lower)):
                y_t, ns = c(x_t, s, t)
                x_t = y_t                                       # encoder.py:225-233: no wrappers on the AttentiveEncoder's cells
                new_lower.append(ns)
            out, ns, new_att, al = attention_wrapper_step(cells[-1], [mech], out_att, x_t, top_state, att, t)
            aligns.append(al[0])
            return out, (tuple(new_lower), (ns, new_att))

        zero = (tuple(c.zero_state(B, dtype) for c in cells[:-1]),
                (cells[-1].zero_state(B, dtype), torch.zeros(B, units[-1], dtype=dtype)))
        outs, st = dynamic_rnn(step, zero, x, lens)
        return EncoderOut(outs, st[1][0], torch.stack(aligns, dim=1))
    if cfg.encoder_type == "unidirectional":
        cells = _make_cells(P, cfg, stream, "fw", units, training, seed, T, lens)
        wrap = "highway" if cfg.highway(stream) else "residual" if cfg.residual(stream) else None   # encoder.py:67-78
        outs, st = dynamic_rnn(_stack_step(cells, wrap, P, f"{stream}/enc/fw"), tuple(c.zero_state(B, dtype) for c in cells), x, lens)
        return EncoderOut(outs, st[-1])
    # bidirectional: two independent stacks, concat at the top only (encoder.py:92-121)
    fw = _make_cells(P, cfg, stream, "fw", units, training, seed, T, lens)
    bw = _make_cells(P, cfg, stream, "bw", units, training, seed, T, lens)
    o_fw, s_fw = dynamic_rnn(_stack_step(fw), tuple(c.zero_state(B, dtype) for c in fw), x, lens)
    o_bw, s_bw = dynamic_rnn(_stack_step(bw), tuple(c.zero_state(B, dtype) for c in bw),
                             _reverse_sequence(x, lens), lens)
    o_bw = _reverse_sequence(o_bw, lens)
    outs = torch.cat([o_fw, o_bw], dim=-1)
    if cfg.cell_type == "lstm":
        c = torch.cat([s_fw[-1][0], s_bw[-1][0]], dim=-1) @ P[f"{stream}/enc/proj_c"]
        h = torch.cat([s_fw[-1][1], s_bw[-1][1]], dim=-1) @ P[f"{stream}/enc/proj_h"]
        return EncoderOut(outs, (c, h))
    h = torch.cat([s_fw[-1][0], s_bw[-1][0]], dim=-1) @ P[f"{stream}/enc/proj"]
    return EncoderOut(outs, (h,))


def au_loss(P, enc: EncoderOut, aus: Tensor, lens: Tensor) -> Tensor:
    """encoder.py:173-189: Dense(2, sigmoid) vs clip(aus,0,3)/3, tf.losses.mean_squared_error with
    weights = sequence mask tiled to [B,T,2] (reduction SUM_BY_NONZERO_WEIGHTS)."""
    pred = torch.sigmoid(enc.outputs @ P["video/au/kernel"] + P["video/au/bias"])
    tgt = torch.clamp(aus, 0.0, 3.0) / 3.0
    T = pred.shape[1]
    w = (torch.arange(T)[None, :] < lens[:, None]).to(pred.dtype)[:, :, None].expand_as(pred)
    num = torch.sum(w * (pred - tgt) ** 2)
    den = torch.sum(w)
    return num / den if float(den) > 0 else num * 0.0


# ----------------------------------------------------------------------------------------
# the model: encoders + decoder (seq2seq.py:30-126)
# ----------------------------------------------------------------------------------------
class _Model:
    def __init__(self, P: Dict[str, Tensor], cfg: OracleConfig, batch: Batch, training: bool, dtype, seed: int = 0, bn_stats=None):
        self.P, self.cfg, self.training, self.seed = P, cfg, training, seed
        bn_stats = bn_stats or {}
        self.bn_updates: dict = {}
        tt = lambda a: None if a is None else torch.as_tensor(np.asarray(a), dtype=dtype)
        ti = lambda a: None if a is None else torch.as_tensor(np.asarray(a), dtype=torch.int64)
        self.enc: Dict[str, EncoderOut] = {}
        self.lens: Dict[str, Tensor] = {}
        self.aux_loss = None
        if cfg.video_units is not None:
            self.lens["video"] = ti(batch.video_len)
            vid = tt(batch.video)
            if cfg.video_processing == "resnet_cnn":            # avsr.py:684-696: [B, T, H, W, C] -> [B, T, cnn_dense_units]
                Bv, Tv = vid.shape[0], vid.shape[1]
                vid = cnn_forward(P, cfg, vid.reshape((Bv * Tv,) + tuple(vid.shape[2:])), training, self.bn_updates).reshape(Bv, Tv, -1)
            self.enc["video"] = encode_stream(P, cfg, "video", vid, self.lens["video"],
                                              training, self.bn_updates, seed=seed, bn_stats=bn_stats.get("video"))
            if cfg.regress_aus and training:
                self.aux_loss = au_loss(P, self.enc["video"], tt(batch.aus), self.lens["video"])
        if cfg.audio_units is not None:
            self.lens["audio"] = ti(batch.audio_len)
            attended = None
            if cfg.architecture == "av_align":
                attended = (self.enc["video"].outputs, self.lens["video"])
            self.enc["audio"] = encode_stream(P, cfg, "audio", tt(batch.audio), self.lens["audio"],
                                              training, self.bn_updates, attended, seed=seed, bn_stats=bn_stats.get("audio"))
        self.B = (batch.audio if batch.audio is not None else batch.video if batch.video is not None else batch.labels).shape[0]
        self.dtype = dtype
        self._init_decoder()

    def _init_decoder(self):
        P, cfg = self.P, self.cfg
        self.cell = _Cell(P, "dec/l0", cfg.cell_type, cfg.decoder_units[0])
        if len(cfg.decoder_units) > 1:
            self.cell = _MultiCell([self.cell] + [_Cell(P, f"dec/l{j}", cfg.cell_type, u) for j, u in enumerate(cfg.decoder_units) if j > 0])
        self.mechs = [
            _Mechanism(P, f"dec/att{i}", t, self.enc[s].outputs, self.lens[s])
            for i, (s, t) in enumerate(cfg.decoder_memories())
        ]
        self.out_att = cfg.output_attention()
        if cfg.architecture == "bimodal":
            # _project_lstm_state_tuple: ONE shared Dense on concat c and on concat h (decoder_bimodal.py:480-490)
            H = cfg.decoder_units[0]
            z = torch.zeros(self.B, H, dtype=self.dtype)
            vs = self.enc["video"].final_state if "video" in self.enc else (z, z)
            as_ = self.enc["audio"].final_state if "audio" in self.enc else (z, z)
            c = torch.cat([vs[0], as_[0]], dim=-1) @ P["dec/state_proj"]
            h = torch.cat([vs[1], as_[1]], dim=-1) @ P["dec/state_proj"]
            self.init_state = (c, h)
        elif cfg.architecture == "lm":                            # lm.py:352-353: cells.zero_state
            z = torch.zeros(self.B, cfg.decoder_units[0], dtype=self.dtype)
            self.init_state = (z, z) if cfg.cell_type == "lstm" else (z,)
        else:
            s = "audio" if "audio" in self.enc else "video"
            self.init_state = self.enc[s].final_state             # decoder_unimodal.py:144-145
        if len(cfg.decoder_units) > 1:
            # decoder_unimodal.py:151-157 / decoder_bimodal.py:159-166: layer 0 from the encoder(s), the layers above start at zero
            self.init_state = tuple(self.init_state) + self.cell.zero_state(self.B, self.dtype)[len(self.init_state):]
        self.att_dim = cfg.decoder_units[-1] * len(self.mechs)

    def step(self, x, state, att, t=0):
        if self.mechs:
            return attention_wrapper_step(self.cell, self.mechs, self.out_att, x, state, att, t)
        out, ns = self.cell(x, state, t)
        return out, ns, att, []

    def logits(self, out):
        return out @ self.P["dec/out/kernel"] + self.P["dec/out/bias"]


def categorical_f32(logits_row: np.ndarray, u: np.float32) -> int:
    """Inverse-CDF draw exactly as the sampling kernel does it (fp32, sequential order)."""
    lg = logits_row.astype(np.float32)
    mx = lg.max()
    p = np.exp(lg - mx, dtype=np.float32)
    tot = np.float32(0.0)
    for v in p:
        tot = np.float32(tot + v)
    target = np.float32(u * tot)
    run = np.float32(0.0)
    for i, v in enumerate(p):
        run = np.float32(run + v)
        if run > target:
            return i
    return len(p) - 1


def forward_train(P: Dict[str, Tensor], cfg: OracleConfig, batch: Batch, dtype=torch.float64, seed: int = 0, bn_stats=None):
    """Train-graph forward: teacher forcing with optional scheduled sampling (ScheduledEmbeddingTrainingHelper,
    decoder_unimodal.py:304-309) and DropoutWrapper'd cells; `seed` = global step keys the stateless RNG.
    Returns logits [B,L,V] and the model."""
    m = _Model(P, cfg, batch, True, dtype, seed, bn_stats=bn_stats)         # bn_stats: {stream: (mean, var, rows)} of a global batch
    labels = torch.as_tensor(batch.labels, dtype=torch.int64)
    ll = torch.as_tensor(batch.labels_len, dtype=torch.int64)
    B, L = labels.shape
    if cfg.use_dropout:
        m.cell.set_dropout(cfg.decoder_dropout, CELL_ID_DECODER, L, seed)
    go = torch.full((B, 1), cfg.go_id, dtype=torch.int64)
    fed = torch.cat([go, labels], dim=1)[:, :L].numpy().copy()           # decoder_unimodal.py:66-68: tokens fed at step t
    state, att = m.init_state, torch.zeros(B, m.att_dim, dtype=dtype)
    steps = int(ll.max())
    logits = []
    for t in range(steps):
        out, ns, natt, _ = m.step(_embedding(P, cfg)[torch.as_tensor(fed[:, t].copy())], state, att, t)
        lg = m.logits(out)
        if cfg.sampling_probability > 0 and t + 1 < L:
            idx = (np.arange(B) * L + t).astype(np.uint32)
            sel = uniform01(seed, STREAM_SS_SELECT, idx) < np.float32(cfg.sampling_probability)
            us = uniform01(seed, STREAM_SS_SAMPLE, idx)
            lgn = lg.detach().numpy()
            for b in range(B):
                if sel[b]:
                    fed[b, t + 1] = categorical_f32(lgn[b], us[b])      # a draw, not argmax; no gradient through it
        fin = (t >= ll)[:, None]            # TrainingHelper: finished once t >= sequence_length
        logits.append(torch.where(fin, torch.zeros_like(lg), lg))          # impute_finished
        state = _map_state(lambda n, s: torch.where(fin, s, n), ns, state)
        att = torch.where(fin, att, natt)
    logits = torch.stack(logits, dim=1)
    if steps < L:
        logits = torch.cat([logits, torch.zeros(B, L - steps, logits.shape[-1], dtype=dtype)], dim=1)
    m.fed_tokens = fed
    return logits, m


def loss_fn(P, cfg: OracleConfig, batch: Batch, logits: Tensor, m: _Model):
    """seq2seq.py:142-190: masked sequence CE + L2 on RNN kernels + au_loss_weight * AU loss."""
    labels = torch.as_tensor(batch.labels, dtype=torch.int64)
    ll = torch.as_tensor(batch.labels_len, dtype=torch.int64)
    w = (torch.arange(labels.shape[1])[None, :] < ll[:, None]).to(logits.dtype)
    V = logits.shape[-1]
    flat = logits.reshape(-1, V)
    onehot = torch.nn.functional.one_hot(labels.reshape(-1), V).to(logits.dtype)
    if cfg.loss_fun is None and cfg.label_smoothing <= 0.0:      # seq2seq.py:147-150: sparse softmax cross-entropy
        ce = torch.nn.functional.cross_entropy(flat, labels.reshape(-1), reduction="none")
    elif cfg.loss_fun is None:
        # seq2seq.py:151-155 -> devel.py:54-61: tf.losses.softmax_cross_entropy(onehot, logits, label_smoothing) keeps its default
        # reduction SUM_BY_NONZERO_WEIGHTS, i.e. it returns ONE SCALAR = mean over all B*L rows (padding included, whose
        # imputed logits are zeros and whose label is 0); sequence_loss then multiplies that scalar by the mask and divides by
        # the mask sum, which gives the scalar back.  Restated as is (TF r1.13 losses_impl.py, recalled).
        soft = onehot * (1.0 - cfg.label_smoothing) + cfg.label_smoothing / V
        scalar = torch.mean(-torch.sum(soft * torch.log_softmax(flat, dim=-1), dim=-1))
        ce = scalar.expand(flat.shape[0])
    elif cfg.loss_fun in ("focal_loss", "mc_loss"):              # devel.py:12-51 (gamma = 2)
        p = torch.clamp(torch.softmax(flat, dim=-1), 1e-7, 1.0 - 1e-7)
        if cfg.loss_fun == "focal_loss":
            ce = torch.sum(-onehot * (1.0 - p) ** 2.0 * torch.log(p) - (1 - onehot) * p ** 2.0 * torch.log(1.0 - p), dim=1)
        else:
            ce = torch.sum(-onehot * torch.log(p) - (1 - onehot) * torch.log(1.0 - p), dim=1)
    else:
        raise ValueError('Unknown loss function {}'.format(cfg.loss_fun))      # seq2seq.py:163
    ce = ce.reshape(labels.shape)
    seq = torch.sum(ce * w) / (torch.sum(w) + 1e-12)
    total = seq
    if cfg.recurrent_l2 is not None:
        for k in l2_names(P, cfg):
            total = total + cfg.recurrent_l2 * 0.5 * torch.sum(P[k] ** 2)
    if cfg.video_units is not None and cfg.video_processing == "resnet_cnn":
        for k in cnn_l2_names(P):                                # seq2seq.py:180-184
            total = total + 0.001 * 0.5 * torch.sum(P[k] ** 2)
        # the same collection also holds the input Dense layers' l2(0.0001) regularisers (encoder.py:164): they reach the loss
        # ONLY on this branch, i.e. only when a CNN front-end is configured -- a quirk of the reference, restated as is
        for k in P:
            if "/dense" in k and k.endswith("/kernel"):
                total = total + 0.0001 * 0.5 * torch.sum(P[k] ** 2)
    if cfg.regress_aus and m.aux_loss is not None:
        total = total + cfg.au_loss_weight * m.aux_loss
    return total, seq


def to_torch(P_np: Dict[str, np.ndarray], dtype=torch.float64, requires_grad=False) -> Dict[str, Tensor]:
    out = {}
    for k, v in P_np.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        if requires_grad and not k.endswith(NON_TRAINABLE):
            t.requires_grad_(True)
        out[k] = t
    return out


def lr_at(cfg: OracleConfig, step: int) -> float:
    """seq2seq.py:259-280: constant lr or tf.train.cosine_decay_restarts(lr, global_step, first_decay_steps) with its
    defaults t_mul=2, m_mul=1, alpha=0 (recalled from TF r1.13 learning_rate_decay_v2.py), then the linear warm-up."""
    lr = cfg.learning_rate
    if cfg.lr_decay_steps:
        frac = step / float(cfg.lr_decay_steps)
        i_restart = np.floor(np.log(1.0 - frac * (1.0 - 2.0)) / np.log(2.0))
        sum_r = (1.0 - 2.0 ** i_restart) / (1.0 - 2.0)
        within = (frac - sum_r) / 2.0 ** i_restart
        lr *= 0.5 * (1.0 + np.cos(np.pi * within))
    if cfg.warmup_steps:
        lr *= min(1.0, (step + 1) / float(cfg.warmup_steps))
    return lr


def train_step(P_np: Dict[str, np.ndarray], opt: Optional[dict], cfg: OracleConfig, batch: Batch,
               dtype=torch.float64):
    """One full train step: fwd, BPTT, global-norm clip, Adam, BN moving stats.
    Returns dict(loss, seq_loss, global_norm, grads, params, opt, logits)."""
    P = to_torch(P_np, dtype, requires_grad=True)
    RELU_MARGIN[0] = float("inf")
    logits, m = forward_train(P, cfg, batch, dtype, seed=(opt["step"] if opt else 0))
    relu_margin = RELU_MARGIN[0]
    loss, seq = loss_fn(P, cfg, batch, logits, m)
    names = trainable_names(P)
    grads = torch.autograd.grad(loss, [P[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(P[k])) for k, g in zip(names, grads)}
    gnorm = torch.sqrt(sum(torch.sum(g * g) for g in grads.values()))
    scale = 1.0
    if cfg.clip_gradients:
        scale = cfg.max_gradient_norm / max(float(gnorm), cfg.max_gradient_norm)
    if opt is None:
        opt = {"step": 0, "m": {k: np.zeros_like(P_np[k], dtype=np.float64) for k in names},
               "v": {k: np.zeros_like(P_np[k], dtype=np.float64) for k in names}}
    step = opt["step"]
    lr = lr_at(cfg, step)
    t = step + 1
    b1, b2, eps = 0.9, 0.999, 1e-8
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    newP = {k: np.array(v, copy=True) for k, v in P_np.items()}
    new_opt = {"step": t, "m": {}, "v": {}}
    for k in names:
        g = grads[k].detach().numpy().astype(np.float64) * scale
        p0 = P_np[k].astype(np.float64)
        if cfg.optimiser == "Momentum":                     # tf.train.MomentumOptimizer(lr, 0.9, use_nesterov=False)
            acc = 0.9 * opt["m"][k] + g
            new_opt["m"][k], new_opt["v"][k] = acc, opt["v"][k]
            newP[k] = (p0 - lr * acc).astype(P_np[k].dtype)
            continue
        mm = b1 * opt["m"][k] + (1 - b1) * g
        vv = b2 * opt["v"][k] + (1 - b2) * g * g
        new_opt["m"][k], new_opt["v"][k] = mm, vv
        if cfg.optimiser == "AdamW":                        # contrib.opt.AdamWOptimizer: var -= weight_decay * var, then Adam
            p0 = p0 - cfg.weight_decay * p0
        num = b1 * mm + (1 - b1) * g if cfg.optimiser == "Nadam" else mm      # NadamOptimizer: ApplyAdam(use_nesterov=True)
        newP[k] = (p0 - lr_t * num / (np.sqrt(vv) + eps)).astype(P_np[k].dtype)
    if cfg.batch_normalisation:                             # seq2seq.py:241-250: UPDATE_OPS ride with the train op only then
        for k, v in m.bn_updates.items():                   # (the CNN's batch norms exist either way; their moving stats then stay put)
            newP[k] = v.detach().numpy().astype(P_np[k].dtype)
    au_term = float((cfg.au_loss_weight * m.aux_loss).detach()) if (cfg.regress_aus and m.aux_loss is not None) else 0.0
    return {"loss": float(loss.detach()), "seq_loss": float(seq.detach()), "au_term": au_term, "global_norm": float(gnorm.detach()),
            "grads": {k: g.detach().numpy() for k, g in grads.items()}, "params": newP, "opt": new_opt,
            "logits": logits.detach().numpy(), "fed_tokens": m.fed_tokens, "relu_margin": relu_margin}


@torch.no_grad()
def lm_likelihoods(P_np: Dict[str, np.ndarray], cfg: OracleConfig, batch: Batch, dtype=torch.float64) -> np.ndarray:
    """avsr.LM evaluate graph (lm.py:362-401): TrainingHelper (teacher forcing, no dropout) and
    sequence_loss(average_across_batch=False, average_across_timesteps=True): per utterance
    sum_t CE[b,t] w[b,t] / (sum_t w[b,t] + 1e-12) -- the reference calls these `average_log_likelihoods`."""
    import dataclasses
    ecfg = dataclasses.replace(cfg, use_dropout=False, sampling_probability=0.0)
    logits, _ = forward_train(to_torch(P_np, dtype), ecfg, batch, dtype)
    labels = torch.as_tensor(batch.labels, dtype=torch.int64)
    ll = torch.as_tensor(batch.labels_len, dtype=torch.int64)
    w = (torch.arange(labels.shape[1])[None, :] < ll[:, None]).to(dtype)
    ce = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), reduction="none").reshape(labels.shape)
    return (torch.sum(ce * w, dim=1) / (torch.sum(w, dim=1) + 1e-12)).numpy()


@torch.no_grad()
def greedy_decode(P_np: Dict[str, np.ndarray], cfg: OracleConfig, batch: Batch, max_steps: Optional[int] = None,
                  dtype=torch.float64, return_logits: bool = False, return_alignments: bool = False):
    """Eval graph: BN moving stats, GreedyEmbeddingHelper, dynamic_decode(impute_finished=True)
    (decoder_unimodal.py:176-217; decoder_bimodal.py:279-323).  Returns int32 [B, T_out].
    return_alignments: also {"decoder": [alpha [B, T_out, T_mem] per mechanism], "encoder": AV-Align alpha [B, T_a, T_v] | None}
    (alignment_history, decoder_unimodal.py:273-290); rows of steps after an utterance finished are reported as zeros
    (TF keeps writing whatever the un-imputed cell call produced there -- don't-care values under the image)."""
    P = to_torch(P_np, dtype)
    m = _Model(P, cfg, batch, False, dtype)
    B = m.B
    max_steps = cfg.max_label_length if max_steps is None else max_steps
    tok = torch.full((B,), cfg.go_id, dtype=torch.int64)
    state, att = m.init_state, torch.zeros(B, m.att_dim, dtype=dtype)
    finished = torch.zeros(B, dtype=torch.bool)
    ids, lgs, als = [], [], []
    for t in range(max_steps):
        out, ns, natt, al = m.step(_embedding(P, cfg)[tok], state, att)
        als.append([torch.where(finished[:, None], torch.zeros_like(a), a) for a in al])
        lg = m.logits(out)
        sample = torch.argmax(lg, dim=-1)
        f = finished[:, None]
        ids.append(torch.where(finished, torch.zeros_like(sample), sample))
        lgs.append(torch.where(f, torch.zeros_like(lg), lg))
        state = _map_state(lambda n, s: torch.where(f, s, n), ns, state)
        att = torch.where(f, att, natt)
        tok = sample
        finished = finished | (sample == cfg.eos_id)
        if bool(finished.all()):
            break
    ids = torch.stack(ids, dim=1).to(torch.int32).numpy()
    if return_alignments:
        dec = [torch.stack([a[i] for a in als], dim=1).numpy() for i in range(len(als[0]))]
        enc = None
        if cfg.architecture == "av_align":
            enc = m.enc["audio"].alignments.numpy()
            enc = enc * (np.arange(enc.shape[1])[None, :, None] < np.asarray(batch.audio_len)[:, None, None])
        return ids, {"decoder": dec, "encoder": enc}
    if return_logits:
        return ids, torch.stack(lgs, dim=1).numpy()
    return ids


@torch.no_grad()
def encoder_outputs(P_np, cfg: OracleConfig, batch: Batch, training: bool, dtype=torch.float64):
    """Convenience for kernel-level parity tests: encoder memories + final states."""
    P = to_torch(P_np, dtype)
    m = _Model(P, cfg, batch, training, dtype)
    return {s: (e.outputs.numpy(), tuple(x.numpy() for x in e.final_state)) for s, e in m.enc.items()}


# ----------------------------------------------------------------------------------------
# beam search (contrib.seq2seq.BeamSearchDecoder as configured by decoder_unimodal.py:248-271 /
# decoder_bimodal.py:358-381).  Restated from the published r1.13 algorithm (beam_search_decoder.py:
# _beam_search_step, _get_scores / _length_penalty, _mask_probs, finalize -> gather_tree); SURVEY A12 rates
# the recall of its details "low confidence".  Since round 6 the per-step bookkeeping is PINNED: the reference tree holds one output
# of TensorFlow's BeamSearchDecoder (avsr/visualise/00025.html: scores, ids, parents of 19 steps x 10 beams) and tests/test_beam_trace.py
# replays it through beam_candidates / beam_advance below (and tests/test_gpu_beam.py through the HIP beam step); three other
# readings of `_beam_search_step` are shown NOT to reproduce it.  gather_tree and the model around the step remain unpinned.
# ----------------------------------------------------------------------------------------
def gather_tree(step_ids: np.ndarray, parent_ids: np.ndarray, max_len: np.ndarray, end_token: int) -> np.ndarray:
    """step_ids / parent_ids [T, B, K] -> beams [T, B, K] (gather_tree op semantics)."""
    T, B, K = step_ids.shape
    out = np.full((T, B, K), end_token, dtype=np.int32)
    for b in range(B):
        ml = min(int(max_len[b]), T)
        if ml <= 0:
            continue
        for k in range(K):
            parent = parent_ids[ml - 1, b, k]
            out[ml - 1, b, k] = step_ids[ml - 1, b, k]
            for level in range(ml - 2, -1, -1):
                out[level, b, k] = step_ids[level, b, parent]
                parent = parent_ids[level, b, parent]
            seen = False
            for level in range(ml):
                if seen:
                    out[level, b, k] = end_token
                elif out[level, b, k] == end_token:
                    seen = True
    return out


@torch.no_grad()
def beam_candidates(logp, finished, lengths, step_lp, w: float, eos: int):
    """The scoring half of one contrib.seq2seq.BeamSearchDecoder step (`_beam_search_step`, called through decoder_unimodal.py:248-271 /
    decoder_bimodal.py:358-381): accumulated log-probabilities `total` [B, K, V] and length-penalised `scores` [B, K * V] of all
    continuations.  A finished beam continues with EOS only, at log-probability 0 (`_mask_probs`); an EOS continuation does not count
    towards the length its score is normalised by, every other token does (`lengths_to_add`); penalty ((5 + len) / 6) ^ w.
    PINNED against TensorFlow's own output: tests/test_beam_trace.py replays the reference's sample search avsr/visualise/00025.html
    (190 nodes of `BeamSearchDecoderOutput.scores / predicted_ids / parent_ids`, written by avsr.py:472-487) through this function."""
    V = step_lp.shape[-1]
    fin_row = torch.full((V,), torch.finfo(torch.float32).min, dtype=step_lp.dtype)
    fin_row[eos] = 0.0
    step_lp = torch.where(finished[:, :, None], fin_row[None, None, :], step_lp)     # _mask_probs
    total = logp[:, :, None] + step_lp
    add = torch.ones(V, dtype=torch.int64)
    add[eos] = 0
    new_len = lengths[:, :, None] + add[None, None, :] * (~finished)[:, :, None].to(torch.int64)
    penalty = ((5.0 + new_len.to(step_lp.dtype)) / 6.0) ** w
    return total, (total / penalty).reshape(total.shape[0], -1)


def beam_advance(total, finished, lengths, order, V: int, eos: int):
    """The state half of the step: the beams selected by `order` [B, K] (indices into K * V) become the new hypotheses.  A beam that
    was not finished BEFORE this step gets one position longer -- the step that emits EOS included, so a finished beam's score is
    divided by a penalty one position larger from the step AFTER its EOS on (the one-time shift visible in the reference's trace)."""
    B = total.shape[0]
    word, parent = order % V, order // V
    logp = torch.gather(total.reshape(B, -1), 1, order)
    prev_fin = torch.gather(finished, 1, parent)
    lengths = torch.gather(lengths, 1, parent) + (~prev_fin).to(torch.int64)
    return logp, prev_fin | (word == eos), lengths


def beam_search_decode(P_np: Dict[str, np.ndarray], cfg: OracleConfig, batch: Batch, beam_width: int = 10,
                       length_penalty_weight: Optional[float] = None, max_steps: Optional[int] = None, dtype=torch.float64,
                       return_all: bool = False, return_trace: bool = False, follow=None):
    """Returns predicted ids of beam 0, int32 [B, T_out] (`outputs.predicted_ids[:, :, 0]`, decoder_unimodal.py:269).
    length_penalty_weight defaults to the reference's 0.6 (unimodal / av_align) or 0.5 (bimodal).

    follow = (step_ids [T, B, K], parent_ids [T, B, K]) of ANOTHER implementation's search (test aid, tests/test_gpu_beam.py): at every
    step the oracle scores all K * V candidates from the current state, records how far the followed implementation's j-th selection
    lies from the oracle's own j-th best score (`follow_dev` [T, B]: 0 when they are the same candidates in the same order, within an
    fp32 implementation's rounding when it ordered a near-tie the other way, large when it picked a wrong candidate), whether the
    selections were identical (`follow_same` [T, B]) and whether they were K distinct candidates, and then CONTINUES FROM THE FOLLOWED
    SELECTION -- so every step of the other search is checked, not only those before its first near-tie."""
    P = to_torch(P_np, dtype)
    m = _Model(P, cfg, batch, False, dtype)
    B, K, V, eos = m.B, beam_width, cfg.vocab_size, cfg.eos_id
    w = length_penalty_weight if length_penalty_weight is not None else (0.5 if cfg.architecture == "bimodal" else 0.6)
    max_steps = cfg.max_label_length if max_steps is None else max_steps
    tile = lambda t: t.repeat_interleave(K, dim=0)                       # seq2seq.tile_batch
    for mech in m.mechs:
        mech.values, mech.keys, mech.mask = tile(mech.values), tile(mech.keys), tile(mech.mask)
    state = tuple(tile(s) for s in m.init_state)
    att = torch.zeros(B * K, m.att_dim, dtype=dtype)
    tok = torch.full((B * K,), cfg.go_id, dtype=torch.int64)
    logp = torch.full((B, K), -float("inf"), dtype=dtype)
    logp[:, 0] = 0.0
    finished = torch.zeros(B, K, dtype=torch.bool)
    lengths = torch.zeros(B, K, dtype=torch.int64)
    step_ids, parent_ids = [], []
    min_gap = torch.full((B,), float("inf"), dtype=dtype)
    step_gaps = []
    f_dev, f_same, f_distinct, f_short = [], [], [], False
    for t in range(max_steps):
        out, state, att, _ = m.step(_embedding(P, cfg)[tok], state, att, t)
        step_lp = torch.log_softmax(m.logits(out), dim=-1).reshape(B, K, V)
        total, scores = beam_candidates(logp, finished, lengths, step_lp, w, eos)
        # tf.nn.top_k: descending, ties -> lower index first
        order_all = torch.argsort(scores, dim=1, descending=True, stable=True)
        order = order_all[:, :K]
        # test aid: the smallest gap between consecutive DISTINCT scores among the best K + 1 candidates (an fp32 implementation may
        # legitimately order two candidates closer than its rounding noise differently; exact ties follow the index rule for both)
        top = torch.gather(scores, 1, order_all[:, :K + 1])
        gap = top[:, :-1] - top[:, 1:]
        gap = torch.where((top[:, :-1] == top[:, 1:]) | torch.isnan(gap), torch.full_like(gap, float("inf")), gap)
        min_gap = torch.minimum(min_gap, gap.min(dim=1).values)
        step_gaps.append(gap.min(dim=1).values.numpy())
        if follow is not None:
            if t >= follow[0].shape[0]:                      # the followed search stopped earlier than this one would
                f_short = True
                break
            eng = torch.as_tensor(follow[1][t].astype(np.int64)) * V + torch.as_tensor(follow[0][t].astype(np.int64))      # [B, K]
            s_eng, s_ref = torch.gather(scores, 1, eng), torch.gather(scores, 1, order)
            both_inf = torch.isinf(s_eng) & torch.isinf(s_ref) & (s_eng == s_ref)
            dev = torch.where(both_inf, torch.zeros_like(s_ref), (s_eng - s_ref).abs() / torch.clamp(s_ref.abs(), min=1.0))
            dev = torch.where(torch.isnan(dev), torch.full_like(dev, float("inf")), dev)
            f_dev.append(dev.max(dim=1).values.numpy())
            f_same.append((eng == order).all(dim=1).numpy())
            srt = torch.sort(eng, dim=1).values
            f_distinct.append(((srt[:, 1:] != srt[:, :-1]).all(dim=1) if K > 1 else torch.ones(B, dtype=torch.bool)).numpy())
            order = eng
        word = order % V
        parent = order // V
        logp, finished, lengths = beam_advance(total, finished, lengths, order, V, eos)
        rows = (torch.arange(B)[:, None] * K + parent).reshape(-1)
        state = tuple(s[rows] for s in state)
        att = att[rows]
        tok = word.reshape(-1)
        step_ids.append(word.numpy().astype(np.int32))
        parent_ids.append(parent.numpy().astype(np.int32))
        if bool(finished.all()):
            break
    sid, pid = np.stack(step_ids), np.stack(parent_ids)
    beams = gather_tree(sid, pid, lengths.max(dim=1).values.numpy(), eos)               # [T, B, K]
    ids = np.ascontiguousarray(beams.transpose(1, 0, 2))                                 # [B, T, K]
    if return_trace:        # per-step selections before gather_tree ([T, B, K]) and per-step near-tie gaps ([T, B]): see tests/test_gpu_beam.py
        tr = dict(step_ids=sid, parent_ids=pid, gaps=np.stack(step_gaps)[:sid.shape[0]])
        if follow is not None:
            tr.update(follow_dev=np.stack(f_dev), follow_same=np.stack(f_same), follow_distinct=np.stack(f_distinct), follow_short=f_short)
        return ids, logp.numpy(), lengths.numpy(), tr
    if return_all:
        return ids, logp.numpy(), lengths.numpy(), min_gap.numpy()
    return ids[:, :, 0]
