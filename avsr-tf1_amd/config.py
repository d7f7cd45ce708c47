"""Hyper-parameters the hot path reads (mirror of the HParams bag, avsr/avsr.py:150-198)."""
from dataclasses import dataclass, replace
from typing import List, Optional, Tuple

LUONG_TYPES = ("luong", "scaled_luong")
BAHDANAU_TYPES = ("bahdanau", "normed_bahdanau")
ATT_CODE = {"luong": 0, "scaled_luong": 1, "bahdanau": 2, "normed_bahdanau": 3}
CELL_ID_DECODER = 40


def round4(n: int) -> int:
    return (n + 3) // 4 * 4


def encoder_cell_id(stream, direction, layer):
    """RNG stream family of a DropoutWrapper'd encoder cell (stream = cell_id*4 + {0 input, 1 state, 2 output})."""
    return 1 + ((0 if stream == "video" else 1) * 2 + (0 if direction == "fw" else 1)) * 8 + layer


@dataclass
class ModelConfig:
    architecture: str = "unimodal"
    encoder_type: str = "unidirectional"
    cell_type: str = "lstm"
    video_units: Optional[Tuple[int, ...]] = None      # encoder_units_per_layer[0]; None = no video stream
    audio_units: Optional[Tuple[int, ...]] = (256, 256, 256)
    decoder_units: Tuple[int, ...] = (256,)
    attention_type: Tuple[Tuple[str, ...], Tuple[str, ...]] = (("scaled_luong",), ("scaled_luong",))
    enable_attention: bool = True
    embedding_size: int = 128
    vocab_size: int = 31
    go_id: int = 30
    eos_id: int = 29
    video_feat: int = 128
    audio_feat: int = 80
    batch_normalisation: bool = True
    regress_aus: bool = False
    au_loss_weight: float = 10.0
    recurrent_l2: Optional[float] = 1e-4
    clip_gradients: bool = True
    max_gradient_norm: float = 1.0
    learning_rate: float = 1e-3
    warmup_steps: int = 750
    optimiser: str = "Adam"                   # Adam | Nadam | AdamW | Momentum (seq2seq.py:195-218)
    weight_decay: float = 1e-4                # AdamW only (avsr.py:45)
    loss_fun: Optional[str] = None            # None | 'focal_loss' | 'mc_loss' (seq2seq.py:147-163, avsr/devel.py:12-51)
    label_smoothing: float = 0.0              # > 0: tf.losses.softmax_cross_entropy path (seq2seq.py:151-155)
    lr_decay_steps: int = 0                   # lr_decay=('cosine_restarts', N) (seq2seq.py:266-270); 0 = constant lr
    max_label_length: int = 150
    use_dropout: bool = False
    video_dropout: Tuple[float, float, float] = (0.9, 0.9, 0.9)     # keep probabilities (input, state, output), avsr.py:52-54
    audio_dropout: Tuple[float, float, float] = (0.9, 0.9, 0.9)
    decoder_dropout: Tuple[float, float, float] = (0.9, 0.9, 0.9)
    sampling_probability: float = 0.0
    # visual front-end (avsr/avsr.py:24, :33-34): "features" = records already hold cnn_dense_units-d vectors,
    # "resnet_cnn" = lip crops [B, T, H, W, C] through video.resnet_cnn (cnn.py)
    video_processing: str = "features"
    cnn_filters: Tuple[int, ...] = (8, 16, 32, 64)
    cnn_dense_units: int = 128
    video_hw: Tuple[int, int, int] = (36, 36, 3)
    input_dense_layers: Tuple[int, ...] = (0,)                      # avsr/avsr.py:38, encoder.py:148-171
    encoder_weight_sharing: bool = False                            # cells.py:77: encoder layers >= 2 reuse layer 1's cell
    instance_normalisation: bool = False                            # encoder.py:51-55: instance_norm after the batch norm
    highway_encoder: bool = False                                   # cells.py:89-90: HighwayWrapper on encoder layers > 0 (wins over residual)
    residual_encoder: bool = False                                  # cells.py:91-92: ResidualWrapper on encoder layers > 0
    one_hot_embedding: bool = False      # set by engine(): embedding_size <= 0 -> tf.eye(vocab_size) decoder inputs (decoder_unimodal.py:76-77)

    # -- same helpers/validation rules as the reference wiring (error types as in the reference) --
    def streams(self) -> List[str]:
        s = []
        if self.video_units is not None:
            s.append("video")
        if self.audio_units is not None:
            s.append("audio")
        return s

    def wrapped(self, stream: str) -> bool:
        """Do residual_encoder / highway_encoder / encoder_weight_sharing reach this stream's stack?  Only the unidirectional
        branch of Seq2SeqEncoder hands them to build_rnn_layers (encoder.py:67-78); the bidirectional branch builds _fw_cells /
        _bw_cells without them (encoder.py:92-108) and so does AttentiveEncoder for the AV-Align audio stack (encoder.py:225-233):
        there the reference silently ignores the flags, and so does this engine."""
        return self.encoder_type == "unidirectional" and not (self.architecture == "av_align" and stream == "audio")

    def highway(self, stream: str) -> bool:
        return self.highway_encoder and self.wrapped(stream)

    def residual(self, stream: str) -> bool:
        return self.residual_encoder and not self.highway_encoder and self.wrapped(stream)      # cells.py:89-92: highway wins

    def shared_layer(self, stream: str, l: int) -> int:
        """Index of the encoder layer whose variables layer l uses (cells.py:77 quirk: `layer > 1` reuses cell_list[-1], so
        layers 0 and 1 stay distinct and every layer from 2 up is layer 1's cell)."""
        return 1 if (self.encoder_weight_sharing and self.wrapped(stream) and l > 1) else l

    def loss_code(self) -> int:
        """avsr_seq_loss_fun's loss_fun argument."""
        if self.loss_fun is None:
            return 1 if self.label_smoothing > 0.0 else 0
        if self.loss_fun not in ("focal_loss", "mc_loss"):
            raise ValueError('Unknown loss function {}'.format(self.loss_fun))          # seq2seq.py:163
        return 2 if self.loss_fun == "focal_loss" else 3

    def directions(self) -> List[str]:
        return ["fw", "bw"] if self.encoder_type == "bidirectional" else ["fw"]

    def units(self, stream):
        return self.video_units if stream == "video" else self.audio_units

    def one_hot(self) -> bool:
        """decoder_unimodal.py:76-77: a non-positive embedding_size falls back to one-hot decoder inputs (no embedding variable)."""
        return self.one_hot_embedding or self.embedding_size <= 0

    def emb_width(self) -> int:
        return self.embedding_size if self.embedding_size > 0 else self.vocab_size

    def engine(self) -> "ModelConfig":
        """The configuration the kernels run: every feature / unit / embedding width rounded up to a multiple of 4 (16-byte rows,
        whole MFMA unit groups).  Padding entries of every variable are zero and stay zero (a padded LSTM / GRU unit has zero
        weights and bias: its candidate is tanh(0) = 0, so its cell and output stay 0; every consumer's rows for it are zero, so
        it receives a zero gradient and so do its weights) - the padded model computes exactly the unpadded one.
        `params.embed` / `params.extract` convert between the two layouts; the reference's shapes are what is imported / exported."""
        r = lambda t: None if t is None else tuple(round4(u) for u in t)
        dense = self.input_dense_layers if self.input_dense_layers[0] <= 0 else tuple(round4(u) for u in self.input_dense_layers)
        return replace(self, video_units=r(self.video_units), audio_units=r(self.audio_units), decoder_units=r(self.decoder_units),
                       embedding_size=round4(self.emb_width()), video_feat=round4(self.video_feat), audio_feat=round4(self.audio_feat),
                       input_dense_layers=dense, one_hot_embedding=self.one_hot())

    def feat(self, stream):
        return self.video_feat if stream == "video" else self.audio_feat

    def layer0_in(self, stream: str) -> int:
        """Width of the first encoder layer's input: the last input Dense layer if any, else the feature size."""
        return self.input_dense_layers[-1] if self.input_dense_layers[0] > 0 else self.feat(stream)

    def memory_depth(self, stream: str) -> int:
        return self.units(stream)[-1] * (2 if self.encoder_type == "bidirectional" else 1)

    def decoder_memories(self) -> List[Tuple[str, str]]:
        """[(stream, attention_type)] in AttentionWrapper order (decoder_bimodal.py:184-223: video first)."""
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
        mems = self.decoder_memories()
        return bool(mems) and mems[-1][1] in LUONG_TYPES

    def validate(self):
        if self.architecture == "lm":                                                 # avsr.LM (lm.py:275-471): labels only, no encoders
            if self.video_units is not None or self.audio_units is not None:
                raise ValueError("the language model has no encoders")
        elif self.architecture not in ("unimodal", "bimodal", "av_align"):
            raise Exception("Unknown architecture")                                   # seq2seq.py:66
        if self.encoder_type not in ("unidirectional", "bidirectional"):
            raise Exception("Allowed encoder types: `unidirectional`, `bidirectional`")  # encoder.py:146
        if self.cell_type not in ("lstm", "gru"):
            raise Exception("cell type not supported: {}".format(self.cell_type))      # cells.py:44
        for types in self.attention_type:
            for t in types:
                if t not in ATT_CODE:
                    raise Exception("unknown attention mechanism")                    # attention.py:86
        if self.architecture == "bimodal" and self.cell_type != "lstm":
            raise ValueError("the bimodal decoder needs LSTM state tuples (decoder_bimodal.py:130-142, :484-485)")
        if self.architecture == "av_align":
            if self.encoder_type != "unidirectional":
                raise ValueError("AttentiveEncoder implements only `unidirectional` (encoder.py:229)")
            if self.video_units is None or self.audio_units is None:
                raise ValueError("av_align needs both a video and an audio stream")
        self.loss_code()
        if self.optimiser not in ("Adam", "Nadam", "AdamW", "Momentum"):
            raise Exception('Unsupported optimiser, try Adam')                            # seq2seq.py:218
        for st in self.streams():                 # the wrapper flags only where the reference applies them (see `wrapped`)
            u = self.units(st)
            if self.highway(st) and self.encoder_weight_sharing:
                raise NotImplementedError("highway_encoder with encoder_weight_sharing")
            if self.highway(st) or self.residual(st):
                if len(set(u)) != 1:
                    raise ValueError("residual_encoder needs equal layer widths")
            if self.encoder_weight_sharing and self.wrapped(st):
                if len(u) > 2 and (len(set(u[1:])) != 1 or u[0] != u[1]):
                    raise ValueError("encoder_weight_sharing needs equal layer sizes: layers >= 2 reuse layer 1's kernel")
        if self.video_processing not in ("features", "resnet_cnn"):
            raise Exception("unknown visual content")                                   # avsr/avsr.py:713 (2dconv_cnn / 3dconv_cnn: not built)
        if self.video_units is not None and self.video_processing == "resnet_cnn":
            if self.video_feat != self.cnn_dense_units:
                raise ValueError("video_feat must equal cnn_dense_units when the CNN front-end produces the video features")
            if any(c % 4 for c in self.cnn_filters) or self.cnn_dense_units % 4 or len(self.cnn_filters) < 1:
                raise ValueError("cnn_filters / cnn_dense_units must be multiples of 4 for the HIP engine")
        if self.input_dense_layers[0] > 0 and any(u <= 0 for u in self.input_dense_layers):
            raise ValueError("input_dense_layers must be positive")
        if len(set(self.decoder_units)) != 1 or len(self.decoder_units) > 4:
            raise NotImplementedError("multi-layer decoders: up to 4 layers of equal width")
        if not self.streams() and self.architecture != "lm":
            raise Exception("labels are None")                                         # seq2seq.py:94
        dims = [self.decoder_units[0]]
        for s in self.streams():
            dims += list(self.units(s)) + [self.feat(s)]
        if any(d <= 0 for d in dims):
            raise ValueError("feature / unit sizes must be positive")       # other widths: padded inside the engine (`engine()`)
