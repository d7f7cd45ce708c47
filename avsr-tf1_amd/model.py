"""Host-side model: the wiring of avsr/seq2seq.py `Seq2SeqModel` (encoders -> AV-Align / dual attention ->
decoder -> loss -> BPTT -> clip -> Adam) expressed as a sequence of C-ABI calls into libavsr_hip.so.

All arithmetic runs in hand-written HIP kernels (csrc/); this file only owns buffers (torch tensors
as containers) and the order of calls.  There is no CPU fallback: without the library / a GPU every
entry point raises.

Reference call sites mirrored here:
  encoders            avsr/seq2seq.py:30-68  -> avsr/encoder.py:37-55 (BN), :67-143 (uni/bi RNN), :173-189 (AU loss)
  AV-Align            avsr/encoder.py:224-294
  decoder init state  avsr/decoder_unimodal.py:126-157, avsr/decoder_bimodal.py:125-166, :480-490
  decoder train       avsr/decoder_unimodal.py:299-352, avsr/decoder_bimodal.py:227-277
  greedy decode       avsr/decoder_unimodal.py:176-217, avsr/decoder_bimodal.py:279-326
  loss / optimiser    avsr/seq2seq.py:135-257, :259-280
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional

import os

import numpy as np
import torch

from . import ops, params as PR
from ._lib import AttnRnn, RnnStack
from .config import ATT_CODE, BAHDANAU_TYPES, CELL_ID_DECODER, LUONG_TYPES, ModelConfig, encoder_cell_id


@dataclass
class Batch:
    """Device-side BatchedData (avsr/io_utils.py:8-18).  float32 [B,T,F] inputs, int32 lengths/labels."""
    audio: Optional[torch.Tensor] = None
    audio_len: Optional[torch.Tensor] = None
    video: Optional[torch.Tensor] = None
    video_len: Optional[torch.Tensor] = None
    aus: Optional[torch.Tensor] = None
    labels: Optional[torch.Tensor] = None
    labels_len: Optional[torch.Tensor] = None

    @staticmethod
    def from_numpy(b, device="cuda"):
        def f(a, dt):
            return None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(device).contiguous()
        return Batch(f(getattr(b, "audio", None), torch.float32), f(getattr(b, "audio_len", None), torch.int32),
                     f(getattr(b, "video", None), torch.float32), f(getattr(b, "video_len", None), torch.int32),
                     f(getattr(b, "aus", None), torch.float32), f(getattr(b, "labels", None), torch.int32),
                     f(getattr(b, "labels_len", None), torch.int32))


class Ref:
    """A named slice of a flat device buffer."""

    def __init__(self, t, off, shape):
        self.t, self.off, self.shape = t, int(off), tuple(shape)
        self.n = int(np.prod(shape))

    def mat(self, ld=None, row0=0, col0=0):
        ld = self.shape[-1] if ld is None else ld
        return ops.mat(self.t, ld, offset=self.off + row0 * ld + col0)

    def view(self):
        return self.t[self.off:self.off + self.n].view(*self.shape)


class SeqBuf:
    """[B, lead + T + trail, D] sequence buffer; time t lives in slot lead + t; guard slots stay zero."""

    def __init__(self, B, T, D, lead, trail, device):
        self.B, self.T, self.D, self.lead = B, T, D, lead
        self.slots = lead + T + trail
        self.t = torch.zeros(B, self.slots, D, device=device)
        self.sb, self.st = self.slots * D, D

    def off(self, dt=0, col=0):
        return (self.lead + dt) * self.D + col

    def mat(self, dt=0, col=0):
        return ops.mat(self.t, self.D, T=self.T, ldo=self.sb, offset=self.off(dt, col))


def _splitk(M, N, K):
    return ops.auto_splitk(M, N, K)


class Seq2SeqModel:
    def __init__(self, cfg: ModelConfig, device="cuda", seed=0, weights: Optional[Dict[str, np.ndarray]] = None):
        cfg.validate()
        if not torch.cuda.is_available():
            raise RuntimeError("avsr_tf1_amd needs an MI355X GPU: the HIP engine has no CPU fallback")
        from . import _lib
        _lib.load()
        # the kernels run the 4-padded configuration (config.py `engine()`); cfg_tf keeps the reference's shapes for import / export
        self.cfg_tf, self.dev = cfg, torch.device(device)
        self.cfg = cfg = cfg.engine()
        self.gru = cfg.cell_type == "gru"
        # one-launch persistent encoder forward (csrc/rnn_persist.hip); process-wide engine switch
        # bits: 1 agent-scope forward | 2 XCD-local forward + fused BPTT | 4 split BPTT (measured slower on c4: 3.0 vs 2.7 ms,
        # kept selectable); 0 = per-step launches only
        pm = int(os.environ.get("AVSR_PERSISTENT_RNN", "3"))
        self.persistent_rnn = pm != 0
        ops.rnn_set_persistent(self.persistent_rnn, device=device, mode=pm or 3)
        # one-launch fused persistent decoder / AV-Align attentive layer forward (csrc/dec_persist.hip); needs the sync scratch above
        self.fused_decode = self.persistent_rnn and os.environ.get("AVSR_FUSED_DECODE", "1") != "0"
        self.G = 2 if self.gru else 4                       # gate pre-activations per unit of the main cell kernel
        self.inv, self.inv_tf = PR.inventory(cfg), PR.inventory(self.cfg_tf)
        self.seg, self.seg_tf = PR.segments(cfg), PR.segments(self.cfg_tf)
        # ---- flat parameter storage (engine layout) ------------------------------------------------
        self._train_off, self._stat_off = OrderedDict(), OrderedDict()
        nt = ns = 0
        for name, (shape, kind, _init) in self.inv.items():
            n = (int(np.prod(shape)) + 3) // 4 * 4      # keep every tensor 16-byte aligned
            if name.endswith(PR.NON_TRAINABLE):
                self._stat_off[name] = ns
                ns += n
            else:
                self._train_off[name] = nt
                nt += n
        z = lambda n, dt=torch.float32: torch.zeros(max(n, 4), dtype=dt, device=self.dev)
        self.params, self.adam_m, self.adam_v = z(nt), z(nt), z(nt)
        # the batch loss lives in the 4-float tail of the gradient buffer: a data-parallel trainer sums gradients AND loss over the
        # ranks with ONE all-reduce of grads_and_loss (parallel.py)
        # ... and, behind the loss, a mirror of the non-trainable state (batch-norm moving statistics): under data parallelism each rank
        # writes stats / world there at the end of the backward pass and reads the rank AVERAGE back before the update, so the replicas'
        # moving statistics stay identical even for the batch norms that normalise with per-rank statistics (the lip CNN's)
        self.grads_and_loss = z(max(nt, 4) + 4 + max(ns, 4))
        self.grads = self.grads_and_loss[:max(nt, 4)]
        self.stats = z(ns)
        self._redo_pass, self._stats_sink = False, None
        self.n_stats, self._stats_mirror_off = ns, max(nt, 4) + 4
        self.dp_world = 1                         # set by DataParallelTrainer
        self.n_train = nt
        self.step = z(1, torch.int32)[:1]
        # RNG key of the stateless dropout / sampling masks: global step + seed_offset (data parallel: rank << 24, so that row i of
        # different ranks does not draw the same masks); refreshed at the start of every train-graph forward
        self.seed = z(1, torch.int32)[:1]
        self.seed_offset = 0
        self.P = {n: Ref(self.params, o, self._eshape(n)) for n, o in self._train_off.items()}
        self.Gr = {n: Ref(self.grads, o, self._eshape(n)) for n, o in self._train_off.items()}
        self.S = {n: Ref(self.stats, o, self.inv[n][0]) for n, o in self._stat_off.items()}
        self.l2_segments = [(self._train_off[n], int(np.prod(self.inv[n][0]))) for n in self._train_off if PR.is_l2(n)]
        self.cnn_l2_segments = [(self._train_off[n], int(np.prod(self.inv[n][0]))) for n in self._train_off if PR.is_cnn_l2(n)]
        self.use_cnn = cfg.video_units is not None and cfg.video_processing == "resnet_cnn"
        self.dense_l2_segments = [(self._train_off[n], int(np.prod(self.inv[n][0]))) for n in self._train_off if PR.is_dense_l2(n)]
        self.bn_sync = None                       # set by bn_sync_enable() under data parallelism
        self.n_dense = len(cfg.input_dense_layers) if cfg.input_dense_layers[0] > 0 else 0
        # ---- derived transposed operands -----------------------------------------------------------
        self._tjobs, self.Tr = [], {}
        tn = 0
        for name in self._train_off:
            if name.endswith(("/kernel", "/query_kernel", "/layer_kernel", "/gates_kernel", "/cand_kernel")) and \
                    not name.startswith(("video/au", "audio/au", "video/cnn/")):
                if name.endswith("memory_kernel"):
                    continue
                r, c = self._eshape(name)
                self._tjobs.append((name, tn, r, c))
                tn += (r * c + 3) // 4 * 4
        self.derived = z(tn)
        for name, off, r, c in self._tjobs:
            self.Tr[name] = Ref(self.derived, off, (c, r))
        self._ws_cache = OrderedDict()
        # workspaces kept (least recently used dropped): bucketed training cycles through more (B, T_a, T_v, L) shapes than 8 -- a miss
        # allocates and zero-fills several GB (c4 with the lip CNN: ~6.5 GB per shape), a tenth of a step's time
        self.max_cached_shapes = int(os.environ.get("AVSR_WS_CACHE", "12"))
        self._ws_pinned = set()                  # keys whose buffers a captured hipGraph points into: never evicted (parallel.py)
        self._dropping = False
        self.au_scale = 1.0
        self.au_external = False     # data parallel: the AU loss is normalised by the all-reduced frame count in dp_norm[1]
        self.onehot = None
        if cfg.one_hot():                       # decoder_unimodal.py:76-77: tf.eye(vocab_size) rows as decoder inputs, not a variable
            self.onehot = torch.zeros(cfg.vocab_size, cfg.embedding_size, device=self.dev)
            self.onehot[:, :cfg.vocab_size] = torch.eye(cfg.vocab_size, device=self.dev)
        self.load_tf_weights(weights if weights is not None else PR.initialise(self.cfg_tf, seed))
        self.scratch = z(1 << 22)
        self.gemm_ws = None
        self._ensure_gemm_ws()               # split-K scratch, also used by forward GEMMs with few output tiles
        self.loss = self.grads_and_loss[max(nt, 4):max(nt, 4) + 1]
        self.gnorm = z(1)[:1]
        self.dp_norm = z(4)          # [sum(mask) of the sequence loss, AU frame-unit count]: what the DP trainer all-reduces per step
        self.denom, self.au_total = self.dp_norm[0:1], self.dp_norm[1:2]

    # ------------------------------------------------------------------------------------------------
    def _eshape(self, name):
        return self.inv[name][0]

    def _to_engine(self, name, a):
        """Reference-shaped array (TF layout) -> flat engine layout: zero padding to the engine widths, then the gate interleave."""
        shape, kind, _i = self.inv_tf[name]
        a = PR.embed(self.seg_tf[name], self.seg[name], np.asarray(a, dtype=np.float32).reshape(shape))
        return PR.to_engine(kind, a).reshape(-1)

    def _emb_t(self):
        t, o = self._emb()
        return t[o:]

    def _emb(self):
        """(tensor, element offset) of the decoder input table: the embedding variable, or the constant one-hot rows."""
        return (self.onehot, 0) if self.onehot is not None else (self.params, self.P["dec/embedding"].off)

    def load_tf_weights(self, W: Dict[str, np.ndarray]):
        """Import a {name: array} dict in TF layout (same names as oracle / export_tf_weights)."""
        for name, (shape, kind, _i) in self.inv_tf.items():
            e = torch.from_numpy(self._to_engine(name, W[name])).to(self.dev)
            ref = self.S[name] if name in self.S else self.P[name]
            ref.t[ref.off:ref.off + e.numel()].copy_(e)
        self._refresh_derived()

    def load_flat(self, buf, W: Dict[str, np.ndarray]):
        """Fill an optimiser-slot buffer (same layout as params) from a TF-layout dict."""
        for name in self.inv_tf:
            if name in self._train_off and name in W:
                e = torch.from_numpy(self._to_engine(name, W[name])).to(self.dev)
                o = self._train_off[name]
                buf[o:o + e.numel()].copy_(e)

    def export_tf_weights(self, which="params") -> Dict[str, np.ndarray]:
        src = {"params": self.params, "grads": self.grads, "adam_m": self.adam_m, "adam_v": self.adam_v}[which]
        host = src.detach().cpu().numpy()
        out = OrderedDict()
        for name, (shape, kind, _i) in self.inv.items():
            n = int(np.prod(shape))
            if name in self._stat_off:
                if which == "params":
                    o = self._stat_off[name]
                    out[name] = PR.extract(self.seg_tf[name], self.seg[name], self.stats[o:o + n].cpu().numpy().reshape(shape)).copy()
                continue
            o = self._train_off[name]
            out[name] = PR.extract(self.seg_tf[name], self.seg[name], PR.from_engine(kind, host[o:o + n].reshape(shape)))
        return out

    def _refresh_derived(self):
        jobs = [(self.params, self._train_off[n], self.derived, off, r, c) for n, off, r, c in self._tjobs]
        if jobs:
            ops.transpose(jobs)

    # ------------------------------------------------------------------------------------------------
    # workspace
    def _get_ws(self, B, Ta, Tv, L, greedy):
        key = (B, Ta, Tv, L, greedy)
        if key in self._ws_cache:
            self._ws_cache[key] = self._ws_cache.pop(key)          # most recently used last
            return self._ws_cache[key]
        while len(self._ws_cache) - len(self._ws_pinned & set(self._ws_cache)) >= self.max_cached_shapes:   # bucketed training visits many shapes
            victim = next((k for k in self._ws_cache if k not in self._ws_pinned), None)   # least recently used, not pinned
            if victim is None:
                break
            self._ws_cache.pop(victim)
        cfg, dev = self.cfg, self.dev
        z = lambda *s: torch.zeros(*s, device=dev)
        ws = {"enc": {}}
        ndir = len(cfg.directions())
        for s in cfg.streams():
            T = Ta if s == "audio" else Tv
            F, units = cfg.feat(s), cfg.units(s)
            attentive = cfg.architecture == "av_align" and s == "audio"
            nplain = len(units) - 1 if attentive else len(units)
            F0 = cfg.layer0_in(s)                                     # width of the first RNN layer's input (after the input Dense stack)
            E = {"T": T, "F": F, "F0": F0, "units": units, "nplain": nplain, "attentive": attentive}
            if self.n_dense:                                          # encoder.py:148-171: pre-activations, outputs and their gradients
                E["dense"] = [dict(z=z(B * T, u), a=z(B * T, u), da=z(B * T, u)) for u in cfg.input_dense_layers]
            E["xn"], E["dxn"], E["xhat"] = z(B * T, F), z(B * T, F), z(B * T, F)
            if self.cfg_tf.feat(s) != F:
                E["xpad"] = z(B, T, F)                                # the batch's features, zero columns up to the engine width
            if cfg.instance_normalisation:
                E["xi"], E["in_mean"], E["in_invstd"], E["in_dg"], E["in_db"] = z(B * T, F), z(B, F), z(B, F), z(B, F), z(B, F)
            if s == "video" and self.use_cnn:
                from .cnn import LipCNN
                E["cnn"] = LipCNN(self, B * T)                       # lip crops -> F = cnn_dense_units features
                E["dfeat"] = z(B * T, F)
            if cfg.use_dropout:
                E["xd"] = {d: z(B * T, F0) for d in cfg.directions()}    # layer-0 input after each direction's input mask
                E["dx_tmp"] = z(B * T, F0)
            E["mean_own"], E["invstd"] = z(F), z(F)
            E["mean"] = E["mean_own"]
            Dm = units[-1] * ndir
            if not attentive:
                E["mem"] = SeqBuf(B, T, Dm, 1, 1, dev)
                E["dmem"] = SeqBuf(B, T, Dm, 1, 1, dev)
            E["layers"] = {}
            for di, d in enumerate(cfg.directions()):
                for l in range(nplain):
                    u = units[l]
                    Ld = {"gates": z(B, T, u, 4), "cs": z(B, T, u), "state": z(6 * B * u), "dgates": z(B, T, u, 4),
                          "dstate": z(14 * B * u), "hf": z(B, u), "cf": z(B, u), "dhf": z(B, u), "dcf": z(B, u)}
                    top = (l == len(units) - 1)
                    if top:
                        Ld["out"], Ld["col"], Ld["dout"] = E["mem"], di * u, E["dmem"]
                    else:
                        Ld["out"], Ld["col"] = SeqBuf(B, T, u, 1, 1, dev), 0
                        Ld["dout"] = SeqBuf(B, T, u, 1, 1, dev) if (attentive and l == nplain - 1) else None
                    if self.gru:
                        Ld["rh"], Ld["dpc"] = z(B, T, u), z(B, T, u)
                    if cfg.use_dropout or (cfg.residual(s) and l > 0):
                        Ld["hs_seq"] = SeqBuf(B, T, u, 1, 1, dev)   # the recurrent h as consumed (residual: the output record holds h + x)
                    if cfg.use_dropout:
                        if not top or attentive:
                            Ld["xt_seq"] = SeqBuf(B, T, u, 1, 1, dev)
                    if cfg.highway(s):
                        # HighwayWrapper stacks run layer by layer (every input projection hoisted): `out` is the layer's emitted
                        # (highway) output, `hout` the cell's own output, `dy` / `dhout` their gradients, cpre the carry pre-activation
                        Ld["dy"] = E["dmem"] if top else SeqBuf(B, T, u, 1, 1, dev)
                        if l > 0:
                            Ld.update(hout=SeqBuf(B, T, u, 1, 1, dev), dhout=SeqBuf(B, T, u, 1, 1, dev), cpre=z(B * T, u), dcpre=z(B * T, u),
                                      dxtmp=z(B * T, u))
                            if cfg.use_dropout:
                                Ld["xd"] = z(B * T, u)
                    E["layers"][(d, l)] = Ld
            H = cfg.decoder_units[0]
            E["c_dec"], E["h_dec"], E["dc_dec"], E["dh_dec"] = z(B, H), z(B, H), z(B, H), z(B, H)
            if s == "video" and cfg.regress_aus:
                E["au_z"], E["au_dz"], E["au_row"] = z(B * T, 2), z(B * T, 2), z(B * T)
            ws["enc"][s] = E
        if cfg.architecture == "av_align":
            A = ws["enc"]["audio"]
            u = cfg.audio_units[-1]
            in_w = cfg.audio_units[-2] if len(cfg.audio_units) > 1 else cfg.layer0_in("audio")
            A["blk"] = self._make_block(ws, B, Ta, u, in_w, [("video", cfg.attention_type[0][0])], "audio/enc/fw/l%d" % (len(cfg.audio_units) - 1),
                                        ["audio/enc/att0"], Tv=Tv, Ta=Ta, greedy=False)
        Ldec = L
        ws["dec"] = self._make_block(ws, B, Ldec, cfg.decoder_units[0], cfg.embedding_size, cfg.decoder_memories(), "dec/l0",
                                     ["dec/att%d" % i for i in range(len(cfg.decoder_memories()))], Tv=Tv, Ta=Ta, greedy=greedy)
        D = ws["dec"]
        V = cfg.vocab_size
        D["xemb"], D["dxemb"] = z(B * Ldec, cfg.embedding_size), z(B * Ldec, cfg.embedding_size)
        D["logits"], D["dlogits"], D["row_loss"] = z(B, Ldec, V), z(B, Ldec, V), z(B * Ldec)
        D["fed"] = torch.zeros(B, Ldec, dtype=torch.int32, device=dev)
        D["ids"] = torch.zeros(B, Ldec, dtype=torch.int32, device=dev)
        D["tok"] = torch.zeros(B, dtype=torch.int32, device=dev)
        D["nunf"] = torch.zeros(1, dtype=torch.int32, device=dev)
        D["nunf_prev"] = torch.zeros(1, dtype=torch.int32, device=dev)
        D["steplen"] = torch.zeros(B, dtype=torch.int32, device=dev)
        ws["B"], ws["L"] = B, L
        self._ws_cache[key] = ws
        return ws

    def prepare_workspace(self, batch):
        """Allocate (or fetch) every buffer a train step on a batch of this shape needs, without launching anything.  The data-parallel
        trainer calls it ahead of a shape's first pass so that an out-of-memory surfaces before any collective of the step."""
        B, L = batch.labels.shape
        Ta = batch.audio.shape[1] if batch.audio is not None else 0
        Tv = batch.video.shape[1] if batch.video is not None else 0
        self._ensure_gemm_ws()
        return self._get_ws(B, Ta, Tv, L, False)

    def pin_workspace(self, ws):
        """A captured graph holds raw pointers into this workspace: exempt it from the LRU eviction of _get_ws until unpinned."""
        for k, v in self._ws_cache.items():
            if v is ws:
                self._ws_pinned.add(k)
                return k
        return None

    def unpin_workspace(self, key):
        self._ws_pinned.discard(key)

    def _make_block(self, ws, B, L, H, E, mems, cell_prefix, att_prefixes, Tv, Ta, greedy, mem_B=None):
        """mem_B: rows of the attention memories when they are shared by several block rows (beam search: one memory per utterance,
        beam_width hypotheses attending it); default = one memory per row."""
        cfg, dev = self.cfg, self.dev
        z = lambda *s: torch.zeros(*s, device=dev)
        A = H * len(mems)
        mem_B = B if mem_B is None else mem_B
        blk = {"B": B, "L": L, "H": H, "E": E, "A": A, "cell": cell_prefix, "mems": [], "mem_B": mem_B}
        if cell_prefix.startswith("dec/"):
            blk["cell_id"], blk["keep"] = CELL_ID_DECODER, cfg.decoder_dropout
        else:
            blk["cell_id"], blk["keep"] = encoder_cell_id("audio", "fw", len(cfg.audio_units) - 1), cfg.audio_dropout
        if self.gru:
            blk["rh"], blk["dpc"] = z(B, L, H), z(B, L, H)
        if cfg.use_dropout:
            blk["hs_seq"] = SeqBuf(B, L, H, 1, 0, dev)
            if A:
                blk["attd"] = SeqBuf(B, L, A, 1, 0, dev)
        blk.update(gates=z(B, L, H, 4), cs=z(B, L, H), cell_out=SeqBuf(B, L, H, 1, 0, dev), state=z(4 * B * H),
                   dgates=z(B, L, H, 4), dstate=z(12 * B * H), dq=z(B, L, H), dh0=z(B, H), dc0=z(B, H),
                   hf=z(B, H), cf=z(B, H), dcell_ext=z(B, L, H))
        if A:
            blk.update(att=SeqBuf(B, L, A, 1, 0, dev), datt=z(B, L, A), datt_ext=z(B, L, A))
        # scratch of the fused persistent decode kernel (quarter softmax partials, split-K logits); LSTM, 1-2 mechanisms only
        if mems and len(mems) <= 2 and not self.gru:
            blk["fused_ws"] = z(ops.attn_rnn_fused_ws_floats(B, len(mems), 256))
        blk["extra"] = []
        if cell_prefix == "dec/l0":
            # multi-layer decoder cell (MultiRNNCell, decoder_unimodal.py:101-108): layers 1.. above the attention-fed one; the top
            # layer's output record IS cell_out, the attention-fed layer records into out0
            n_extra = len(cfg.decoder_units) - 1
            if n_extra:
                blk["out0"] = SeqBuf(B, L, H, 1, 0, dev)
            for j in range(1, n_extra + 1):
                X = dict(prefix="dec/l%d" % j, cell_id=CELL_ID_DECODER + j, gates=z(B, L, H, 4), cs=z(B, L, H), state=z(4 * B * H),
                         dgates=z(B, L, H, 4), dstate=z(12 * B * H), out=(blk["cell_out"] if j == n_extra else SeqBuf(B, L, H, 1, 0, dev)))
                if cfg.use_dropout:
                    X["hs_seq"], X["xin_seq"] = SeqBuf(B, L, H, 1, 0, dev), SeqBuf(B, L, H, 1, 0, dev)
                blk["extra"].append(X)
        for (stream, att_type), pre in zip(mems, att_prefixes):
            T = Ta if stream == "audio" else Tv
            D = cfg.memory_depth(stream)
            chunk = 64 if T > 64 else max(16, (T + 1) // 2)
            while (T + chunk - 1) // chunk > 16:
                chunk *= 2
            nc = (T + chunk - 1) // chunk
            m = {"stream": stream, "type": att_type, "prefix": pre, "T": T, "D": D, "chunk": chunk, "nc": nc}
            # Memories wider than 256 (the [fw | bw] outputs of bidirectional 256-unit encoders) are attended through their PROJECTION:
            # the attention layer is linear, att = h.W_h + (sum_t alpha_t v_t).W_ctx = h.W_h + sum_t alpha_t (v_t.W_ctx), so the loop
            # attends pvals = values.W_ctx [B,T,H] (one GEMM per pass) with the attention layer [W_h ; I]: same attention vector, half
            # the bytes per frame, and the block fits the fused persistent decode kernels (values resident in LDS, D <= 256).  The
            # `ctx` / `dctx` records then hold the projected context and its gradient (= d attention); d W_ctx and d values come from
            # d pvals after the loop.
            m["proj"] = D > 256 and H <= 256
            Dv = H if m["proj"] else D
            m["Dv"] = Dv
            if m["proj"]:
                eye = torch.eye(H, device=dev)
                m.update(pvals=z(mem_B, T, H), dpvals=z(mem_B, T, H), eye=eye, watt_p=torch.cat([z(H, H), eye], 0).contiguous(),
                         watt_p_t=torch.cat([z(H, H), eye], 1).contiguous())
            m.update(keys=z(mem_B, T, H), dkeys=z(mem_B, T, H), scores=z(B, L, T), dscores=z(B, L, T), ctx=z(B, L, Dv), dctx=z(B, L, Dv),
                     pstat=z(L, 2, nc, B), pctx=z(nc, B, Dv), pdq=z(nc, B, H), rowdot=z(B * L))
            if att_type in BAHDANAU_TYPES:
                m.update(pq=z(B, L, H), dpq=z(B, L, H), vn=z(H), dvn=z(H), dv_part=z(((T + 15) // 16) * B, H))
            blk["mems"].append(m)
        return blk

    # ------------------------------------------------------------------------------------------------
    # encoders
    def _kn(self, prefix):
        """(main kernel, main bias) parameter names of a cell: LSTM kernel / GRU gate kernel."""
        return (prefix + "/gates_kernel", prefix + "/gates_bias") if self.gru else (prefix + "/kernel", prefix + "/bias")

    def _keeps(self, s):
        return self.cfg.video_dropout if s == "video" else self.cfg.audio_dropout

    def _sdrop(self, s):
        """DropoutWrapper active for this stream's encoder cells in the current pass?"""
        return self._dropping and min(self._keeps(s)) < 1.0

    def _bdrop(self, blk):
        return self._dropping and min(blk["keep"]) < 1.0

    def _rnn_stack(self, ws, s, d, B, len_t, backward=False):
        cfg = self.cfg
        E = ws["enc"][s]
        st = RnnStack()
        st.B, st.T, st.reverse, st.n_layers, st.cell = B, E["T"], int(d == "bw"), E["nplain"], int(self.gru)
        st.len = ops.fptr(len_t)
        drop = self._sdrop(s)
        if drop:
            k = self._keeps(s)
            st.seed = ops.fptr(self.seed)
            st.keep_in, st.keep_state, st.keep_out = k
            st.cell_id_base = encoder_cell_id(s, d, 0)
            if E["attentive"]:                   # the attention-wrapped top layer consumes this stack through xt_seq
                st.consumer_keep = k[0]
                st.consumer_stream = encoder_cell_id(s, d, E["nplain"]) * 4
                st.consumer_width = E["units"][-2] + E["units"][-1]
        i = E["F0"]
        for l in range(E["nplain"]):
            u = E["units"][l]
            Ld = E["layers"][(d, l)]
            name, bname = self._kn(f"{s}/enc/{d}/l{cfg.shared_layer(s, l)}")
            Ly = st.layer[l]
            Ly.units, Ly.in_dim, Ly.hoisted, Ly.out_col = u, i, int(l == 0), Ld["col"]
            Ly.wt = ops.fptr(self.derived, self.Tr[name].off)
            Ly.w = ops.fptr(self.params, self.P[name].off)
            Ly.bias = ops.fptr(self.params, self.P[bname].off)
            if self.gru:
                cn = f"{s}/enc/{d}/l{cfg.shared_layer(s, l)}/cand_kernel"
                Ly.wt2, Ly.w2 = ops.fptr(self.derived, self.Tr[cn].off), ops.fptr(self.params, self.P[cn].off)
                Ly.bias2 = ops.fptr(self.params, self.P[f"{s}/enc/{d}/l{cfg.shared_layer(s, l)}/cand_bias"].off)
                Ly.rh_seq, Ly.dgates2 = ops.fptr(Ld["rh"]), ops.fptr(Ld["dpc"])
            Ly.gates, Ly.cs = ops.fptr(Ld["gates"]), ops.fptr(Ld["cs"])
            Ly.out, Ly.ld_out = ops.fptr(Ld["out"].t), Ld["out"].D
            Ly.state, Ly.h_final, Ly.c_final = ops.fptr(Ld["state"]), ops.fptr(Ld["hf"]), ops.fptr(Ld["cf"])
            Ly.dgates, Ly.dstate = ops.fptr(Ld["dgates"]), ops.fptr(Ld["dstate"])
            Ly.residual = int(cfg.residual(s) and l > 0)
            if drop or Ly.residual:
                Ly.hs_seq = ops.fptr(Ld["hs_seq"].t)
            if drop:
                if "xt_seq" in Ld:
                    Ly.xt_seq = ops.fptr(Ld["xt_seq"].t)
            if backward and Ld["dout"] is not None:
                Ly.dout, Ly.ld_dout, Ly.dout_col = ops.fptr(Ld["dout"].t), Ld["dout"].D, Ld["col"]
            i = u
        if backward and not E["attentive"]:
            top = E["layers"][(d, E["nplain"] - 1)]
            st.dh_final, st.dc_final = ops.fptr(top["dhf"]), (None if self.gru else ops.fptr(top["dcf"]))
        return st

    # ---- HighwayWrapper encoders (cells.py:89-90): layer-by-layer execution with every input projection hoisted ----
    def _rnn_stack_single(self, ws, s, d, l, B, len_t, backward=False):
        """One-layer stack descriptor for layer l of (stream, direction): input projection already in `gates`, cell output into
        `hout` (layer 0: straight into `out`), external output gradient from `dhout` (layer 0: `dy`)."""
        cfg = self.cfg
        E = ws["enc"][s]
        Ld = E["layers"][(d, l)]
        u = E["units"][l]
        st = RnnStack()
        st.B, st.T, st.reverse, st.n_layers, st.cell = B, E["T"], int(d == "bw"), 1, int(self.gru)
        st.len = ops.fptr(len_t)
        drop = self._sdrop(s)
        if drop:
            st.seed = ops.fptr(self.seed)
            st.keep_in, st.keep_state, st.keep_out = self._keeps(s)
            st.cell_id_base = encoder_cell_id(s, d, l)
        name, bname = self._kn(f"{s}/enc/{d}/l{l}")
        Ly = st.layer[0]
        Ly.units, Ly.in_dim, Ly.hoisted = u, (E["F0"] if l == 0 else u), 1
        Ly.wt, Ly.w = ops.fptr(self.derived, self.Tr[name].off), ops.fptr(self.params, self.P[name].off)
        Ly.bias = ops.fptr(self.params, self.P[bname].off)
        if self.gru:                             # candidate kernel; its hoisted input part sits in the c~ record (`cs`)
            cn = f"{s}/enc/{d}/l{l}/cand_kernel"
            Ly.wt2, Ly.w2 = ops.fptr(self.derived, self.Tr[cn].off), ops.fptr(self.params, self.P[cn].off)
            Ly.bias2 = ops.fptr(self.params, self.P[f"{s}/enc/{d}/l{l}/cand_bias"].off)
            Ly.rh_seq, Ly.dgates2 = ops.fptr(Ld["rh"]), ops.fptr(Ld["dpc"])
        Ly.gates, Ly.cs = ops.fptr(Ld["gates"]), ops.fptr(Ld["cs"])
        if l == 0:
            Ly.out, Ly.ld_out, Ly.out_col = ops.fptr(Ld["out"].t), Ld["out"].D, Ld["col"]
        else:
            Ly.out, Ly.ld_out, Ly.out_col = ops.fptr(Ld["hout"].t), u, 0
        Ly.state, Ly.h_final, Ly.c_final = ops.fptr(Ld["state"]), ops.fptr(Ld["hf"]), ops.fptr(Ld["cf"])
        Ly.dgates, Ly.dstate = ops.fptr(Ld["dgates"]), ops.fptr(Ld["dstate"])
        if drop:
            Ly.hs_seq = ops.fptr(Ld["hs_seq"].t)
        if backward:
            if l == 0:
                Ly.dout, Ly.ld_dout, Ly.dout_col = ops.fptr(Ld["dy"].t), Ld["dy"].D, Ld["col"]
            else:
                Ly.dout, Ly.ld_dout, Ly.dout_col = ops.fptr(Ld["dhout"].t), u, 0
            if l == E["nplain"] - 1 and not E["attentive"]:
                st.dh_final, st.dc_final = ops.fptr(Ld["dhf"]), (None if self.gru else ops.fptr(Ld["dcf"]))
        return st

    def _highway_x(self, E, d, l):
        """Row view of layer l's RAW input (what the HighwayWrapper carries through): the emitted output of the layer below."""
        Lo = E["layers"][(d, l - 1)]
        return Lo["out"].mat(0, Lo["col"])

    def _encode_highway(self, ws, B):
        cfg = self.cfg
        hw_streams = [s for s in cfg.streams() if cfg.highway(s)]
        nmax = max(ws["enc"][s]["nplain"] for s in hw_streams)
        for l in range(nmax):
            stacks = []
            for s in hw_streams:
                E = ws["enc"][s]
                if l >= E["nplain"]:
                    continue
                T, u, F0 = E["T"], E["units"][l], E["F0"]
                for d in cfg.directions():
                    Ld = E["layers"][(d, l)]
                    in_w = F0 if l == 0 else u
                    x = ops.mat(E["xin0"], F0) if l == 0 else self._highway_x(E, d, l)
                    xin = x
                    if self._sdrop(s):           # DropoutWrapper input mask of this cell
                        xin = ops.mat(E["xd"][d] if l == 0 else Ld["xd"], in_w)
                        ops.dropout_rows(x, xin, B * T, in_w, self.seed, encoder_cell_id(s, d, l) * 4, self._keeps(s)[0], in_w)
                    Wk, G = self.P[self._kn(f"{s}/enc/{d}/l{l}")[0]], self.G
                    ops.gemm(xin, Wk.mat(G * u), ops.mat(Ld["gates"], G * u), B * T, G * u, in_w)
                    if self.gru:                 # candidate kernel's input part, hoisted into the c~ record
                        ops.gemm(xin, self.P[f"{s}/enc/{d}/l{l}/cand_kernel"].mat(u), ops.mat(Ld["cs"], u), B * T, u, in_w)
                    stacks.append(self._rnn_stack_single(ws, s, d, l, B, E["len"]))
            self._run_stacks(stacks, ops.rnn_fwd)
            if l == 0:
                continue
            for s in hw_streams:
                E = ws["enc"][s]
                if l >= E["nplain"]:
                    continue
                T, u = E["T"], E["units"][l]
                for d in cfg.directions():
                    Ld = E["layers"][(d, l)]
                    x = self._highway_x(E, d, l)
                    pre = f"{s}/enc/{d}/l{l}"
                    ops.gemm(x, self.P[pre + "/carry_w"].mat(u), ops.mat(Ld["cpre"], u), B * T, u, u, bias=self._pp(pre + "/carry_b"))
                    ops.highway_fwd(x, Ld["hout"].mat(0), ops.mat(Ld["cpre"], u, T=T, ldo=T * u), Ld["out"].mat(0, Ld["col"]), E["len"], B, T, u)

    def _encode_highway_backward(self, ws, B):
        """Top-down, layer by layer: highway gate backward, one-layer BPTT, weight gradients, input gradient into the layer below."""
        cfg = self.cfg
        hw_streams = [s for s in cfg.streams() if cfg.highway(s)]
        nmax = max(ws["enc"][s]["nplain"] for s in hw_streams)
        for l in reversed(range(nmax)):
            stacks = []
            for s in hw_streams:
                E = ws["enc"][s]
                if l >= E["nplain"]:
                    continue
                T, u = E["T"], E["units"][l]
                for d in cfg.directions():
                    Ld = E["layers"][(d, l)]
                    if l > 0:
                        Lo = E["layers"][(d, l - 1)]
                        x, pre = self._highway_x(E, d, l), f"{s}/enc/{d}/l{l}"
                        cp, dcp = ops.mat(Ld["cpre"], u, T=T, ldo=T * u), ops.mat(Ld["dcpre"], u, T=T, ldo=T * u)
                        dy_below = Lo["dy"].mat(0, Lo["col"])
                        # bidirectional top layers write into column halves of one memory gradient: never accumulate across directions here
                        ops.highway_bwd(x, Ld["hout"].mat(0), cp, Ld["dy"].mat(0, Ld["col"]), Ld["dhout"].mat(0), dcp, dy_below, E["len"], B, T, u,
                                        accumulate_dx=False)
                        self._gemm_tn(x, ops.mat(Ld["dcpre"], u), self.Gr[pre + "/carry_w"].mat(u), u, u, B * T)
                        ops.colsum(ops.mat(Ld["dcpre"], u), B * T, u, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[pre + "/carry_b"].off)
                        ops.gemm(ops.mat(Ld["dcpre"], u), self.P[pre + "/carry_w"].mat(u), dy_below, B * T, u, u, trans_b=1, beta=1.0)
                    stacks.append(self._rnn_stack_single(ws, s, d, l, B, E["len"], backward=True))
            self._run_stacks(stacks, ops.rnn_bwd)
            for s in hw_streams:
                E = ws["enc"][s]
                if l >= E["nplain"]:
                    continue
                T, u, F0 = E["T"], E["units"][l], E["F0"]
                drop = self._sdrop(s)
                for d in cfg.directions():
                    Ld = E["layers"][(d, l)]
                    kname, bname = self._kn(f"{s}/enc/{d}/l{l}")
                    G = self.G
                    Gk, dg = self.Gr[kname], ops.mat(Ld["dgates"], G * u)
                    in_w = F0 if l == 0 else u
                    if l == 0:
                        a_x = ops.mat(E["xd"][d] if drop else E["xin0"], F0)
                    else:
                        a_x = ops.mat(Ld["xd"], u) if drop else self._highway_x(E, d, l)
                    self._gemm_tn(a_x, dg, Gk.mat(G * u), in_w, G * u, B * T)
                    sh = 1 if d == "bw" else -1
                    hrec = Ld["hs_seq"].mat(sh) if drop else (Ld["out"].mat(sh, Ld["col"]) if l == 0 else Ld["hout"].mat(sh))
                    self._gemm_tn(hrec, dg, Gk.mat(G * u, row0=in_w), u, G * u, B * T)
                    ops.colsum(dg, B * T, G * u, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[bname].off)
                    if self.gru:                 # candidate kernel: inputs [x ; r*h]
                        cpre = f"{s}/enc/{d}/l{l}"
                        Gc, dpc = self.Gr[cpre + "/cand_kernel"], ops.mat(Ld["dpc"], u)
                        self._gemm_tn(a_x, dpc, Gc.mat(u), in_w, u, B * T)
                        self._gemm_tn(ops.mat(Ld["rh"], u), dpc, Gc.mat(u, row0=in_w), u, u, B * T)
                        ops.colsum(dpc, B * T, u, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[cpre + "/cand_bias"].off)
                    if l > 0:                    # gradient of the cell's (masked) input -> the layer below's emitted output
                        Lo = E["layers"][(d, l - 1)]
                        dy_below = Lo["dy"].mat(0, Lo["col"])
                        tgt, beta = (ops.mat(Ld["dxtmp"], u), 0.0) if drop else (dy_below, 1.0)
                        ops.gemm(dg, self.P[kname].mat(G * u), tgt, B * T, u, G * u, trans_b=1, beta=beta)
                        if self.gru:
                            ops.gemm(ops.mat(Ld["dpc"], u), self.P[f"{s}/enc/{d}/l{l}/cand_kernel"].mat(u), tgt, B * T, u, u, trans_b=1, beta=1.0)
                        if drop:
                            ops.dropout_rows(ops.mat(Ld["dxtmp"], u), dy_below, B * T, u, self.seed, encoder_cell_id(s, d, l) * 4,
                                             self._keeps(s)[0], u, accumulate=True)

    # ---- sync batch-norm of the encoder inputs across data-parallel ranks (SURVEY 8(e) collective (3)) ----
    def bn_sync_enable(self):
        """Called by DataParallelTrainer when world > 1.  Streams whose input BN is synchronised: the feature inputs.
        (A CNN-fed stream keeps per-rank statistics: its input gradient would need an all-reduce inside BPTT.)
        The per-stream row counts ride at the tail of the first buffer so that one all-reduce carries sums and counts
        (ranks may hold different B and T)."""
        cfg = self.cfg
        streams = [s for s in cfg.streams() if cfg.batch_normalisation and not (s == "video" and self.use_cnn)]
        if not streams:
            return None
        off, n = {}, 0
        for s in streams:
            off[s], n = n, n + cfg.feat(s)
        z = lambda k: torch.zeros(k, dtype=torch.float32, device=self.dev)
        buf = z(n + len(streams))
        self.bn_sync = dict(streams=streams, off=off, sum=buf, sq=z(n), mean=z(n), rows=[buf[n + i:n + i + 1] for i in range(len(streams))])
        return self.bn_sync

    def _fit_width(self, E, x, s):
        """Reference-width features -> the workspace's copy with zero padding columns up to the engine width (config.py `engine()`)."""
        F = E["F"]
        if "xpad" in E and x.shape[-1] != F:
            Ft = self.cfg_tf.feat(s)
            assert x.shape == E["xpad"].shape[:2] + (Ft,) and x.is_contiguous() and x.dtype == torch.float32
            ops.dropout_rows(ops.mat(x, Ft), ops.mat(E["xpad"], F), x.shape[0] * x.shape[1], Ft, None, 0, 1.0, Ft)
            x = E["xpad"]
        return x

    def _bn_sync_x(self, batch, s):
        x = batch.video if s == "video" else batch.audio
        F = self.cfg.feat(s)
        if x.shape[-1] != F:
            B, L = batch.labels.shape
            ws = self._get_ws(B, batch.audio.shape[1] if batch.audio is not None else 0, batch.video.shape[1] if batch.video is not None else 0,
                              L, False)
            x = self._fit_width(ws["enc"][s], x, s)
        assert x.is_contiguous() and x.dtype == torch.float32 and x.shape[-1] == F
        return x, x.shape[0] * x.shape[1], F

    def dp_sync_pack(self, batch):
        """ONE small collective per data-parallel step: this rank's [sum(mask) of the sequence loss, AU frame-unit count | per
        synchronised stream: sum x, sum x^2 (fp64), rows] in one fp64 buffer the trainer all-reduces; dp_sync_unpack() then turns the
        global sums into the operands the step reads (dp_norm; mean / centred squares / rows of the input batch norms)."""
        bs = self.bn_sync
        streams = bs["streams"] if bs else []
        if getattr(self, "_dp_buf", None) is None:
            offs, n = [], 2
            for s in streams:
                offs.append(n)
                n += 2 * self.cfg.feat(s) + 1
            self._dp_buf, self._dp_offs = torch.zeros(n, dtype=torch.float64, device=self.dev), offs
        buf = self._dp_buf
        buf[0:1].copy_(self.local_loss_denominator(batch))
        buf[1:2].copy_(self.local_au_count(batch))
        for i, s in enumerate(streams):
            x, rows, F = self._bn_sync_x(batch, s)
            o = self._dp_offs[i]
            ops.batchnorm_sync_moments(x, rows, F, buf[o:o + 2 * F], self.scratch)
            buf[o + 2 * F:o + 2 * F + 1].fill_(float(rows))
        self.au_scale, self.au_external = 1.0, True
        return buf

    def dp_sync_unpack(self):
        bs = self.bn_sync
        streams = bs["streams"] if bs else []
        jobs = []
        for i, s in enumerate(streams):
            F, o = self.cfg.feat(s), bs["off"][s]
            jobs.append((self._dp_offs[i], F, bs["mean"][o:o + F], bs["sq"][o:o + F], bs["rows"][i]))
        ops.dp_sync_unpack(self._dp_buf, self.dp_norm, jobs)

    def bn_sync_sums(self, batch):
        """Phase 1: local sum over rows of every synchronised stream + its local row count; returns the buffer to all-reduce."""
        bs = self.bn_sync
        for i, s in enumerate(bs["streams"]):
            x, rows, F = self._bn_sync_x(batch, s)
            o = bs["off"][s]
            ops.batchnorm_sync_sum(x, rows, F, bs["sum"][o:o + F], self.scratch)
            bs["rows"][i].fill_(float(rows))
        return bs["sum"]

    def bn_sync_squares(self, batch):
        """Phase 2 (after the all-reduce of phase 1): global mean, local centred squares; returns the buffer to all-reduce."""
        bs = self.bn_sync
        for i, s in enumerate(bs["streams"]):
            x, rows, F = self._bn_sync_x(batch, s)
            o = bs["off"][s]
            ops.batchnorm_sync_sqsum(x, rows, F, bs["sum"][o:o + F], bs["rows"][i], bs["mean"][o:o + F], bs["sq"][o:o + F], self.scratch)
        return bs["sq"]

    def _encode(self, ws, batch: Batch, training: bool):
        cfg, B = self.cfg, ws["B"]
        self._dropping = bool(cfg.use_dropout and training)       # cells.py:46: DropoutWrapper only in the train graph
        stacks = []
        for s in cfg.streams():
            E = ws["enc"][s]
            T, F = E["T"], E["F"]
            x = batch.video if s == "video" else batch.audio
            len_t = batch.video_len if s == "video" else batch.audio_len
            if "cnn" in E:                                           # avsr/avsr.py:684-696: frames -> visual features
                Hh, Ww, Cc = cfg.video_hw
                assert x.shape == (B, T, Hh, Ww, Cc) and x.is_contiguous() and x.dtype == torch.float32
                x = E["cnn"].forward(x.view(B * T, Hh, Ww, Cc), training).view(B, T, F)
            if "cnn" not in E:
                x = self._fit_width(E, x, s)                          # reference-width features: copied next to zero padding columns
            assert x.shape == (B, T, F) and x.is_contiguous() and x.dtype == torch.float32
            E["x"], E["len"] = x, len_t
            if cfg.batch_normalisation:
                sync = getattr(self, "cnn_bn_sync", None) if (training and "cnn" in E) else None
                if sync is not None:
                    # data parallel with sync_cnn_bn: the CNN-fed stream's input batch norm takes the statistics of the GLOBAL batch too --
                    # its input only exists inside the step, so its fp64 moments get their own small all-reduce here
                    if "sync64" not in E:
                        E["sync64"] = torch.zeros(2 * F + 1, dtype=torch.float64, device=self.dev)
                        E["sync_mean"], E["sync_sq"], E["sync_rows"] = (torch.zeros(F, device=self.dev), torch.zeros(F, device=self.dev),
                                                                       torch.zeros(1, device=self.dev))
                    buf = E["sync64"]
                    ops.batchnorm_sync_moments(x, B * T, F, buf[:2 * F], self.scratch)
                    buf[2 * F:2 * F + 1].fill_(float(B * T))
                    sync(buf)
                    ops.dp_sync_unpack(buf, None, [(0, F, E["sync_mean"], E["sync_sq"], E["sync_rows"])])
                    ops.batchnorm_sync_apply(x, E["xn"], B * T, F, self._pp(f"{s}/bn/gamma"), self._pp(f"{s}/bn/beta"),
                                             self._sp(f"{s}/bn/moving_mean"), self._sp(f"{s}/bn/moving_variance"),
                                             E["sync_mean"], E["sync_sq"], E["sync_rows"], E["invstd"])
                    E["mean"] = E["sync_mean"]
                elif training and self.bn_sync is not None and s in self.bn_sync["streams"]:
                    # statistics of the GLOBAL batch: mean / centred squares were all-reduced by the trainer (bn_sync_*)
                    o = self.bn_sync["off"][s]
                    ops.batchnorm_sync_apply(x, E["xn"], B * T, F, self._pp(f"{s}/bn/gamma"), self._pp(f"{s}/bn/beta"),
                                             self._sp(f"{s}/bn/moving_mean"), self._sp(f"{s}/bn/moving_variance"),
                                             self.bn_sync["mean"][o:o + F], self.bn_sync["sq"][o:o + F],
                                             self.bn_sync["rows"][self.bn_sync["streams"].index(s)], E["invstd"])
                    E["mean"] = self.bn_sync["mean"][o:o + F]
                else:
                    E["mean"] = E["mean_own"]
                    ops.batchnorm_fwd(x, E["xn"], B * T, F, self._pp(f"{s}/bn/gamma"), self._pp(f"{s}/bn/beta"),
                                      self._sp(f"{s}/bn/moving_mean"), self._sp(f"{s}/bn/moving_variance"),
                                      E["mean"], E["invstd"], training, self.scratch)
                E["xin"] = E["xn"]
            else:
                E["xin"] = x
            if cfg.instance_normalisation:       # contrib.layers.instance_norm over the time axis (encoder.py:51-55)
                E["in_x"] = E["xin"]
                ops.instnorm_fwd(E["xin"], E["xi"], B, T, F, self._pp(f"{s}/in/gamma"), self._pp(f"{s}/in/beta"), E["in_mean"], E["in_invstd"])
                E["xin"] = E["xi"]
            F0 = E["F0"]
            E["xin0"], E["dxin0"] = E["xin"], E["dxn"]
            if self.n_dense:                     # Dense(units, selu, use_bias=False) stack between BN and the RNN
                a_prev, w_prev = E["xin"], F
                for k, Dn in enumerate(E["dense"]):
                    u = cfg.input_dense_layers[k]
                    ops.gemm(ops.mat(a_prev, w_prev), self.P[f"{s}/dense{k}/kernel"].mat(u), ops.mat(Dn["z"], u), B * T, u, w_prev)
                    ops.selu(Dn["z"], Dn["a"], B * T * u)
                    a_prev, w_prev = Dn["a"], u
                E["xin0"], E["dxin0"] = a_prev, E["dense"][-1]["da"]
            if E["nplain"] == 0 or cfg.highway(s):
                continue
            for d in cfg.directions():
                u0 = E["units"][0]
                W0 = self.P[self._kn(f"{s}/enc/{d}/l0")[0]]
                xin = E["xin0"]
                if self._sdrop(s):               # DropoutWrapper input mask of the layer-0 cell of this direction
                    xin = E["xd"][d]
                    ops.dropout_rows(ops.mat(E["xin0"], F0), ops.mat(xin, F0), B * T, F0, self.seed, encoder_cell_id(s, d, 0) * 4,
                                     self._keeps(s)[0], F0)
                G = self.G
                ops.gemm(ops.mat(xin, F0), W0.mat(G * u0), ops.mat(E["layers"][(d, 0)]["gates"], G * u0), B * T, G * u0, F0)
                if self.gru:                     # candidate kernel's input part, hoisted into the c~ record
                    ops.gemm(ops.mat(xin, F0), self.P[f"{s}/enc/{d}/l0/cand_kernel"].mat(u0), ops.mat(E["layers"][(d, 0)]["cs"], u0), B * T, u0, F0)
                stacks.append(self._rnn_stack(ws, s, d, B, len_t))
        self._run_stacks(stacks, ops.rnn_fwd)
        if any(cfg.highway(s) for s in cfg.streams()):
            self._encode_highway(ws, B)
        for s in cfg.streams():
            E = ws["enc"][s]
            if E["attentive"]:
                self._av_align_forward(ws, batch, training)
                continue
            self._final_state_fwd(ws, s)
            if s == "video" and cfg.regress_aus and training:
                Wau = self.P["video/au/kernel"]
                ops.gemm(E["mem"].mat(), Wau.mat(2), ops.mat(E["au_z"], 2), B * E["T"], 2, E["mem"].D, bias=self._pp("video/au/bias"))
                ops.au_loss(E["au_z"], batch.aus, E["len"], E["au_row"], E["au_dz"], B, E["T"], cfg.au_loss_weight * self.au_scale,
                            total_count=self.au_total if self.au_external else None)

    def persistent_flagged(self):
        """Read-only form of check_persistent(): did a persistent kernel flag the last pass on THIS rank?"""
        return bool(self.persistent_rnn and ops.rnn_persistent_error())

    def check_persistent(self, disable=True, force=False):
        """Synchronise and read the persistent kernels' sticky error word (a bounded device-side wait expired: some
        workgroups were not co-resident).  Returns True if the last results are invalid; the persistent path is then
        switched off so that the caller can simply redo the pass through the per-step launches.  force=True: another
        data-parallel rank flagged its pass -- switch off here as well so that every rank redoes the pass the same way."""
        if not self.persistent_rnn:
            return False
        if not force and not ops.rnn_persistent_error():
            return False
        if disable:
            import warnings
            warnings.warn("avsr_tf1_amd: persistent RNN kernel timed out; falling back to per-step launches")
            self.persistent_rnn = False
            self.fused_decode = False
            ops.rnn_set_persistent(False)
            ops.rnn_persistent_clear()
        return True

    @staticmethod
    def _run_stacks(stacks, fn):
        if not stacks:
            return
        if sum(s.n_layers for s in stacks) <= 8 and len(stacks) <= 4:
            fn(stacks)
            return
        for s in stacks:
            fn([s])

    def _pp(self, name):
        r = self.P[name]
        return r.t[r.off:r.off + r.n]

    def _gp(self, name):
        r = self.Gr[name]
        return r.t[r.off:r.off + r.n]

    def _sp(self, name):
        """The non-trainable buffer `name` (batch-norm moving statistics) as the kernels update it.  While a flagged pass is being REDONE
        (redoing()) the updates go to a scratch copy: every batch norm sits upstream of the persistent kernels, so the flagged pass
        has already applied this step's (valid) update, and a second one would move the averages twice in one step."""
        r = self.S[name]
        if self._redo_pass:
            if self._stats_sink is None:
                self._stats_sink = torch.empty_like(self.stats)
            ops.copy_(self._stats_sink[r.off:r.off + r.n], r.t[r.off:r.off + r.n])      # (the kernels read the old value to blend it)
            return self._stats_sink[r.off:r.off + r.n]
        return r.t[r.off:r.off + r.n]

    def redoing(self):
        """Context of a pass that repeats one whose persistent kernels flagged (trainer / decode redo paths): see _sp."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            self._redo_pass = True
            try:
                yield
            finally:
                self._redo_pass = False
        return ctx()

    def _final_state_fwd(self, ws, s):
        """uni: last layer's (c, h) (decoder_unimodal.py:144-145); bi: Dense on concat fw|bw (encoder.py:133-138)."""
        cfg, B = self.cfg, ws["B"]
        E = ws["enc"][s]
        top = len(E["units"]) - 1
        u, H = E["units"][-1], cfg.decoder_units[0]
        if cfg.encoder_type == "unidirectional":
            Lt = E["layers"][("fw", top)]
            E["c_fin"], E["h_fin"] = (None if self.gru else Lt["cf"]), Lt["hf"]
            return
        for nm, key, dst in ((("proj", "hf", "h_dec"),) if self.gru else (("proj_c", "cf", "c_dec"), ("proj_h", "hf", "h_dec"))):
            Pm = self.P[f"{s}/enc/{nm}"]
            for di, d in enumerate(cfg.directions()):
                ops.gemm(ops.mat(E["layers"][(d, top)][key], u), Pm.mat(H, row0=di * u), ops.mat(E[dst], H), B, H, u,
                         beta=0.0 if di == 0 else 1.0)
        E["c_fin"], E["h_fin"] = (None if self.gru else E["c_dec"]), E["h_dec"]

    def _final_state_bwd(self, ws, s, dc, dh):
        """dc, dh: [B, Hdec] gradient wrt the stream's final (c, h) handed to the decoder."""
        cfg, B = self.cfg, ws["B"]
        E = ws["enc"][s]
        top = len(E["units"]) - 1
        u, H = E["units"][-1], cfg.decoder_units[0]
        if E["attentive"]:
            E["blk"]["dcf_in"], E["blk"]["dhf_in"] = dc, dh
            return
        if cfg.encoder_type == "unidirectional":
            Lt = E["layers"][("fw", top)]
            if not self.gru and dc.data_ptr() != Lt["dcf"].data_ptr():
                ops.copy_(Lt["dcf"], dc)
            if dh.data_ptr() != Lt["dhf"].data_ptr():
                ops.copy_(Lt["dhf"], dh)
            return
        for nm, key, dkey, g in ((("proj", "hf", "dhf", dh),) if self.gru else (("proj_c", "cf", "dcf", dc), ("proj_h", "hf", "dhf", dh))):
            Pm, Gm = self.P[f"{s}/enc/{nm}"], self.Gr[f"{s}/enc/{nm}"]
            for di, d in enumerate(cfg.directions()):
                Lt = E["layers"][(d, top)]
                ops.gemm(ops.mat(g, H), Pm.mat(H, row0=di * u), ops.mat(Lt[dkey], u), B, u, H, trans_b=1)
                ops.gemm(ops.mat(Lt[key], u), ops.mat(g, H), Gm.mat(H, row0=di * u), u, H, B, trans_a=1, beta=1.0)

    def _encode_backward(self, ws, batch: Batch):
        cfg, B = self.cfg, ws["B"]
        self._ensure_gemm_ws()
        # AU loss gradient into the video memory
        if "video" in ws["enc"] and cfg.regress_aus:
            E = ws["enc"]["video"]
            T, D = E["T"], E["mem"].D
            ops.gemm(ops.mat(E["au_dz"], 2), self.P["video/au/kernel"].mat(2), E["dmem"].mat(), B * T, D, 2, trans_b=1, beta=1.0)
            self._gemm_tn(E["mem"].mat(), ops.mat(E["au_dz"], 2), self.Gr["video/au/kernel"].mat(2), D, 2, B * T)
            ops.colsum(ops.mat(E["au_dz"], 2), B * T, 2, self.grads, self.scratch, beta=1.0, out_offset=self.Gr["video/au/bias"].off)
        if cfg.architecture == "av_align":
            self._av_align_backward(ws, batch)       # needs the complete gradient of the audio memory; fills video dmem
        stacks = []
        for s in cfg.streams():
            E = ws["enc"][s]
            if E["nplain"] == 0 or cfg.highway(s):
                continue
            for d in cfg.directions():
                stacks.append(self._rnn_stack(ws, s, d, B, E["len"], backward=True))
        self._run_stacks(stacks, ops.rnn_bwd)
        if any(cfg.highway(s) for s in cfg.streams()):
            self._encode_highway_backward(ws, B)
        for s in cfg.streams():
            E = ws["enc"][s]
            T, F, F0 = E["T"], E["F"], E["F0"]
            first = True
            for d in cfg.directions():
                i = F0
                for l in range(0 if cfg.highway(s) else E["nplain"]):
                    u = E["units"][l]
                    Ld = E["layers"][(d, l)]
                    kname, bname = self._kn(f"{s}/enc/{d}/l{cfg.shared_layer(s, l)}")
                    Gk = self.Gr[kname]
                    G = self.G
                    dg = ops.mat(Ld["dgates"], G * u)
                    drop = self._sdrop(s)
                    if l == 0:
                        a_x = ops.mat(E["xd"][d] if drop else E["xin0"], F0)
                    elif drop:
                        a_x = E["layers"][(d, l - 1)]["xt_seq"].mat(0)
                    else:
                        a_x = E["layers"][(d, l - 1)]["out"].mat(0, E["layers"][(d, l - 1)]["col"])
                    nct = (G * u + 127) // 128
                    gt = (((i + 127) // 128) + ((u + 127) // 128)) * nct if not self.gru else None     # tiles of the launch below
                    with ops.gemm_group():       # the row blocks of one layer's kernel gradient(s): independent, one launch
                        self._gemm_tn(a_x, dg, Gk.mat(G * u), i, G * u, B * T, group_tiles=gt)
                        sh = 1 if d == "bw" else -1
                        a_h = Ld["hs_seq"].mat(sh) if (drop or (cfg.residual(s) and l > 0)) else Ld["out"].mat(sh, Ld["col"])
                        # (the bias gradient = column sums of d gates rides in this launch: the tiles of dg pass through it anyway)
                        self._gemm_tn(a_h, dg, Gk.mat(G * u, row0=i), u, G * u, B * T, colsum=(self.grads, self.Gr[bname].off), group_tiles=gt)
                        if self.gru:                 # candidate kernel: inputs [x ; r*h]
                            Gc = self.Gr[f"{s}/enc/{d}/l{cfg.shared_layer(s, l)}/cand_kernel"]
                            dpc = ops.mat(Ld["dpc"], u)
                            self._gemm_tn(a_x, dpc, Gc.mat(u), i, u, B * T)
                            self._gemm_tn(ops.mat(Ld["rh"], u), dpc, Gc.mat(u, row0=i), u, u, B * T)
                            ops.colsum(dpc, B * T, u, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[f"{s}/enc/{d}/l{cfg.shared_layer(s, l)}/cand_bias"].off)
                    i = u
                if E["nplain"] > 0 and (cfg.batch_normalisation or "cnn" in E or self.n_dense or cfg.instance_normalisation):
                    u0, G = E["units"][0], self.G
                    W0 = self.P[self._kn(f"{s}/enc/{d}/l0")[0]]
                    L0 = E["layers"][(d, 0)]
                    tgt = E["dx_tmp"] if self._sdrop(s) else E["dxin0"]
                    beta0 = 0.0 if (self._sdrop(s) or first) else 1.0
                    ops.gemm(ops.mat(L0["dgates"], G * u0), W0.mat(G * u0), ops.mat(tgt, F0), B * T, F0, G * u0, trans_b=1, beta=beta0)
                    if self.gru:
                        ops.gemm(ops.mat(L0["dpc"], u0), self.P[f"{s}/enc/{d}/l0/cand_kernel"].mat(u0), ops.mat(tgt, F0), B * T, F0, u0,
                                 trans_b=1, beta=1.0)
                    if self._sdrop(s):
                        ops.dropout_rows(ops.mat(E["dx_tmp"], F0), ops.mat(E["dxin0"], F0), B * T, F0, self.seed,
                                         encoder_cell_id(s, d, 0) * 4, self._keeps(s)[0], F0, accumulate=not first)
                    first = False
            if self.n_dense:
                # back through the input Dense stack: d z = d a * selu'(z);  d W += a_prev^T d z;  d a_prev = d z W^T
                for k in reversed(range(self.n_dense)):
                    Dn, u = E["dense"][k], cfg.input_dense_layers[k]
                    a_prev, w_prev = (E["dense"][k - 1]["a"], cfg.input_dense_layers[k - 1]) if k else (E["xin"], F)
                    ops.selu_bwd(Dn["z"], Dn["da"], Dn["z"], B * T * u)          # in place: z is not needed again
                    self._gemm_tn(ops.mat(a_prev, w_prev), ops.mat(Dn["z"], u), self.Gr[f"{s}/dense{k}/kernel"].mat(u), w_prev, u, B * T)
                    g_prev = E["dense"][k - 1]["da"] if k else E["dxn"]
                    ops.gemm(ops.mat(Dn["z"], u), self.P[f"{s}/dense{k}/kernel"].mat(u), ops.mat(g_prev, w_prev), B * T, w_prev, u, trans_b=1)
            if cfg.instance_normalisation:       # dxn holds d(instance-norm output): turn it into d(input) in place
                ops.instnorm_bwd(E["in_x"], E["dxn"], self._pp(f"{s}/in/gamma"), E["in_mean"], E["in_invstd"], E["dxn"], E["in_dg"], E["in_db"],
                                 B, T, F)
                ops.colsum(ops.mat(E["in_dg"], F), B, F, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[f"{s}/in/gamma"].off)
                ops.colsum(ops.mat(E["in_db"], F), B, F, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[f"{s}/in/beta"].off)
            if cfg.batch_normalisation:
                # (a 1-layer attentive encoder wrote dxn in _av_align_backward)
                ops.batchnorm_xhat(E["x"], E["mean"], E["invstd"], E["xhat"], B * T, F)
                ops.colsum(ops.mat(E["dxn"], F), B * T, F, self.grads, self.scratch, b=ops.mat(E["xhat"], F), beta=1.0,
                           out_offset=self.Gr[f"{s}/bn/gamma"].off)
                ops.colsum(ops.mat(E["dxn"], F), B * T, F, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[f"{s}/bn/beta"].off)
            if "cnn" in E:                       # gradient wrt the visual features, then through the CNN
                sync = getattr(self, "cnn_bn_sync", None)
                if cfg.batch_normalisation and sync is not None:
                    # sync_cnn_bn: dx = gamma*invstd*(dy - mean(dy) - xhat*mean(dy*xhat)) with the means over the GLOBAL batch.  This rank's
                    # sums are what the two column-sum launches above left in the gradient buffer (d beta | d gamma, accumulated from
                    # zero); their all-reduced copy gives the three coefficient vectors of dx = k1*dy + k2*x + k3 (a handful of [F]-sized
                    # torch ops: this mode launches eagerly)
                    ops.colsum_batch_flush(self.scratch)      # (the column sums of a half-pass are collected: run the ones queued so far)
                    ob, og = self.Gr[f"{s}/bn/beta"].off, self.Gr[f"{s}/bn/gamma"].off
                    red = torch.cat([self.grads[ob:ob + F], self.grads[og:og + F], torch.full((1,), float(B * T), device=self.dev)]).to(torch.float64)
                    sync(red)
                    n = red[2 * F]
                    g64, is64, m64 = (self.params[self.P[f"{s}/bn/gamma"].off:self.P[f"{s}/bn/gamma"].off + F].to(torch.float64),
                                      E["invstd"].to(torch.float64), E["mean"].to(torch.float64))
                    a, b = red[:F] / n, red[F:2 * F] / n
                    E["bn_k"] = torch.cat([g64 * is64, -g64 * is64 * is64 * b, -g64 * is64 * a + g64 * is64 * is64 * b * m64]).to(torch.float32).contiguous()
                    ops.bn_bwd_apply(E["dxn"], E["x"], E["bn_k"], E["dfeat"], B * T, F)
                    E["cnn"].backward(E["dfeat"])
                elif cfg.batch_normalisation:
                    ops.batchnorm_bwd(E["x"], E["dxn"], self._pp(f"{s}/bn/gamma"), self._pp(f"{s}/bn/beta"), E["mean"], E["invstd"],
                                      E["dfeat"], None, None, B * T, F, 0, self.scratch)
                    E["cnn"].backward(E["dfeat"])
                else:
                    E["cnn"].backward(E["dxn"])

    def _ensure_gemm_ws(self):
        if self.gemm_ws is None:
            self.gemm_ws = torch.empty(48 << 20, device=self.dev)
        ops.set_gemm_workspace(self.gemm_ws)

    def _gemm_tn(self, A, Bm, Cm, M, N, K, beta=1.0, colsum=None, group_tiles=None):
        """C (+)= A^T B with K = number of (b,t) rows: split-K so the small M x N output still fills the chip.
        colsum=(tensor, offset): the column sums of B (the layer's bias gradient) are accumulated there by the same launch.
        group_tiles: 128 x 128 output tiles of ALL the GEMMs this one shares a grouped launch with (itself included): the K split is
        then chosen for the launch, not the GEMM -- two 256 x 1024 x 32000 gradients split 48 ways each are 1536 workgroups = two uneven
        rounds of the chip's 768 slots (361 us measured); split 24 ways they are one round (294 us)."""
        sk = _splitk(M, N, K)
        if group_tiles and K >= 16384:
            sk = max(2, min(768 // int(group_tiles), K // 256))
        elif group_tiles and K >= 1024:             # (the decoder block's B * L rows: at least four 16-deep K tiles per slice)
            sk = max(1, min(768 // int(group_tiles), K // 64))
        while sk > 1 and sk * (M * N + (N if colsum is not None else 0)) > self.gemm_ws.numel():
            sk //= 2
        ops.gemm(A, Bm, Cm, M, N, K, trans_a=1, beta=beta, splitk=sk, workspace=self.gemm_ws, colsum=colsum, colsum_beta=1.0)

    # ------------------------------------------------------------------------------------------------
    # attention-wrapped LSTM block (decoder, AV-Align top layer)
    def _mem_desc(self, ws, stream):
        """(values SeqBuf-like view, grad view, len) of a stream's encoder memory as seen by attention."""
        E = ws["enc"][stream]
        if not E["attentive"]:
            return dict(t=E["mem"].t, off=E["mem"].off(), sb=E["mem"].sb, st=E["mem"].st, vmat=E["mem"].mat(),
                        gt=E["dmem"].t, goff=E["dmem"].off(), gsb=E["dmem"].sb, gmat=E["dmem"].mat(), len=E["len"])
        blk = E["blk"]
        if blk["mems"][0]["type"] in LUONG_TYPES:     # encoder output = attention vector (output_attention=True)
            buf, g = blk["att"], blk["datt_ext"]
        else:
            buf, g = blk["cell_out"], blk["dcell_ext"]
        D = buf.D
        return dict(t=buf.t, off=buf.off(), sb=buf.sb, st=buf.st, vmat=buf.mat(), gt=g, goff=0, gsb=buf.T * D,
                    gmat=ops.mat(g, D), len=E["len"])

    def _block_desc(self, ws, blk, steplen, mode, h0, c0, with_bwd):
        cfg = self.cfg
        B, L, H, E, A = blk["B"], blk["L"], blk["H"], blk["E"], blk["A"]
        d = AttnRnn()
        d.B, d.L, d.H, d.E, d.n_mech, d.V, d.mode = B, L, H, E, len(blk["mems"]), cfg.vocab_size, mode
        d.go_id, d.eos_id = cfg.go_id, cfg.eos_id
        d.steplen = ops.fptr(steplen)
        kname, bname = self._kn(blk["cell"])
        d.wt, d.w = ops.fptr(self.derived, self.Tr[kname].off), ops.fptr(self.params, self.P[kname].off)
        d.bias = ops.fptr(self.params, self.P[bname].off)
        if self.gru:
            cn = blk["cell"] + "/cand_kernel"
            d.cell = 1
            d.wt2, d.w2 = ops.fptr(self.derived, self.Tr[cn].off), ops.fptr(self.params, self.P[cn].off)
            d.bias2 = ops.fptr(self.params, self.P[blk["cell"] + "/cand_bias"].off)
            d.rh_seq, d.dgates2 = ops.fptr(blk["rh"]), ops.fptr(blk["dpc"])
        d.gates, d.cs, d.cell_out = ops.fptr(blk["gates"]), ops.fptr(blk["cs"]), ops.fptr(blk["cell_out"].t)
        d.att = ops.fptr(blk["att"].t) if A else None
        d.h0, d.c0, d.state = ops.fptr(h0), ops.fptr(c0), ops.fptr(blk["state"])
        d.h_final, d.c_final = ops.fptr(blk["hf"]), ops.fptr(blk["cf"])
        for i, m in enumerate(blk["mems"]):
            md = self._mem_desc(ws, m["stream"])
            M = d.mech[i]
            pre = m["prefix"]
            M.type, M.T, M.D, M.chunk = ATT_CODE[m["type"]], m["T"], m["Dv"], m["chunk"]
            M.len, M.keys = ops.fptr(md["len"]), ops.fptr(m["keys"])
            if m["proj"]:
                M.values, M.values_sb, M.values_st = ops.fptr(m["pvals"]), m["T"] * H, H
            else:
                M.values, M.values_sb, M.values_st = ops.fptr(md["t"], md["off"]), md["sb"], md["st"]
            if m["type"] == "scaled_luong":
                M.g = ops.fptr(self.params, self.P[pre + "/g"].off)
            if m["type"] in BAHDANAU_TYPES:
                if m["type"] == "normed_bahdanau":
                    M.v, M.bq = ops.fptr(m["vn"]), ops.fptr(self.params, self.P[pre + "/b"].off)
                else:
                    M.v = ops.fptr(self.params, self.P[pre + "/v"].off)
                M.wq_t = ops.fptr(self.derived, self.Tr[pre + "/query_kernel"].off)
                M.wq = ops.fptr(self.params, self.P[pre + "/query_kernel"].off)
                M.pq, M.dpq = ops.fptr(m["pq"]), ops.fptr(m["dpq"])
            if m["proj"]:
                M.watt_t, M.watt = ops.fptr(m["watt_p_t"]), ops.fptr(m["watt_p"])
            else:
                M.watt_t = ops.fptr(self.derived, self.Tr[pre + "/layer_kernel"].off)
                M.watt = ops.fptr(self.params, self.P[pre + "/layer_kernel"].off)
            M.scores, M.ctx, M.pstat, M.pctx = ops.fptr(m["scores"]), ops.fptr(m["ctx"]), ops.fptr(m["pstat"]), ops.fptr(m["pctx"])
            if with_bwd:
                M.dscores, M.dctx, M.pdq = ops.fptr(m["dscores"]), ops.fptr(m["dctx"]), ops.fptr(m["pdq"])
        if self._bdrop(blk) and mode != 1:
            keep = blk["keep"]
            d.seed = ops.fptr(self.seed)
            d.keep_in, d.keep_state, d.keep_out = keep
            d.cell_id = blk["cell_id"]
            d.hs_seq = ops.fptr(blk["hs_seq"].t)
            d.attd = ops.fptr(blk["attd"].t) if A else None
        if with_bwd:
            d.dgates, d.dstate, d.dq = ops.fptr(blk["dgates"]), ops.fptr(blk["dstate"]), ops.fptr(blk["dq"])
            d.datt = ops.fptr(blk["datt"]) if A else None
            d.dh0, d.dc0 = ops.fptr(blk["dh0"]), ops.fptr(blk["dc0"])
        if blk.get("fused_ws") is not None and self.fused_decode:
            d.fused_ws, d.fused_ws_floats = ops.fptr(blk["fused_ws"]), blk["fused_ws"].numel()
        d.n_extra = len(blk["extra"])
        if d.n_extra:
            d.out0 = ops.fptr(blk["out0"].t)
        for j, X in enumerate(blk["extra"]):
            Xd = d.extra[j]
            kn, bn = self._kn(X["prefix"])
            Xd.wt, Xd.w, Xd.bias = ops.fptr(self.derived, self.Tr[kn].off), ops.fptr(self.params, self.P[kn].off), ops.fptr(self.params, self.P[bn].off)
            Xd.gates, Xd.cs, Xd.out, Xd.state = ops.fptr(X["gates"]), ops.fptr(X["cs"]), ops.fptr(X["out"].t), ops.fptr(X["state"])
            Xd.cell_id = X["cell_id"]
            if self._bdrop(blk) and mode != 1:
                Xd.hs_seq, Xd.xin_seq = ops.fptr(X["hs_seq"].t), ops.fptr(X["xin_seq"].t)
            if with_bwd:
                Xd.dgates, Xd.dstate = ops.fptr(X["dgates"]), ops.fptr(X["dstate"])
        return d

    def _block_prepare(self, ws, blk):
        """Per-batch attention memory preparation: keys = values . W_mem (attention.py memory_layer)."""
        B, H = blk.get("mem_B", blk["B"]), blk["H"]
        with ops.gemm_group():                   # the memories' GEMMs are independent of each other: one launch
            for m in blk["mems"]:
                md = self._mem_desc(ws, m["stream"])
                pre = m["prefix"]
                ops.gemm(md["vmat"], self.P[pre + "/memory_kernel"].mat(H), ops.mat(m["keys"], H), B * m["T"], H, m["D"])
                if m["proj"]:
                    Wl = self.P[pre + "/layer_kernel"]
                    ops.gemm(md["vmat"], Wl.mat(H, row0=H), ops.mat(m["pvals"], H), B * m["T"], H, m["D"])          # pvals = values . W_ctx
                    ops.copy_(m["watt_p"].view(-1)[:H * H], Wl.t[Wl.off:Wl.off + H * H])                              # [W_h ; I]
                    ops.gemm(Wl.mat(H), ops.mat(m["eye"], H), ops.mat(m["watt_p_t"], 2 * H), H, H, H, trans_a=1)      # [W_h^T | I]
                if m["type"] == "normed_bahdanau":
                    ops.normed_v(self._pp(pre + "/v"), self._pp(pre + "/g"), m["vn"], H)

    def _block_backward(self, ws, blk, desc, xin_mat, dxin_mat, dxin_beta, out_att):
        """attention-RNN BPTT + every deferred (post-loop) gradient GEMM of the block.
        xin_mat: Mat over the [B*L, E] hoisted inputs; dxin_mat: where d(inputs) goes (or None)."""
        cfg = self.cfg
        B, L, H, E, A = blk["B"], blk["L"], blk["H"], blk["E"], blk["A"]
        self._ensure_gemm_ws()
        ops.attn_rnn_bwd(desc)
        rows = B * L
        co = blk["cell_out"]
        mems = list(enumerate(blk["mems"]))
        # The per-memory gradient GEMMs are small (a few workgroups each) and independent across memories: they are issued in phases,
        # every phase ONE grouped launch (ops.gemm_group): attention-layer kernels | alignments (kernels) | d values, d keys |
        # memory-layer gradients.  Two GEMMs that accumulate into the same matrix never share a phase.
        with ops.gemm_group():
            for i, m in mems:
                pre, D = m["prefix"], m["D"]
                datt_m = ops.mat(blk["datt"], A, offset=i * H)
                Gl = self.Gr[pre + "/layer_kernel"]
                self._gemm_tn(co.mat(0), datt_m, Gl.mat(H), H, H, rows)                 # rows 0..H: cell_out part
                if not m["proj"]:
                    self._gemm_tn(ops.mat(m["ctx"], D), datt_m, Gl.mat(H, row0=H), D, H, rows)   # rows H..H+D: context part
        for i, m in mems:
            pre, T = m["prefix"], m["T"]
            md = self._mem_desc(ws, m["stream"])
            luong = m["type"] in LUONG_TYPES
            g_t = self._pp(pre + "/g") if m["type"] == "scaled_luong" else None
            # scores -> alpha (in place); rowdot = sum_t ds * raw  (d g for scaled_luong)
            ops.attn_alpha_rows(m["scores"], m["dscores"], md["len"], desc_steplen(desc), g_t if luong else None, m["rowdot"], B, L, T)
            if m["type"] == "scaled_luong":
                ops.reduce_scalar(m["rowdot"], rows, self.grads, accumulate=True, out_offset=self.Gr[pre + "/g"].off)
        with ops.gemm_group():
            for i, m in mems:
                pre, T, D = m["prefix"], m["T"], m["D"]
                md = self._mem_desc(ws, m["stream"])
                if m["proj"]:
                    # d pvals[b,t,:] = sum_l alpha[b,l,t] * dctx'[b,l,:]
                    ops.gemm(ops.mat(m["scores"], T), ops.mat(m["dctx"], H), ops.mat(m["dpvals"], H), T, H, L,
                             trans_a=1, batch=B, strides=(L * T, L * H, T * H))
                else:
                    # d values[b,t,:] += sum_l alpha[b,l,t] * dctx[b,l,:]        (batched over b)
                    ops.gemm(ops.mat(m["scores"], T), ops.mat(m["dctx"], D), ops.mat(md["gt"], md["st"], offset=md["goff"]), T, D, L,
                             trans_a=1, beta=1.0, batch=B, strides=(L * T, L * D, md["gsb"]))
                if m["type"] in LUONG_TYPES:
                    # d keys[b,t,:] = g * sum_l ds[b,l,t] * cell_out[b,l,:]
                    g_t = self._pp(pre + "/g") if m["type"] == "scaled_luong" else None
                    ops.gemm(ops.mat(m["dscores"], T), ops.mat(co.t, H, offset=co.off(0)), ops.mat(m["dkeys"], H), T, H, L,
                             trans_a=1, batch=B, strides=(L * T, co.sb, T * H), alpha_dev=g_t)
        for i, m in mems:
            if m["type"] in LUONG_TYPES:
                continue
            pre, T = m["prefix"], m["T"]
            md = self._mem_desc(ws, m["stream"])
            v_t = m["vn"] if m["type"] == "normed_bahdanau" else self._pp(pre + "/v")
            bq = self._pp(pre + "/b") if m["type"] == "normed_bahdanau" else None
            ops.bahdanau_dkeys(m["keys"], m["pq"], L * H, H, m["dscores"], v_t, bq, md["len"], m["dkeys"], m["dv_part"], B, L, T, H)
            nblk = m["dv_part"].shape[0]
            if m["type"] == "normed_bahdanau":
                ops.colsum(ops.mat(m["dv_part"], H), nblk, H, m["dvn"], self.scratch)
                ops.normed_v_bwd(self._pp(pre + "/v"), self._pp(pre + "/g"), m["dvn"], self._gp(pre + "/v"), self._gp(pre + "/g"), H)
                ops.colsum(ops.mat(m["dpq"], H), rows, H, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[pre + "/b"].off)
            else:
                ops.colsum(ops.mat(m["dv_part"], H), nblk, H, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[pre + "/v"].off)
        with ops.gemm_group():
            for i, m in mems:
                pre, D = m["prefix"], m["D"]
                md = self._mem_desc(ws, m["stream"])
                if m["proj"]:
                    # d W_ctx = values^T . d pvals;  d values += d pvals . W_ctx^T
                    Wl, Gl = self.P[pre + "/layer_kernel"], self.Gr[pre + "/layer_kernel"]
                    self._gemm_tn(md["vmat"], ops.mat(m["dpvals"], H), Gl.mat(H, row0=H), D, H, B * m["T"])
                    ops.gemm(ops.mat(m["dpvals"], H), Wl.mat(H, row0=H), md["gmat"], B * m["T"], D, H, trans_b=1, beta=1.0)
                if m["type"] not in LUONG_TYPES:
                    self._gemm_tn(co.mat(0), ops.mat(m["dpq"], H), self.Gr[pre + "/query_kernel"].mat(H), H, H, rows)
                # memory_layer: d W_mem = values^T . d keys
                self._gemm_tn(md["vmat"], ops.mat(m["dkeys"], H), self.Gr[pre + "/memory_kernel"].mat(H), D, H, B * m["T"])
        with ops.gemm_group():
            for i, m in mems:                     # memory_layer: d values += d keys . W_mem^T (after the projected-context term above)
                pre, D = m["prefix"], m["D"]
                md = self._mem_desc(ws, m["stream"])
                ops.gemm(ops.mat(m["dkeys"], H), self.P[pre + "/memory_kernel"].mat(H), md["gmat"], B * m["T"], D, H, trans_b=1, beta=1.0)
        # cell kernel: rows [0:E] inputs, [E:E+A] previous attention, [E+A:] previous h
        kname, bname = self._kn(blk["cell"])
        Gk, G = self.Gr[kname], self.G
        dg = ops.mat(blk["dgates"], G * H)
        drop = self._bdrop(blk)
        a_att = (blk["attd"] if drop else blk["att"]).mat(-1) if A else None
        out0 = blk["out0"] if blk["extra"] else co               # output record of the attention-fed layer
        a_h = (blk["hs_seq"] if drop else out0).mat(-1)
        below = out0
        for X in blk["extra"]:                                    # MultiRNNCell layers above: kernel rows [0:H] input, [H:2H] previous h
            kx, bx = self._kn(X["prefix"])
            dgx = ops.mat(X["dgates"], 4 * H)
            self._gemm_tn((X["xin_seq"] if drop else below).mat(0), dgx, self.Gr[kx].mat(4 * H), H, 4 * H, rows)
            self._gemm_tn((X["hs_seq"] if drop else X["out"]).mat(-1), dgx, self.Gr[kx].mat(4 * H, row0=H), H, 4 * H, rows)
            ops.colsum(dgx, rows, 4 * H, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[bx].off)
            below = X["out"]
        nct = (G * H + 127) // 128
        gt = (((E + 127) // 128) + ((A + 127) // 128 if A else 0) + ((H + 127) // 128)) * nct if not self.gru else None
        with ops.gemm_group():                   # the row blocks of the cell kernel's gradient and d inputs: independent
            self._gemm_tn(xin_mat, dg, Gk.mat(G * H), E, G * H, rows, group_tiles=gt)
            if A:
                self._gemm_tn(a_att, dg, Gk.mat(G * H, row0=E), A, G * H, rows, group_tiles=gt)
            self._gemm_tn(a_h, dg, Gk.mat(G * H, row0=E + A), H, G * H, rows, group_tiles=gt)
            ops.colsum(dg, rows, G * H, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[bname].off)
            if dxin_mat is not None and not self.gru:
                ops.gemm(dg, self.P[kname].mat(G * H), dxin_mat, rows, E, G * H, trans_b=1, beta=dxin_beta)
        if dxin_mat is not None and self.gru:
            ops.gemm(dg, self.P[kname].mat(G * H), dxin_mat, rows, E, G * H, trans_b=1, beta=dxin_beta)
        if self.gru:                             # candidate kernel: inputs [x ; attention ; r*h]
            cn = blk["cell"] + "/cand_kernel"
            Gc, dpc = self.Gr[cn], ops.mat(blk["dpc"], H)
            self._gemm_tn(xin_mat, dpc, Gc.mat(H), E, H, rows)
            if A:
                self._gemm_tn(a_att, dpc, Gc.mat(H, row0=E), A, H, rows)
            self._gemm_tn(ops.mat(blk["rh"], H), dpc, Gc.mat(H, row0=E + A), H, H, rows)
            ops.colsum(dpc, rows, H, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[blk["cell"] + "/cand_bias"].off)
            if dxin_mat is not None:
                ops.gemm(dpc, self.P[cn].mat(H), dxin_mat, rows, E, H, trans_b=1, beta=1.0)

    # ------------------------------------------------------------------------------------------------
    # AV-Align: attention-wrapped top audio layer over the video memory (encoder.py:265-290)
    def _av_align_forward(self, ws, batch, training):
        cfg, B = self.cfg, ws["B"]
        E = ws["enc"]["audio"]
        blk = E["blk"]
        T, H, Ein = E["T"], blk["H"], blk["E"]
        kname = self._kn(blk["cell"])[0]
        if self._sdrop("audio") and E["nplain"] == 0:
            ops.dropout_rows(ops.mat(E["xin0"], E["F0"]), ops.mat(E["xd"]["fw"], E["F0"]), B * T, E["F0"], self.seed, blk["cell_id"] * 4,
                             blk["keep"][0], Ein + blk["A"])
        xin = self._av_xin(E)
        ops.gemm(xin, self.P[kname].mat(self.G * H), ops.mat(blk["gates"], self.G * H), B * T, self.G * H, Ein)
        if self.gru:
            ops.gemm(xin, self.P[blk["cell"] + "/cand_kernel"].mat(H), ops.mat(blk["cs"], H), B * T, H, Ein)
        self._block_prepare(ws, blk)
        blk["desc"] = self._block_desc(ws, blk, E["len"], 0, None, None, with_bwd=training)
        blk["desc"].prof_tag = 1                 # timed as the attentive encoder layer, not as a decoder (bench.py roofline classes)
        ops.attn_rnn_fwd(blk["desc"], 0, T)
        E["c_fin"], E["h_fin"] = (None if self.gru else blk["cf"]), blk["hf"]

    def _av_xin(self, E):
        """Hoisted input of the attention-wrapped layer (already carrying that cell's input mask under dropout)."""
        if E["nplain"] == 0:
            return ops.mat(E["xd"]["fw"] if self._sdrop("audio") else E["xin0"], E["F0"])
        Ld = E["layers"][("fw", E["nplain"] - 1)]
        return (Ld["xt_seq"] if self._sdrop("audio") else Ld["out"]).mat(0)

    def _av_align_backward(self, ws, batch):
        cfg, B = self.cfg, ws["B"]
        E = ws["enc"]["audio"]
        blk = E["blk"]
        d = blk["desc"]
        luong = blk["mems"][0]["type"] in LUONG_TYPES
        d.datt_ext = ops.fptr(blk["datt_ext"]) if luong else None
        d.dcell_ext = None if luong else ops.fptr(blk["dcell_ext"])
        d.dh_final, d.dc_final = ops.fptr(blk["dhf_in"]), ops.fptr(blk["dcf_in"])
        if E["nplain"] == 0:
            dxin, beta = ops.mat(E["dx_tmp"] if self._sdrop("audio") else E["dxin0"], E["F0"]), 0.0
        else:
            Ld = E["layers"][("fw", E["nplain"] - 1)]
            dxin, beta = Ld["dout"].mat(0), 0.0
        self._block_backward(ws, blk, d, self._av_xin(E), dxin, beta, luong)
        if self._sdrop("audio"):                 # gradient of the DROPPED input -> gradient of the layer below's output
            keep, W = blk["keep"][0], blk["E"] + blk["A"]
            if E["nplain"] == 0:
                ops.dropout_rows(dxin, ops.mat(E["dxin0"], E["F0"]), B * E["T"], E["F0"], self.seed, blk["cell_id"] * 4, keep, W)
            else:
                ops.dropout_rows(dxin, dxin, B * E["T"], blk["E"], self.seed, blk["cell_id"] * 4, keep, W)

    # ------------------------------------------------------------------------------------------------
    # decoder
    def _decoder_init_state(self, ws):
        """unimodal / av_align: encoder final (c,h) used directly; bimodal: ONE shared Dense on concat c and on
        concat h (decoder_bimodal.py:480-490); a missing stream contributes zeros (:129-142)."""
        cfg, B = self.cfg, ws["B"]
        D = ws["dec"]
        H = cfg.decoder_units[0]
        if cfg.architecture == "lm":                         # lm.py:352-353: MultiRNNCell.zero_state
            if "h0buf" not in D:
                D["c0buf"], D["h0buf"] = torch.zeros(B, H, device=self.dev), torch.zeros(B, H, device=self.dev)
            D["h0"], D["c0"] = D["h0buf"], (None if self.gru else D["c0buf"])
            return
        if cfg.architecture != "bimodal":
            s = "audio" if "audio" in ws["enc"] else "video"
            E = ws["enc"][s]
            D["h0"], D["c0"] = E["h_fin"], E["c_fin"]        # GRU: c_fin is None (state = h only)
            return
        if "c0buf" not in D:
            D["c0buf"], D["h0buf"] = torch.zeros(B, H, device=self.dev), torch.zeros(B, H, device=self.dev)
        SP = self.P["dec/state_proj"]
        first = True
        for si, s in enumerate(("video", "audio")):           # per stream ONE launch for (c, h); the second stream accumulates
            if s not in ws["enc"]:
                continue
            with ops.gemm_group():
                for key, dst in (("c_fin", "c0buf"), ("h_fin", "h0buf")):
                    ops.gemm(ops.mat(ws["enc"][s][key], H), SP.mat(H, row0=si * H), ops.mat(D[dst], H), B, H, H, beta=0.0 if first else 1.0)
            first = False
        D["h0"], D["c0"] = D["h0buf"], D["c0buf"]

    def _decoder_init_state_bwd(self, ws):
        cfg, B = self.cfg, ws["B"]
        D = ws["dec"]
        H = cfg.decoder_units[0]
        if cfg.architecture == "lm":
            return
        if cfg.architecture != "bimodal":
            s = "audio" if "audio" in ws["enc"] else "video"
            self._final_state_bwd(ws, s, D["dc0"], D["dh0"])
            return
        SP, GSP = self.P["dec/state_proj"], self.Gr["dec/state_proj"]
        present = [(si, s) for si, s in enumerate(("video", "audio")) if s in ws["enc"]]
        tgt = {}
        for si, s in present:                     # plain unidirectional encoders of the decoder's width: the products land where the encoder
            E = ws["enc"][s]                      # BPTT reads its final-state gradient (no copy launches behind them)
            direct = (not E["attentive"]) and cfg.encoder_type == "unidirectional" and E["units"][-1] == H
            Lt = E["layers"][("fw", len(E["units"]) - 1)] if direct else None
            tgt[s] = (Lt["dcf"] if (direct and not self.gru) else E["dc_dec"], Lt["dhf"] if direct else E["dh_dec"])
        with ops.gemm_group():                    # d (c, h) of every stream: independent
            for si, s in present:
                for key, g, dst in (("c_fin", D["dc0"], tgt[s][0]), ("h_fin", D["dh0"], tgt[s][1])):
                    ops.gemm(ops.mat(g, H), SP.mat(H, row0=si * H), ops.mat(dst, H), B, H, H, trans_b=1)
        for key, g in (("c_fin", D["dc0"]), ("h_fin", D["dh0"])):      # the c and the h term of a stream accumulate into the same rows
            with ops.gemm_group():
                for si, s in present:
                    ops.gemm(ops.mat(ws["enc"][s][key], H), ops.mat(g, H), GSP.mat(H, row0=si * H), H, H, B, trans_a=1, beta=1.0)
        for si, s in present:
            self._final_state_bwd(ws, s, tgt[s][0], tgt[s][1])

    def _out_vec(self, D):
        """what the output Dense consumes: attention (Luong family) or the cell output (Bahdanau family)."""
        if self.cfg.output_attention():
            return D["att"].mat(0), D["A"]
        return D["cell_out"].mat(0), D["H"]

    # ------------------------------------------------------------------------------------------------
    # public API
    def forward_train(self, batch: Batch, compute_denom=True):
        """Train-graph forward: encoders, teacher-forced decoder, logits, loss (stays on device)."""
        cfg = self.cfg
        B, L = batch.labels.shape
        Ta = batch.audio.shape[1] if batch.audio is not None else 0
        Tv = batch.video.shape[1] if batch.video is not None else 0
        ws = self._get_ws(B, Ta, Tv, L, False)
        self._cur = (ws, batch)
        ops.colsum_batch_abort()                 # (a backward pass that raised half-way must not leave its collection open)
        self._refresh_derived()
        ops.add_int(self.step, int(self.seed_offset), self.seed)
        self._encode(ws, batch, True)
        D = ws["dec"]
        H, E, V = D["H"], D["E"], cfg.vocab_size
        self._decoder_init_state(ws)
        sampling = cfg.sampling_probability > 0
        A = D["A"]
        drop_in = self._dropping and cfg.decoder_dropout[0] < 1.0
        xm = ops.mat(D["xemb"], E)
        # decoder inputs = embedding of the GO-prefixed labels (all steps when teacher forcing, only step 0 when sampling)
        ops.embed_labels(self._emb_t(), batch.labels, cfg.go_id, D["xemb"], D["fed"], B, L, E, 1 if sampling else L)
        if drop_in:
            if sampling:     # only row (b, 0): address rows with stride L*E, mask index (b*L + 0)*(E+A) + e
                ops.dropout_rows(ops.mat(D["xemb"], L * E), ops.mat(D["xemb"], L * E), B, E, self.seed, CELL_ID_DECODER * 4,
                                 cfg.decoder_dropout[0], L * (E + A))
            else:
                ops.dropout_rows(xm, xm, B * L, E, self.seed, CELL_ID_DECODER * 4, cfg.decoder_dropout[0], E + A)
        if not sampling:
            ops.gemm(xm, self.P[self._kn("dec/l0")[0]].mat(self.G * H), ops.mat(D["gates"], self.G * H), B * L, self.G * H, E)
            if self.gru:
                ops.gemm(xm, self.P["dec/l0/cand_kernel"].mat(H), ops.mat(D["cs"], H), B * L, H, E)
        self._block_prepare(ws, D)
        D["desc"] = d = self._block_desc(ws, D, batch.labels_len, 2 if sampling else 0, D["h0"], D["c0"], with_bwd=True)
        if sampling:
            d.output_attention = int(cfg.output_attention())
            d.seed = ops.fptr(self.seed)
            d.sampling_prob = cfg.sampling_probability
            if not self._bdrop(D):
                d.keep_in = d.keep_state = d.keep_out = 1.0
                d.cell_id = CELL_ID_DECODER
            d.embedding = ops.fptr(*self._emb())
            d.wout_t = ops.fptr(self.derived, self.Tr["dec/out/kernel"].off)
            d.bout = ops.fptr(self.params, self.P["dec/out/bias"].off)
            d.logits, d.xs, d.labels, d.fed = ops.fptr(D["logits"]), ops.fptr(D["xemb"]), ops.fptr(batch.labels), ops.fptr(D["fed"])
        ops.attn_rnn_fwd(d, 0, L)
        if not sampling:
            ov, O = self._out_vec(D)
            ops.gemm(ov, self.P["dec/out/kernel"].mat(V), ops.mat(D["logits"], V), B * L, V, O, bias=self._pp("dec/out/bias"))
        ops.seq_loss(D["logits"], batch.labels, batch.labels_len, self.denom, compute_denom, D["row_loss"], D["dlogits"], B, L, V,
                     loss_fun=cfg.loss_code(), label_smoothing=cfg.label_smoothing)
        ops.reduce_scalar(D["row_loss"], B * L, self.loss)
        if cfg.regress_aus and "video" in ws["enc"]:
            Ev = ws["enc"]["video"]
            ops.reduce_scalar(Ev["au_row"], B * Ev["T"], self.loss, accumulate=True)
        return D["logits"]

    def local_loss_denominator(self, batch: Batch):
        """What this rank contributes to the loss normaliser the data-parallel trainer all-reduces: the number of valid label
        steps (seq2seq.py:165-171), or the number of label ROWS when label smoothing makes the loss a mean over all rows."""
        B, L = batch.labels.shape
        if self.cfg.loss_code() == 1:
            return torch.full((1,), float(B * L), device=self.dev)
        return batch.labels_len.clamp(0, L).sum().to(torch.float32).reshape(1)

    def local_au_count(self, batch: Batch):
        """This rank's share of the AU loss normaliser: 2 units per valid video frame (encoder.py:173-189)."""
        if not (self.cfg.regress_aus and batch.video is not None):
            return torch.zeros(1, device=self.dev)
        return (2.0 * batch.video_len.clamp(0, batch.video.shape[1]).sum()).to(torch.float32).reshape(1)

    def sequence_likelihoods(self, batch: Batch):
        """Teacher-forced forward, then the per-utterance average step loss [B] (the LM's evaluate graph, lm.py:362-401:
        `average_log_likelihoods`).  Uses this engine's own dropout / sampling settings: build the evaluation engine with
        use_dropout=False, sampling_probability=0 as the reference builds its evaluate graph."""
        self.forward_train(batch)
        ws, _ = self._cur
        D = ws["dec"]
        if "utt_loss" not in D:
            D["utt_loss"] = torch.zeros(ws["B"], device=self.dev)
        ops.seq_loss_per_utterance(D["row_loss"], batch.labels_len, self.denom, D["utt_loss"], ws["B"], ws["L"])
        return D["utt_loss"]

    def backward_decoder(self):
        """First half of the backward pass: zeroes the gradient buffer, output layer, decoder BPTT and every decoder-side weight gradient
        (also the gradients flowing into the encoder memories / final states, which backward_encoders() continues from)."""
        cfg = self.cfg
        ws, batch = self._cur
        B, L = ws["B"], ws["L"]
        D = ws["dec"]
        H, E, A, V = D["H"], D["E"], D["A"], cfg.vocab_size
        self._ensure_gemm_ws()
        zs = [self.grads] + [ws["enc"][s]["dmem"].t for s in cfg.streams() if "dmem" in ws["enc"][s]]
        if cfg.architecture == "av_align":
            zs += [ws["enc"]["audio"]["blk"]["datt_ext"], ws["enc"]["audio"]["blk"]["dcell_ext"]]
        ops.zero_multi(zs)                        # one engine launch (every fill of the step is an engine kernel)
        ops.colsum_batch_begin(self.grads)        # bias gradients: collected, run in two launches when the half-pass ends
        # output layer
        ov, O = self._out_vec(D)
        dl = ops.mat(D["dlogits"], V)
        oa = cfg.output_attention()
        dext = D["datt_ext"] if oa else D["dcell_ext"]
        with ops.gemm_group():
            self._gemm_tn(ov, dl, self.Gr["dec/out/kernel"].mat(V), O, V, B * L)
            ops.colsum(dl, B * L, V, self.grads, self.scratch, beta=1.0, out_offset=self.Gr["dec/out/bias"].off)
            ops.gemm(dl, self.P["dec/out/kernel"].mat(V), ops.mat(dext, O), B * L, O, V, trans_b=1)
        d = D["desc"]
        d.datt_ext = ops.fptr(dext) if oa else None
        d.dcell_ext = None if oa else ops.fptr(dext)
        self._block_backward(ws, D, d, ops.mat(D["xemb"], E), ops.mat(D["dxemb"], E), 0.0, oa)
        if self._dropping and cfg.decoder_dropout[0] < 1.0:
            dm = ops.mat(D["dxemb"], E)
            ops.dropout_rows(dm, dm, B * L, E, self.seed, CELL_ID_DECODER * 4, cfg.decoder_dropout[0], E + A)
        if self.onehot is None:
            ops.embed_grad(D["dxemb"], D["fed"], self._gp("dec/embedding"), B, L, E, V, self.scratch)
        self._decoder_init_state_bwd(ws)
        ops.colsum_batch_end(self.scratch)        # the decoder's block of the gradient buffer is final here (decoder_grad_bucket)

    def backward_encoders(self):
        """Second half of the backward pass: encoder BPTT, the lip CNN, every encoder-side weight gradient."""
        ws, batch = self._cur
        ops.colsum_batch_begin(self.grads)
        try:
            self._encode_backward(ws, batch)
        finally:
            ops.colsum_batch_end(self.scratch)
        self._dp_stats_pack()

    def backward(self):
        """BPTT through decoder and encoders; leaves the full gradient in self.grads (engine layout)."""
        self.backward_decoder()
        self.backward_encoders()

    def _dp_stats_pack(self):
        """Data parallel: this rank's share (1 / world) of the moving statistics into the all-reduced buffer's tail."""
        if self.dp_world > 1 and self.n_stats > 0:
            ops.colsum(ops.mat(self.stats, self.n_stats), 1, self.n_stats, self.grads_and_loss, self.scratch, alpha=1.0 / self.dp_world,
                       out_offset=self._stats_mirror_off)

    def _dp_stats_unpack(self):
        if self.dp_world > 1 and self.n_stats > 0:
            ops.colsum(ops.mat(self.grads_and_loss, self.n_stats, offset=self._stats_mirror_off), 1, self.n_stats, self.stats, self.scratch)

    def decoder_grad_bucket(self):
        """(lo, hi) of the flat gradient buffer holding exactly the decoder's parameters (`dec/...`: embedding, cell, attention
        mechanisms, output layer, state bridge) -- final once backward_decoder() has run, so a data-parallel trainer can reduce it while
        backward_encoders() is still computing.  None when the layout does not keep them in one block."""
        segs = [(g.off, g.off + g.n, n) for n, g in self.Gr.items()]
        dec = [(lo, hi) for lo, hi, n in segs if n.startswith("dec/")]
        if not dec:
            return None
        lo, hi = min(x[0] for x in dec), max(x[1] for x in dec)
        if any(lo <= a < hi and not n.startswith("dec/") for a, b, n in segs):
            return None
        return lo, hi

    def apply_update(self):
        """L2 on the RNN kernels, global-norm clip, Adam, LR warm-up (seq2seq.py:175-178, :195-199, :245-257)."""
        cfg = self.cfg
        self._dp_stats_unpack()                  # (data parallel: the rank average of the moving statistics, summed with the gradients)
        if cfg.recurrent_l2 is not None:
            ops.l2_regularise(self.l2_segments, self.params, self.grads, cfg.recurrent_l2, self.loss, self.scratch)
        if self.use_cnn:                         # conv2d kernel_regularizer l2(0.001), seq2seq.py:180-184
            ops.l2_regularise(self.cnn_l2_segments, self.params, self.grads, 1e-3, self.loss, self.scratch)
            if self.dense_l2_segments:           # the input Dense layers' l2(0.0001) sit in the same collection (reference quirk)
                ops.l2_regularise(self.dense_l2_segments, self.params, self.grads, 1e-4, self.loss, self.scratch)
        ops.global_norm(self.grads, self.n_train, self.gnorm, self.scratch)
        ops.adam_step(self.params, self.grads, self.adam_m, self.adam_v, self.n_train, self.gnorm, self.step,
                      cfg.learning_rate, cfg.warmup_steps, cfg.max_gradient_norm if cfg.clip_gradients else 0.0,
                      first_decay_steps=cfg.lr_decay_steps, optimiser=cfg.optimiser, weight_decay=cfg.weight_decay)

    def train_step(self, batch: Batch):
        """One `session.run([train_op, batch_loss, global_norm])` (avsr/avsr.py:265-271); returns device scalars."""
        self.forward_train(batch)
        self.backward()
        self.apply_update()
        return self.loss, self.gnorm

    def beam_search_decode(self, *args, **kw):
        """See _beam_search_decode.  If a persistent kernel's bounded wait expired during the pass (workgroups not co-resident) the
        results are invalid: check_persistent() has then switched the one-launch paths off and the pass is redone with one launch
        per step (the ids written to .mlf files and error rates never come from a flagged pass)."""
        out = self._beam_search_decode(*args, **kw)
        if self.check_persistent():
            out = self._beam_search_decode(*args, **kw)
        return out

    def greedy_decode(self, *args, **kw):
        """See _greedy_decode; redone through the per-step launches if a persistent kernel flagged its pass (as above)."""
        out = self._greedy_decode(*args, **kw)
        if self.check_persistent():
            out = self._greedy_decode(*args, **kw)
        return out

    def _beam_search_decode(self, batch: Batch, beam_width: int = 10, length_penalty_weight: Optional[float] = None,
                            max_steps: Optional[int] = None, check_every: int = 8, return_all: bool = False):
        """Eval graph with BeamSearchDecoder (decoder_unimodal.py:222-271, decoder_bimodal.py:328-381): ids of beam 0,
        int32 [B, T_out]; positions after the first EOS hold EOS (gather_tree).  length_penalty_weight defaults to the
        reference's 0.6 (unimodal / av_align) or 0.5 (bimodal)."""
        cfg, K = self.cfg, int(beam_width)
        if K < 1 or K > 64 or K * cfg.vocab_size > 1024:
            # beam_step_kernel keeps the K * V candidates of an utterance in registers, four per thread of one workgroup
            raise ValueError("beam search: beam_width must be in 1..64 with beam_width * vocabulary <= 1024 (got %d x %d); "
                             "the reference's default width 10 fits every shipped unit list" % (K, cfg.vocab_size))
        B = (batch.audio if batch.audio is not None else batch.video if batch.video is not None else batch.labels).shape[0]
        L = cfg.max_label_length if max_steps is None else max_steps
        w = length_penalty_weight if length_penalty_weight is not None else (0.5 if cfg.architecture == "bimodal" else 0.6)
        Ta = batch.audio.shape[1] if batch.audio is not None else 0
        Tv = batch.video.shape[1] if batch.video is not None else 0
        ws = self._get_ws(B, Ta, Tv, 1, True)                  # encoders at batch B (decoder block of this ws is unused)
        self._refresh_derived()
        self._encode(ws, batch, False)
        # tile_batch (attention.py:100-106): the decoder block runs on B*K rows and the final states are repeated K times; the MEMORIES
        # are not copied -- hypothesis row r attends memory row r // K (avsr_attn_rnn.mem_shared): keys are computed once per
        # utterance and the K hypotheses of an utterance read the same bytes (tiled: K x 75 MB streamed from HBM every step)
        R, V, dev, H = B * K, cfg.vocab_size, self.dev, cfg.decoder_units[0]
        mems = cfg.decoder_memories()
        ck = (B, K, L, Ta, Tv)
        cache = getattr(self, "_beam_ws", None)
        if cache is None or cache[0] != ck:              # buffers of the last beam-search shape are kept (a decode allocates ~100)
            wsb = {"enc": {s: {} for s in cfg.streams()}, "B": R, "L": L}
            D = None
            logp0 = torch.full((2, B, K), float("-inf"), device=dev)
            logp0[0, :, 0] = 0.0
            X = dict(logp0=logp0, logp=torch.empty_like(logp0), fin=torch.zeros(2, R, dtype=torch.int32, device=dev),
                     ln=torch.zeros(2, R, dtype=torch.int32, device=dev), sid=torch.zeros(L, R, dtype=torch.int32, device=dev),
                     pid=torch.zeros(L, R, dtype=torch.int32, device=dev), prow0=torch.arange(R, dtype=torch.int32, device=dev),
                     prow=torch.zeros(R, dtype=torch.int32, device=dev),
                     c_dec={s: torch.zeros(R, H, device=dev) for s in cfg.streams()}, h_dec={s: torch.zeros(R, H, device=dev) for s in cfg.streams()})
            self._beam_ws = cache = (ck, wsb, X)
        _ck, wsb, X = cache
        for s in cfg.streams():
            E = ws["enc"][s]
            md = self._mem_desc(ws, s)
            src = E["mem"] if not E["attentive"] else (E["blk"]["att"] if E["blk"]["mems"][0]["type"] in LUONG_TYPES else E["blk"]["cell_out"])
            Eb = wsb["enc"][s]
            Eb.update({"attentive": False, "mem": src, "dmem": src, "len": md["len"], "T": E["T"], "units": E["units"],
                       "h_fin": E["h_fin"].repeat_interleave(K, dim=0).contiguous(),
                       "c_fin": None if E["c_fin"] is None else E["c_fin"].repeat_interleave(K, dim=0).contiguous(),
                       "c_dec": X["c_dec"][s], "h_dec": X["h_dec"][s]})
        if "dec" not in wsb:
            D = self._make_block(wsb, R, L, H, cfg.embedding_size, mems, "dec/l0",
                                 ["dec/att%d" % i for i in range(len(mems))], Tv=Tv, Ta=Ta, greedy=True, mem_B=B)
            wsb["dec"] = D
            D["logits"] = torch.zeros(R, L, V, device=dev)
            D["tok"] = torch.zeros(R, dtype=torch.int32, device=dev)
            D["nunf"] = torch.zeros(L, dtype=torch.int32, device=dev)
            D["steplen"] = torch.full((R,), L, dtype=torch.int32, device=dev)
        D = wsb["dec"]
        D["tok"].fill_(cfg.go_id)
        D["nunf"].fill_(1)
        logp, fin, ln, sid, pid, prow = X["logp"], X["fin"], X["ln"], X["sid"], X["pid"], X["prow"]
        logp.copy_(X["logp0"])
        prow.copy_(X["prow0"])
        ops.zero_multi([fin, ln])
        self._decoder_init_state(wsb)
        self._block_prepare(wsb, D)
        d = self._block_desc(wsb, D, D["steplen"], 3, D["h0"], D["c0"], with_bwd=False)
        d.output_attention = int(cfg.output_attention())
        d.embedding = ops.fptr(*self._emb())
        d.wout_t = ops.fptr(self.derived, self.Tr["dec/out/kernel"].off)
        d.bout = ops.fptr(self.params, self.P["dec/out/bias"].off)
        d.logits, d.tok, d.n_unfinished = ops.fptr(D["logits"]), ops.fptr(D["tok"]), ops.fptr(D["nunf"])
        d.beam_width, d.length_penalty, d.mem_shared = K, float(w), 1
        d.beam_logp, d.beam_fin, d.beam_len = ops.fptr(logp), ops.fptr(fin), ops.fptr(ln)
        d.step_ids, d.parent_ids, d.parent_rows = ops.fptr(sid), ops.fptr(pid), ops.fptr(prow)
        # steps are launched in chunks of check_every; the "every beam finished" flag of a chunk is read while the NEXT chunk is already
        # queued (the read would otherwise leave the GPU idle for a host round trip per chunk).  A chunk past the end is harmless:
        # a beam step whose predecessor left no unfinished beam hands its input state through unchanged (beam_step_kernel), and T
        # below comes from the per-step counters.
        fr = self._flag_reader()
        l, pending = 0, False
        while l < L:
            l1 = min(L, l + check_every)
            ops.attn_rnn_fwd(d, l, l1)
            if pending and fr.value() == 0:
                l = l1
                break
            fr.request(D["nunf"][l1 - 1:l1])
            pending, l = True, l1
        # dynamic_decode stops right after the first step at which every beam is finished
        hist = D["nunf"][:l].cpu().numpy()
        done = np.nonzero(hist == 0)[0]
        T = int(done[0]) + 1 if len(done) else l
        out = torch.zeros(B, T, K, dtype=torch.int32, device=dev)
        ops.beam_gather_tree(sid, pid, ln[T & 1], out, B, K, T, cfg.eos_id)     # lengths after step T-1 live at parity T&1
        self._last_beam = (D, T)
        if return_all:
            return out
        return out[:, :, 0].contiguous()

    def _flag_reader(self):
        if getattr(self, "_fr", None) is None:
            self._fr = _FlagReader(self.dev)
        return self._fr

    def _greedy_decode(self, batch: Batch, max_steps: Optional[int] = None, check_every: int = 8):
        """Eval graph with GreedyEmbeddingHelper (decoder_unimodal.py:176-217): int32 ids [B, T_out], zeros after EOS."""
        cfg = self.cfg
        B = (batch.audio if batch.audio is not None else batch.video if batch.video is not None else batch.labels).shape[0]
        L = cfg.max_label_length if max_steps is None else max_steps
        Ta = batch.audio.shape[1] if batch.audio is not None else 0
        Tv = batch.video.shape[1] if batch.video is not None else 0
        ws = self._get_ws(B, Ta, Tv, L, True)
        self._refresh_derived()
        self._encode(ws, batch, False)
        D = ws["dec"]
        self._decoder_init_state(ws)
        self._block_prepare(ws, D)
        D["steplen"].fill_(L)
        D["tok"].fill_(cfg.go_id)
        D["ids"].zero_()
        D["logits"].zero_()          # a group of the fused kernel that exits early leaves its later steps unwritten: zeros, not a previous batch's logits
        d = self._block_desc(ws, D, D["steplen"], 1, D["h0"], D["c0"], with_bwd=False)
        d.output_attention = int(cfg.output_attention())
        d.embedding = ops.fptr(*self._emb())
        d.wout_t = ops.fptr(self.derived, self.Tr["dec/out/kernel"].off)
        d.bout = ops.fptr(self.params, self.P["dec/out/bias"].off)
        d.logits, d.ids, d.tok, d.n_unfinished = ops.fptr(D["logits"]), ops.fptr(D["ids"]), ops.fptr(D["tok"]), ops.fptr(D["nunf"])
        # chunks of check_every steps; a chunk's "unfinished" count is read after the next chunk has been queued (no idle GPU while the
        # host waits).  Steps past the end change nothing: finished rows are frozen (impute_finished) and t_out is the longest row.
        # The fused persistent decode kernel stops by itself, group by group, once every utterance of a group has emitted EOS
        # (dec_persist.hip): all maximum_iterations steps are then ONE launch and the host never looks at the device in between.
        if self.fused_decode and ops.attn_rnn_fused_fwd_active(d):
            check_every = L
        fr = self._flag_reader()
        l, pending = 0, False
        while l < L:
            l1 = min(L, l + check_every)
            ops.attn_rnn_fwd(d, l, l1)
            if pending and fr.value() == 0:      # all utterances had emitted EOS by the end of the previous chunk
                l = l1
                break
            ops.copy_(D["nunf_prev"], D["nunf"])                         # this chunk's count (the next call resets the counter)
            fr.request(D["nunf_prev"])
            pending, l = True, l1
        t_out = min(int(D["steplen"].max().item()), l)   # dynamic_decode stops once every utterance has finished
        self._last_greedy = (ws, t_out)
        self._last_align = None
        return D["ids"][:, :t_out].contiguous()

    def attention_alignments(self):
        """alignment_history of the LAST greedy_decode (decoder_unimodal.py:273-290, decoder_bimodal.py:447-475,
        encoder.py:296-310): {"decoder": [alpha [B, T_out, T_mem] per mechanism, video first], "encoder": alpha
        [B, T_a, T_v] of the AV-Align layer or None}.  The raw scores the attention kernels kept are normalised in place
        (masked softmax over the valid memory frames); steps after an utterance finished are rows of zeros."""
        if self._last_align is not None:
            return self._last_align
        ws, t_out = self._last_greedy
        out = {"decoder": [], "encoder": None}

        def alphas(blk, steplen):
            res = []
            for m in blk["mems"]:
                md = self._mem_desc(ws, m["stream"])
                g_t = self._pp(m["prefix"] + "/g") if m["type"] == "scaled_luong" else None
                ops.attn_alpha_rows(m["scores"], m["scores"], md["len"], steplen, g_t, None, blk["B"], blk["L"], m["T"])
                res.append(m["scores"].view(blk["B"], blk["L"], m["T"]))
            return res
        D = ws["dec"]
        out["decoder"] = [a[:, :t_out] for a in alphas(D, D["steplen"])]
        if self.cfg.architecture == "av_align":
            E = ws["enc"]["audio"]
            out["encoder"] = alphas(E["blk"], E["len"])[0]
        self._last_align = out
        return out


class _FlagReader:
    """Reads a device int32 word for the host WITHOUT draining the main stream: the word is copied to page-locked memory on a side
    stream behind an event, so a decode loop can queue its next chunk of steps before it looks at the previous chunk's "all finished"
    counter (a plain .item() waits for everything queued so far, the next chunk included)."""

    def __init__(self, device):
        self.side = torch.cuda.Stream(device=device)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.done = torch.cuda.Event()

    def request(self, word):
        """Queue the read of `word` (a 1-element int32 device tensor) as of everything queued on the current stream so far."""
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            self.host.copy_(word, non_blocking=True)
            self.done.record()

    def value(self):
        self.done.synchronize()
        return int(self.host[0])


def desc_steplen(desc):
    return _PtrView(desc.steplen)


class _PtrView:
    """Wraps a raw device address so it can be passed where ops.fptr() expects a tensor."""

    def __init__(self, addr):
        self.addr = addr
        self.is_cuda = True

    def data_ptr(self):
        return self.addr

    def element_size(self):
        return 4
