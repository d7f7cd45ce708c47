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
from .model_base import Batch, Ref, SeqBuf, _FlagReader, _PtrView, _splitk, desc_steplen  # noqa: F401  (re-exported: the public names live here)
from .model_decoder import DecoderMixin
from .model_encoder import EncoderMixin


class Seq2SeqModel(EncoderMixin, DecoderMixin):
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
                if self.gru:                      # r*h record and d(candidate pre-activation) record of a GRU layer
                    X["rh"], X["dpc"] = z(B, L, H), z(B, L, H)
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
