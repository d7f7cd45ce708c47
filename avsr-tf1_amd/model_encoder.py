"""Encoder half of the engine (mixin of model.Seq2SeqModel): per-stack descriptors of avsr_rnn_fwd / _bwd, the HighwayWrapper stacks, input batch norm (and
its data-parallel synchronisation), the encoders' forward and backward passes with every deferred weight / input gradient.
Reference: avsr/encoder.py:14-196 (Seq2SeqEncoder), avsr/cells.py:61-102 (build_rnn_layers)."""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional
import os
import numpy as np
import torch
from . import ops, params as PR
from ._lib import AttnRnn, RnnStack
from .config import ATT_CODE, BAHDANAU_TYPES, CELL_ID_DECODER, LUONG_TYPES, ModelConfig, encoder_cell_id
from .model_base import Batch, Ref, SeqBuf, _FlagReader, _PtrView, _splitk, desc_steplen  # noqa: F401


class EncoderMixin:
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
