"""Decoder half of the engine (mixin of model.Seq2SeqModel): the attention-wrapped block descriptor (avsr_attn_rnn) shared by the decoder and the AV-Align
attentive layer, its backward pass, the decoder's initial state, and the evaluation decodes (greedy, beam search, alignment history).
Reference: avsr/decoder_unimodal.py, avsr/decoder_bimodal.py, avsr/attention.py:132-191, avsr/encoder.py:224-294."""
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


class DecoderMixin:
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
            if self.gru:
                cn = X["prefix"] + "/cand_kernel"
                Xd.wt2, Xd.w2 = ops.fptr(self.derived, self.Tr[cn].off), ops.fptr(self.params, self.P[cn].off)
                Xd.bias2 = ops.fptr(self.params, self.P[X["prefix"] + "/cand_bias"].off)
                Xd.rh_seq, Xd.dgates2 = ops.fptr(X["rh"]), ops.fptr(X["dpc"])
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
            dgx = ops.mat(X["dgates"], G * H)
            x_in = (X["xin_seq"] if drop else below).mat(0)
            self._gemm_tn(x_in, dgx, self.Gr[kx].mat(G * H), H, G * H, rows)
            self._gemm_tn((X["hs_seq"] if drop else X["out"]).mat(-1), dgx, self.Gr[kx].mat(G * H, row0=H), H, G * H, rows)
            ops.colsum(dgx, rows, G * H, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[bx].off)
            if self.gru:                                          # candidate kernel of the layer: rows [0:H] input, [H:2H] r*h
                Gcx, dpcx = self.Gr[X["prefix"] + "/cand_kernel"], ops.mat(X["dpc"], H)
                self._gemm_tn(x_in, dpcx, Gcx.mat(H), H, H, rows)
                self._gemm_tn(ops.mat(X["rh"], H), dpcx, Gcx.mat(H, row0=H), H, H, rows)
                ops.colsum(dpcx, rows, H, self.grads, self.scratch, beta=1.0, out_offset=self.Gr[X["prefix"] + "/cand_bias"].off)
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
