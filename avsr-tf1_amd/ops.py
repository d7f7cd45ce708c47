"""Thin Python wrappers over the C ABI: torch tensors are only containers for device memory."""
import ctypes as C

import torch

from . import _lib
from ._lib import (AttnMech, AttnRnn, GemmDesc, Mat, RnnLayer, RnnStack, TransposeJob, check, stream_ptr)


def fptr(t, offset=0):
    """Device address of element `offset` of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda, "HIP engine buffers must live on the GPU"
    return t.data_ptr() + t.element_size() * int(offset)


def mat(t, ld, T=0, ldo=0, offset=0):
    assert t.dtype == torch.float32
    return Mat(fptr(t, offset), int(ld), int(T), 0, int(ldo))


def _L():
    return _lib.load()


def _s():
    return stream_ptr()


_gemm_ws = None


def set_gemm_workspace(t):
    """Register the split-K scratch (float32 device tensor) that `gemm(..., splitk=None)` may use."""
    global _gemm_ws
    _gemm_ws = t


def auto_splitk(M, N, K, batch=1):
    """Split-K factor for a GEMM whose M x N tile grid alone cannot fill the chip (tools/gemm_bench.py on MI355X):
    one 128x128 workgroup walks K at ~1.5 us per 16-deep tile, so few-tile GEMMs are latency-bound unless K is cut.
    Aim at ~768 workgroups, keep >= 64 of K per slice, never split when the grid already has >= 128 tiles."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * max(1, batch)
    if 128 <= tiles < 384 and K >= 512:      # one partial wave of workgroups walking a long K (e.g. 32000 x 80 x 1024: 250 tiles, 64 K tiles each)
        sk = min(768 // tiles, K // 256)
        return sk if sk >= 2 else 1
    if tiles >= 128:
        return 1
    if K < 65536:
        sk = min(768 // tiles, K // 64, 64)
    else:                        # weight gradients over 10^5..10^7 rows (encoder BPTT, convolutions): ~4 workgroups per CU
        sk = min(1024 // tiles, K // 256, 2048)
    return sk if sk >= 2 else 1


_gemm_group = None        # [descs], workspace cursor while a `with gemm_group():` block collects independent GEMMs


class gemm_group:
    """`with ops.gemm_group():` -- the GEMMs issued inside are INDEPENDENT of each other (no output is an operand or the output of
    another): they are collected and launched side by side by avsr_gemm_batch on exit (one launch per operand-layout class + one
    split-K reduction launch).  Split-K slabs of the collected GEMMs take consecutive regions of the shared workspace.  Nested
    blocks join the outer one.  AVSR_GEMM_GROUP=0 turns the collection off (every GEMM is launched where it is issued)."""

    def __enter__(self):
        global _gemm_group
        self.outer = _gemm_group is not None
        if not self.outer and _GROUP_ON:
            _gemm_group = {"descs": [], "cursor": 0, "keep": []}
        return self

    def __exit__(self, et, ev, tb):
        global _gemm_group
        if self.outer or _gemm_group is None:
            return False
        grp, _gemm_group = _gemm_group, None
        if et is None and grp["descs"]:
            n = len(grp["descs"])
            arr = (GemmDesc * n)(*grp["descs"])
            check(_L().avsr_gemm_batch(arr, n, _s()), "avsr_gemm_batch")
        return False


import os as _os
_GROUP_ON = _os.environ.get("AVSR_GEMM_GROUP", "1") != "0"


def gemm(A, B, Cm, M, N, K, trans_a=False, trans_b=False, alpha=1.0, beta=0.0, bias=None,
         batch=1, strides=(0, 0, 0), splitk=None, workspace=None, alpha_dev=None, colsum=None, colsum_beta=0.0):
    """C = alpha*op(A)*op(B) + beta*C + bias.  A, B, Cm are `Mat` views (see `mat`).
    splitk=None picks the factor (auto_splitk) when a workspace is available, else 1.
    colsum=(tensor, offset): also tensor[offset + n] = colsum_beta * tensor[offset + n] + sum_k B[k, n] (trans_b False, batch 1)."""
    d = GemmDesc()
    d.A, d.B, d.C = A, B, Cm
    d.bias = fptr(bias)
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.trans_a, d.trans_b = int(trans_a), int(trans_b)
    d.alpha, d.beta = float(alpha), float(beta)
    d.batch = int(batch)
    d.stride_a, d.stride_b, d.stride_c = [int(s) for s in strides]
    d.alpha_dev = fptr(alpha_dev)
    cs_extra = 0
    if colsum is not None:
        assert not trans_b and batch == 1
        d.colsum, d.colsum_beta = fptr(colsum[0], colsum[1]), float(colsum_beta)
        cs_extra = N
    if splitk is None:
        workspace = workspace if workspace is not None else _gemm_ws
        splitk = auto_splitk(M, N, K, batch) if workspace is not None else 1
        while splitk > 1 and batch * splitk * (M * N + cs_extra) > workspace.numel():
            splitk //= 2
    d.splitk = int(splitk)
    grp = _gemm_group
    if splitk > 1:
        need = batch * splitk * (M * N + cs_extra)
        assert workspace is not None and workspace.numel() >= need, "split-K workspace too small"
        if grp is not None:
            # concurrent entries need disjoint slabs: carve consecutive regions; when the workspace is used up the collected
            # GEMMs are launched and the collection restarts
            if grp["cursor"] + need > workspace.numel() or (grp["descs"] and grp.get("ws_ptr") not in (None, workspace.data_ptr())):
                _flush_group(grp)
            grp["ws_ptr"] = workspace.data_ptr()
            d.workspace = fptr(workspace, grp["cursor"])
            d.workspace_floats = workspace.numel() - grp["cursor"]
            grp["cursor"] += (need + 3) // 4 * 4
        else:
            d.workspace = fptr(workspace)
            d.workspace_floats = workspace.numel()
    if grp is not None:
        if len(grp["descs"]) >= 48:
            _flush_group(grp)
        grp["descs"].append(d)
        grp["keep"] += [A, B, Cm]
        return
    check(_L().avsr_gemm(C.byref(d), _s()), "avsr_gemm")


def _flush_group(grp):
    if grp["descs"]:
        n = len(grp["descs"])
        arr = (GemmDesc * n)(*grp["descs"])
        check(_L().avsr_gemm_batch(arr, n, _s()), "avsr_gemm_batch")
    grp["descs"], grp["cursor"], grp["keep"] = [], 0, []


def rnn_fwd(stacks):
    arr = (RnnStack * len(stacks))(*stacks)
    check(_L().avsr_rnn_fwd(arr, len(stacks), _s()), "avsr_rnn_fwd")


def rnn_bwd(stacks):
    arr = (RnnStack * len(stacks))(*stacks)
    check(_L().avsr_rnn_bwd(arr, len(stacks), _s()), "avsr_rnn_bwd")


def attn_rnn_fwd(desc, l_begin, l_end):
    check(_L().avsr_attn_rnn_fwd(C.byref(desc), int(l_begin), int(l_end), _s()), "avsr_attn_rnn_fwd")


def attn_rnn_bwd(desc):
    check(_L().avsr_attn_rnn_bwd(C.byref(desc), _s()), "avsr_attn_rnn_bwd")


def beam_search_step(logits, n_utt, beam_width, V, step, eos_id, length_penalty_weight, logp_in, fin_in, len_in, logp_out, fin_out, len_out,
                     tok, parent_rows, step_ids, parent_ids, n_unfinished, x=None, x_stride=0, O=0, wout_t=None, bout=None):
    """One BeamSearchDecoder step on given logits, or (x given) with the output layer inside (decoder_unimodal.py:248-271); see include/avsr_hip.h."""
    check(_L().avsr_beam_search_step(fptr(logits), n_utt, beam_width, V, step, eos_id, float(length_penalty_weight), fptr(logp_in), fptr(fin_in),
                                     fptr(len_in), fptr(logp_out), fptr(fin_out), fptr(len_out), fptr(tok), fptr(parent_rows), fptr(step_ids),
                                     fptr(parent_ids), fptr(n_unfinished), fptr(x), x_stride, O, fptr(wout_t), fptr(bout), _s()),
          "avsr_beam_search_step")


def beam_gather_tree(step_ids, parent_ids, beam_len, out, n_utt, beam_width, T, eos_id):
    check(_L().avsr_beam_gather_tree(fptr(step_ids), fptr(parent_ids), fptr(beam_len), fptr(out), n_utt, beam_width, T, eos_id, _s()),
          "avsr_beam_gather_tree")


def attn_alpha_rows(scores, dscores, mem_len, steplen, g, rowdot, B, L, T):
    check(_L().avsr_attn_alpha_rows(fptr(scores), fptr(dscores), fptr(mem_len), fptr(steplen), fptr(g), fptr(rowdot),
                                    B, L, T, _s()), "avsr_attn_alpha_rows")


def bahdanau_dkeys(keys, pq, pq_sb, pq_sl, dscores, v, bq, mem_len, dkeys, dv_part, B, L, T, H):
    check(_L().avsr_bahdanau_dkeys(fptr(keys), fptr(pq), pq_sb, pq_sl, fptr(dscores), fptr(v), fptr(bq),
                                   fptr(mem_len), fptr(dkeys), fptr(dv_part), B, L, T, H, _s()), "avsr_bahdanau_dkeys")


def transpose(jobs):
    """jobs: list of (src_tensor, src_off, dst_tensor, dst_off, rows, cols)."""
    arr = (TransposeJob * len(jobs))()
    for i, (s, so, d, do, r, c) in enumerate(jobs):
        arr[i] = TransposeJob(fptr(s, so), fptr(d, do), int(r), int(c))
    check(_L().avsr_transpose(arr, len(jobs), _s()), "avsr_transpose")


_colsum_batch = None      # (gradient buffer the deferred sums write into, [jobs]) while a backward pass collects its bias gradients


def colsum_batch_begin(grads):
    """From here on, column sums whose destination is `grads` (bias / gamma / beta gradients: read by nothing before the optimiser)
    are collected instead of launched; colsum_batch_flush() runs them all in two launches (avsr_colsum_multi)."""
    global _colsum_batch
    _colsum_batch = (grads, [])


def colsum_batch_flush(scratch):
    global _colsum_batch
    if _colsum_batch is None:
        return
    grads, jobs = _colsum_batch
    _colsum_batch = (grads, [])
    if jobs:
        from ._lib import ColsumJob
        arr = (ColsumJob * len(jobs))(*jobs)
        check(_L().avsr_colsum_multi(arr, len(jobs), fptr(scratch), scratch.numel(), _s()), "avsr_colsum_multi")


def colsum_batch_abort():
    global _colsum_batch
    _colsum_batch = None


def colsum_batch_end(scratch):
    global _colsum_batch
    colsum_batch_flush(scratch)
    _colsum_batch = None


def colsum(a, rows, F, out, scratch, b=None, alpha=1.0, beta=0.0, out_offset=0):
    if _colsum_batch is not None and out is _colsum_batch[0]:
        from ._lib import ColsumJob, Mat
        dst = fptr(out, out_offset)
        dups = [j for j in _colsum_batch[1] if j.out == dst]
        dup = bool(dups)
        if dup and (beta != 1.0 or any(j.beta != 1.0 for j in dups)):
            # an overwriting job on either side: keep the program order (the queued job first, then this one in a fresh batch)
            colsum_batch_flush(scratch)
            dup = False
        if not dup:
            _colsum_batch[1].append(ColsumJob(a, b if b is not None else Mat(None, 0, 0, 0, 0), dst, int(rows), int(F),
                                              float(alpha), float(beta)))
            return
        # a second ACCUMULATING sum into the same destination (shared encoder weights, cells.py:77: several layers add into one
        # bias): jobs of one launch run concurrently, so this one is launched now, on its own -- additions commute
    check(_L().avsr_colsum(C.byref(a), C.byref(b) if b is not None else None, rows, F, alpha, beta,
                           fptr(out, out_offset), fptr(scratch), scratch.numel(), _s()), "avsr_colsum")


def batchnorm_fwd(x, y, rows, F, gamma, beta, mov_mean, mov_var, save_mean, save_invstd, training, scratch):
    check(_L().avsr_batchnorm_fwd(fptr(x), fptr(y), rows, F, fptr(gamma), fptr(beta), fptr(mov_mean), fptr(mov_var),
                                  fptr(save_mean), fptr(save_invstd), int(training), fptr(scratch), scratch.numel(),
                                  _s()), "avsr_batchnorm_fwd")


def batchnorm_sync_sum(x, rows, F, sum_out, scratch):
    check(_L().avsr_batchnorm_sync_sum(fptr(x), rows, F, fptr(sum_out), fptr(scratch), scratch.numel(), _s()), "avsr_batchnorm_sync_sum")


def batchnorm_sync_sqsum(x, rows, F, sum_global, total_rows, mean_out, sq_out, scratch):
    check(_L().avsr_batchnorm_sync_sqsum(fptr(x), rows, F, fptr(sum_global), fptr(total_rows), fptr(mean_out), fptr(sq_out),
                                         fptr(scratch), scratch.numel(), _s()), "avsr_batchnorm_sync_sqsum")


def batchnorm_sync_apply(x, y, rows, F, gamma, beta, mov_mean, mov_var, mean, sq_global, total_rows, invstd_out, eps=1e-3, momentum=0.99,
                         relu=0):
    check(_L().avsr_batchnorm_sync_apply(fptr(x), fptr(y), rows, F, fptr(gamma), fptr(beta), fptr(mov_mean), fptr(mov_var), fptr(mean),
                                         fptr(sq_global), fptr(total_rows), fptr(invstd_out), float(eps), float(momentum), int(relu), _s()),
          "avsr_batchnorm_sync_apply")


def batchnorm_sync_moments(x, rows, F, out64, scratch):
    assert out64.dtype == torch.float64
    check(_L().avsr_batchnorm_sync_moments(fptr(x), rows, F, out64.data_ptr(), fptr(scratch), scratch.numel(), _s()), "avsr_batchnorm_sync_moments")


def dp_sync_unpack(buf64, dp_norm, streams):
    """streams: [(offset in buf64, F, mean, sq, rows)] float32 destinations."""
    n = len(streams)
    off = (C.c_int32 * max(n, 1))(*[int(s[0]) for s in streams])
    Fs = (C.c_int32 * max(n, 1))(*[int(s[1]) for s in streams])
    mk = lambda k: (C.c_void_p * max(n, 1))(*[s[k].data_ptr() for s in streams])
    check(_L().avsr_dp_sync_unpack(buf64.data_ptr(), fptr(dp_norm), n, off, Fs, mk(2), mk(3), mk(4), _s()), "avsr_dp_sync_unpack")


def batchnorm_xhat(x, mean, invstd, xhat, rows, F):
    check(_L().avsr_batchnorm_xhat(fptr(x), fptr(mean), fptr(invstd), fptr(xhat), rows, F, _s()), "avsr_batchnorm_xhat")


def embed_labels(emb, labels, go_id, out, fed, B, L, E, n_steps=None):
    check(_L().avsr_embed_labels(fptr(emb), fptr(labels), go_id, fptr(out), fptr(fed), B, L, E, L if n_steps is None else n_steps,
                                 _s()), "avsr_embed_labels")


def embed_grad(dx, fed, demb, B, L, E, V, scratch):
    check(_L().avsr_embed_grad(fptr(dx), fptr(fed), fptr(demb), B, L, E, V, fptr(scratch), scratch.numel(), _s()), "avsr_embed_grad")


def dropout_rows(x, y, rows, cols, seed, stream_id, keep, idx_width, idx_coff=0, accumulate=False):
    """y (+)= x * mask / keep; x, y are Mat views; mask index = r*idx_width + idx_coff + c."""
    check(_L().avsr_dropout_rows(C.byref(x), C.byref(y), rows, cols, fptr(seed), int(stream_id), float(keep), int(idx_width),
                                 int(idx_coff), int(accumulate), _s()), "avsr_dropout_rows")


LOSS_FUN = {None: 0, "label_smoothing": 1, "focal_loss": 2, "mc_loss": 3}


def seq_loss(logits, labels, labels_len, denom, compute_denom, row_loss, dlogits, B, L, V, loss_fun=0, label_smoothing=0.0):
    check(_L().avsr_seq_loss_fun(fptr(logits), fptr(labels), fptr(labels_len), fptr(denom), int(compute_denom),
                                 fptr(row_loss), fptr(dlogits), B, L, V, int(loss_fun), float(label_smoothing), _s()), "avsr_seq_loss_fun")


def au_loss(z, aus, lens, row_loss, dz, B, T, weight, total_count=None):
    check(_L().avsr_au_loss_dp(fptr(z), fptr(aus), fptr(lens), fptr(row_loss), fptr(dz), B, T, float(weight), fptr(total_count), _s()),
          "avsr_au_loss_dp")


def normed_v(v, g, vn, H):
    check(_L().avsr_normed_v(fptr(v), fptr(g), fptr(vn), H, _s()), "avsr_normed_v")


def normed_v_bwd(v, g, dvn, dv, dg, H):
    check(_L().avsr_normed_v_bwd(fptr(v), fptr(g), fptr(dvn), fptr(dv), fptr(dg), H, _s()), "avsr_normed_v_bwd")


def reduce_scalar(part, n, out, do_sqrt=False, accumulate=False, scale=1.0, out_offset=0):
    check(_L().avsr_reduce_scalar(fptr(part), n, fptr(out, out_offset), int(do_sqrt), int(accumulate), float(scale),
                                  _s()), "avsr_reduce_scalar")


def l2_regularise(segments, params, grads, l2, loss_accum, scratch):
    n = len(segments)
    if n == 0:
        return
    off = (C.c_int64 * n)(*[int(o) for o, _ in segments])
    cnt = (C.c_int64 * n)(*[int(c) for _, c in segments])
    check(_L().avsr_l2_regularise(off, cnt, n, fptr(params), fptr(grads), float(l2), fptr(loss_accum), fptr(scratch),
                                  _s()), "avsr_l2_regularise")


def global_norm(grads, n, norm_out, scratch, grad_scale=1.0):
    check(_L().avsr_global_norm(fptr(grads), n, float(grad_scale), fptr(norm_out), fptr(scratch), _s()),
          "avsr_global_norm")


def seq_loss_per_utterance(row_loss, labels_len, denom, out, B, L):
    check(_L().avsr_seq_loss_per_utterance(fptr(row_loss), fptr(labels_len), fptr(denom), fptr(out), B, L, _s()),
          "avsr_seq_loss_per_utterance")


def highway_fwd(x, h, cpre, y, lens, B, T, H):
    check(_L().avsr_highway_fwd(C.byref(x), C.byref(h), C.byref(cpre), C.byref(y), fptr(lens), B, T, H, _s()), "avsr_highway_fwd")


def highway_bwd(x, h, cpre, dy, dh, dcpre, dx, lens, B, T, H, accumulate_dx=False):
    check(_L().avsr_highway_bwd(C.byref(x), C.byref(h), C.byref(cpre), C.byref(dy), C.byref(dh), C.byref(dcpre), C.byref(dx), fptr(lens),
                                B, T, H, int(accumulate_dx), _s()), "avsr_highway_bwd")


def copy_(dst, src):
    """dst = src (same shape, 4-byte dtype, both contiguous) by an engine kernel -- never a D2D memcpy (see include/avsr_hip.h)."""
    assert dst.is_cuda and src.is_cuda and dst.numel() == src.numel() and dst.element_size() == 4 and src.element_size() == 4
    assert dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype
    check(_L().avsr_copy_words(dst.data_ptr(), src.data_ptr(), dst.numel(), _s()), "avsr_copy_words")


def zero_(dst):
    assert dst.is_cuda and dst.element_size() == 4 and dst.is_contiguous()
    check(_L().avsr_zero_words(dst.data_ptr(), dst.numel(), _s()), "avsr_zero_words")


def zero_multi(tensors):
    """Zero several 4-byte-element buffers with one engine launch per eight of them."""
    ts = [t for t in tensors if t is not None and t.numel()]
    if not ts:
        return
    for t in ts:
        assert t.is_cuda and t.element_size() == 4 and t.is_contiguous()
    ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    cnt = (C.c_int64 * len(ts))(*[t.numel() for t in ts])
    check(_L().avsr_zero_multi(ptrs, cnt, len(ts), _s()), "avsr_zero_multi")


def add_int(a, b, out):
    """out[0] = a[0] + b (int32, on the device)."""
    check(_L().avsr_add_int(fptr(a), int(b), fptr(out), _s()), "avsr_add_int")


def instnorm_fwd(x, y, B, T, F, gamma, beta, mean_out, invstd_out, eps=1e-6):
    check(_L().avsr_instnorm_fwd(fptr(x), fptr(y), B, T, F, fptr(gamma), fptr(beta), fptr(mean_out), fptr(invstd_out), float(eps), _s()),
          "avsr_instnorm_fwd")


def instnorm_bwd(x, dy, gamma, mean, invstd, dx, dgamma_part, dbeta_part, B, T, F):
    check(_L().avsr_instnorm_bwd(fptr(x), fptr(dy), fptr(gamma), fptr(mean), fptr(invstd), fptr(dx), fptr(dgamma_part), fptr(dbeta_part),
                                 B, T, F, _s()), "avsr_instnorm_bwd")


OPTIMISER = {"Adam": 0, "Nadam": 1, "AdamW": 2, "Momentum": 3}


def adam_step(params, grads, m, v, n, gnorm, step, lr, warmup_steps, clip_norm, grad_scale=1.0, first_decay_steps=0, optimiser="Adam",
              weight_decay=0.0):
    check(_L().avsr_optimiser_step(fptr(params), fptr(grads), fptr(m), fptr(v), n, fptr(gnorm), fptr(step), float(lr),
                                   int(warmup_steps), int(first_decay_steps), float(clip_norm), float(grad_scale), OPTIMISER[optimiser],
                                   float(weight_decay), _s()), "avsr_optimiser_step")


PROF_KINDS = ("gemm", "step_lstm_fwd", "step_lstm_bwd", "step_dense", "attn_fwd", "attn_bwd", "rnn_persist_fwd", "rnn_persist_bwd", "dec_persist_fwd", "dec_persist_bwd",
              "conv_fwd", "conv_bwd_data", "conv_bwd_weight", "align_persist_fwd", "align_persist_bwd")


def prof_begin(max_launches=65536):
    check(_L().avsr_prof_begin(int(max_launches)), "avsr_prof_begin")


def prof_end():
    """{kind: (launch count, total ms, algorithmic FLOPs or 0)} since prof_begin (synchronises the device)."""
    cnt = (C.c_int32 * len(PROF_KINDS))()
    ms = (C.c_float * len(PROF_KINDS))()
    fl = (C.c_double * len(PROF_KINDS))()
    check(_L().avsr_prof_end(cnt, ms, fl), "avsr_prof_end")
    return {k: (int(cnt[i]), float(ms[i]), float(fl[i])) for i, k in enumerate(PROF_KINDS)}


_persist_sync = None
_persist_scratch = None


def rnn_set_persistent(on, device="cuda", ints=1 << 20, mode=3, scratch_floats=64 << 20):
    """Enable / disable the one-launch persistent execution of avsr_rnn_fwd / avsr_rnn_bwd (see include/avsr_hip.h).
    mode: bit 0 agent-scope forward, bit 1 XCD-local forward + fused BPTT, bit 2 split BPTT (uses a float scratch)."""
    global _persist_sync, _persist_scratch
    check(_L().avsr_rnn_set_persistent_mode(int(mode)), "avsr_rnn_set_persistent_mode")
    if on:
        if _persist_sync is None:
            _persist_sync = torch.zeros(ints, dtype=torch.int32, device=device)
        if (mode & 6) and _persist_scratch is None:
            # mode 2: the K-split BPTT kernel's partial-gradient slabs (2 x H/16 x B x H floats per cell) and its helpers' dx records
            # ([B, T, H] per layer edge); mode 4: the split BPTT's dx operands
            _persist_scratch = torch.zeros(scratch_floats, dtype=torch.float32, device=device)
        if _persist_scratch is not None:
            check(_L().avsr_rnn_set_persistent_scratch(_persist_scratch.data_ptr(), _persist_scratch.numel()), "avsr_rnn_set_persistent_scratch")
        _persist_sync[:1].zero_()
        check(_L().avsr_rnn_set_persistent(_persist_sync.data_ptr(), ints), "avsr_rnn_set_persistent")
    else:
        check(_L().avsr_rnn_set_persistent(None, 0), "avsr_rnn_set_persistent")


def rnn_persistent_clear():
    """Reset the sticky error word (after the caller has switched the persistent paths off and is about to redo the pass)."""
    if _persist_sync is not None:
        _persist_sync[:1].zero_()


def attn_rnn_fused_ws_floats(B, n_mech, Dmax=256):
    return int(_L().avsr_attn_rnn_fused_ws_floats(int(B), int(n_mech), int(Dmax)))


def attn_rnn_fused_eligible(desc):
    """Would avsr_attn_rnn_fwd run this block as the one-launch fused persistent decode kernel (csrc/dec_persist.hip)?"""
    return bool(_L().avsr_attn_rnn_fused_eligible(C.byref(desc)))


def attn_rnn_fused_fwd_active(desc):
    """Will avsr_attn_rnn_fwd run the fused forward kernel NOW (eligible + switched on + sync scratch registered)?"""
    return bool(_L().avsr_attn_rnn_fused_fwd_active(C.byref(desc)))


def attn_rnn_set_fused(on):
    check(_L().avsr_attn_rnn_set_fused(int(on)), "avsr_attn_rnn_set_fused")   # 0 off, 1 / True both, 2 forward only, 3 backward only


def attn_rnn_set_beam_kernel(on):
    """Beam search kernels: 1 / True all beam-shaped kernels (default), 2 the K-hypotheses-per-workgroup attention kernel only, 0 the
    general kernels everywhere (include/avsr_hip.h)."""
    check(_L().avsr_attn_rnn_set_beam_kernel(int(on)), "avsr_attn_rnn_set_beam_kernel")


def rnn_persistent_error():
    """Sticky flag: a device-side bounded wait of the persistent kernel expired (results of that call are invalid)."""
    return bool(_persist_sync is not None and int(_persist_sync[:1].item()) != 0)


# ---- lip-crop CNN front-end helpers (csrc/conv.hip) -------------------------------------------------------------------
def batchnorm_fwd_ex(x, y, rows, F, gamma, beta, mov_mean, mov_var, save_mean, save_invstd, training, eps, momentum, relu, scratch, bessel=1):
    check(_L().avsr_batchnorm_fwd_ex(fptr(x), fptr(y), rows, F, fptr(gamma), fptr(beta), fptr(mov_mean), fptr(mov_var), fptr(save_mean),
                                     fptr(save_invstd), int(training), float(eps), float(momentum), int(relu), int(bessel), fptr(scratch),
                                     scratch.numel(), _s()), "avsr_batchnorm_fwd_ex")


def batchnorm_bwd(x, dy, gamma, beta, mean, invstd, dx, dgamma, dbeta, rows, F, relu, scratch, dx_beta=0.0):
    check(_L().avsr_batchnorm_bwd(fptr(x), fptr(dy), fptr(gamma), fptr(beta), fptr(mean), fptr(invstd), fptr(dx), fptr(dgamma), fptr(dbeta),
                                  rows, F, int(relu), float(dx_beta), fptr(scratch), scratch.numel(), _s()), "avsr_batchnorm_bwd")


def im2col(x, col, N, H, W, Cc, kh, kw, stride, pad_t, pad_l, Ho, Wo):
    check(_L().avsr_im2col(fptr(x), fptr(col), N, H, W, Cc, kh, kw, stride, pad_t, pad_l, Ho, Wo, _s()), "avsr_im2col")


def col2im(dcol, dx, N, H, W, Cc, kh, kw, stride, pad_t, pad_l, Ho, Wo, beta=0.0):
    check(_L().avsr_col2im(fptr(dcol), fptr(dx), N, H, W, Cc, kh, kw, stride, pad_t, pad_l, Ho, Wo, float(beta), _s()), "avsr_col2im")


def relu(x, y, n):
    check(_L().avsr_relu(fptr(x), fptr(y), int(n), _s()), "avsr_relu")


def relu_bwd(y, dy, dx, n):
    check(_L().avsr_relu_bwd(fptr(y), fptr(dy), fptr(dx), int(n), _s()), "avsr_relu_bwd")


def add(a, b, out, n):
    check(_L().avsr_add(fptr(a), fptr(b), fptr(out), int(n), _s()), "avsr_add")


def conv3x3_supported(Ci, Co, H, W):
    return bool(_L().avsr_conv3x3_supported(Ci, Co, H, W))


def conv3x3(x, w, bias, y, N, H, W, Ci, Co, stride, pad_t, pad_l, Ho, Wo, flip=0, beta=0.0):
    check(_L().avsr_conv3x3(fptr(x), fptr(w), fptr(bias), fptr(y), N, H, W, Ci, Co, stride, pad_t, pad_l, Ho, Wo, int(flip), float(beta), _s()),
          "avsr_conv3x3")


def conv_desc(N, H, W, Ci, Co, k, stride, pad_t, pad_l, Ho, Wo, bn=None):
    """avsr_conv_desc; bn = (scale, shift) device vectors when the input is normalised by the loader."""
    from ._lib import ConvDesc
    d = ConvDesc(N, H, W, Ci, Co, k, stride, pad_t, pad_l, Ho, Wo, 0, None, None)
    if bn is not None:
        d.bn_scale, d.bn_shift = fptr(bn[0]), fptr(bn[1])
    return d


def conv_supported(d):
    return bool(_L().avsr_conv_supported(C.byref(d)))


def conv_fwd(d, x, w, bias, y, res=None, res_bn=None, stats=None):
    """Returns the number of statistic partial rows written (0 without stats)."""
    n = C.c_int32(0)
    check(_L().avsr_conv_fwd(C.byref(d), fptr(x), fptr(w), fptr(bias), fptr(res), fptr(res_bn[0]) if res_bn else None,
                             fptr(res_bn[1]) if res_bn else None, fptr(y), fptr(stats), C.byref(n), _s()), "avsr_conv_fwd")
    return int(n.value)


def conv_bwd_data(d, dy, w, dx, beta=0.0):
    check(_L().avsr_conv_bwd_data(C.byref(d), fptr(dy), fptr(w), fptr(dx), float(beta), _s()), "avsr_conv_bwd_data")


def conv_bwd_data_bn_supported(d):
    return bool(_L().avsr_conv_bwd_data_bn_supported(C.byref(d)))


def conv_bwd_data_bn(d, dy, w, dx, beta=0.0, acc=None, bn_x=None, bn=None, stats=None):
    """Data gradient with an accumulate source and / or the batch-norm backward epilogue; returns the number of partial rows."""
    n = C.c_int32(0)
    check(_L().avsr_conv_bwd_data_bn(C.byref(d), fptr(dy), fptr(w), fptr(dx), float(beta), fptr(acc), fptr(bn_x),
                                     fptr(bn[0]) if bn else None, fptr(bn[1]) if bn else None, fptr(stats), C.byref(n), _s()), "avsr_conv_bwd_data_bn")
    return int(n.value)


def bn_bwd_finalize(part, nparts, Cn, count, mean, invstd, gamma, dgamma, dbeta, k, grad_beta=1.0):
    check(_L().avsr_bn_bwd_finalize(fptr(part), int(nparts), int(Cn), int(count), fptr(mean), fptr(invstd), fptr(gamma), fptr(dgamma), fptr(dbeta),
                                    float(grad_beta), fptr(k), _s()), "avsr_bn_bwd_finalize")


def bn_partials_f64(part, nparts, Cn, out64):
    check(_L().avsr_bn_partials_f64(fptr(part), int(nparts), int(Cn), out64.data_ptr(), _s()), "avsr_bn_partials_f64")


def bn_finalize_f64(sums64, Cn, eps, momentum, mean, invstd, mov_mean, mov_var, gamma=None, beta=None, scale=None, shift=None):
    check(_L().avsr_bn_finalize_f64(sums64.data_ptr(), int(Cn), float(eps), float(momentum), fptr(mean), fptr(invstd), fptr(mov_mean), fptr(mov_var),
                                    fptr(gamma), fptr(beta), fptr(scale), fptr(shift), _s()), "avsr_bn_finalize_f64")


def bn_bwd_finalize_f64(local64, global64, Cn, mean, invstd, gamma, dgamma, dbeta, k, grad_beta=1.0):
    check(_L().avsr_bn_bwd_finalize_f64(local64.data_ptr(), global64.data_ptr(), int(Cn), fptr(mean), fptr(invstd), fptr(gamma), fptr(dgamma),
                                        fptr(dbeta), float(grad_beta), fptr(k), _s()), "avsr_bn_bwd_finalize_f64")


def bn_eval_affine(gamma, beta, mov_mean, mov_var, eps, scale, shift, Cn):
    check(_L().avsr_bn_eval_affine(fptr(gamma), fptr(beta), fptr(mov_mean), fptr(mov_var), float(eps), fptr(scale), fptr(shift), int(Cn), _s()),
          "avsr_bn_eval_affine")


def bn_bwd_apply(dz, x, k, dx, rows, Cn, beta=0.0):
    check(_L().avsr_bn_bwd_apply(fptr(dz), fptr(x), fptr(k), fptr(dx), int(rows), int(Cn), float(beta), _s()), "avsr_bn_bwd_apply")


def bn_bwd_stage1(dy, x, dz, rows, Cn, part, scale=None, shift=None, y=None):
    """dz = dy * [relu(scale*x + shift) > 0] (or [y > 0]) and its partial sums [nparts][2*Cn]; returns nparts."""
    n = C.c_int32(0)
    check(_L().avsr_bn_bwd_stage1(fptr(dy), fptr(x), fptr(scale), fptr(shift), fptr(y), fptr(dz), int(rows), int(Cn), fptr(part), C.byref(n), _s()),
          "avsr_bn_bwd_stage1")
    return int(n.value)


def conv_bwd_weight(d, x, dy, dw, dbias, scratch, beta=1.0):
    check(_L().avsr_conv_bwd_weight(C.byref(d), fptr(x), fptr(dy), fptr(dw), fptr(dbias), float(beta), fptr(scratch), scratch.numel(), _s()),
          "avsr_conv_bwd_weight")


def conv_bwd_weight_bn(d, x, dz, y, k, dw, dbias, scratch, beta=1.0, dx_out=None):
    """Weight gradient with dy = k1*dz + k2*y + k3 (the batch-norm backward of the convolution's own output) evaluated in the operand fetch;
    dx_out: also stored there for the layer's data gradient."""
    check(_L().avsr_conv_bwd_weight_bn(C.byref(d), fptr(x), fptr(dz), fptr(y), fptr(k), fptr(dx_out), fptr(dw), fptr(dbias), float(beta),
                                       fptr(scratch), scratch.numel(), _s()), "avsr_conv_bwd_weight_bn")


def conv_bwd_weight_bn_supported(d):
    return bool(_L().avsr_conv_bwd_weight_bn_supported(C.byref(d)))


def slab_defer_begin():
    """From here to slab_defer_end() the conv_bwd_weight calls of this thread record their final slab reductions instead of launching
    them: one launch at the end (every call needs its own scratch region)."""
    check(_L().avsr_slab_defer_begin(), "avsr_slab_defer_begin")


def slab_defer_end():
    check(_L().avsr_slab_defer_end(_s()), "avsr_slab_defer_end")


def bn_finalize(part, nparts, Cn, count, eps, momentum, mean, invstd, mov_mean, mov_var, gamma=None, beta=None, scale=None, shift=None):
    check(_L().avsr_bn_finalize(fptr(part), int(nparts), int(Cn), int(count), float(eps), float(momentum), fptr(mean), fptr(invstd),
                                fptr(mov_mean), fptr(mov_var), fptr(gamma), fptr(beta), fptr(scale), fptr(shift), _s()), "avsr_bn_finalize")


def batchnorm_apply(x, y, rows, F, gamma, beta, mean, invstd, relu):
    check(_L().avsr_batchnorm_apply(fptr(x), fptr(y), rows, F, fptr(gamma), fptr(beta), fptr(mean), fptr(invstd), int(relu), _s()),
          "avsr_batchnorm_apply")


def conv3x3_bwd_data_s2(dy, w, dx, N, H, W, Ci, Co, pad_t, pad_l, Ho, Wo, beta=0.0):
    check(_L().avsr_conv3x3_bwd_data_s2(fptr(dy), fptr(w), fptr(dx), N, H, W, Ci, Co, pad_t, pad_l, Ho, Wo, float(beta), _s()),
          "avsr_conv3x3_bwd_data_s2")


def conv3x3_bwd_weight(x, dy, dw, N, H, W, Ci, Co, stride, pad_t, pad_l, Ho, Wo, scratch, beta=1.0):
    check(_L().avsr_conv3x3_bwd_weight(fptr(x), fptr(dy), fptr(dw), N, H, W, Ci, Co, stride, pad_t, pad_l, Ho, Wo, float(beta), fptr(scratch),
                                       scratch.numel(), _s()), "avsr_conv3x3_bwd_weight")


def selu(z, y, n):
    check(_L().avsr_selu(fptr(z), fptr(y), int(n), _s()), "avsr_selu")


def selu_bwd(z, dy, dz, n):
    check(_L().avsr_selu_bwd(fptr(z), fptr(dy), fptr(dz), int(n), _s()), "avsr_selu_bwd")
