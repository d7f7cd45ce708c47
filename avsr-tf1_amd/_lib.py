"""ctypes binding of libavsr_hip.so (C ABI declared in include/avsr_hip.h).

The product path has NO CPU fallback: if the shared library is missing or an entry point fails,
an exception is raised.  torch must be imported before the library is loaded so that the HIP
runtime (libamdhip64.so.7) already mapped by torch is the one our kernels launch on.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: shares torch's HIP runtime instance)

from . import build as _build

MAX_LAYERS = 4
MAX_MECH = 4
MAX_STACKS = 4

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class AvsrError(RuntimeError):
    pass


class Mat(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int64), ("T", C.c_int32), ("pad_", C.c_int32), ("ldo", C.c_int64)]


class GemmDesc(C.Structure):
    _fields_ = [("A", Mat), ("B", Mat), ("C", Mat), ("bias", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("trans_a", C.c_int32), ("trans_b", C.c_int32),
                ("alpha", C.c_float), ("beta", C.c_float), ("batch", C.c_int32),
                ("stride_a", C.c_int64), ("stride_b", C.c_int64), ("stride_c", C.c_int64),
                ("alpha_dev", C.c_void_p),
                ("splitk", C.c_int32), ("pad_", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_floats", C.c_int64),
                ("colsum", C.c_void_p), ("colsum_beta", C.c_float), ("pad2_", C.c_int32)]


class RnnLayer(C.Structure):
    _fields_ = [("units", C.c_int32), ("in_dim", C.c_int32), ("hoisted", C.c_int32), ("out_col", C.c_int32),
                ("wt", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
                ("gates", C.c_void_p), ("cs", C.c_void_p), ("out", C.c_void_p), ("ld_out", C.c_int64),
                ("state", C.c_void_p), ("h_final", C.c_void_p), ("c_final", C.c_void_p),
                ("dgates", C.c_void_p), ("dstate", C.c_void_p), ("dout", C.c_void_p), ("ld_dout", C.c_int64),
                ("dout_col", C.c_int32), ("residual", C.c_int32), ("hs_seq", C.c_void_p), ("xt_seq", C.c_void_p),
                ("wt2", C.c_void_p), ("w2", C.c_void_p), ("bias2", C.c_void_p), ("rh_seq", C.c_void_p), ("dgates2", C.c_void_p)]


class RnnStack(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("reverse", C.c_int32), ("n_layers", C.c_int32),
                ("cell", C.c_int32), ("pad_", C.c_int32),
                ("len", C.c_void_p), ("dh_final", C.c_void_p), ("dc_final", C.c_void_p),
                ("seed", C.c_void_p), ("keep_in", C.c_float), ("keep_state", C.c_float), ("keep_out", C.c_float),
                ("consumer_keep", C.c_float), ("cell_id_base", C.c_int32), ("consumer_stream", C.c_int32),
                ("consumer_width", C.c_int32), ("pad2_", C.c_int32),
                ("layer", RnnLayer * MAX_LAYERS)]


class AttnMech(C.Structure):
    _fields_ = [("type", C.c_int32), ("T", C.c_int32), ("D", C.c_int32), ("chunk", C.c_int32),
                ("len", C.c_void_p), ("keys", C.c_void_p), ("values", C.c_void_p),
                ("values_sb", C.c_int64), ("values_st", C.c_int64),
                ("g", C.c_void_p), ("v", C.c_void_p), ("bq", C.c_void_p),
                ("wq_t", C.c_void_p), ("wq", C.c_void_p), ("watt_t", C.c_void_p), ("watt", C.c_void_p),
                ("scores", C.c_void_p), ("ctx", C.c_void_p), ("pq", C.c_void_p), ("pstat", C.c_void_p),
                ("pctx", C.c_void_p), ("dscores", C.c_void_p), ("dctx", C.c_void_p), ("dpq", C.c_void_p),
                ("pdq", C.c_void_p)]


class DecLayer(C.Structure):
    _fields_ = [("wt", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("gates", C.c_void_p), ("cs", C.c_void_p),
                ("out", C.c_void_p), ("state", C.c_void_p), ("hs_seq", C.c_void_p), ("xin_seq", C.c_void_p),
                ("dgates", C.c_void_p), ("dstate", C.c_void_p), ("cell_id", C.c_int32), ("pad_", C.c_int32),
                ("wt2", C.c_void_p), ("w2", C.c_void_p), ("bias2", C.c_void_p), ("rh_seq", C.c_void_p), ("dgates2", C.c_void_p)]


MAX_DEC_EXTRA = 3


class AttnRnn(C.Structure):
    _fields_ = [("B", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("E", C.c_int32),
                ("n_mech", C.c_int32), ("output_attention", C.c_int32), ("V", C.c_int32), ("mode", C.c_int32),
                ("go_id", C.c_int32), ("eos_id", C.c_int32),
                ("steplen", C.c_void_p), ("wt", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
                ("gates", C.c_void_p), ("cs", C.c_void_p), ("cell_out", C.c_void_p), ("att", C.c_void_p),
                ("h0", C.c_void_p), ("c0", C.c_void_p), ("state", C.c_void_p),
                ("h_final", C.c_void_p), ("c_final", C.c_void_p),
                ("mech", AttnMech * MAX_MECH),
                ("embedding", C.c_void_p), ("wout_t", C.c_void_p), ("bout", C.c_void_p),
                ("logits", C.c_void_p), ("ids", C.c_void_p), ("tok", C.c_void_p), ("n_unfinished", C.c_void_p),
                ("dgates", C.c_void_p), ("dstate", C.c_void_p), ("datt", C.c_void_p), ("dq", C.c_void_p),
                ("datt_ext", C.c_void_p), ("dcell_ext", C.c_void_p), ("dh0", C.c_void_p), ("dc0", C.c_void_p),
                ("dh_final", C.c_void_p), ("dc_final", C.c_void_p),
                ("seed", C.c_void_p), ("keep_in", C.c_float), ("keep_state", C.c_float), ("keep_out", C.c_float),
                ("sampling_prob", C.c_float), ("cell_id", C.c_int32), ("pad3_", C.c_int32),
                ("hs_seq", C.c_void_p), ("attd", C.c_void_p), ("xs", C.c_void_p), ("labels", C.c_void_p), ("fed", C.c_void_p),
                ("cell", C.c_int32), ("pad4_", C.c_int32), ("wt2", C.c_void_p), ("w2", C.c_void_p), ("bias2", C.c_void_p),
                ("rh_seq", C.c_void_p), ("dgates2", C.c_void_p),
                ("beam_width", C.c_int32), ("mem_shared", C.c_int32), ("length_penalty", C.c_float), ("pad6_", C.c_float),
                ("beam_logp", C.c_void_p), ("beam_fin", C.c_void_p), ("beam_len", C.c_void_p), ("step_ids", C.c_void_p),
                ("parent_ids", C.c_void_p), ("parent_rows", C.c_void_p),
                ("n_extra", C.c_int32), ("prof_tag", C.c_int32), ("out0", C.c_void_p), ("extra", DecLayer * MAX_DEC_EXTRA),
                ("fused_ws", C.c_void_p), ("fused_ws_floats", C.c_int64)]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "H", "W", "Ci", "Co", "k", "stride", "pad_t", "pad_l", "Ho", "Wo", "pad_")] + \
               [("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p)]


class ColsumJob(C.Structure):
    _fields_ = [("a", Mat), ("b", Mat), ("out", C.c_void_p), ("rows", C.c_int32), ("F", C.c_int32), ("alpha", C.c_float), ("beta", C.c_float)]


class TransposeJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]


_STRUCTS = {"avsr_dec_layer": DecLayer, "avsr_mat": Mat, "avsr_gemm_desc": GemmDesc, "avsr_rnn_layer": RnnLayer, "avsr_rnn_stack": RnnStack,
            "avsr_conv_desc": ConvDesc, "avsr_attn_mech": AttnMech, "avsr_attn_rnn": AttnRnn, "avsr_transpose_job": TransposeJob,
            "avsr_colsum_job": ColsumJob}

EXPORTS = ["avsr_abi_version", "avsr_sizeof", "avsr_gemm", "avsr_gemm_batch", "avsr_rnn_fwd", "avsr_rnn_bwd", "avsr_rnn_set_persistent", "avsr_rnn_set_persistent_mode", "avsr_rnn_set_persistent_scratch", "avsr_attn_rnn_fwd",
           "avsr_attn_rnn_fused_ws_floats", "avsr_attn_rnn_fused_eligible", "avsr_attn_rnn_fused_fwd_active", "avsr_attn_rnn_set_fused", "avsr_attn_rnn_set_beam_kernel", "avsr_conv_set_mfma", "avsr_conv_supported", "avsr_conv_fwd", "avsr_conv_bwd_data", "avsr_conv_bwd_weight", "avsr_bn_finalize", "avsr_batchnorm_apply", "avsr_conv_bwd_data_bn", "avsr_conv_bwd_data_bn_supported", "avsr_conv_bwd_weight_bn", "avsr_conv_bwd_weight_bn_supported", "avsr_bn_bwd_finalize", "avsr_bn_bwd_apply", "avsr_bn_bwd_stage1", "avsr_bn_eval_affine", "avsr_bn_partials_f64", "avsr_bn_finalize_f64", "avsr_bn_bwd_finalize_f64",
           "avsr_attn_rnn_bwd", "avsr_beam_gather_tree", "avsr_beam_search_step", "avsr_attn_alpha_rows", "avsr_bahdanau_dkeys", "avsr_transpose", "avsr_slab_defer_begin", "avsr_slab_defer_end", "avsr_colsum",
           "avsr_batchnorm_fwd", "avsr_batchnorm_fwd_ex", "avsr_batchnorm_bwd", "avsr_batchnorm_xhat", "avsr_im2col", "avsr_col2im",
           "avsr_relu", "avsr_relu_bwd", "avsr_add", "avsr_selu", "avsr_selu_bwd", "avsr_conv3x3_supported", "avsr_conv3x3", "avsr_conv3x3_bwd_data_s2",
           "avsr_conv3x3_bwd_weight", "avsr_embed_labels", "avsr_embed_grad", "avsr_dropout_rows", "avsr_seq_loss",
           "avsr_au_loss", "avsr_au_loss_dp", "avsr_normed_v", "avsr_normed_v_bwd", "avsr_reduce_scalar", "avsr_l2_regularise",
           "avsr_global_norm", "avsr_adam_step", "avsr_adam_step_decay", "avsr_copy_words", "avsr_zero_words", "avsr_zero_multi", "avsr_add_int", "avsr_colsum_multi", "avsr_highway_fwd", "avsr_highway_bwd", "avsr_optimiser_step", "avsr_instnorm_fwd", "avsr_instnorm_bwd", "avsr_seq_loss_fun", "avsr_seq_loss_per_utterance", "avsr_batchnorm_sync_sum", "avsr_batchnorm_sync_sqsum", "avsr_batchnorm_sync_apply", "avsr_batchnorm_sync_moments", "avsr_dp_sync_unpack", "avsr_prof_begin", "avsr_prof_end"]

_lib = None


def lib_path():
    return _build.LIB


def load():
    """Load (building first if sources are newer and hipcc is available).  Raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    # AVSR_LIB: load another build of the library (A/B timing of kernel variants: tools/build_variant.py); never set in product use
    path = os.environ.get("AVSR_LIB") or _build.LIB
    if path == _build.LIB and _build.needs_build():
        try:
            _build.build()
        except Exception as e:  # no hipcc on the box: fall through to a prebuilt .so if present
            if not os.path.exists(path):
                raise AvsrError("libavsr_hip.so is missing and could not be built: %s" % e)
    lib = C.CDLL(path)
    lib.avsr_abi_version.restype = C.c_int
    lib.avsr_sizeof.restype = C.c_int64
    lib.avsr_sizeof.argtypes = [C.c_char_p]
    for sym in EXPORTS:
        if not hasattr(lib, sym):
            raise AvsrError("libavsr_hip.so does not export %s" % sym)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    sigs = {
        "avsr_gemm": [C.POINTER(GemmDesc), vp],
        "avsr_gemm_batch": [C.POINTER(GemmDesc), i32, vp],
        "avsr_rnn_fwd": [C.POINTER(RnnStack), i32, vp],
        "avsr_rnn_bwd": [C.POINTER(RnnStack), i32, vp],
        "avsr_rnn_set_persistent": [vp, i64],
        "avsr_rnn_set_persistent_mode": [i32],
        "avsr_rnn_set_persistent_scratch": [vp, i64],
        "avsr_attn_rnn_fwd": [C.POINTER(AttnRnn), i32, i32, vp],
        "avsr_attn_rnn_fused_eligible": [C.POINTER(AttnRnn)],
        "avsr_attn_rnn_fused_fwd_active": [C.POINTER(AttnRnn)],
        "avsr_attn_rnn_set_fused": [i32],
        "avsr_attn_rnn_set_beam_kernel": [i32],
        "avsr_conv_set_mfma": [i32],
        "avsr_conv_supported": [C.POINTER(ConvDesc)],
        "avsr_conv_fwd": [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i32), vp],
        "avsr_conv_bwd_data": [C.POINTER(ConvDesc), vp, vp, vp, f32, vp],
        "avsr_conv_bwd_weight": [C.POINTER(ConvDesc), vp, vp, vp, vp, f32, vp, i64, vp],
        "avsr_conv_bwd_data_bn": [C.POINTER(ConvDesc), vp, vp, vp, f32, vp, vp, vp, vp, vp, C.POINTER(i32), vp],
        "avsr_conv_bwd_data_bn_supported": [C.POINTER(ConvDesc)],
        "avsr_conv_bwd_weight_bn": [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, f32, vp, i64, vp],
        "avsr_conv_bwd_weight_bn_supported": [C.POINTER(ConvDesc)],
        "avsr_bn_bwd_finalize": [vp, i32, i32, i64, vp, vp, vp, vp, vp, f32, vp, vp],
        "avsr_bn_bwd_apply": [vp, vp, vp, vp, i64, i32, f32, vp],
        "avsr_bn_bwd_stage1": [vp, vp, vp, vp, vp, vp, i64, i32, vp, C.POINTER(i32), vp],
        "avsr_bn_partials_f64": [vp, i32, i32, vp, vp],
        "avsr_bn_finalize_f64": [vp, i32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "avsr_bn_bwd_finalize_f64": [vp, vp, i32, vp, vp, vp, vp, vp, f32, vp, vp],
        "avsr_bn_eval_affine": [vp, vp, vp, vp, f32, vp, vp, i32, vp],
        "avsr_bn_finalize": [vp, i32, i32, i64, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "avsr_batchnorm_apply": [vp, vp, i32, i32, vp, vp, vp, vp, i32, vp],
        "avsr_attn_rnn_bwd": [C.POINTER(AttnRnn), vp],
        "avsr_beam_gather_tree": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "avsr_beam_search_step": [vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp],
        "avsr_attn_alpha_rows": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
        "avsr_bahdanau_dkeys": [vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "avsr_transpose": [C.POINTER(TransposeJob), i32, vp],
        "avsr_colsum": [C.POINTER(Mat), C.POINTER(Mat), i32, i32, f32, f32, vp, vp, i64, vp],
        "avsr_batchnorm_fwd": [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, i64, vp],
        "avsr_batchnorm_xhat": [vp, vp, vp, vp, i32, i32, vp],
        "avsr_batchnorm_fwd_ex": [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, f32, f32, i32, i32, vp, i64, vp],
        "avsr_batchnorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp, i64, vp],
        "avsr_im2col": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
        "avsr_col2im": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp],
        "avsr_relu": [vp, vp, i64, vp],
        "avsr_relu_bwd": [vp, vp, vp, i64, vp],
        "avsr_add": [vp, vp, vp, i64, vp],
        "avsr_selu": [vp, vp, i64, vp],
        "avsr_selu_bwd": [vp, vp, vp, i64, vp],
        "avsr_conv3x3_supported": [i32, i32, i32, i32],
        "avsr_conv3x3": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp],
        "avsr_conv3x3_bwd_data_s2": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp],
        "avsr_conv3x3_bwd_weight": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, i64, vp],
        "avsr_embed_labels": [vp, vp, i32, vp, vp, i32, i32, i32, i32, vp],
        "avsr_embed_grad": [vp, vp, vp, i32, i32, i32, i32, vp, i64, vp],
        "avsr_dropout_rows": [C.POINTER(Mat), C.POINTER(Mat), i32, i32, vp, i32, f32, i32, i32, i32, vp],
        "avsr_seq_loss": [vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp],
        "avsr_seq_loss_fun": [vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, f32, vp],
        "avsr_au_loss": [vp, vp, vp, vp, vp, i32, i32, f32, vp],
        "avsr_au_loss_dp": [vp, vp, vp, vp, vp, i32, i32, f32, vp, vp],
        "avsr_normed_v": [vp, vp, vp, i32, vp],
        "avsr_normed_v_bwd": [vp, vp, vp, vp, vp, i32, vp],
        "avsr_reduce_scalar": [vp, i32, vp, i32, i32, f32, vp],
        "avsr_l2_regularise": [C.POINTER(i64), C.POINTER(i64), i32, vp, vp, f32, vp, vp, vp],
        "avsr_global_norm": [vp, i64, f32, vp, vp, vp],
        "avsr_adam_step": [vp, vp, vp, vp, i64, vp, vp, f32, i32, f32, f32, vp],
        "avsr_batchnorm_sync_sum": [vp, i32, i32, vp, vp, i64, vp],
        "avsr_batchnorm_sync_sqsum": [vp, i32, i32, vp, vp, vp, vp, vp, i64, vp],
        "avsr_batchnorm_sync_apply": [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, i32, vp],
        "avsr_batchnorm_sync_moments": [vp, i32, i32, vp, vp, i64, vp],
        "avsr_dp_sync_unpack": [vp, vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp],
        "avsr_seq_loss_per_utterance": [vp, vp, vp, vp, i32, i32, vp],
        "avsr_instnorm_fwd": [vp, vp, i32, i32, i32, vp, vp, vp, vp, f32, vp],
        "avsr_instnorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
        "avsr_optimiser_step": [vp, vp, vp, vp, i64, vp, vp, f32, i32, i32, f32, f32, i32, f32, vp],
        "avsr_highway_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
        "avsr_highway_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "avsr_copy_words": [vp, vp, i64, vp],
        "avsr_zero_words": [vp, i64, vp],
        "avsr_zero_multi": [C.POINTER(vp), C.POINTER(i64), i32, vp],
        "avsr_add_int": [vp, i32, vp, vp],
        "avsr_colsum_multi": [C.POINTER(ColsumJob), i32, vp, i64, vp],
        "avsr_slab_defer_begin": [], "avsr_slab_defer_end": [vp],
        "avsr_adam_step_decay": [vp, vp, vp, vp, i64, vp, vp, f32, i32, i32, f32, f32, vp],
        "avsr_prof_begin": [i32],
        "avsr_prof_end": [C.POINTER(i32), C.POINTER(f32), C.POINTER(C.c_double)],
    }
    for name, at in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = at
        fn.restype = C.c_int
    lib.avsr_attn_rnn_fused_ws_floats.argtypes = [i32, i32, i32]
    lib.avsr_attn_rnn_fused_ws_floats.restype = C.c_int64
    for name, st in _STRUCTS.items():
        n = lib.avsr_sizeof(name.encode())
        if n != C.sizeof(st):
            raise AvsrError("ABI mismatch for %s: C %d vs ctypes %d" % (name, n, C.sizeof(st)))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise AvsrError("%s failed with code %d" % (what, rc))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream
