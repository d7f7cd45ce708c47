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
                ("splitk", C.c_int32), ("pad_", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_floats", C.c_int64)]


class RnnLayer(C.Structure):
    _fields_ = [("units", C.c_int32), ("in_dim", C.c_int32), ("hoisted", C.c_int32), ("out_col", C.c_int32),
                ("wt", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
                ("gates", C.c_void_p), ("cs", C.c_void_p), ("out", C.c_void_p), ("ld_out", C.c_int64),
                ("state", C.c_void_p), ("h_final", C.c_void_p), ("c_final", C.c_void_p),
                ("dgates", C.c_void_p), ("dstate", C.c_void_p), ("dout", C.c_void_p), ("ld_dout", C.c_int64),
                ("dout_col", C.c_int32), ("pad_", C.c_int32)]


class RnnStack(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("reverse", C.c_int32), ("n_layers", C.c_int32),
                ("cell", C.c_int32), ("pad_", C.c_int32),
                ("len", C.c_void_p), ("dh_final", C.c_void_p), ("dc_final", C.c_void_p),
                ("layer", RnnLayer * MAX_LAYERS)]


_STRUCTS = {"avsr_mat": Mat, "avsr_gemm_desc": GemmDesc, "avsr_rnn_layer": RnnLayer, "avsr_rnn_stack": RnnStack}

_lib = None


def lib_path():
    return _build.LIB


def load():
    """Load (building first if sources are newer and hipcc is available).  Raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if _build.needs_build():
        try:
            _build.build()
        except Exception as e:  # no hipcc on the box: fall through to a prebuilt .so if present
            if not os.path.exists(path):
                raise AvsrError("libavsr_hip.so is missing and could not be built: %s" % e)
    lib = C.CDLL(path)
    lib.avsr_abi_version.restype = C.c_int
    lib.avsr_sizeof.restype = C.c_int64
    lib.avsr_sizeof.argtypes = [C.c_char_p]
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
