"""ctypes binding of libavsr_io.so (include/avsr_io.h, csrc/io_native.c): the native SequenceExample indexer + batch filler of the
input pipeline.  Host-side helper (gcc, no GPU): when the library cannot be built or loaded the pipeline uses its python parser, which
gives the same batches (tests/test_io.py compares the two)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "io_native.c")
LIB = os.path.join(_HERE, "csrc", "libavsr_io.so")
INC = os.path.join(os.path.dirname(_HERE), "include")
NFIELD = 16                      # int64 fields of avsr_io_rec
F = {n: i for i, n in enumerate(("slow", "input_length", "labels_length", "fn_off", "fn_len", "in_off", "in_stride", "in_T", "in_F",
                                 "aus_off", "aus_stride", "aus_T", "aus_F", "lab_off", "lab_stride", "lab_n"))}
_lib = None
_failed = False
THREADS = max(1, min(8, (os.cpu_count() or 2) // 2))


def build(force=False):
    """gcc -O3 -shared; returns the library path.  Raises when gcc is missing or the compile fails."""
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(INC, "avsr_io.h"))):
        return LIB
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found: the native input-pipeline helper cannot be built")
    # compile to a private temporary file and rename it into place: under torchrun several ranks may get here at once, and a rank
    # must never CDLL a library another rank is still writing (os.replace is atomic on one file system)
    tmp = "%s.%d.tmp" % (LIB, os.getpid())
    try:
        subprocess.check_call([gcc, "-O3", "-std=c99", "-Wall", "-shared", "-fPIC", "-pthread", "-I", INC, SRC, "-o", tmp])
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


def load():
    """The library, or None when it is unavailable (the caller then parses in python)."""
    global _lib, _failed
    if _lib is not None or _failed:
        return _lib
    if os.environ.get("AVSR_IO_NATIVE") == "0":
        _failed = True
        return None
    try:
        lib = C.CDLL(build())
        if lib.avsr_io_abi_version() != 2:
            lib = C.CDLL(build(force=True))                  # a stale build of an older ABI: rebuild once
            if lib.avsr_io_abi_version() != 2:
                raise RuntimeError("libavsr_io.so: ABI version mismatch")
        pp, pl = C.POINTER(C.c_void_p), C.POINTER(C.c_int64)
        lib.avsr_io_index.argtypes = [C.c_int32, pp, pl, C.c_void_p, C.c_int32]
        lib.avsr_io_fill_f32.argtypes = [C.c_int32, pp, pl, pl, pl, pl, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32]
        lib.avsr_io_fill_labels.argtypes = [C.c_int32, pp, pl, pl, pl, C.c_int32, C.c_void_p, C.c_int64]
        _lib = lib
    except Exception as e:
        # not silent: ranks that ended up on different parsers would be hard to notice (the batches are the same, the speed is not)
        import warnings
        warnings.warn("avsr_tf1_amd: native input-pipeline helper unavailable (%s: %s); using the python parser" % (type(e).__name__, e))
        _failed = True
    return _lib


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(C.POINTER(C.c_int64))


def _ptrs(payloads):
    """Pointer array over payloads that are `bytes` or memory-mapped record windows (io_utils._Rec: `.addr`).  The caller keeps the
    payloads alive for the duration of the native call."""
    n = len(payloads)
    arr = (C.c_void_p * n)()
    for i, p in enumerate(payloads):
        a = getattr(p, "addr", None)
        arr[i] = a if a is not None else C.cast(C.c_char_p(p), C.c_void_p).value
    return arr


def index(payloads):
    """[bytes] -> int64 array [n, NFIELD] (fields: F)."""
    n = len(payloads)
    out = np.zeros((n, NFIELD), np.int64)
    lens, lp = _i64([len(p) for p in payloads])
    _lib.avsr_io_index(n, _ptrs(payloads), lp, out.ctypes.data, THREADS)
    return out


def fill_f32(payloads, off, stride, steps, step_floats, Tmax, row_shape, out=None):
    """Zero-padded [n, Tmax, *row_shape] float32 batch from the records' value regions; `out`: a C-contiguous float32 array of that
    shape to fill (a window of a reused, possibly page-locked buffer) instead of a fresh allocation."""
    n = len(payloads)
    row = int(np.prod(row_shape)) if len(row_shape) else 1
    if out is not None:
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape == (n, Tmax) + tuple(row_shape)
        dst = out
    else:
        dst = np.empty((n, Tmax) + tuple(row_shape), np.float32)      # the helper writes every byte (values, then zero padding)
    (o, op), (s, sp), (t, tp), (ln, lp) = _i64(off), _i64(stride), _i64(steps), _i64([len(p) for p in payloads])
    _lib.avsr_io_fill_f32(n, _ptrs(payloads), lp, op, sp, tp, int(step_floats), dst.ctypes.data, int(Tmax), row, THREADS)
    return dst


def fill_labels(payloads, off, stride, cnt, eos, Lmax):
    n = len(payloads)
    dst = np.zeros((n, Lmax), np.int32)
    (o, op), (s, sp), (c, cp) = _i64(off), _i64(stride), _i64(cnt)
    _lib.avsr_io_fill_labels(n, _ptrs(payloads), op, sp, cp, int(eos), dst.ctypes.data, int(Lmax))
    return dst
