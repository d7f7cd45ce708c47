"""Thin Python wrappers over the C ABI: torch tensors are only containers for device memory."""
import ctypes as C

import torch

from . import _lib
from ._lib import GemmDesc, Mat, RnnLayer, RnnStack, check, ptr, stream_ptr


def fptr(t, offset=0):
    """Device address of element `offset` of a float32 tensor."""
    if t is None:
        return None
    assert t.dtype == torch.float32 and t.is_cuda
    return t.data_ptr() + 4 * int(offset)


def mat(t, ld, T=0, ldo=0, offset=0):
    return Mat(fptr(t, offset), int(ld), int(T), 0, int(ldo))


def gemm(A, B, Cm, M, N, K, trans_a=False, trans_b=False, alpha=1.0, beta=0.0, bias=None,
         batch=1, strides=(0, 0, 0), splitk=1, workspace=None):
    """C = alpha*op(A)*op(B) + beta*C + bias.  A, B, Cm are `Mat` views (see `mat`)."""
    d = GemmDesc()
    d.A, d.B, d.C = A, B, Cm
    d.bias = fptr(bias)
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.trans_a, d.trans_b = int(trans_a), int(trans_b)
    d.alpha, d.beta = float(alpha), float(beta)
    d.batch = int(batch)
    d.stride_a, d.stride_b, d.stride_c = [int(s) for s in strides]
    d.splitk = int(splitk)
    if splitk > 1:
        need = batch * splitk * M * N
        assert workspace is not None and workspace.numel() >= need, "split-K workspace too small"
        d.workspace = fptr(workspace)
        d.workspace_floats = workspace.numel()
    check(_lib.load().avsr_gemm(C.byref(d), C.c_void_p(stream_ptr())), "avsr_gemm")


def rnn_fwd(stacks):
    arr = (RnnStack * len(stacks))(*stacks)
    check(_lib.load().avsr_rnn_fwd(arr, C.c_int32(len(stacks)), C.c_void_p(stream_ptr())), "avsr_rnn_fwd")


def rnn_bwd(stacks):
    arr = (RnnStack * len(stacks))(*stacks)
    check(_lib.load().avsr_rnn_bwd(arr, C.c_int32(len(stacks)), C.c_void_p(stream_ptr())), "avsr_rnn_bwd")
