"""TF-layout <-> engine-layout weight conversion (numpy, host side).

TF stores `LSTMCell.kernel` as [in+H, 4H] with gate blocks i, j, f, o (rnn_cell_impl.py); the
engine interleaves gates per unit (column u*4+g) so a 16-column MFMA tile holds 4 whole units."""
import numpy as np


def lstm_kernel_to_engine(W):
    K, H4 = W.shape
    H = H4 // 4
    return np.ascontiguousarray(W.reshape(K, 4, H).transpose(0, 2, 1).reshape(K, H4))


def lstm_kernel_from_engine(W):
    K, H4 = W.shape
    H = H4 // 4
    return np.ascontiguousarray(W.reshape(K, H, 4).transpose(0, 2, 1).reshape(K, H4))


def lstm_bias_to_engine(b):
    H = b.shape[0] // 4
    return np.ascontiguousarray(b.reshape(4, H).T.reshape(4 * H))


def lstm_bias_from_engine(b):
    H = b.shape[0] // 4
    return np.ascontiguousarray(b.reshape(H, 4).T.reshape(4 * H))
