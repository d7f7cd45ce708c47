"""Containers and small helpers shared by the engine's modules (model.py, model_encoder.py, model_decoder.py): the device batch, views into the flat
parameter buffers, sequence buffers with guard slots, the flag reader of the decode loops."""
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
