"""Data-parallel algebra on CPU: world_size 2, gloo.  The HIP model is replaced by an oracle-backed stand-in that
exposes the same attributes DataParallelTrainer drives (denom / grads / loss / forward_train / backward /
apply_update), so the test exercises exactly the collective logic that runs on RCCL:
  sharded utterances + all-reduced loss normaliser + summed gradients  ==  one process on the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from avsr_tf1_amd.model import Batch
from avsr_tf1_amd.parallel import DataParallelTrainer
from oracle import avsr_oracle as O


def _cfg():
    return O.OracleConfig(architecture="bimodal", video_units=(8,), audio_units=(8, 8), decoder_units=(8,), embedding_size=4,
                          video_feat=4, audio_feat=8, batch_normalisation=True, regress_aus=False)


class OracleBackedModel:
    def __init__(self, cfg, W):
        self.cfg, self.W, self.opt = cfg, {k: v.copy() for k, v in W.items()}, None
        self.names = O.trainable_names(W)
        self.sizes = [W[k].size for k in self.names]
        self.grads = torch.zeros(sum(self.sizes), dtype=torch.float64)
        self.denom, self.loss, self.gnorm = torch.zeros(1), torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)
        self.au_scale = 1.0
        self.bn_sync = None

    # the three sync-BN hooks DataParallelTrainer drives (model.py: bn_sync_enable / bn_sync_sums / bn_sync_squares)
    def bn_sync_enable(self):
        feats = {"video": self.cfg.video_feat, "audio": self.cfg.audio_feat}
        streams = [s for s in ("video", "audio") if getattr(self.cfg, s + "_units") is not None]
        n = sum(feats[s] for s in streams)
        self.bn_sync = dict(streams=streams, feats=feats, sum=torch.zeros(n + len(streams), dtype=torch.float64),
                            sq=torch.zeros(n, dtype=torch.float64), n=n)
        return self.bn_sync

    def _rows(self, batch, s):
        x = getattr(batch, s).to(torch.float64)
        return x.reshape(-1, x.shape[-1])

    def bn_sync_sums(self, batch):
        bs, o = self.bn_sync, 0
        for i, s in enumerate(bs["streams"]):
            x = self._rows(batch, s)
            bs["sum"][o:o + x.shape[1]] = x.sum(0)
            bs["sum"][bs["n"] + i] = x.shape[0]
            o += x.shape[1]
        return bs["sum"]

    def bn_sync_squares(self, batch):
        bs, o = self.bn_sync, 0
        bs["mean"] = {}
        for i, s in enumerate(bs["streams"]):
            x = self._rows(batch, s)
            bs["mean"][s] = bs["sum"][o:o + x.shape[1]] / bs["sum"][bs["n"] + i]
            bs["sq"][o:o + x.shape[1]] = ((x - bs["mean"][s]) ** 2).sum(0)
            o += x.shape[1]
        return bs["sq"]

    def _bn_stats(self):
        if self.bn_sync is None:
            return None
        bs, o, out = self.bn_sync, 0, {}
        for i, s in enumerate(bs["streams"]):
            F, n = bs["feats"][s], float(bs["sum"][bs["n"] + i])
            out[s] = (bs["mean"][s].clone(), bs["sq"][o:o + F] / n, n)
            o += F
        return out

    def forward_train(self, batch, compute_denom=True):
        nb = O.Batch(**{k: (None if getattr(batch, k) is None else getattr(batch, k).numpy())
                        for k in ("audio", "audio_len", "video", "video_len", "aus", "labels", "labels_len")})
        self._P = O.to_torch(self.W, torch.float64, requires_grad=True)
        logits, _m = O.forward_train(self._P, self.cfg, nb, bn_stats=self._bn_stats())
        self._bn_updates = _m.bn_updates
        labels = torch.as_tensor(nb.labels, dtype=torch.int64)
        ll = torch.as_tensor(nb.labels_len, dtype=torch.int64)
        w = (torch.arange(labels.shape[1])[None, :] < ll[:, None]).to(torch.float64)
        if compute_denom:
            self.denom.copy_(w.sum().to(torch.float32).reshape(1))
        ce = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), reduction="none").reshape(labels.shape)
        self._loss = torch.sum(ce * w) / (self.denom.to(torch.float64)[0] + 1e-12)
        self.loss.copy_(self._loss.detach().reshape(1))

    def backward(self):
        gs = torch.autograd.grad(self._loss, [self._P[k] for k in self.names], allow_unused=True)
        flat = [(g if g is not None else torch.zeros_like(self._P[k])).reshape(-1) for g, k in zip(gs, self.names)]
        self.grads.copy_(torch.cat(flat))

    def apply_update(self):
        cfg = self.cfg
        g = dict(zip(self.names, torch.split(self.grads.clone(), self.sizes)))
        for k in O.l2_names(self.W, cfg):
            g[k] = g[k] + cfg.recurrent_l2 * torch.tensor(self.W[k], dtype=torch.float64).reshape(-1)
        gn = torch.sqrt(sum(torch.sum(x * x) for x in g.values()))
        self.gnorm.copy_(gn.reshape(1))
        scale = cfg.max_gradient_norm / max(float(gn), cfg.max_gradient_norm)
        if self.opt is None:
            self.opt = {"step": 0, "m": {k: np.zeros(self.W[k].size) for k in self.names}, "v": {k: np.zeros(self.W[k].size) for k in self.names}}
        t = self.opt["step"] + 1
        lr_t = O.lr_at(cfg, self.opt["step"]) * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        for k in self.names:
            gi = g[k].numpy() * scale
            self.opt["m"][k] = 0.9 * self.opt["m"][k] + 0.1 * gi
            self.opt["v"][k] = 0.999 * self.opt["v"][k] + 0.001 * gi * gi
            upd = lr_t * self.opt["m"][k] / (np.sqrt(self.opt["v"][k]) + 1e-8)
            self.W[k] = (self.W[k].astype(np.float64) - upd.reshape(self.W[k].shape)).astype(np.float32)
        self.opt["step"] = t
        for k, v in self._bn_updates.items():                    # UPDATE_OPS: moving statistics of the (global) batch
            self.W[k] = v.detach().numpy().astype(np.float32)


class FusedSyncModel(OracleBackedModel):
    """The same stand-in with the hooks of the fused transport (model.py dp_sync_pack / dp_sync_unpack / grads_and_loss): ONE small
    fp64 collective carrying the loss normaliser and the batch-norm moments (variance = E[x^2] - mean^2 in fp64), and the loss summed
    with the gradients in the tail of one buffer."""

    def __init__(self, cfg, W):
        super().__init__(cfg, W)
        n = sum(self.sizes)
        self.grads_and_loss = torch.zeros(n + 4, dtype=torch.float64)
        self.grads = self.grads_and_loss[:n]
        self.loss = self.grads_and_loss[n:n + 1]
        self.calls = []

    def dp_sync_pack(self, batch):
        bs = self.bn_sync
        streams = bs["streams"] if bs else []
        L = batch.labels.shape[1]
        parts = [batch.labels_len.clamp(0, L).sum().to(torch.float64).reshape(1), torch.zeros(1, dtype=torch.float64)]
        for s in streams:
            x = self._rows(batch, s)
            parts += [x.sum(0), (x * x).sum(0), torch.tensor([float(x.shape[0])], dtype=torch.float64)]
        self._buf = torch.cat(parts)
        self.calls.append("pack")
        return self._buf

    def dp_sync_unpack(self):
        self.denom.copy_(self._buf[0:1].to(torch.float32))
        bs, o = self.bn_sync, 2
        if bs:
            bs["mean"], self._var = {}, {}
            for s in bs["streams"]:
                F = bs["feats"][s]
                n = float(self._buf[o + 2 * F])
                mean = self._buf[o:o + F] / n
                bs["mean"][s] = mean
                self._var[s] = (torch.clamp(self._buf[o + F:o + 2 * F] / n - mean * mean, min=0.0), n)
                o += 2 * F + 1
        self.calls.append("unpack")

    def _bn_stats(self):
        if self.bn_sync is None:
            return None
        return {s: (self.bn_sync["mean"][s].clone(), self._var[s][0], self._var[s][1]) for s in self.bn_sync["streams"]}


def _shard(b, lo, hi):
    def f(a, dt):
        return None if a is None else torch.as_tensor(np.ascontiguousarray(a[lo:hi]), dtype=dt)
    return Batch(f(b.audio, torch.float32), f(b.audio_len, torch.int32), f(b.video, torch.float32), f(b.video_len, torch.int32),
                 f(b.aus, torch.float32), f(b.labels, torch.int32), f(b.labels_len, torch.int32))


def _worker(rank, world, port, out_dir, fused=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = _cfg()
    W = O.init_params(cfg, seed=9)
    full = O.synthetic_batch(cfg, B=4, T_a=7, T_v=4, L=5, ragged=True)
    per = 4 // world
    model = (FusedSyncModel if fused else OracleBackedModel)(cfg, W)
    trainer = DataParallelTrainer(model, dist, use_graph=False)
    ncoll = [0]
    real = dist.all_reduce

    def counting(*a, **k):
        ncoll[0] += 1
        return real(*a, **k)
    dist.all_reduce = counting
    for _ in range(2):
        trainer.train_step(_shard(full, rank * per, (rank + 1) * per))
    dist.all_reduce = real
    if fused:                                                   # two collectives per step: the packed normalisers, gradients + loss
        assert ncoll[0] == 4 and model.calls == ["pack", "unpack"] * 2, (ncoll, model.calls)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), gnorm=model.gnorm.numpy(), **model.W)
    dist.destroy_process_group()


@pytest.mark.parametrize("fused", [False, True])
def test_two_rank_data_parallel_equals_single_process(tmp_path, fused):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), fused), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    cfg = _cfg()
    W = O.init_params(cfg, seed=9)
    full = O.synthetic_batch(cfg, B=4, T_a=7, T_v=4, L=5, ragged=True)
    a = O.train_step(W, None, cfg, full)
    b = O.train_step(a["params"], a["opt"], cfg, full)
    assert abs(float(r0["gnorm"][0]) - b["global_norm"]) < 1e-9
    for k in O.trainable_names(W):
        assert np.array_equal(r0[k], r1[k]), k                  # replicas stay bit-identical
        assert np.abs(r0[k] - b["params"][k]).max() < 1e-6, k   # and equal the single-process step on the whole batch
    for k in W:                                                 # sync-BN: moving statistics are those of the GLOBAL batch
        if "moving_" in k:
            assert np.array_equal(r0[k], r1[k]) and np.abs(r0[k] - b["params"][k]).max() < 1e-6, k


# ---- out-of-memory while a rank prepares a new shape's capture: the collective schedule must stay shared ----------------
class _TinyModel:
    """Minimal stand-in: gradient = batch mean of the labels, one parameter; records every pass."""

    def __init__(self, oom_on_prepare):
        self.grads_and_loss = torch.zeros(8, dtype=torch.float64)
        self.grads, self.loss = self.grads_and_loss[:4], self.grads_and_loss[4:5]
        self.gnorm, self.denom = torch.zeros(1, dtype=torch.float64), torch.zeros(1)
        self.param = torch.zeros(4, dtype=torch.float64)
        self.oom_on_prepare, self.prepared, self.passes, self.flag_checks = oom_on_prepare, 0, 0, 0

    def prepare_workspace(self, batch):
        self.prepared += 1
        if self.oom_on_prepare and self.prepared == 1:
            raise torch.cuda.OutOfMemoryError("simulated")

    def forward_train(self, batch, compute_denom=True):
        self.passes += 1
        self._x = batch.labels.to(torch.float64).sum()

    def backward(self):
        self.grads.fill_(float(self._x))
        self.loss.fill_(float(self._x))

    def apply_update(self):
        self.param -= 0.1 * self.grads
        self.gnorm.copy_(self.grads.norm().reshape(1))

    def check_persistent(self, disable=True, force=False):
        self.flag_checks += 1
        return False


class _FakeGraph:
    def __init__(self, fn):           # (a capture records, it does not execute)
        self.fn = fn

    def replay(self):
        self.fn()


def _oom_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the capture path's GPU plumbing, replaced for the CPU run: the test is about WHICH collectives each rank issues
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda: None
    DataParallelTrainer._drain = staticmethod(lambda: None)
    DataParallelTrainer._copy_into = staticmethod(lambda dst, src: dst.copy_(src))
    model = _TinyModel(oom_on_prepare=(rank == 1))
    trainer = DataParallelTrainer(model, dist, use_graph=True, check_every_step=True)
    trainer._capture = lambda fn: _FakeGraph(fn)
    ncoll, real = [0], dist.all_reduce

    def counting(*a, **k):
        ncoll[0] += 1
        return real(*a, **k)
    dist.all_reduce = counting
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt)
    for step in range(4):
        b = Batch(z(2, 3, 4), z(2, dt=torch.int32), None, None, None, torch.full((2, 5), rank + step + 1, dtype=torch.int32),
                  torch.full((2,), 5, dtype=torch.int32))
        trainer.train_step(b)
    dist.all_reduce = real
    np.savez(os.path.join(out_dir, "oom%d.npz" % rank), param=model.param.numpy(), ncoll=ncoll[0], passes=model.passes,
             graph=int(trainer.use_graph), checks=model.flag_checks)
    dist.destroy_process_group()


def test_out_of_memory_on_one_rank_keeps_the_collective_schedule(tmp_path):
    """Rank 1 runs out of memory while allocating the shape it is about to capture; rank 0 captures and replays.  Both must issue the
    same collectives every step (a hang or a mismatched reduction otherwise) and end with identical parameters."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_oom_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "oom0.npz"), np.load(tmp_path / "oom1.npz")
    assert int(r0["graph"]) == 1 and int(r1["graph"]) == 0            # rank 1 dropped to eager launches, rank 0 replays its graph
    assert int(r0["ncoll"]) == int(r1["ncoll"])
    assert int(r1["passes"]) == 4 and int(r0["passes"]) == 4
    assert np.array_equal(r0["param"], r1["param"])
    # 4 steps, gradient of step s = sum over ranks of 10 * (rank + s + 1)
    want = -0.1 * sum(10.0 * ((0 + s + 1) + (1 + s + 1)) for s in range(4))
    assert np.allclose(r0["param"], want)


def test_graph_cache_admission_does_not_thrash(monkeypatch):
    """More recurring batch shapes than graph slots (bucketed training on ragged data): least-recently-used replacement would recapture
    on almost every step.  The trainer admits a new shape to a FULL cache only once it has been seen clearly more often than the least
    often seen captured shape; everything else launches eagerly.  Round-robin over 5 shapes with 2 slots: the captures stop."""
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(DataParallelTrainer, "_drain", staticmethod(lambda: None))
    monkeypatch.setattr(DataParallelTrainer, "_copy_into", staticmethod(lambda dst, src: dst.copy_(src)))
    model = _TinyModel(oom_on_prepare=False)
    trainer = DataParallelTrainer(model, None, use_graph=True, check_every_step=False, graph_after=2, max_graphs=2)
    captures = [0]

    def cap(fn):
        captures[0] += 1
        return _FakeGraph(fn)
    trainer._capture = cap
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt)
    mk = lambda T: Batch(z(2, T, 4), z(2, dt=torch.int32), None, None, None, torch.ones(2, 5, dtype=torch.int32), torch.full((2,), 5, dtype=torch.int32))
    for step in range(100):
        trainer.train_step(mk(3 + step % 5))
    assert len(trainer._graphs) == 2
    assert captures[0] == 4, captures[0]                # two shapes x (forward+backward graph, update graph): captured once, never again
    assert model.passes == 100                           # every step ran exactly one pass (a capture step's pass is its eager one)
    # a shape that becomes dominant is admitted: it displaces the least often seen captured shape
    for step in range(40):
        trainer.train_step(mk(9))
    assert trainer._key(mk(9)) in trainer._graphs and captures[0] == 6
