"""Data-parallel training by utterance + hipGraph replay of the train step.

The reference is single-device (`num_gpus` is deprecated and ignored, avsr/avsr.py:67,:127).  This is
the MI355X-native addition SURVEY.md 8(e) describes: one process per GPU, each rank runs the whole
hot path on its shard of the utterances, and the only collectives are
  (1) a scalar all-reduce of sum(mask) so every rank normalises the sequence loss by the GLOBAL token
      count (seq2seq.py:165-171), before the backward pass;
  (2) ONE all-reduce (sum) of the flat fp32 gradient buffer (RCCL over xGMI) between BPTT and the
      clip/Adam update, so global-norm clipping and Adam see identical gradients on every rank.
  (3) sync batch-norm of the encoder inputs (encoder.py:44-50 computes the statistics over the whole batch):
      two small all-reduces before the forward pass -- per-feature sums with the row counts, then the centred
      squares -- so mean / variance / moving averages are those of the GLOBAL batch on every rank.  The BN
      gamma / beta gradients ride in (2).  `sync_bn=False` keeps per-rank statistics; the batch-norms inside the
      lip-crop CNN (and the input BN of a CNN-fed stream) are always per rank.
The AU regression term is a masked mean over the GLOBAL batch too: its frame count rides with sum(mask) in one 4-float all-reduce.

Launch overhead: a train step is ~1.3k dependent kernel launches; they are captured once per batch
shape into a hipGraph (torch.cuda.CUDAGraph is only the capture/replay plumbing -- every node is one
of our kernels or a memset/memcpy) and replayed.
"""
from collections import OrderedDict

import torch

_FIELDS = ("audio", "audio_len", "video", "video_len", "aus", "labels", "labels_len")


class DataParallelTrainer:
    MAX_GRAPHS = 24          # captured shapes kept (bucketed training visits many): least recently used is dropped with its buffers

    def __init__(self, model, dist=None, use_graph=True, sync_bn=True, force_collectives=False, check_every_step=True, graph_after=1,
                 max_graphs=None, sync_cnn_bn=None):
        self.model, self.dist = model, dist
        if max_graphs is not None:               # bucketed training on large batches: every captured shape pins a workspace and a graph pool
            self.MAX_GRAPHS = max(1, int(max_graphs))
        self.world = dist.get_world_size() if dist is not None else 1
        # force_collectives: issue every collective even at world size 1 (exercises the RCCL path on a single-GPU box)
        self.collective = self.world > 1 or bool(force_collectives and dist is not None)
        if hasattr(model, "dp_world"):
            model.dp_world = self.world           # the moving statistics travel (averaged) in the gradient all-reduce's tail
        self.use_graph = use_graph
        self.mode = "eager"
        self._want_graph = use_graph
        self._graphs = OrderedDict()            # key -> (forward+backward graph, update graph, pinned workspace key)
        # graph_after: a batch shape is captured at its graph_after-th sighting and launched eagerly before (bucketed training:
        # most (B, T_a, T_v, L) shapes of an epoch never come back, and capturing one costs a second pass of python launches plus a
        # pinned workspace; the shapes that do repeat -- full buckets at the maximum lengths -- are replayed from then on)
        self.graph_after = max(1, int(graph_after))
        self._seen = OrderedDict()
        self._checked = False
        # read the persistent kernels' sticky "wait expired" flag before EVERY update (one small host sync per step) and redo the
        # pass through the per-step launches if it is set.  Default ON: a flagged pass holds invalid activations, and training on
        # them silently is worse than one host round trip per step; bench.py switches it off (first step only) for the timed loop.
        self.check_every_step = bool(check_every_step)
        import os
        self.drain_around_collectives = os.environ.get("AVSR_DP_DRAIN", "1") != "0"
        self.drain_after_collectives = os.environ.get("AVSR_DP_DRAIN", "1") == "2"
        if self.collective and self.use_graph and os.environ.get("AVSR_DP_GRAPH") == "0":
            # Escape hatch.  Round 1 replayed graphs that still held memset / memcpy nodes around the collectives and got inf / NaN
            # gradients.  With every device fill / copy a kernel, two engine ranks replaying their graphs
            # around the collectives are bit-identical to the same two ranks launching eagerly over 12, 50 and 200 steps at the
            # benchmark size (tools/dp_full_check.py, AVSR_DP_GRAPH=1 vs unset), so graphs are the default at every world size;
            # the stream is still drained around each collective (AVSR_DP_DRAIN).
            self.use_graph = False
            self.mode = "eager (AVSR_DP_GRAPH=0)"
        # Gradient all-reduce in two buckets (AVSR_DP_OVERLAP=1): the decoder's block of the flat buffer is final after the first half
        # of the backward pass, so its all-reduce runs on a side stream under the encoder BPTT / lip-CNN backward; the rest follows
        # at the end.  Opt-in: the side-stream collective then shares the GPU with the persistent encoder kernels, which cannot be
        # exercised on this one-GPU box; the two-rank tests run it over gloo.
        self._bucket = None
        if self.collective and os.environ.get("AVSR_DP_OVERLAP") == "1" and hasattr(model, "decoder_grad_bucket"):
            self._bucket = model.decoder_grad_bucket()
        self._side = torch.cuda.Stream() if (self._bucket and torch.cuda.is_available()) else None
        self._pending = None
        self._static = {}
        if self.world > 1 and hasattr(model, "seed_offset"):
            model.seed_offset = dist.get_rank() << 24      # decorrelate the ranks' dropout / sampling masks
        model.au_scale = 1.0 / self.world          # stand-in models without dp_norm: the AU term is averaged over ranks
        self.sync_bn = bool(sync_bn and self.collective and getattr(model, "bn_sync_enable", None) and model.bn_sync_enable())
        # Opt-in (sync_cnn_bn=True or AVSR_DP_SYNC_CNN_BN=1): the batch norms INSIDE the lip CNN and the input batch norm of the CNN-fed
        # stream normalise with the statistics of the GLOBAL batch (video.py:4-14 on one device sees the whole batch): two ranks then
        # reproduce one engine on the whole batch from lip crops.  Their statistics are sequentially dependent (layer k's moments are
        # taken over layer k-1's normalised output), so each of the 8 batch norms needs its own small all-reduce in the forward pass and
        # one in the backward pass -- 16 collectives INSIDE the step, which therefore launches eagerly (no captured graph).  Default
        # off: per-rank statistics, replicas bit-identical, the deviation stated in the bench JSON.
        if sync_cnn_bn is None:
            sync_cnn_bn = os.environ.get("AVSR_DP_SYNC_CNN_BN") == "1"
        self.sync_cnn_bn = bool(sync_cnn_bn and self.collective and getattr(model, "use_cnn", False))
        if self.sync_cnn_bn:
            model.cnn_bn_sync = lambda t: dist.all_reduce(t)
            self.use_graph = False
            self.mode = "eager (sync_cnn_bn: 16 small collectives inside the step)"

    # -- helpers ----------------------------------------------------------------------------------
    @staticmethod
    def _key(batch):
        """Shapes of every field (None / not-None included: a batch with `aus` and one without are different graphs)."""
        return tuple((None if getattr(batch, n, None) is None else tuple(getattr(batch, n).shape)) for n in _FIELDS)

    def _stage(self, key, batch):
        """The static buffers the captured graph reads: trainer-OWNED clones of the first batch of a shape (the caller's tensors are
        never adopted -- a caller that keeps an epoch resident on the GPU must not find its first batch overwritten).  Every batch
        is copied in by a kernel after draining the stream (see _drain); feeding the static batch object itself back (what a
        benchmark loop does with the object returned by `static_batch`) costs no copy and puts nothing between the graph launches."""
        st = self._static.get(key)
        if st is None:
            st = type(batch)(**{n: (None if getattr(batch, n, None) is None else getattr(batch, n).clone()) for n in _FIELDS})
            self._static[key] = st
            return st
        todo = [(getattr(st, n), getattr(batch, n)) for n in _FIELDS
                if getattr(batch, n) is not None and getattr(batch, n).data_ptr() != getattr(st, n).data_ptr()]
        if todo:
            self._drain()
            for dst, src in todo:
                self._copy_into(dst, src)
        return st

    def static_batch(self, batch):
        """The trainer-owned buffers a batch of this shape is staged into (created from `batch` if the shape is new).  Training on the
        returned object skips the staging copy."""
        return self._stage(self._key(batch), batch)

    def _evict_graphs(self):
        while len(self._graphs) > self.MAX_GRAPHS:
            key, (_ga, _gb, wskey) = self._graphs.popitem(last=False)
            self._static.pop(key, None)
            unpin = getattr(self.model, "unpin_workspace", None)
            if unpin and wskey is not None:
                unpin(wskey)

    @staticmethod
    def _drain():
        """Host-wait for everything queued on the stream.  Measured on ROCm 7.0 / MI355X (tools/graph_queue_probe.py, tools/dp_full_check.py): with
        several steps queued, a LARGE eagerly launched kernel or D2D copy between two launches of the captured graphs let the
        next launch start before the previous one had finished -- overlapping train steps, wrong results, persistent-kernel
        waits expiring and GPU memory faults (a tiny kernel in between, or graph launches alone at any depth, were fine).  So
        eager work that has to sit between graph launches (batch staging, the RCCL all-reduces) is only issued on a drained
        stream; the cost is one host round trip per step in those modes."""
        torch.cuda.current_stream().synchronize()

    @staticmethod
    def _copy_into(dst, src):
        """dst = src through an engine KERNEL on the current stream, not a device-to-device memcpy.  Measured on ROCm 7.0 /
        MI355X: a large D2D memcpy (DMA-engine path; the 75 MB lip-crop batch) enqueued between two launches of a captured
        graph let the second launch start before the first had finished once several steps were queued -- overlapping
        steps, wrong results, GPU memory faults.  A kernel stays on the compute queue and keeps the stream order."""
        from . import ops
        assert src.shape == dst.shape and src.dtype == dst.dtype and src.is_contiguous() and dst.is_contiguous()
        if src.element_size() == 4:
            ops.copy_(dst, src)                                          # word copy: bit-exact for float32 and int32 alike
        else:
            dst.copy_(src)

    def _fwd_bwd(self, batch):
        self.model.forward_train(batch, compute_denom=not self.collective)
        self.model.backward()

    # collective mode with the overlapped bucket: the backward pass in two halves with the first all-reduce started in between
    def _phase1(self, batch):
        self.model.forward_train(batch, compute_denom=False)
        self.model.backward_decoder()

    def _phase2(self):
        self.model.backward_encoders()

    def _start_bucket(self):
        lo, hi = self._bucket
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            self._pending = self.dist.all_reduce(self.model.grads[lo:hi], async_op=True)

    def _reduce_grads(self):
        """Sum the flat gradient buffer over the ranks: everything, or what the bucket started earlier does not cover.  The batch
        loss rides in the buffer's tail (model.grads_and_loss): one collective instead of two."""
        m = self.model
        full = getattr(m, "grads_and_loss", None)
        self._loss_reduced = full is not None
        if self._bucket is None:
            self.dist.all_reduce(full if full is not None else m.grads)
            return
        lo, hi = self._bucket
        if lo > 0:
            self.dist.all_reduce(m.grads[:lo])
        tail = full if full is not None else m.grads
        if hi < tail.numel():
            self.dist.all_reduce(tail[hi:])
        if self._pending is not None:
            self._pending.wait()                       # the current stream waits for the side-stream collective
            self._pending = None

    def _fwd_bwd_collective(self, batch):
        if self._bucket is None:
            self._fwd_bwd(batch)
            return
        self._phase1(batch)
        self._start_bucket()
        self._phase2()

    def _persistent_failed(self):
        """Did a persistent kernel flag the pass -- on ANY rank?  The flag is a per-rank observation but the redo issues collectives
        (the overlapped bucket's all-reduce), so the decision must be the same everywhere: the local flags are MAX-reduced first
        (one 4-byte collective, only on steps that check) and a rank that was fine redoes its pass together with the one that was not;
        ranks deciding alone would issue different collective sequences -- a hang or mismatched reductions."""
        chk = getattr(self.model, "check_persistent", None)
        if not chk:
            return False
        if not (self.collective and self.world > 1):
            return bool(chk())
        probe = getattr(self.model, "persistent_flagged", None)
        local = bool(probe()) if probe else bool(chk(disable=False))
        flag = torch.tensor([1.0 if local else 0.0], device=self.model.grads.device if hasattr(self.model, "grads") else "cpu")
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
        if float(flag.item()) == 0.0:
            return False
        chk(force=True)                                   # every rank switches to the per-step launches and clears its flag
        return True

    def _capture(self, fn):
        # thread_local: only THIS thread's calls are checked against the capture -- the input pipeline's prefetch thread may allocate a
        # page-locked buffer (hipHostMalloc) at any time, which the default global mode would reject and thereby invalidate the capture
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            fn()
        return g

    # -- one training step ---------------------------------------------------------------------------
    def train_step(self, batch):
        m, dist = self.model, self.dist
        key = self._key(batch)
        if self.collective and self.use_graph and self.drain_around_collectives:
            self._drain()
        fused_sync = self.collective and hasattr(m, "dp_sync_pack") and (self.sync_bn or getattr(m, "bn_sync", None) is None)
        if fused_sync:
            # ONE small collective: both loss normalisers and the fp64 moments of every synchronised input batch norm
            buf = m.dp_sync_pack(batch)
            dist.all_reduce(buf)
            m.dp_sync_unpack()
        elif self.collective:
            # global loss normaliser: sum over ALL ranks of min(labels_len, L)
            local = getattr(m, "local_loss_denominator", None)
            if local is not None:
                self._copy_into(m.denom, local(batch))
            else:
                L = batch.labels.shape[1]
                m.denom.copy_(batch.labels_len.clamp(0, L).sum().to(torch.float32).reshape(1))   # CPU stand-in models (tests)
            if getattr(m, "dp_norm", None) is not None:      # one collective for both normalisers: sum(mask) and the AU frame count
                self._copy_into(m.au_total, m.local_au_count(batch))
                m.au_scale, m.au_external = 1.0, True
                dist.all_reduce(m.dp_norm)
            else:
                dist.all_reduce(m.denom)
            if self.sync_bn:
                dist.all_reduce(m.bn_sync_sums(batch))
                dist.all_reduce(m.bn_sync_squares(batch))
        # whether this step reads the persistent kernels' flag must not depend on the path a rank takes (the check is a collective at
        # world size > 1): decided from rank-independent state only
        do_check = self.check_every_step or not self._checked
        self._checked = True
        eager_now = not self.use_graph
        n = 0
        if self.use_graph:
            # sightings per shape (halved every 1024 steps so that a new phase of a curriculum can displace the old one's shapes);
            # no bookkeeping once the trainer launches eagerly for good (sync_cnn_bn, AVSR_DP_GRAPH=0, a failed capture)
            n = self._seen.pop(key, 0) + 1
            self._seen[key] = n
            while len(self._seen) > 4096:
                self._seen.popitem(last=False)
            self._steps_seen = getattr(self, "_steps_seen", 0) + 1
            if self._steps_seen % 1024 == 0:
                for k in list(self._seen):
                    self._seen[k] = (self._seen[k] + 1) // 2
        if self.use_graph and key not in self._graphs:
            eager_now = n < self.graph_after
            if not eager_now and len(self._graphs) >= self.MAX_GRAPHS:
                # Cache full.  Capturing costs two extra passes of host launches, an instantiation and a pinned workspace: with more
                # recurring shapes than slots, least-recently-used replacement recaptures for ever (measured: AVSR.train on utterances of
                # 450-500 frames ran at 0.6-0.8x its eager rate).  A new shape displaces the LEAST OFTEN seen captured shape, and only
                # once it has been seen clearly more often (graph_after = 1, capture at first sight, included: a shape seen once does
                # not displace one that has been replayed); otherwise it keeps launching eagerly.
                victim = min(self._graphs, key=lambda k: self._seen.get(k, 0))
                if n <= self._seen.get(victim, 0) + self.graph_after:
                    eager_now = True
                else:
                    self._drop_graph(victim)
        if eager_now:
            self._eager_pass(batch, do_check)
            m.apply_update()
            return m.loss, m.gnorm
        if self.collective and self.use_graph and self.drain_after_collectives:
            torch.cuda.synchronize()
        st = self._stage(key, batch)
        gr = self._graphs.get(key)
        if gr is None:
            # one eager step allocates every workspace, then capture.  The eager step issues EXACTLY the collectives of every other
            # path (bucket, remainder, loss): ranks may reach the capture step of a shape at different times (their shard sizes, hence
            # their shape keys and sighting counts, can differ), and a rank capturing must pair up with one that replays or runs eagerly.
            # A new shape's buffers are allocated BEFORE the pass (model.prepare_workspace): an out-of-memory there happens ahead of every
            # collective of the step, so the rank can drop its graphs and run the ordinary eager pass -- the same collective sequence,
            # the same do_check -- while the other ranks capture or replay.  Allocations later in the pass are only retried on a single
            # rank: under collectives the aborted pass may already have started the bucket all-reduce, and a retry would pair a second
            # set of reductions with the other ranks' one.
            # (Under collectives an out-of-memory anywhere else in the step -- the lazily allocated synchronisation buffers in front of
            # the step's first all-reduce, an allocation later in the pass -- is FATAL for the job: the failing rank raises while the
            # others block in a collective; the launcher's timeout ends them.  A rank cannot leave a collective schedule on its own.)
            prep = getattr(m, "prepare_workspace", None)
            try:
                if prep is not None:
                    prep(st)
            except torch.cuda.OutOfMemoryError:
                self._oom_to_eager()
                self._eager_pass(batch, do_check)
                m.apply_update()
                return m.loss, m.gnorm
            if self.collective:
                self._eager_pass(st, do_check)
            else:
                try:
                    self._eager_pass(st, do_check)
                except torch.cuda.OutOfMemoryError:
                    self._oom_to_eager()
                    with self._redoing():      # the aborted pass may have reached the batch norms: the retry must not move their averages again
                        self._eager_pass(batch, do_check)
                    m.apply_update()
                    return m.loss, m.gnorm
            m.apply_update()
            torch.cuda.synchronize()
            try:
                if self._bucket is not None:           # two graphs: the bucket's all-reduce starts between them
                    ga = (self._capture(lambda: self._phase1(st)), self._capture(self._phase2))
                else:
                    ga = self._capture(lambda: self._fwd_bwd(st))
                gb = self._capture(m.apply_update)
                # the graphs hold raw pointers into this shape's workspace: pin it against the model's LRU eviction
                pin = getattr(m, "pin_workspace", None)
                wskey = pin(m._cur[0]) if (pin and getattr(m, "_cur", None)) else None
                self._graphs[key] = gr = (ga, gb, wskey)
                self._evict_graphs()
                self.mode = "hipgraph"
            except Exception as e:  # capture unsupported: stay eager (still the HIP engine, just host-launched)
                self.use_graph = False
                self.mode = "eager (graph capture failed: %s)" % type(e).__name__
            return m.loss, m.gnorm
        ga, gb, _ = gr
        self._graphs.move_to_end(key)
        if isinstance(ga, tuple):
            ga[0].replay()
            if self.drain_around_collectives:
                self._drain()
            self._start_bucket()
            ga[1].replay()
        else:
            ga.replay()
        if do_check and self._persistent_failed():
            # a persistent kernel's bounded wait expired inside the replay: the activations are invalid.  check_persistent() has
            # switched the one-launch paths off; drop the captured graphs (they contain those launches) and redo the step eagerly.
            self._drop_graphs()
            self.use_graph = False
            self.mode = "eager (persistent kernel flagged a pass)"
            if self._pending is not None:
                self._pending.wait()
                self._pending = None
            with self._redoing():
                self._fwd_bwd_collective(st) if self.collective else self._fwd_bwd(st)
        if self.collective:
            if self.drain_around_collectives:          # the gradient all-reduce is a large eager kernel between two graph launches
                self._drain()
            self._reduce_grads()
            self._reduce_loss()
            if self.drain_after_collectives:
                torch.cuda.synchronize()
        if self.use_graph:
            gb.replay()
        else:
            m.apply_update()
        return m.loss, m.gnorm

    def _eager_pass(self, batch, do_check):
        """Forward + backward + gradient / loss reduction by host launches, with the collective schedule every path shares:
        [bucket all-reduce on the side stream], [flag MAX-reduce iff do_check], remainder of the gradients (+ loss)."""
        if self.collective:
            self._fwd_bwd_collective(batch)
        else:
            self._fwd_bwd(batch)
        if do_check and self._persistent_failed():       # a persistent kernel was not co-resident: redo through the launch path
            if self._pending is not None:                # the bucket reduced an invalid pass: finish it, then redo everything
                self._pending.wait()
                self._pending = None
            with self._redoing():
                self._fwd_bwd_collective(batch) if self.collective else self._fwd_bwd(batch)
        if self.collective:
            self._reduce_grads()
            self._reduce_loss()

    def _redoing(self):
        """The pass being repeated has already moved the batch-norm averages once (model.redoing): the repeat must not move them again."""
        import contextlib
        ctx = getattr(self.model, "redoing", None)
        return ctx() if ctx else contextlib.nullcontext()

    def _oom_to_eager(self):
        """Too many captured shapes alive (each pins a workspace and a private graph pool): drop them all and stay eager."""
        self._drop_graphs()
        self._static.clear()
        self.use_graph = False
        self.mode = "eager (out of memory while allocating a shape's workspace; captured graphs dropped)"
        torch.cuda.empty_cache()

    def _drop_graph(self, key):
        _ga, _gb, wskey = self._graphs.pop(key)
        self._static.pop(key, None)
        unpin = getattr(self.model, "unpin_workspace", None)
        if unpin and wskey is not None:
            unpin(wskey)

    def _drop_graphs(self):
        unpin = getattr(self.model, "unpin_workspace", None)
        for _k, (_ga, _gb, wskey) in self._graphs.items():
            if unpin and wskey is not None:
                unpin(wskey)
        self._graphs.clear()

    def _reduce_loss(self):
        """The forward pass leaves this rank's share of the batch loss (its cross-entropy sum over the GLOBAL token count, its share
        of the AU term): summed over the ranks it is the loss the reference reports; L2 joins once, in apply_update, on every rank."""
        if getattr(self, "_loss_reduced", False):
            self._loss_reduced = False                      # summed with the gradients (grads_and_loss)
            return
        if getattr(self.model, "loss", None) is not None and torch.is_tensor(self.model.loss):
            self.dist.all_reduce(self.model.loss)
