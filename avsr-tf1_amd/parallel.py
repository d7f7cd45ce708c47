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
import torch


class DataParallelTrainer:
    def __init__(self, model, dist=None, use_graph=True, sync_bn=True, force_collectives=False, check_every_step=False):
        self.model, self.dist = model, dist
        self.world = dist.get_world_size() if dist is not None else 1
        # force_collectives: issue every collective even at world size 1 (exercises the RCCL path on a single-GPU box)
        self.collective = self.world > 1 or bool(force_collectives and dist is not None)
        self.use_graph = use_graph
        self.mode = "eager"
        self._want_graph = use_graph
        self._graphs = {}
        self._checked = False
        # eager mode only: read the persistent kernels' sticky "wait expired" flag before EVERY update (one small host sync per
        # step) and redo the pass through the per-step launches if it is set; off by default (first step only) for benchmarking
        self.check_every_step = bool(check_every_step)
        import os
        self.drain_around_collectives = os.environ.get("AVSR_DP_DRAIN", "1") != "0"
        self.drain_after_collectives = os.environ.get("AVSR_DP_DRAIN", "1") == "2"
        if self.collective and self.use_graph and os.environ.get("AVSR_DP_GRAPH") != "1":
            # Measured with two real engine ranks at the benchmark size (tools/graph_queue_probe.py, tools/dp_full_check.py): replaying the captured
            # graphs around collectives gave inf / NaN gradients within 24 steps however the stream was drained, while eager
            # launches are exact (and cost 1-2 % on this GPU-bound step).  Collective mode therefore launches eagerly.
            self.use_graph = False
            self.mode = "eager (captured graphs are not replayed around collectives)"
        self._static = {}
        if self.world > 1 and hasattr(model, "seed_offset"):
            model.seed_offset = dist.get_rank() << 24      # decorrelate the ranks' dropout / sampling masks
        model.au_scale = 1.0 / self.world          # stand-in models without dp_norm: the AU term is averaged over ranks
        self.sync_bn = bool(sync_bn and self.collective and getattr(model, "bn_sync_enable", None) and model.bn_sync_enable())

    # -- helpers ----------------------------------------------------------------------------------
    @staticmethod
    def _key(batch):
        return tuple((None if t is None else tuple(t.shape)) for t in
                     (batch.audio, batch.video, batch.labels))

    def _stage(self, key, batch):
        """The static buffers the captured graph reads.  The first batch of a shape is adopted as is (its tensors become the
        static buffers: a caller that keeps feeding the same tensors pays no copy and, more importantly, puts nothing between
        the graph launches); any other batch is copied in by a kernel, after draining the stream (see _drain)."""
        st = self._static.get(key)
        if st is None:
            self._static[key] = st = batch
            return st
        todo = [(getattr(st, name), getattr(batch, name)) for name in ("audio", "audio_len", "video", "video_len", "aus", "labels", "labels_len")
                if getattr(batch, name) is not None and getattr(batch, name).data_ptr() != getattr(st, name).data_ptr()]
        if todo:
            self._drain()
            for dst, src in todo:
                self._copy_into(dst, src)
        return st

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

    def _persistent_failed(self):
        chk = getattr(self.model, "check_persistent", None)
        return bool(chk and chk())

    def _capture(self, fn):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g

    # -- one training step ---------------------------------------------------------------------------
    def train_step(self, batch):
        m, dist = self.model, self.dist
        key = self._key(batch)
        if self.collective and self.use_graph and self.drain_around_collectives:
            self._drain()
        if self.collective:
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
        if not self.use_graph:
            self._fwd_bwd(batch)
            if not self._checked or self.check_every_step:   # make sure the persistent kernels were co-resident
                self._checked = True
                if self._persistent_failed():
                    self._fwd_bwd(batch)
            if self.collective:
                dist.all_reduce(m.grads)
            m.apply_update()
            return m.loss, m.gnorm
        if self.collective and self.use_graph and self.drain_after_collectives:
            torch.cuda.synchronize()
        st = self._stage(key, batch)
        gr = self._graphs.get(key)
        if gr is None:
            # one eager step allocates every workspace; then capture
            self._fwd_bwd(st)
            if self._persistent_failed():              # persistent kernels not co-resident: redo through the launch path
                self._fwd_bwd(st)
            if self.collective:
                dist.all_reduce(m.grads)
            m.apply_update()
            torch.cuda.synchronize()
            try:
                ga = self._capture(lambda: self._fwd_bwd(st))
                gb = self._capture(m.apply_update)
                self._graphs[key] = gr = (ga, gb)
                self.mode = "hipgraph"
            except Exception as e:  # capture unsupported: stay eager (still the HIP engine, just host-launched)
                self.use_graph = False
                self.mode = "eager (graph capture failed: %s)" % type(e).__name__
            return m.loss, m.gnorm
        ga, gb = gr
        ga.replay()
        if self.collective:
            if self.drain_around_collectives:          # the 13 MB gradient all-reduce is a large eager kernel between two graph launches
                self._drain()
            dist.all_reduce(m.grads)
            if self.drain_after_collectives:
                torch.cuda.synchronize()
        gb.replay()
        return m.loss, m.gnorm
