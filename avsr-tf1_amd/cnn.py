"""Lip-crop CNN front-end: `video.resnet_cnn` (avsr/video.py:143-195, wired at avsr/avsr.py:684-696) on the HIP engine.

[N, H, W, C] frames (N = B*T, zero padding frames included, exactly as `cnn_layers` reshapes them, video.py:224-233)
-> conv3x3(C->f0)+bias -> BN-ReLU -> residual block (identity shortcut, no leading BN) -> one strided residual block
per further filter count (BN-ReLU, 1x1/2 projection shortcut on the un-normalised input, conv3x3/2, BN-ReLU, conv3x3,
add) -> VALID conv over the remaining map -> cnn_dense_units, ReLU.  BN: epsilon 1e-5, momentum 0.98 (video.py:8-11);
every conv kernel carries l2(0.001) (video.py:26, summed into the loss at seq2seq.py:180-184).

Convolutions run on the frame-resident MFMA kernels of csrc/conv_mfma.hip (avsr_conv_fwd / _bwd_data / _bwd_weight) with three
fusions around them: the producing convolution's epilogue adds the residual and emits the batch-norm statistics of what it wrote;
the batch norm itself is never materialised in training -- avsr_bn_finalize turns the statistics into per-channel scale / shift and
the CONSUMING convolution (forward and weight gradient) applies max(x*scale + shift, 0) while it stages frames in LDS; the weight
gradient kernel also produces the bias gradient.  Shapes those kernels do not cover fall back to the direct VALU kernels
(avsr_conv3x3*) or to avsr_im2col + avsr_gemm (the TF kernel [kh,kw,cin,cout] IS the [kh*kw*cin, cout] operand; data gradient =
transposed GEMM + avsr_col2im, weight gradient = split-K TN GEMM), with BN through avsr_batchnorm_fwd_ex / avsr_batchnorm_bwd.
This file only owns buffers and the op order; all arithmetic is in csrc/conv_mfma.hip, conv_direct.hip, conv.hip, gemm.hip,
elementwise.hip."""
import os

import torch

from . import ops


def same_pad(n, k, s):
    """TF 'SAME': out = ceil(n / s); total = max((out-1)*s + k - n, 0); the odd pixel goes after (bottom / right)."""
    out = (n + s - 1) // s
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2


def layout(hw, filters, dense):
    """Op list of resnet_cnn.  ('conv', name, src, dst, k, stride, cin, cout) | ('bnrelu', name, src, dst, c) |
    ('add', name, a, b, dst) | ('flatten', name, src, dst, kh, kw, cin, cout);  shapes[name] = (H, W, C) of every map."""
    H, W, C = hw
    f = list(filters)
    shapes = {"in": (H, W, C)}
    ops_ = []

    def conv(name, src, dst, k, s, cout):
        h, w, cin = shapes[src]
        ho, _ = same_pad(h, k, s)
        wo, _ = same_pad(w, k, s)
        shapes[dst] = (ho, wo, cout)
        ops_.append(("conv", name, src, dst, k, s, cin, cout))

    def bnrelu(name, src, dst):
        shapes[dst] = shapes[src]
        ops_.append(("bnrelu", name, src, dst, shapes[src][2]))

    def add(name, a, b, dst):
        shapes[dst] = shapes[a]
        ops_.append(("add", name, a, b, dst))

    conv("layer0", "in", "a0", 3, 1, f[0])
    bnrelu("layer0_bn", "a0", "b0")
    conv("res_block_0_conv1", "b0", "r0a", 3, 1, f[0])
    bnrelu("res_block_0_second_bn", "r0a", "r0b")
    conv("res_block_0_conv2", "r0b", "r0c", 3, 1, f[0])
    add("res_block_0", "r0c", "b0", "x0")
    prev = "x0"
    for i, c in enumerate(f[1:], start=1):
        n = "res_block_%d" % i
        bnrelu(n + "_first_bn", prev, n + "_p")
        conv(n + "_shortcut", prev, n + "_s", 1, 2, c)
        conv(n + "_conv1", n + "_p", n + "_a", 3, 2, c)
        bnrelu(n + "_second_bn", n + "_a", n + "_b")
        conv(n + "_conv2", n + "_b", n + "_c", 3, 1, c)
        add(n, n + "_c", n + "_s", "x%d" % i)
        prev = "x%d" % i
    h, w, cin = shapes[prev]
    shapes["out"] = (1, 1, dense)
    ops_.append(("flatten", "flatten", prev, "out", h, w, cin, dense))
    return ops_, shapes


def param_shapes(hw, filters, dense):
    """[(name, tf_shape, role)] in graph order; role: conv_kernel | bias | gamma | beta | moving_mean | moving_variance."""
    out = []
    for op in layout(hw, filters, dense)[0]:
        if op[0] == "conv":
            _, name, _, _, k, _, cin, cout = op
            out += [(name + "/kernel", (k, k, cin, cout), "conv_kernel"), (name + "/bias", (cout,), "bias")]
        elif op[0] == "flatten":
            _, name, _, _, kh, kw, cin, cout = op
            out += [(name + "/kernel", (kh, kw, cin, cout), "conv_kernel"), (name + "/bias", (cout,), "bias")]
        elif op[0] == "bnrelu":
            c = op[4]
            out += [(op[1] + "/gamma", (c,), "gamma"), (op[1] + "/beta", (c,), "beta"),
                    (op[1] + "/moving_mean", (c,), "moving_mean"), (op[1] + "/moving_variance", (c,), "moving_variance")]
    return out


class LipCNN:
    BN_EPS, BN_MOMENTUM, L2 = 1e-5, 0.98, 1e-3

    def __init__(self, model, N, prefix="video/cnn/"):
        cfg = model.cfg
        self.m, self.N, self.pre = model, N, prefix
        self.ops, self.shapes = layout(cfg.video_hw, cfg.cnn_filters, cfg.cnn_dense_units)
        dev = model.dev
        z = lambda *s: torch.zeros(*s, device=dev)
        self.maps, self.gmaps, self.col, self.bn, self.direct, self.mfma = {}, {}, {}, {}, set(), {}
        self._sync_bufs = {}
        max_col = 4
        for op in self.ops:
            kind = op[0]
            if kind == "conv":
                _, name, src, dst, k, s, cin, cout = op
                ho, wo, _ = self.shapes[dst]
                h, w, _ = self.shapes[src]
                geo = (N, h, w, cin, cout, k, s, same_pad(h, k, s)[1], same_pad(w, k, s)[1], ho, wo)
                if ops.conv_supported(ops.conv_desc(*geo)):          # frame-resident MFMA kernels
                    self.mfma[name] = geo
                elif k == 3 and ops.conv3x3_supported(cin, cout, h, w):   # shallow wide layers: direct kernels, no im2col operand
                    self.direct.add(name)
                else:
                    self.col[name] = z(N * ho * wo, k * k * cin)
                    max_col = max(max_col, self.col[name].numel())
            elif kind == "bnrelu":
                c = op[4]
                self.bn[op[1]] = (z(c), z(c), z(c), z(c))          # batch mean, inverse std (training statistics); scale, shift
            if kind == "flatten":
                self.pre_act = z(N, op[7])
        # how many ops read each map: the gradient of a single-consumer input of a residual add is the add's output gradient
        # itself (aliased, no copy); maps with several consumers own a buffer their contributions accumulate into
        self.consumers = {}
        for op in self.ops:
            for src in ((op[2],) if op[0] != "add" else (op[2], op[3])):
                self.consumers[src] = self.consumers.get(src, 0) + 1
        self.alias = {}
        for op in self.ops:
            if op[0] == "add":
                for t in (op[2], op[3]):
                    if self.consumers.get(t, 0) == 1:
                        self.alias[t] = op[4]
        # fused conv epilogues (csrc/conv_mfma.hip): a conv whose output feeds a batch norm emits the statistics' partial sums, and
        # the second conv of a residual block adds the shortcut itself (the separate add pass and its operand map disappear)
        self.stat_buf = {}                                        # map name -> partial sums [512][2C] of the conv that produced it
        self.fuse_add = {}                                        # conv name -> (shortcut map, add output map) when the add is fused
        self.skip_add = set()
        bn_src = {op[2] for op in self.ops if op[0] == "bnrelu"}
        by_dst = {op[3]: op for op in self.ops if op[0] == "conv"}
        for op in self.ops:
            if op[0] == "add":
                _, name, a, b, dst = op
                prod = by_dst.get(a)
                if prod is not None and prod[1] in self.mfma and self.consumers.get(a, 0) == 1:
                    self.fuse_add[prod[1]] = (b, dst)
                    self.skip_add.add(name)
        for op in self.ops:
            if op[0] == "conv" and op[1] in self.mfma:
                out = self.fuse_add[op[1]][1] if op[1] in self.fuse_add else op[3]
                if out in bn_src:
                    self.stat_buf[out] = z(512 * 2 * op[7])
        self.stat_rows = {}
        # a batch norm whose every reader is an MFMA conv (as input) or a fused add (as the shortcut operand) is applied by those
        # readers' loaders: its output map is never written in training (self.lazy[dst] = (pre-BN map, scale, shift) per forward)
        self.lazy_ok = set()
        for op in self.ops:
            if op[0] != "bnrelu":
                continue
            dst, ok = op[3], op[4] % 4 == 0
            for o in self.ops:
                if o[0] == "conv" and o[2] == dst:
                    ok = ok and o[1] in self.mfma
                elif o[0] == "add" and dst in (o[2], o[3]):
                    ok = ok and o[1] in self.skip_add and o[3] == dst
                elif o[0] in ("bnrelu", "flatten") and o[2] == dst:
                    ok = False
            if ok:
                self.lazy_ok.add(op[1])
        self.lazy = {}
        # Batch-norm backward fused into the data gradient that produces the gradient of the BN's output (csrc/conv_mfma.hip,
        # avsr_conv_bwd_data_bn): the LAST contribution to that gradient in the reverse pass comes from the output's FIRST reader in
        # the graph; when that reader is an MFMA convolution with a single-launch data gradient its epilogue applies the ReLU mask and
        # emits sum dz / sum dz*x, and the statistics pass over two maps disappears.  An earlier contribution over a residual
        # connection (exactly one other reader, a fused add) is not copied into the gradient map first: the convolution reads it
        # where it lies (`acc`).
        self.bnb, self.bnb_conv, self.bnb_stat, self.bnb_k, self.bnb_rows, self.acc_src = {}, {}, {}, {}, {}, {}
        self.acc_ok = set()
        for op in self.ops:
            if op[0] != "bnrelu" or op[1] not in self.lazy_ok:
                continue
            name, dst, c = op[1], op[3], op[4]
            readers = [o for o in self.ops if (o[0] in ("conv", "bnrelu", "flatten") and o[2] == dst) or (o[0] == "add" and dst in (o[2], o[3]))]
            first = readers[0] if readers else None
            if first is None or first[0] != "conv" or first[1] not in self.mfma or first[2] != dst:
                continue
            if not ops.conv_bwd_data_bn_supported(ops.conv_desc(*self.mfma[first[1]])):
                continue
            self.bnb[name], self.bnb_conv[first[1]] = first[1], name
            self.bnb_stat[name], self.bnb_k[name] = z(512 * 2 * c), z(3 * c)
            if len(readers) == 2 and readers[1][0] == "add":
                self.acc_ok.add(dst)
        # Batch-norm backward folded into the weight gradient of the convolution that PRODUCED the batch norm's input (round 5): when that
        # convolution has no data gradient (its source is the crops) the gradient of its output has exactly one reader -- its weight
        # gradient -- which evaluates dx = k1*dz + k2*x + k3 while it fetches the operand (avsr_conv_bwd_weight_bn): the stand-alone
        # avsr_bn_bwd_apply pass over three 199 MB maps of layer 0 disappears.
        self.fold_wg = {}                                        # bn name -> producing conv name
        for op in self.ops:
            if op[0] != "bnrelu" or op[1] not in self.bnb:
                continue
            prod = by_dst.get(op[2])
            lvl = int(os.environ.get("AVSR_CNN_FOLD", "2"))      # 0 off, 1 layers without a data gradient only, 2 every single-reader case
            if lvl and prod is not None and prod[1] in self.mfma and self.consumers.get(op[2], 0) == 1 and \
                    (prod[2] == "in" or lvl >= 2) and ops.conv_bwd_weight_bn_supported(ops.conv_desc(*self.mfma[prod[1]])):
                # (a producer WITH a data gradient: the weight gradient -- issued first -- also writes the evaluated gradient for it)
                self.fold_wg[op[1]] = prod[1]
        self.fold_src = {}
        for name, (h, w, c) in self.shapes.items():
            if name == "in":
                continue
            self.maps[name] = z(N, h, w, c)
            if name not in self.alias:
                self.gmaps[name] = z(N, h, w, c)
        for t, dst in self.alias.items():
            self.gmaps[t] = self.gmaps[dst]
        self.dcol = torch.empty(max_col, device=dev)              # d col of the layer being differentiated (transient)
        # Weight-gradient slabs: every MFMA convolution gets its own region, so that the twelve final reductions of a backward pass can
        # run as ONE launch at its end (ops.slab_defer_begin / _end) instead of one 5 us launch behind every convolution.
        self.wg_off, off = {}, 0
        for op in self.ops:
            if op[0] == "conv" and op[1] in self.mfma:
                _, name, _src, _dst, k, _s, cin, cout = op
                self.wg_off[name] = off
                # the kernel's own rule (conv_bwd_weight_impl): slabs above 2048 floats come from at most 256 workgroups (one per CU:
                # there the partial slabs, not the staging, are the traffic), smaller ones from up to four workgroups per CU
                # (a launch covers the whole kernel or a group of its taps: whichever of the two caps needs more floats)
                slab = max(k * k * cin * cout + cout, 12 * cin * 16 + 16)
                off += max(256 * slab, 1024 * min(slab, 2048))
        # ONE buffer per model, shared by the LipCNN of every workspace shape (the region sizes do not depend on N; the slabs are live
        # only inside one backward pass, and passes -- eager or replayed graphs -- are serialised on the engine's stream): a buffer
        # per shape was ~180 MB x (8 cached + 8 pinned) shapes
        shared = getattr(model, "_cnn_wg_scratch", None)
        if shared is None or shared.numel() < max(off, 4):
            shared = model._cnn_wg_scratch = torch.empty(max(off, 4), device=dev)
        self.wg_scratch = shared

    def _sync_buf(self, name, c):
        """fp64 [sum | sum of squares | rows] + a second [2c] for this rank's own sums: what one batch norm all-reduces (sync_cnn_bn)."""
        b = self._sync_bufs.get(name)
        if b is None:
            b = self._sync_bufs[name] = torch.zeros(4 * c + 2, dtype=torch.float64, device=self.m.dev)
        return b

    # parameters live in the model's flat buffers
    def _p(self, n):
        return self.m.P[self.pre + n]

    def _g(self, n):
        return self.m.Gr[self.pre + n]

    def _pv(self, n):
        return self.m._pp(self.pre + n)

    @staticmethod
    def _copy(src, dst):
        """dst = src with an engine kernel.  (A torch copy_ of these 100+ MB maps is captured as a D2D memcpy node; back-to-back
        replays of a graph holding such nodes were observed to overlap on ROCm 7.0 -- wrong results and GPU memory faults.)"""
        ops.copy_(dst, src)

    def _src(self, name):
        """(map to read, (scale, shift) | None): a lazily normalised map is read through its pre-BN map."""
        lz = self.lazy.get(name)
        return (self.maps[lz[0]], (lz[1], lz[2])) if lz else (self.maps[name], None)

    def forward(self, frames, training):
        m, N = self.m, self.N
        H, W, C = self.shapes["in"]
        assert frames.shape == (N, H, W, C) and frames.is_contiguous() and frames.dtype == torch.float32
        self.maps["in"] = frames
        self.training = training
        self.lazy = {}
        for op in self.ops:
            kind = op[0]
            if kind == "conv":
                _, name, src, dst, k, s, cin, cout = op
                h, w, _ = self.shapes[src]
                ho, pt = same_pad(h, k, s)
                wo, pl = same_pad(w, k, s)
                kw_ = self._p(name + "/kernel")
                if name in self.mfma:
                    x, bn = self._src(src)
                    res, res_bn, out = None, None, dst
                    if name in self.fuse_add:
                        res, res_bn = self._src(self.fuse_add[name][0])
                        out = self.fuse_add[name][1]
                    stats = self.stat_buf.get(out) if training else None
                    n = ops.conv_fwd(ops.conv_desc(*self.mfma[name], bn=bn), x, kw_.t[kw_.off:], self._pv(name + "/bias"), self.maps[out],
                                     res, res_bn, stats)
                    if stats is not None:
                        self.stat_rows[out] = n
                    else:
                        self.stat_rows.pop(out, None)
                    continue
                self.stat_rows.pop(dst, None)
                if name in self.direct:
                    ops.conv3x3(self.maps[src], kw_.t[kw_.off:], self._pv(name + "/bias"), self.maps[dst], N, h, w, cin, cout, s, pt, pl, ho, wo)
                    continue
                col = self.col[name]
                ops.im2col(self.maps[src], col, N, h, w, cin, k, k, s, pt, pl, ho, wo)
                rows, K = N * ho * wo, k * k * cin
                ops.gemm(ops.mat(col, K), kw_.mat(cout), ops.mat(self.maps[dst], cout), rows, cout, K, bias=self._pv(name + "/bias"))
            elif kind == "bnrelu":
                _, name, src, dst, c = op
                h, w, _ = self.shapes[src]
                mean, invstd, scale, shift = self.bn[name]
                # seq2seq.py:241-250: the UPDATE_OPS (moving averages) only run with the train op under batch_normalisation=True
                upd = not training or m.cfg.batch_normalisation
                sync = getattr(m, "cnn_bn_sync", None) if training else None
                if sync is not None and not self.stat_rows.get(src):
                    raise NotImplementedError("sync_cnn_bn: batch norm %s takes its statistics outside the fused convolution epilogues" % name)
                if training and self.stat_rows.get(src) and sync is not None:
                    # data parallel, opt-in: statistics of the GLOBAL batch.  This rank's fp64 sums + its row count -> one small all-reduce
                    # -> the same finalisation from the global sums (moving averages included: identical on every rank)
                    buf = self._sync_buf(name, c)
                    ops.bn_partials_f64(self.stat_buf[src], self.stat_rows[src], c, buf)
                    buf[2 * c:2 * c + 1].fill_(float(N * h * w))
                    sync(buf[:2 * c + 1])
                    ops.bn_finalize_f64(buf, c, self.BN_EPS, self.BN_MOMENTUM, mean, invstd,
                                        m._sp(self.pre + name + "/moving_mean") if upd else None,
                                        m._sp(self.pre + name + "/moving_variance") if upd else None,
                                        self._pv(name + "/gamma"), self._pv(name + "/beta"), scale, shift)
                elif training and self.stat_rows.get(src):
                    # statistics came with the producing convolution's epilogue: finalise (fp64 merge); no statistic passes, and no
                    # normalisation pass either when every reader applies scale / shift in its loader
                    ops.bn_finalize(self.stat_buf[src], self.stat_rows[src], c, N * h * w, self.BN_EPS, self.BN_MOMENTUM, mean, invstd,
                                    m._sp(self.pre + name + "/moving_mean") if upd else None,
                                    m._sp(self.pre + name + "/moving_variance") if upd else None,
                                    self._pv(name + "/gamma"), self._pv(name + "/beta"), scale, shift)
                if training and self.stat_rows.get(src):
                    if name in self.lazy_ok:
                        self.lazy[dst] = (src, scale, shift)
                    else:
                        ops.batchnorm_apply(self.maps[src], self.maps[dst], N * h * w, c, self._pv(name + "/gamma"), self._pv(name + "/beta"),
                                            mean, invstd, 1)
                    continue
                if not training and name in self.lazy_ok:
                    # evaluation graph: the same loader-applied affine, from the moving statistics -- no normalised map is written
                    ops.bn_eval_affine(self._pv(name + "/gamma"), self._pv(name + "/beta"), m._sp(self.pre + name + "/moving_mean"),
                                       m._sp(self.pre + name + "/moving_variance"), self.BN_EPS, scale, shift, c)
                    self.lazy[dst] = (src, scale, shift)
                    continue
                ops.batchnorm_fwd_ex(self.maps[src], self.maps[dst], N * h * w, c, self._pv(name + "/gamma"), self._pv(name + "/beta"),
                                     m._sp(self.pre + name + "/moving_mean") if upd else None,
                                     m._sp(self.pre + name + "/moving_variance") if upd else None, mean, invstd,
                                     training, self.BN_EPS, self.BN_MOMENTUM, 1, m.scratch, bessel=1)
            elif kind == "add":
                _, name, a, b, dst = op
                if name in self.skip_add:                                  # done by the producing convolution's epilogue
                    continue
                ops.add(self.maps[a], self.maps[b], self.maps[dst], self.maps[dst].numel())
            else:
                _, name, src, dst, kh, kw, cin, cout = op
                K = kh * kw * cin
                ops.gemm(ops.mat(self.maps[src], K), self._p(name + "/kernel").mat(cout), ops.mat(self.pre_act, cout), N, cout, K,
                         bias=self._pv(name + "/bias"))
                ops.relu(self.pre_act, self.maps[dst], N * cout)
        return self.maps["out"].view(N, -1)

    def backward(self, dfeat):
        """dfeat [N, cnn_dense_units]: gradient of the loss wrt the CNN output.  Accumulates into the model's gradient buffer."""
        m, N = self.m, self.N
        written = set()
        deferred = []                                                     # (map, thunk): contributions that can only accumulate

        def target(name):
            """(gradient map of `name`, beta): first contribution overwrites, later ones accumulate."""
            beta = 1.0 if name in written else 0.0
            written.add(name)
            return self.gmaps[name], beta

        gout = self.gmaps["out"].view(N, -1)
        self._copy(dfeat, gout)
        written.add("out")
        self.bnb_rows, self.acc_src, self.fold_src = {}, {}, {}
        ops.slab_defer_begin()
        try:
            for op in reversed(self.ops):
                kind = op[0]
                if kind == "flatten":
                    _, name, src, dst, kh, kw, cin, cout = op
                    K = kh * kw * cin
                    dpre = self.pre_act                                           # overwritten in place: d(pre-activation)
                    ops.relu_bwd(self.maps[dst], self.gmaps[dst], dpre, N * cout)
                    m._gemm_tn(ops.mat(self.maps[src], K), ops.mat(dpre, cout), self._g(name + "/kernel").mat(cout), K, cout, N)
                    ops.colsum(ops.mat(dpre, cout), N, cout, m.grads, m.scratch, beta=1.0, out_offset=self._g(name + "/bias").off)
                    g, beta = target(src)
                    ops.gemm(ops.mat(dpre, cout), self._p(name + "/kernel").mat(cout), ops.mat(g, K), N, K, cout, trans_b=1, beta=beta)
                elif kind == "add":
                    _, name, a, b, dst = op
                    for t in (a, b):
                        if t in self.alias:                                        # same buffer as the add's output gradient
                            written.add(t)
                            continue
                        if t in self.acc_ok and t not in written and t in self.lazy:
                            # first of two contributions, the second being a fused data gradient: it reads this one in place
                            written.add(t)
                            self.acc_src[t] = self.gmaps[dst]
                            continue
                        g, beta = target(t)
                        if beta:
                            ops.add(g, self.gmaps[dst], g, g.numel())
                        else:
                            self._copy(self.gmaps[dst], g)
                elif kind == "bnrelu":
                    _, name, src, dst, c = op
                    h, w, _ = self.shapes[src]
                    mean, invstd = self.bn[name][:2]
                    g, beta = target(src)
                    gg, gb = self._g(name + "/gamma"), self._g(name + "/beta")
                    sync = getattr(m, "cnn_bn_sync", None)
                    if sync is not None and name not in self.bnb_rows:
                        # the output gradient was assembled by several launches (the four-class 3x3/2 data gradient of a wide layer): stage 1
                        # -- ReLU mask + partial sums -- as a pass of its own, then the same global finalisation as the fused layers
                        if c % 4 or c > 1024:
                            raise NotImplementedError("sync_cnn_bn: batch norm %s (%d channels)" % (name, c))
                        if name not in self.bnb_stat:
                            self.bnb_stat[name] = torch.zeros(512 * 2 * c, device=m.dev)
                            self.bnb_k[name] = torch.zeros(3 * c, device=m.dev)
                        lz = self.lazy.get(dst)
                        self.bnb_rows[name] = ops.bn_bwd_stage1(self.gmaps[dst], self.maps[src], self.gmaps[dst], N * h * w, c, self.bnb_stat[name],
                                                                scale=lz[1] if lz else None, shift=lz[2] if lz else None,
                                                                y=None if lz else self.maps[dst])
                    if name in self.bnb_rows and sync is not None:
                        # global means of dz and dz * xhat for the input gradient; d gamma / d beta keep this rank's share
                        buf = self._sync_buf(name, c)
                        loc = buf[2 * c + 1:4 * c + 1]
                        ops.bn_partials_f64(self.bnb_stat[name], self.bnb_rows.pop(name), c, loc)
                        buf[:2 * c].copy_(loc)
                        buf[2 * c:2 * c + 1].fill_(float(N * h * w))
                        sync(buf[:2 * c + 1])
                        ops.bn_bwd_finalize_f64(loc, buf, c, mean, invstd, self._pv(name + "/gamma"), gg.t[gg.off:gg.off + c], gb.t[gb.off:gb.off + c],
                                                self.bnb_k[name], grad_beta=0.0)
                        if name in self.fold_wg and beta == 0.0:
                            self.fold_src[src] = (self.gmaps[dst], self.maps[src], self.bnb_k[name], g)
                        else:
                            ops.bn_bwd_apply(self.gmaps[dst], self.maps[src], self.bnb_k[name], g, N * h * w, c, beta=beta)
                        continue
                    if name in self.bnb_rows:            # stage 1 ran in the producing data gradient's epilogue: gmaps[dst] holds dz
                        ops.bn_bwd_finalize(self.bnb_stat[name], self.bnb_rows.pop(name), c, N * h * w, mean, invstd, self._pv(name + "/gamma"),
                                            gg.t[gg.off:gg.off + c], gb.t[gb.off:gb.off + c], self.bnb_k[name], grad_beta=0.0)
                        if name in self.fold_wg and beta == 0.0:
                            # the first reader of this gradient -- the producing convolution's weight gradient -- evaluates it in its loader
                            # (and stores it for the data gradient, if the layer has one)
                            self.fold_src[src] = (self.gmaps[dst], self.maps[src], self.bnb_k[name], g)
                        else:
                            ops.bn_bwd_apply(self.gmaps[dst], self.maps[src], self.bnb_k[name], g, N * h * w, c, beta=beta)
                        continue
                    ops.batchnorm_bwd(self.maps[src], self.gmaps[dst], self._pv(name + "/gamma"), self._pv(name + "/beta"), mean, invstd, g,
                                      gg.t[gg.off:gg.off + c], gb.t[gb.off:gb.off + c], N * h * w, c, 1, m.scratch, dx_beta=beta)
                else:
                    _, name, src, dst, k, s, cin, cout = op
                    h, w, _ = self.shapes[src]
                    ho, pt = same_pad(h, k, s)
                    wo, pl = same_pad(w, k, s)
                    rows, K = N * ho * wo, k * k * cin
                    dy = ops.mat(self.gmaps[dst], cout)
                    gk, kw_, gb = self._g(name + "/kernel"), self._p(name + "/kernel"), self._g(name + "/bias")
                    if name in self.mfma:
                        x, bn = self._src(src)
                        d = ops.conv_desc(*self.mfma[name], bn=bn)
                        fold = self.fold_src.pop(dst, None)
                        if fold is not None:
                            ops.conv_bwd_weight_bn(d, x, fold[0], fold[1], fold[2], gk.t[gk.off:], gb.t[gb.off:], self.wg_scratch[self.wg_off[name]:],
                                                   dx_out=fold[3] if src != "in" else None)
                        else:
                            ops.conv_bwd_weight(d, x, self.gmaps[dst], gk.t[gk.off:], gb.t[gb.off:], self.wg_scratch[self.wg_off[name]:])
                        fused_bn = self.bnb_conv.get(name)
                        if src != "in" and fused_bn is not None and src in self.lazy:
                            g, beta = target(src)
                            pre, sc, sh = self.lazy[src]
                            self.bnb_rows[fused_bn] = ops.conv_bwd_data_bn(d, self.gmaps[dst], kw_.t[kw_.off:], g, beta=beta,
                                                                           acc=self.acc_src.pop(src, None), bn_x=self.maps[pre], bn=(sc, sh),
                                                                           stats=self.bnb_stat[fused_bn])
                        elif src != "in":
                            def data_grad(d=d, src=src, dst=dst, kw_=kw_):
                                g, beta = target(src)
                                ops.conv_bwd_data(d, self.gmaps[dst], kw_.t[kw_.off:], g, beta=beta)
                            if k == 1 and s == 2 and src not in written:          # reaches only the even pixels: must accumulate
                                deferred.append((src, data_grad))
                            else:
                                data_grad()
                    elif name in self.direct:
                        ops.conv3x3_bwd_weight(self.maps[src], self.gmaps[dst], gk.t[gk.off:], N, h, w, cin, cout, s, pt, pl, ho, wo, m.scratch)
                        ops.colsum(dy, rows, cout, m.grads, m.scratch, beta=1.0, out_offset=gb.off)
                        if src != "in":
                            g, beta = target(src)
                            if s == 1:       # the same kernel on dy with the kernel flipped and transposed
                                ops.conv3x3(self.gmaps[dst], kw_.t[kw_.off:], None, g, N, ho, wo, cout, cin, 1, 1, 1, h, w, flip=1, beta=beta)
                            else:
                                ops.conv3x3_bwd_data_s2(self.gmaps[dst], kw_.t[kw_.off:], g, N, h, w, cin, cout, pt, pl, ho, wo, beta=beta)
                    else:
                        m._gemm_tn(ops.mat(self.col[name], K), dy, gk.mat(cout), K, cout, rows)
                        ops.colsum(dy, rows, cout, m.grads, m.scratch, beta=1.0, out_offset=gb.off)
                        if src != "in":                                                # pixels are data: no gradient needed
                            dcol = self.dcol[:rows * K]
                            ops.gemm(dy, kw_.mat(cout), ops.mat(dcol, K), rows, K, cout, trans_b=1)
                            g, beta = target(src)
                            ops.col2im(dcol, g, N, h, w, cin, k, k, s, pt, pl, ho, wo, beta=beta)
                for item in list(deferred):
                    if item[0] in written:
                        deferred.remove(item)
                        item[1]()
            # A deferred 1x1/2 data gradient waits for another contribution to its map.  In resnet_cnn every such map is also read by the
            # block's first batch norm, whose backward writes it -- and that backward is what the map's PRODUCER then consumes.  A layout
            # in which the strided shortcut were the map's only reader would reach this point with the producer's backward already run on
            # an unwritten gradient map: refuse it instead of flushing too late.
        finally:
            ops.slab_defer_end()
        assert not deferred, "LipCNN.backward: a map read only by a strided 1x1 shortcut (%s) is not supported" % [d[0] for d in deferred]
