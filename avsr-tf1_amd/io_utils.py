"""TF-free input pipeline with the behaviour of avsr/io_utils.py (tf.data over TFRecord SequenceExamples).

On-disk schema = what avsr/dataset_writer.py writes (:290-311 labels, :439-459 features, :461-498 video):
  data record   context {input_length:int64, input_size:int64 | width,height,channels:int64, filename:bytes}
                feature_lists {inputs: one float list per time step [, aus: float[2] per step]}
  label record  context {unit:bytes, labels_length:int64, filename:bytes}, feature_lists {labels: one int64 per step}
TFRecord framing (tensorflow/core/lib/io/record_writer.cc): u64 length | masked crc32c(length) | payload |
masked crc32c(payload).  The protobuf wire format is decoded by hand (no dependency on tensorflow or protobuf).

Pipeline semantics mirrored from the reference: zip data and label records by position (io_utils.py:99, :189-194),
append EOS and add 1 to labels_length (:81-85), optional max_sentence_length filter (:102-103), shuffle with a
5000-element buffer (:105-106), bucket by input_length // bucket_width with windows of batch_size
(group_by_window, :133-143), zero padded batches with a ragged final batch (:113-127).
No reference .tfrecord exists to test against, so byte compatibility is checked against the protobuf wire format
produced by the official `protobuf` runtime in tests/test_io.py (and CRC32C against its published test vector).
"""
import collections
import random
import os
import struct

import numpy as np


class BatchedData(collections.namedtuple("BatchedData", ("iterator_initializer", "inputs", "inputs_length", "inputs_filenames",
                                                          "labels", "labels_length", "labels_filenames", "payload"))):
    """Same fields as avsr/io_utils.py:8-18; here `inputs` etc. are numpy arrays of ONE batch."""


# ------------------------------------------------------------------------------------------------
# crc32c + TFRecord framing
def _make_crc_table():
    tbl = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tbl.append(c)
    return tbl


_CRC_TABLE = _make_crc_table()


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def read_tfrecord(path, verify_crc=False):
    """Yield the serialized payload of every record of a TFRecord file."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError("truncated TFRecord header in %s" % path)
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if verify_crc and _masked_crc(head[:8]) != lcrc:
                raise IOError("corrupt TFRecord length crc in %s" % path)
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise IOError("truncated TFRecord payload in %s" % path)
            if verify_crc and _masked_crc(data) != struct.unpack("<I", tail)[0]:
                raise IOError("corrupt TFRecord payload crc in %s" % path)
            yield data


class _Rec:
    """One record payload as a window of a memory-mapped TFRecord file: no copy is made when the file is read -- the native filler
    copies the values straight from the page cache into the (pinned) batch buffer.  Behaves like `bytes` where the pipeline needs it
    (len, slicing to bytes, bytes())."""
    __slots__ = ("base", "off", "n")

    def __init__(self, base, off, n):
        self.base, self.off, self.n = base, off, n             # base: read-only uint8 array over the whole mapping

    def __len__(self):
        return self.n

    def __getitem__(self, k):
        if isinstance(k, slice):
            a, b, _ = k.indices(self.n)
            return self.base[self.off + a:self.off + max(a, b)].tobytes()
        return int(self.base[self.off + (k if k >= 0 else self.n + k)])

    def __bytes__(self):
        return self.base[self.off:self.off + self.n].tobytes()

    @property
    def addr(self):
        return self.base.ctypes.data + self.off


def read_tfrecord_mapped(path):
    """Yield a _Rec for every record of a TFRecord file (the file is memory-mapped; framing as read_tfrecord, no crc check).
    Empty files and platforms without mmap fall back to read_tfrecord's bytes."""
    import mmap
    try:
        f = open(path, "rb")
        size = os.fstat(f.fileno()).st_size
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) if size else None
    except (OSError, ValueError):
        mm = None
    if mm is None:
        yield from read_tfrecord(path)
        return
    base = np.frombuffer(mm, dtype=np.uint8)                    # keeps the mapping alive for as long as a record refers to it
    pos = 0
    while pos < size:
        if pos + 12 > size:
            raise IOError("truncated TFRecord header in %s" % path)
        length = int(base[pos:pos + 8].view("<u8")[0])
        if pos + 12 + length + 4 > size:
            raise IOError("truncated TFRecord payload in %s" % path)
        yield _Rec(base, pos + 12, length)
        pos += 12 + length + 4


class TFRecordFileWriter:
    def __init__(self, path):
        self._f = open(path, "wb")

    def write(self, payload: bytes):
        head = struct.pack("<Q", len(payload))
        self._f.write(head + struct.pack("<I", _masked_crc(head)) + payload + struct.pack("<I", _masked_crc(payload)))

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ------------------------------------------------------------------------------------------------
# protobuf wire format (tensorflow/core/example/{example,feature}.proto)
def _varint(buf, i):
    r, s = 0, 0
    while True:
        b = buf[i]
        i += 1
        r |= (b & 0x7F) << s
        if b < 0x80:
            return r, i
        s += 7


def _enc_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fields(buf):
    """Iterate (field_number, wire_type, value) of one message; length-delimited values are memoryview slices."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = buf[i:i + ln]
            i += ln
        elif wt == 5:
            v = buf[i:i + 4]
            i += 4
        elif wt == 1:
            v = buf[i:i + 8]
            i += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _parse_feature(buf):
    """Feature -> ('bytes', [bytes]) | ('float', np.float32[]) | ('int64', np.int64[])."""
    for fn, _wt, v in _fields(buf):
        if fn == 1:
            return "bytes", [bytes(x) for f2, _w, x in _fields(v) if f2 == 1]
        if fn == 2:
            parts = []
            for f2, w2, x in _fields(v):
                if f2 == 1:
                    parts.append(np.frombuffer(x, dtype="<f4"))          # packed (wt 2) or single fixed32 (wt 5)
            return "float", (np.concatenate(parts) if len(parts) != 1 else parts[0]) if parts else np.zeros(0, np.float32)
        if fn == 3:
            vals = []
            for f2, w2, x in _fields(v):
                if f2 != 1:
                    continue
                if w2 == 0:
                    vals.append(x)
                else:
                    j, xb = 0, bytes(x)
                    while j < len(xb):
                        val, j = _varint(xb, j)
                        vals.append(val)
            arr = np.array(vals, dtype=np.uint64).astype(np.int64)
            return "int64", arr
    return "none", None


def _fast_float_steps(x):
    """FeatureList whose steps are equally sized packed float Features (every feature / frame vector of one stream is):
    the steps sit at a constant stride, so the whole [T, k] array is ONE strided view of the payload instead of T parsed messages.
    Returns None when the layout is anything else (the generic parser then handles it)."""
    n = len(x)
    if n < 8 or x[0] != 0x0A:
        return None
    l1, i = _varint(x, 1)                                # Feature length
    stride = i + l1
    if stride <= 0 or n % stride or x[i] != 0x12:        # Feature.float_list
        return None
    l2, j = _varint(x, i + 1)
    if x[j] != 0x0A:                                     # FloatList.value, packed
        return None
    l3, h = _varint(x, j + 1)
    if l3 % 4 or h + l3 != stride or l2 != (h - j) + l3:
        return None
    rows = np.frombuffer(x, dtype=np.uint8).reshape(n // stride, stride)
    if not (rows[:, :h] == rows[0, :h]).all():           # every step carries the same header bytes
        return None
    return np.ascontiguousarray(rows[:, h:]).view("<f4")


def _fast_small_int_steps(x):
    """FeatureList of single small (< 128, one-byte varint) packed int64 values per step -- the label lists: constant stride again."""
    n = len(x)
    if n < 7 or x[0] != 0x0A:
        return None
    l1, i = _varint(x, 1)
    stride = i + l1
    if stride <= 0 or n % stride or x[i] != 0x1A:        # Feature.int64_list
        return None
    l2, j = _varint(x, i + 1)
    if x[j] != 0x0A:                                     # Int64List.value, packed
        return None
    l3, h = _varint(x, j + 1)
    if l3 != 1 or h + 1 != stride:
        return None
    rows = np.frombuffer(x, dtype=np.uint8).reshape(n // stride, stride)
    if not (rows[:, :h] == rows[0, :h]).all() or (rows[:, h] >= 0x80).any():
        return None
    return rows[:, h:h + 1].astype(np.int64)


def parse_sequence_example(payload: bytes):
    """-> (context: {name: value-list}, feature_lists: {name: [value per step]})."""
    buf = memoryview(payload)
    context, flists = {}, {}
    for fn, _wt, v in _fields(buf):
        if fn == 1:                                   # Features context
            for f2, _w, entry in _fields(v):
                if f2 != 1:
                    continue
                key, feat = None, None
                for f3, _w3, x in _fields(entry):
                    if f3 == 1:
                        key = bytes(x).decode("utf-8")
                    elif f3 == 2:
                        feat = _parse_feature(x)
                context[key] = feat[1] if feat else None
        elif fn == 2:                                 # FeatureLists
            for f2, _w, entry in _fields(v):
                if f2 != 1:
                    continue
                key, steps = None, []
                for f3, _w3, x in _fields(entry):
                    if f3 == 1:
                        key = bytes(x).decode("utf-8")
                    elif f3 == 2:                     # FeatureList { repeated Feature feature = 1; }
                        steps = _fast_float_steps(x)
                        if steps is None:
                            steps = _fast_small_int_steps(x)
                        if steps is None:
                            steps = [_parse_feature(fx)[1] for f4, _w4, fx in _fields(x) if f4 == 1]
                flists[key] = steps
    return context, flists


def _ld(fn, payload):
    return _enc_varint((fn << 3) | 2) + _enc_varint(len(payload)) + payload


def _enc_feature(value):
    if isinstance(value, (bytes, str)):
        b = value.encode("utf-8") if isinstance(value, str) else value
        return _ld(1, _ld(1, b))
    arr = np.asarray(value)
    if arr.dtype.kind == "f":
        return _ld(2, _ld(1, arr.astype("<f4").tobytes()))
    return _ld(3, _ld(1, b"".join(_enc_varint(int(x)) for x in arr.reshape(-1))))


def make_sequence_example(context: dict, feature_lists: dict) -> bytes:
    """Serialize with the same field layout tf.train.SequenceExample produces (packed repeated scalars)."""
    ctx = b"".join(_ld(1, _ld(1, k.encode("utf-8")) + _ld(2, _enc_feature(v))) for k, v in context.items())
    fls = b""
    for k, steps in feature_lists.items():
        fl = b"".join(_ld(1, _enc_feature(s)) for s in steps)
        fls += _ld(1, _ld(1, k.encode("utf-8")) + _ld(2, fl))
    return _ld(1, ctx) + _ld(2, fls)


def make_feature_example(sentence_id, inputs):
    """avsr/dataset_writer.py:439-459"""
    inputs = np.asarray(inputs, np.float32)
    return make_sequence_example({"input_length": [len(inputs)], "input_size": [inputs.shape[-1]], "filename": sentence_id},
                                 {"inputs": list(inputs)})


def make_video_example(sentence_id, frames, aus=None):
    """avsr/dataset_writer.py:461-498 (frames [T,H,W[,C]])"""
    frames = np.asarray(frames, np.float32)
    h, w = frames.shape[1], frames.shape[2]
    c = frames.shape[3] if frames.ndim == 4 else 1
    fl = {"inputs": [f.reshape(-1) for f in frames]}
    if aus is not None:
        fl["aus"] = [np.asarray(a, np.float32).reshape(-1) for a in aus]
    return make_sequence_example({"input_length": [len(frames)], "width": [w], "height": [h], "channels": [c], "filename": sentence_id}, fl)


def make_label_example(label_id, labels, unit):
    """avsr/dataset_writer.py:290-311"""
    return make_sequence_example({"unit": unit, "labels_length": [len(labels)], "filename": label_id},
                                 {"labels": [np.asarray([int(l)], np.int64) for l in labels]})


# ------------------------------------------------------------------------------------------------
def create_unit_dict(unit_file):
    """id -> symbol: MASK=0, END=-1, symbols 1..N, EOS=N+1, GO=N+2 (avsr/io_utils.py:354-370)."""
    unit_dict = {"MASK": 0, "END": -1}
    with open(unit_file, "r") as f:
        unit_list = f.read().splitlines()
    idx = 0
    for idx, sub in enumerate(unit_list):
        unit_dict[sub] = idx + 1
    unit_dict["EOS"] = idx + 2
    unit_dict["GO"] = idx + 3
    return {v: k for k, v in unit_dict.items()}


def _get_input_shape_from_record(record):
    """avsr/io_utils.py:309-345: feature-vector stream vs raw video stream, optional Action Units."""
    ctx, fl = parse_sequence_example(next(read_tfrecord(record)))
    content = {}
    if "input_size" in ctx:
        shape, content["stream"] = [int(ctx["input_size"][0])], "feature"
    else:
        ch = int(ctx["channels"][0]) if "channels" in ctx else 1
        shape, content["stream"] = [int(ctx["width"][0]), int(ctx["height"][0]), ch], "video"
    if fl.get("aus") is not None and len(fl["aus"]):
        content["aus"] = True
    return shape, content


def _parse_input(payload, input_shape):
    ctx, fl = parse_sequence_example(payload)
    T = int(ctx["input_length"][0])
    x = np.asarray(fl["inputs"] if isinstance(fl["inputs"], np.ndarray) else np.stack(fl["inputs"]), dtype=np.float32).reshape(
        [T] + list(input_shape)) if T else np.zeros([0] + list(input_shape), np.float32)
    aus_l = fl.get("aus")
    aus = None
    if aus_l is not None and len(aus_l):
        aus = np.asarray(aus_l if isinstance(aus_l, np.ndarray) else np.stack(aus_l), dtype=np.float32).reshape(T, 2)
    return x, aus, T, ctx["filename"][0]


def _parse_labels(payload, eos_id):
    ctx, fl = parse_sequence_example(payload)
    lab = np.array([int(s[0]) for s in fl["labels"]] + [eos_id], dtype=np.int32)      # io_utils.py:81-83
    return lab, int(ctx["labels_length"][0]) + 1, ctx["filename"][0]


def _pad_stack(arrs, T=None):
    T = max(a.shape[0] for a in arrs) if T is None else T
    out = np.zeros((len(arrs), T) + arrs[0].shape[1:], arrs[0].dtype)
    for i, a in enumerate(arrs):
        out[i, :a.shape[0]] = a
    return out


class _Pipeline:
    """shuffle(5000) -> bucket(group_by_window) -> padded_batch over zipped (data streams..., labels)."""

    def __init__(self, data_records, label_record, unit_dict, batch_size, shuffle, bucket_width, max_sentence_length, seed=None,
                 rank=0, world=1):
        # data parallelism by utterance: every rank runs the SAME pipeline (same shuffle seed) -- shuffle, bucket
        # (avsr/io_utils.py:133-147: group_by_window on the input length) and batch exactly as one process would -- and then keeps
        # its contiguous share of every bucketed batch.  "Bucket first, then split": the global batches are the reference's.
        self.rank, self.world = int(rank), int(world)
        if self.world > 1 and shuffle and seed is None:
            raise ValueError("data-parallel input pipelines need a shared shuffle seed (every rank must draw the same order)")
        self.data_records, self.label_record = data_records, label_record
        self.eos = {v: k for k, v in unit_dict.items()}["EOS"]
        self.shapes = [_get_input_shape_from_record(r) for r in data_records]
        self.batch_size, self.shuffle, self.bucket_width, self.max_len = batch_size, shuffle, bucket_width, max_sentence_length
        self.rng = random.Random(seed)
        self.shuffle_buffer = 5000
        from . import _io_native
        self.native = _io_native if _io_native.load() is not None else None
        # reuse_buffers (set by AVSR.train / evaluate, which consume a batch before asking for the one after next): the padded input
        # arrays of a batch are windows of a small ring of host buffers -- page-locked when a GPU is present, so the host-to-device
        # copy is an asynchronous DMA from the very buffer the filler wrote -- instead of fresh allocations (a 75 MB lip-crop batch
        # otherwise costs 18 000 page faults before the first byte is copied).  Off by default: a caller may keep batches.
        self.reuse_buffers = False
        self._ring, self._ring_pos, self.RING = {}, {}, 6

    def _out_buffer(self, key, shape):
        """A float32 array of `shape` inside the next slot of ring `key` (or None: let the filler allocate)."""
        if not self.reuse_buffers:
            return None
        n = int(np.prod(shape))
        slots = self._ring.get(key)
        if slots is None or slots[0][1].size < n:
            cap = max(n, 0 if slots is None else 2 * slots[0][1].size)
            slots = []
            for _ in range(self.RING):
                keep, arr = None, None
                try:
                    import torch
                    if torch.cuda.is_available():
                        keep = torch.empty(cap, dtype=torch.float32, pin_memory=True)
                        arr = keep.numpy()
                except Exception as e:
                    keep = None
                    if not getattr(self, "_pin_warned", False):
                        self._pin_warned = True
                        import warnings
                        warnings.warn("input pipeline: page-locked batch buffer unavailable (%s: %s); falling back to pageable memory "
                                      "-- host-to-device copies become synchronous" % (type(e).__name__, e))
                if arr is None:
                    arr = np.empty(cap, np.float32)
                slots.append((keep, arr))
            self._ring[key], self._ring_pos[key] = slots, 0
        i = self._ring_pos[key]
        self._ring_pos[key] = (i + 1) % self.RING
        return slots[i][1][:n].reshape(shape)

    def _key(self, ex):
        return ex[0][0][2] // self.bucket_width                # first stream's input_length (video for AV)

    # Native path (include/avsr_io.h): the records of a chunk are INDEXED by the C helper (value regions, lengths, file names) and the
    # examples that flow through shuffle / bucket carry (payload, index row) instead of parsed arrays; _batch fills the padded arrays
    # straight from the payloads.  An example is a pair (streams, lab) either way:
    #   stream = (x | None, aus | None, T, filename, payload, row)      lab = (labels | None, labels_length + 1, filename, payload, row, n + 1)
    # with x / labels None for indexed records.  Records the helper does not recognise are parsed by the python parser (same tuples,
    # arrays filled in).
    def _examples_native(self, its):
        N, F = self.native, self.native.F
        ns = len(self.data_records)
        sizes = [int(np.prod(shp[0])) for shp in self.shapes]

        def flush(chunk):
            info = N.index([p for recs in chunk for p in recs])
            for ci, recs in enumerate(chunk):
                rows = info[ci * (ns + 1):(ci + 1) * (ns + 1)]
                lr = rows[ns]
                ok = not rows[:, F["slow"]].any() and lr[F["labels_length"]] >= 0 and lr[F["fn_off"]] >= 0
                for k in range(ns):
                    r = rows[k]
                    # exactly the layout _stack_stream fills from: one step per frame, `size` floats per step (2 for the Action
                    # Units).  A record whose floats are split differently (same product) is legal for the generic parser and is
                    # left to it -- the native filler copies steps[b] * step_floats per utterance and must never see another split.
                    ok = ok and r[F["input_length"]] >= 0 and r[F["fn_off"]] >= 0 and \
                        r[F["in_T"]] == r[F["input_length"]] and r[F["in_F"]] == sizes[k] and \
                        (r[F["aus_T"]] == 0 or (r[F["aus_F"]] == 2 and r[F["aus_T"]] == r[F["input_length"]]))
                if not ok:                                  # unusual layout: the generic parser decides (and raises what it raises)
                    streams = [_parse_input(bytes(p), shp[0]) + (None, None) for p, shp in zip(recs[:-1], self.shapes)]
                    lab = _parse_labels(bytes(recs[-1]), self.eos)
                    lab = lab + (None, None, lab[0].shape[0])
                else:
                    streams = []
                    for k in range(ns):
                        r, pl = rows[k], recs[k]
                        streams.append((None, None, int(r[F["input_length"]]), pl[r[F["fn_off"]]:r[F["fn_off"]] + r[F["fn_len"]]], pl, r))
                    pl = recs[ns]
                    lab = (None, int(lr[F["labels_length"]]) + 1, pl[lr[F["fn_off"]]:lr[F["fn_off"]] + lr[F["fn_len"]]], pl, lr,
                           int(lr[F["lab_n"]]) + 1)
                if self.max_len is not None and not lab[1] < self.max_len:
                    continue
                yield streams, lab

        chunk = []
        for recs in zip(*its):
            chunk.append(recs)
            if len(chunk) == 256:
                yield from flush(chunk)
                chunk = []
        if chunk:
            yield from flush(chunk)

    def _examples(self):
        if self.native is not None:
            # memory-mapped records: the shuffle buffer then holds 5000 (offset, length) windows, not 5000 payload copies, and a value
            # is copied exactly once -- page cache -> batch buffer -- by the filler's threads
            its = [read_tfrecord_mapped(r) for r in self.data_records] + [read_tfrecord_mapped(self.label_record)]
            yield from self._examples_native(its)
            return
        its = [read_tfrecord(r) for r in self.data_records] + [read_tfrecord(self.label_record)]
        for recs in zip(*its):
            streams = [_parse_input(p, shp[0]) for p, shp in zip(recs[:-1], self.shapes)]
            lab = _parse_labels(recs[-1], self.eos)
            if self.max_len is not None and not lab[1] < self.max_len:
                continue
            yield streams, lab

    def _shuffled(self):
        if not self.shuffle:
            yield from self._examples()
            return
        buf = []
        for ex in self._examples():
            buf.append(ex)
            if len(buf) >= self.shuffle_buffer:
                yield buf.pop(self.rng.randrange(len(buf)))
        self.rng.shuffle(buf)
        yield from buf

    def _shard(self, exs):
        """This rank's share of a global batch.  A batch with fewer utterances than ranks is processed whole by every rank: the
        summed gradients and the summed loss normaliser then both carry the factor `world`, which cancels."""
        if self.world == 1 or len(exs) < self.world:
            return exs
        n, w, r = len(exs), self.world, self.rank
        lo, hi = (n * r) // w, (n * (r + 1)) // w
        return exs[lo:hi]

    def _stack_stream(self, st, k, T):
        """Zero-padded [n, T, ...] inputs (and Action Units or None) of stream k from its per-utterance entries."""
        shape = tuple(self.shapes[k][0])
        idx = [i for i, s in enumerate(st) if s[0] is None]                    # indexed records: filled by the native helper
        if not idx:
            x = _pad_stack([s[0] for s in st], T)
            aus = _pad_stack([s[1] for s in st], T) if st[0][1] is not None else None
            return x, aus
        N, F = self.native, self.native.F
        pls = [s[4] if s[0] is None else b"" for s in st]
        rows = [s[5] if s[0] is None else None for s in st]
        col = lambda name: [0 if r is None else int(r[F[name]]) for r in rows]
        step = max(col("in_F"))
        x = N.fill_f32(pls, col("in_off"), col("in_stride"), col("in_T"), step, T, shape, out=self._out_buffer((k, "x"), (len(st), T) + shape))
        has_aus = any(r is not None and r[F["aus_T"]] > 0 for r in rows) or any(s[0] is not None and s[1] is not None for s in st)
        aus = N.fill_f32(pls, col("aus_off"), col("aus_stride"), col("aus_T"), 2, T, (2,),
                         out=self._out_buffer((k, "aus"), (len(st), T, 2))) if has_aus else None
        for i, s in enumerate(st):                                              # the few generically parsed ones
            if s[0] is not None:
                x[i, :s[0].shape[0]] = s[0]
                if aus is not None and s[1] is not None:
                    aus[i, :s[1].shape[0]] = s[1]
        return x, aus

    def _stack_labels(self, labs, Lmax):
        if all(e[0] is not None for e in labs):
            return _pad_stack([e[0] for e in labs], Lmax)
        N, F = self.native, self.native.F
        rows = [e[4] if e[0] is None else None for e in labs]
        col = lambda name: [0 if r is None else int(r[F[name]]) for r in rows]
        out = N.fill_labels([e[3] if e[0] is None else b"\0" for e in labs], col("lab_off"), col("lab_stride"), col("lab_n"), self.eos, Lmax)
        for i, e in enumerate(labs):
            if e[0] is not None:
                out[i] = 0
                out[i, :e[0].shape[0]] = e[0]
        return out

    @staticmethod
    def _lab_len(lab):
        return lab[0].shape[0] if lab[0] is not None else lab[5]

    def _batch(self, exs):
        # a rank's shard is padded to the lengths of the GLOBAL batch: padded_batch pads to the longest member of the whole batch, and
        # the input batch-norm takes its statistics over the padded rows too, so the shard must keep that row count
        full = exs
        exs = self._shard(exs)
        Lmax = max(self._lab_len(e[1]) for e in full)
        if not exs[0][0]:                                       # label-only pipelines (language model)
            names = [e[1][2] for e in exs]
            return BatchedData(None, None, None, None, _pad_stack([e[1][0] for e in exs], Lmax), np.array([e[1][1] for e in exs], np.int32),
                               None if names[0] is None else names, None)
        streams = list(zip(*[e[0] for e in exs]))
        fstreams = list(zip(*[e[0] for e in full]))
        # (an indexed record's frame count is its input_length: the consistency of the two was checked when it was indexed)
        Tmax = [max((s[0].shape[0] if s[0] is not None else s[2]) for s in st) for st in fstreams]
        inputs, payload = [], {}
        for k, (st, T) in enumerate(zip(streams, Tmax)):
            x, aus = self._stack_stream(st, k, T)
            inputs.append(x)
            if aus is not None:
                payload["aus"] = aus
        lens = [np.array([s[2] for s in st], np.int32) for st in streams]
        names = [[s[3] for s in st] for st in streams]
        labels = self._stack_labels([e[1] for e in exs], Lmax)
        llen = np.array([e[1][1] for e in exs], np.int32)
        lnames = [e[1][2] for e in exs]
        one = len(inputs) == 1
        return BatchedData(None, inputs[0] if one else tuple(inputs), lens[0] if one else tuple(lens),
                           names[0] if one else tuple(names), labels, llen, lnames, payload)

    def __iter__(self):
        if self.bucket_width == -1:
            cur = []
            for ex in self._shuffled():
                cur.append(ex)
                if len(cur) == self.batch_size:
                    yield self._batch(cur)
                    cur = []
            if cur:
                yield self._batch(cur)
            return
        windows = collections.OrderedDict()
        for ex in self._shuffled():
            key = self._key(ex)
            w = windows.setdefault(key, [])
            w.append(ex)
            if len(w) == self.batch_size:
                yield self._batch(w)
                del windows[key]
        for w in windows.values():                              # group_by_window flushes partial windows at the end
            yield self._batch(w)


def make_iterator_from_one_record(data_record, label_record, unit_dict, batch_size, shuffle=False, reverse_input=False,
                                  bucket_width=-1, num_cores=4, max_sentence_length=None, seed=None, rank=0, world=1):
    """Iterable of BatchedData (avsr/io_utils.py:88-165).  reverse_input is always False in the reference's callers."""
    if reverse_input:
        raise NotImplementedError("reverse_input is never enabled by the reference (avsr/avsr.py:646, :658, :671)")
    return _Pipeline([data_record], label_record, unit_dict, batch_size, shuffle, bucket_width, max_sentence_length, seed, rank, world)


def make_iterator_from_two_records(video_record, audio_record, label_record, batch_size, unit_dict, shuffle=False,
                                   reverse_input=False, bucket_width=-1, num_cores=4, seed=None, rank=0, world=1):
    """Iterable of BatchedData with (video, audio) tuples (avsr/io_utils.py:168-259); buckets on the VIDEO length."""
    return _Pipeline([video_record, audio_record], label_record, unit_dict, batch_size, shuffle, bucket_width, None, seed, rank, world)


class _LabelPipeline(_Pipeline):
    """Label-only batches for avsr.LM: TFRecord labels (EOS appended, shuffle buffer 45000; avsr/io_utils.py:262-308) or a
    text file, one sentence per line split into characters and looked up in the unit list (no EOS, unknown symbols -> -1,
    shuffle buffer 1000000; avsr/io_utils.py:383-440).  Buckets on the label length."""

    def __init__(self, label_record, text_dataset, unit_dict, batch_size, shuffle, bucket_width, seed=None):
        self.label_record, self.text_dataset = label_record, text_dataset
        self.eos = {v: k for k, v in unit_dict.items()}["EOS"]
        self.lookup = {v: k for k, v in unit_dict.items()}
        self.batch_size, self.shuffle, self.bucket_width = batch_size, shuffle, bucket_width
        self.rng = random.Random(seed)
        self.rank, self.world = 0, 1
        self.shuffle_buffer = 45000 if text_dataset is None else 1000000

    def _key(self, ex):
        return ex[1][1] // self.bucket_width

    def _examples(self):
        if self.text_dataset is None:
            for rec in read_tfrecord(self.label_record):
                yield (), _parse_labels(rec, self.eos)
            return
        with open(self.text_dataset, "r") as f:
            for line in f:
                chars = list(line.rstrip("\n"))
                yield (), (np.array([self.lookup.get(c, -1) for c in chars], dtype=np.int32), len(chars), None)


def make_iterator_from_label_record(label_record, batch_size, unit_dict, shuffle=False, reverse_input=False, bucket_width=-1, num_cores=4,
                                    seed=None):
    """Iterable of label-only BatchedData (avsr/io_utils.py:262-308)."""
    return _LabelPipeline(label_record, None, unit_dict, batch_size, shuffle, bucket_width, seed)


def make_iterator_from_text_dataset(text_dataset, batch_size, unit_dict, shuffle=False, bucket_width=-1, num_cores=4, seed=None):
    """Iterable of label-only BatchedData from a text file (avsr/io_utils.py:383-440)."""
    return _LabelPipeline(None, text_dataset, unit_dict, batch_size, shuffle, bucket_width, seed)
