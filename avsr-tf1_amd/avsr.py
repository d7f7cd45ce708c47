"""`AVSR` -- the reference's top-level class (avsr/avsr.py:20-753) on the MI355X engine.

Same constructor keywords and defaults (avsr/avsr.py:21-75), same `train(logfile, num_epochs,
try_restore_latest_checkpoint)` / `evaluate(checkpoint_path, epoch)` behaviour and side-effect files:
`logs/<logfile>` (average loss per epoch, error rates every 10th epoch), `checkpoints/<name>/checkpoint.ckp-<epoch>`
(own .npz format: TF-layout weights + Adam slots + step; epoch parsed from the file name on resume, :233-249),
`predictions/<name>/predicted_epoch_<N>.mlf`.  TensorFlow graphs/sessions/summaries do not exist here.

`video_processing='resnet_cnn'` runs the lip crops through the HIP lip-CNN front-end (cnn.py; avsr/video.py:143-195).
Built besides the defaults: `input_dense_layers`, `instance_normalisation`, `residual_encoder`, `highway_encoder`, `encoder_weight_sharing`, multi-layer
decoders (equal widths), `enable_attention=False`, `loss_fun` / `label_smoothing`, `lr_decay=('cosine_restarts', N)`, the Nadam /
AdamW / Momentum optimisers, `write_attention_alignment` (greedy decoding), one-hot decoder inputs (`embedding_size <= 0`).  Feature / unit /
embedding sizes may be anything (the engine pads to multiples of 4 inside, config.py `engine()`; checkpoints keep the reference's shapes).
Not built (raise explicitly): `precision='float16'`, the `2dconv_cnn` / `3dconv_cnn` front-ends, `'wav'` audio
(non-functional in the reference too), the monotonic attention variants and the non-default cell types.
"""
import glob
import os
import time
from os import makedirs, path

import numpy as np
import torch

from . import ops
from .config import ModelConfig
from .io_utils import (_get_input_shape_from_record, create_unit_dict, make_iterator_from_one_record,
                       make_iterator_from_two_records)
from .model import Batch, Seq2SeqModel
from .parallel import DataParallelTrainer
from .utils import alignment_image, compute_wer, write_png_gray, write_sequences_to_labelfile


class AVSR(object):
    def __init__(self,
                 unit,
                 unit_file=None,
                 video_processing=None,
                 video_train_record=None,
                 video_test_record=None,
                 audio_processing=None,
                 audio_train_record=None,
                 audio_test_record=None,
                 labels_train_record=None,
                 labels_test_record=None,
                 batch_size=(64, 64),
                 cnn_filters=(8, 16, 32, 64),
                 cnn_dense_units=128,
                 regress_aus=False,
                 batch_normalisation=True,
                 instance_normalisation=False,
                 input_dense_layers=(0,),
                 architecture='unimodal',
                 encoder_type='unidirectional',
                 highway_encoder=False,
                 residual_encoder=False,
                 cell_type='lstm',
                 recurrent_l2_regularisation=0.0001,
                 weight_decay=0.0001,
                 encoder_units_per_layer=((256, ), (256, 256, 256)),
                 decoder_units_per_layer=(256,),
                 encoder_weight_sharing=False,
                 enable_attention=True,
                 attention_type=(('scaled_luong',)*1, ('scaled_luong',)*1),
                 use_dropout=True,
                 audio_encoder_dropout_probability=(0.9, 0.9, 0.9),
                 video_encoder_dropout_probability=(0.9, 0.9, 0.9),
                 decoder_dropout_probability=(0.9, 0.9, 0.9),
                 embedding_size=128,
                 sampling_probability_outputs=0.1,
                 label_smoothing=0.0,
                 decoding_algorithm='beam_search',
                 beam_width=10,
                 max_sentence_length=None,
                 optimiser='Adam',
                 learning_rate=0.001,
                 lr_decay=None,
                 loss_fun=None,
                 clip_gradients=True,
                 max_gradient_norm=1.0,
                 num_gpus=1,
                 write_attention_alignment=False,
                 write_beam_search_graphs=False,
                 write_estimated_modality_lags=False,
                 precision='float32',
                 profiling=False,
                 required_grahps=('train', 'eval'),
                 **kwargs):
        self._unit = unit
        self._unit_dict = create_unit_dict(unit_file=unit_file)
        self._video_processing, self._audio_processing = video_processing, audio_processing
        self._records = {
            'train': (video_train_record, audio_train_record, labels_train_record),
            'evaluate': (video_test_record, audio_test_record, labels_test_record)}
        self._batch_size = batch_size
        self._max_sentence_length = max_sentence_length
        self._required_graphs = required_grahps

        for name, val, ok in (("precision", precision, 'float32'),):
            if val != ok:
                raise NotImplementedError("%s=%r is a non-default option of the reference that the HIP engine does not build" % (name, val))
        self._profiling = bool(profiling)
        lr_decay_steps = 0
        if lr_decay is not None:                                                                  # seq2seq.py:263-273
            if lr_decay[0] == 'cosine_restarts':
                lr_decay_steps = int(lr_decay[1])
            else:
                print('learning rate policy not implemented, falling back to constant learning rate')
        if loss_fun not in (None, 'focal_loss', 'mc_loss'):
            raise ValueError('Unknown loss function {}'.format(loss_fun))                           # seq2seq.py:163
        if video_processing is not None and video_processing not in ('features', 'resnet_cnn'):
            if 'cnn' in video_processing:
                raise NotImplementedError("video_processing=%r: only the default `resnet_cnn` front-end is built" % video_processing)
            raise Exception('unknown visual content')                                             # avsr/avsr.py:713
        if audio_processing is not None and audio_processing != 'features':
            raise NotImplementedError("audio_processing=%r (the reference's 'wav' path is non-functional as well)" % audio_processing)
        if decoding_algorithm not in ('greedy', 'beam_search'):
            raise Exception('The only supported algorithms are `greedy` and `beam_search`')     # decoder_unimodal.py:124
        self._decoding_algorithm, self._beam_width = decoding_algorithm, beam_width
        self._write_attention_alignment = bool(write_attention_alignment)
        if self._write_attention_alignment and decoding_algorithm != 'greedy':
            raise NotImplementedError("write_attention_alignment=True needs decoding_algorithm='greedy' (the alignment history is "
                                      "kept by the greedy decode only)")

        reverse = {v: k for k, v in self._unit_dict.items()}
        feats, video_hw = {}, (36, 36, 3)
        for idx, (proc, key) in enumerate(((video_processing, 'video'), (audio_processing, 'audio'))):
            rec = self._records['train'][idx] or self._records['evaluate'][idx]
            if proc is not None:
                shape, _ = _get_input_shape_from_record(rec)
                if key == 'video' and proc == 'resnet_cnn':
                    if len(shape) != 3:
                        raise ValueError("video_processing='resnet_cnn' needs raw [width, height, channels] frames in the video record")
                    video_hw, feats[key] = tuple(shape), cnn_dense_units
                    continue
                if len(shape) != 1:
                    raise ValueError("raw video records need video_processing='resnet_cnn'")
                feats[key] = shape[0]
        self._cfg = ModelConfig(
            architecture=architecture, encoder_type=encoder_type, cell_type=cell_type,
            video_units=tuple(encoder_units_per_layer[0]) if video_processing is not None else None,
            audio_units=tuple(encoder_units_per_layer[1]) if audio_processing is not None else None,
            decoder_units=tuple(decoder_units_per_layer), attention_type=tuple(tuple(t) for t in attention_type),
            enable_attention=enable_attention, embedding_size=embedding_size,
            vocab_size=len(self._unit_dict) - 1, go_id=reverse['GO'], eos_id=reverse['EOS'],
            video_feat=feats.get('video', 128), audio_feat=feats.get('audio', 80),
            batch_normalisation=batch_normalisation, regress_aus=regress_aus,
            au_loss_weight=kwargs.get('au_loss_weight', 10.0),
            recurrent_l2=None if optimiser == 'AdamW' else recurrent_l2_regularisation,     # avsr.py:168
            optimiser=optimiser, weight_decay=weight_decay, clip_gradients=clip_gradients, max_gradient_norm=max_gradient_norm,
            learning_rate=learning_rate, warmup_steps=kwargs.get('warmup_steps', 750), lr_decay_steps=lr_decay_steps, loss_fun=loss_fun, label_smoothing=float(label_smoothing),
            max_label_length={'viseme': 150, 'phoneme': 150, 'character': 150}[unit],
            use_dropout=use_dropout, video_dropout=tuple(video_encoder_dropout_probability),
            audio_dropout=tuple(audio_encoder_dropout_probability), decoder_dropout=tuple(decoder_dropout_probability),
            sampling_probability=sampling_probability_outputs,
            video_processing=video_processing if video_processing is not None else 'features',
            cnn_filters=tuple(cnn_filters), cnn_dense_units=cnn_dense_units, video_hw=video_hw,
            input_dense_layers=tuple(input_dense_layers), encoder_weight_sharing=bool(encoder_weight_sharing), residual_encoder=bool(residual_encoder), highway_encoder=bool(highway_encoder), instance_normalisation=bool(instance_normalisation))
        self._model = Seq2SeqModel(self._cfg, seed=kwargs.get('seed', 0))
        self._shuffle_seed = kwargs.get('shuffle_seed')        # None = a fresh order every run, as tf.data's unseeded shuffle(5000)
        # Data parallelism (the reference's num_gpus is deprecated and ignored, avsr/avsr.py:67,:127): when the process was started
        # under torch.distributed (one process per GPU, `torchrun`), every rank builds the same model, reads the same records,
        # shards each bucketed batch by utterance and all-reduces normalisers / BN statistics / gradients over RCCL (parallel.py).
        import torch.distributed as _dist
        self._epoch_counter = 0
        self._dist = _dist if (_dist.is_available() and _dist.is_initialized() and _dist.get_world_size() > 1) else None
        self._rank = self._dist.get_rank() if self._dist is not None else 0
        self._world = self._dist.get_world_size() if self._dist is not None else 1
        if self._dist is not None and self._shuffle_seed is None:      # every rank must draw the same shuffle order
            seed_t = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64)
            if _dist.get_backend() == 'nccl':
                seed_t = seed_t.cuda()
            _dist.broadcast(seed_t, src=0)
            self._shuffle_seed = int(seed_t.item())
        # bucketed batches: shapes vary per step, but full buckets at the longest lengths come back -- a shape is captured into a hipGraph
        # at its second sighting and replayed from then on (AVSR_TRAIN_GRAPH=0: eager launches only)
        # (profiling=True brackets every launch with an event pair: eager launches)
        self._trainer = DataParallelTrainer(self._model, self._dist,
                                            use_graph=os.environ.get("AVSR_TRAIN_GRAPH", "1") != "0" and not self._profiling,
                                            check_every_step=True, graph_after=2,
                                            # every captured shape pins its workspace (lip-CNN activations: GBs at B = 64) next to the
                                            # 8 unpinned ones of the model's LRU: keep few
                                            max_graphs=int(os.environ.get("AVSR_TRAIN_MAX_GRAPHS", "8")))

    # ------------------------------------------------------------------------------------------------
    def _iterator(self, mode):
        vrec, arec, lrec = self._records[mode]
        bs = self._batch_size[0 if mode == 'train' else 1]
        shuffle = mode == 'train'
        # training batches are sharded by utterance across the data-parallel ranks; evaluation runs whole on every rank
        rank, world = (self._rank, self._world) if mode == 'train' else (0, 1)
        seed = self._shuffle_seed
        if seed is not None:
            seed = seed + self._epoch_counter                     # a new order every epoch; under data parallelism identical on every rank
        if self._video_processing is not None and self._audio_processing is not None:
            return make_iterator_from_two_records(vrec, arec, lrec, bs, self._unit_dict, shuffle=shuffle, bucket_width=45,
                                                  seed=seed, rank=rank, world=world)
        rec = vrec if self._video_processing is not None else arec
        return make_iterator_from_one_record(rec, lrec, self._unit_dict, bs, shuffle=shuffle, bucket_width=45,
                                             max_sentence_length=self._max_sentence_length if self._audio_processing is not None else None,
                                             seed=seed, rank=rank, world=world)

    @staticmethod
    def _prefetched(it, depth=2):
        """Batches of `it`, produced by a helper thread up to `depth` ahead: TFRecord indexing and the padding copies (ctypes calls into
        libavsr_io.so, which release the GIL) run while the launching thread waits for the GPU step -- with the step replayed from a
        hipGraph that thread is idle almost all of the time.  depth + 3 <= the pipeline's buffer ring (io_utils._Pipeline.RING): `depth` queued, one being filled, two held by the training
        loop (the step in flight and the batch whose upload overlaps it)."""
        import queue
        import threading
        q, end = queue.Queue(maxsize=depth), object()
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None

        def run():
            try:
                if dev is not None:
                    torch.cuda.set_device(dev)            # page-locked buffers are allocated here: on THIS rank's device, not on device 0
                for b in it:
                    q.put(b)
                q.put(end)
            except BaseException as e:                        # surfaces in the consumer
                q.put(e)

        threading.Thread(target=run, daemon=True).start()
        while True:
            x = q.get()
            if x is end:
                return
            if isinstance(x, BaseException):
                raise x
            yield x

    def _to_batch(self, bd):
        # (non_blocking: the big input arrays of a training batch live in page-locked ring buffers -- an asynchronous DMA, stream-ordered
        # ahead of the step's kernels; for pageable arrays the flag changes nothing)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).cuda(non_blocking=True)
        b = Batch(labels=t(bd.labels, torch.int32), labels_len=t(bd.labels_length, torch.int32))
        if isinstance(bd.inputs, tuple):
            b.video, b.audio = t(bd.inputs[0], torch.float32), t(bd.inputs[1], torch.float32)
            b.video_len, b.audio_len = t(bd.inputs_length[0], torch.int32), t(bd.inputs_length[1], torch.int32)
            names = bd.inputs_filenames[0]
        elif self._video_processing is not None:
            b.video, b.video_len, names = t(bd.inputs, torch.float32), t(bd.inputs_length, torch.int32), bd.inputs_filenames
        else:
            b.audio, b.audio_len, names = t(bd.inputs, torch.float32), t(bd.inputs_length, torch.int32), bd.inputs_filenames
        if 'aus' in bd.payload:
            b.aus = t(bd.payload['aus'], torch.float32)
        elif self._cfg.regress_aus and b.video is not None:
            raise ValueError("regress_aus=True needs Action Units in the video record")
        return b, names

    # ------------------------------------------------------------------------------------------------
    def save(self, checkpoint_path):
        m = self._model
        blob = {"step": np.array(int(m.step.item()), np.int64)}
        for which in ("params", "adam_m", "adam_v"):
            for k, v in m.export_tf_weights(which).items():
                blob[which + ":" + k] = v
        makedirs(path.dirname(checkpoint_path), exist_ok=True)
        np.savez(checkpoint_path + ".npz", **blob)
        return checkpoint_path

    def restore(self, checkpoint_path):
        z = np.load(checkpoint_path if checkpoint_path.endswith(".npz") else checkpoint_path + ".npz")
        m = self._model
        m.load_tf_weights({k[7:]: z[k] for k in z.files if k.startswith("params:")})
        for which, buf in (("adam_m", m.adam_m), ("adam_v", m.adam_v)):
            W = {k[len(which) + 1:]: z[k] for k in z.files if k.startswith(which + ":")}
            if W:
                m.load_flat(buf, W)
        m.step.fill_(int(z["step"]))

    @staticmethod
    def latest_checkpoint(checkpoint_dir):
        files = glob.glob(path.join(checkpoint_dir, "checkpoint.ckp-*.npz"))
        if not files:
            return None
        best = max(files, key=lambda f: int(f[:-4].split('-')[-1]))
        return best[:-4]

    # ------------------------------------------------------------------------------------------------
    def train(self, logfile, num_epochs=400, try_restore_latest_checkpoint=False):
        checkpoint_dir = path.join('checkpoints', path.split(logfile)[-1])
        checkpoint_path = path.join(checkpoint_dir, 'checkpoint.ckp')
        makedirs(checkpoint_dir, exist_ok=True)
        if path.dirname(logfile):
            makedirs(path.dirname(logfile), exist_ok=True)
        last_epoch = 0
        if try_restore_latest_checkpoint is True:
            try:
                latest_ckp = self.latest_checkpoint(checkpoint_dir)
                last_epoch = int(latest_ckp.split('-')[-1])
                self.restore(latest_ckp)
                print('Restoring checkpoint from epoch {}\n'.format(last_epoch))
            except Exception:
                print('Could not restore from checkpoint, training from scratch!\n')
        lead = self._rank == 0                                  # side-effect files (log, checkpoints, predictions) are written by rank 0
        f = open(logfile if lead else os.devnull, 'a')
        for current_epoch in range(1, num_epochs):            # num_epochs - 1 epochs actually run (avsr.py:253)
            epoch = last_epoch + current_epoch
            self._epoch_counter = epoch
            sum_loss, batches = 0.0, 0
            start = time.time()
            it = self._iterator('train')
            if hasattr(it, "reuse_buffers") and os.environ.get("AVSR_IO_PREFETCH", "1") != "0":
                it.reuse_buffers = True                       # each batch is consumed (copied to the device) before the one after next is asked for
                it = self._prefetched(it, depth=2)            # (+ 2 batches held by the loop below: RING >= 5)
            # The NEXT batch's host-to-device copy (85 MB of lip crops + audio at the benchmark shape: ~1.5 ms of PCIe) runs on a copy
            # stream while the current step computes: it is started before the step is launched (the trainer waits for the GPU inside
            # train_step when it checks the persistent kernels' flag).
            it = iter(it)
            copy_stream = torch.cuda.Stream()

            def upload(bd):
                if bd is None:
                    return None
                with torch.cuda.stream(copy_stream):
                    batch, _names = self._to_batch(bd)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                return batch, ev

            nxt = upload(next(it, None))
            while nxt is not None:                            # end of data = StopIteration (reference: OutOfRangeError)
                batch, ev = nxt
                cur = torch.cuda.current_stream()
                cur.wait_event(ev)
                for t_ in vars(batch).values():               # allocated on the copy stream, consumed on this one
                    if torch.is_tensor(t_):
                        t_.record_stream(cur)
                nxt = upload(next(it, None))                  # (the prefetch thread is normally ahead: no wait here)
                if self._profiling:
                    ops.prof_begin(1 << 16)
                loss, gnorm = self._trainer.train_step(batch)
                if self._profiling:
                    self._write_timeline(ops.prof_end(), epoch, batches)
                batch_loss, global_norm = float(loss.item()), float(gnorm.item())
                sum_loss += batch_loss
                if lead:
                    print('batch: {}, batch loss: {:.2f}, gradient norm: {:.2f}'.format(batches, batch_loss, global_norm))
                batches += 1
            if lead:
                print('epoch time: {}'.format(time.time() - start))
            f.write('Average batch_loss as epoch {} is {}\n'.format(epoch, sum_loss / max(1, batches)))
            f.flush()
            if epoch % 10 == 0:
                save_path = checkpoint_path + '-{}'.format(epoch)
                if lead:
                    self.save(save_path)
                if self._dist is not None:
                    self._dist.barrier()                          # the other ranks restore the file rank 0 has just written
                error_rate = self.evaluate(save_path, epoch)
                for (k, v) in error_rate.items():
                    f.write(k + ': {:.4f}% '.format(v * 100))
                f.write('\n')
                f.flush()
        f.close()

    @staticmethod
    def _write_timeline(prof, epoch, batch):
        """`profiling=True` (avsr/avsr.py:542-550, :274-290 drive tf.profiler with FULL_TRACE and write timelines to /tmp/timelines/):
        here every engine launch of the step is bracketed by a HIP event pair (avsr_prof_begin / avsr_prof_end) and the per-kernel-
        class launch counts, summed milliseconds and algorithmic FLOPs of the step go to /tmp/timelines/timeline_<epoch>_<batch>.json.
        For a kernel-by-kernel trace run the same script under `rocprofv3 --kernel-trace --stats`."""
        import json
        makedirs('/tmp/timelines/', exist_ok=True)
        rows = {k: {"launches": c, "ms": round(ms, 4), "algorithmic_gflop": round(fl / 1e9, 3)} for k, (c, ms, fl) in prof.items() if c}
        with open('/tmp/timelines/timeline_{}_{}.json'.format(epoch, batch), 'w') as f:
            json.dump({"epoch": epoch, "batch": batch, "kernel_classes": rows, "total_ms": round(sum(r["ms"] for r in rows.values()), 4)}, f, indent=1)

    def _write_alignments(self, names, alignments_outdir):
        """`<file>.png` (unimodal / av_align decoder), `<file>_video.png` + `<file>_audio.png` (bimodal), `<file>_av.png`
        (AV-Align cross-modal), one greyscale image [T_memory x T_decoder] per utterance (avsr/avsr.py:404-436)."""
        al = self._model.attention_alignments()
        dec = [a.cpu().numpy() for a in al["decoder"]]
        enc = None if al["encoder"] is None else al["encoder"].cpu().numpy()
        arch = self._cfg.architecture
        for idx in range(len(names)):
            base = path.join(alignments_outdir, names[idx].decode('utf-8'))
            makedirs(path.dirname(base), exist_ok=True)
            if arch == 'bimodal':
                write_png_gray(base + '_video.png', alignment_image(dec[0][idx]))
                write_png_gray(base + '_audio.png', alignment_image(dec[-1][idx]))
            else:
                write_png_gray(base + '.png', alignment_image(dec[0][idx]))
                if arch == 'av_align':
                    write_png_gray(base + '_av.png', alignment_image(enc[idx]))

    def evaluate(self, checkpoint_path, epoch=None, alignments_outdir='./alignments/tmp/', beam_graphs_outdir='./beam_graphs/tmp/'):
        self.restore(checkpoint_path)                         # the path argument is honoured, as in avsr.py:328-331
        predictions_dict, labels_dict = {}, {}
        it = self._iterator('evaluate')
        if hasattr(it, "reuse_buffers") and os.environ.get("AVSR_IO_PREFETCH", "1") != "0":
            it.reuse_buffers = True                       # a batch is decoded (host-synchronised) before the next one is asked for
            it = self._prefetched(it, depth=2)
        for bd in it:
            batch, names = self._to_batch(bd)
            if self._decoding_algorithm == 'beam_search':     # avsr.py:58-59 default: width 10, first beam returned
                ids = self._model.beam_search_decode(batch, beam_width=self._beam_width, max_steps=self._cfg.max_label_length)
            else:
                ids = self._model.greedy_decode(batch, max_steps=self._cfg.max_label_length)
            ids = ids.cpu().numpy()
            if self._write_attention_alignment:
                self._write_alignments(names, alignments_outdir)
            for idx in range(len(names)):
                file = names[idx].decode('utf-8')
                predictions_dict[file] = [self._unit_dict[int(s)] for s in ids[idx]]
                labels_dict[file] = [self._unit_dict[int(s)] for s in bd.labels[idx]]
        uer, uer_dict = compute_wer(predictions_dict, labels_dict)
        error_rate = {self._unit: uer}
        if self._unit == 'character':
            wer, _wer_dict = compute_wer(predictions_dict, labels_dict, split_words=True)
            error_rate['word'] = wer
        if self._rank == 0:
            outdir = path.join('predictions', path.split(path.split(checkpoint_path)[0])[-1])
            makedirs(outdir, exist_ok=True)
            write_sequences_to_labelfile(predictions_dict, path.join(outdir, 'predicted_epoch_{}.mlf'.format(epoch)), labels_dict,
                                         uer_dict, sep=' ' if self._unit == 'phoneme' else '')
        return error_rate
