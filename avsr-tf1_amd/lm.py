"""`avsr.LM` on the HIP engine: the character / phoneme / viseme language model of the reference (avsr/lm.py:15-272 host
class, :275-471 model) -- embedding, a recurrent stack started from the zero state, Dense(V), trained with the same masked
sequence loss, L2 on the recurrent kernels, global-norm clipping and Adam (constant learning rate, no warm-up) as the
recogniser's decoder, with the train graph's DropoutWrapper and scheduled sampling.

It is the recogniser's decoder block with no attention memory (`ModelConfig(architecture='lm')`), so every kernel on its
path is one the AVSR hot path already exercises.  As in the reference there are two engines -- a train one and an evaluate
one (no dropout, TrainingHelper) -- and weights travel between them through the checkpoint file `evaluate()` is given.
Checkpoints: `checkpoints/<logfile name>/checkpoint.ckp-<epoch>.npz`, one per epoch, the newest five kept
(`Saver(max_to_keep=5)`, lm.py:403-404)."""
import glob
import os
import time
from os import makedirs, path

import numpy as np
import torch

from .avsr import AVSR
from .config import ModelConfig
from .io_utils import create_unit_dict, make_iterator_from_label_record, make_iterator_from_text_dataset
from .model import Batch, Seq2SeqModel
from .parallel import DataParallelTrainer


class LM(object):
    def __init__(self,
                 unit,
                 unit_file=None,
                 labels_train_record=None,
                 labels_test_record=None,
                 text_dataset=None,
                 batch_size=(64, 64),
                 cell_type='lstm',
                 recurrent_l2_regularisation=0.0001,
                 decoder_units_per_layer=(256,),
                 use_dropout=True,
                 decoder_dropout_probability=(0.9, 0.9, 0.9),
                 embedding_size=128,
                 sampling_probability_outputs=0.1,
                 optimiser='Adam',
                 learning_rate=0.001,
                 clip_gradients=True,
                 max_gradient_norm=1.0,
                 precision='float32',
                 required_grahps=('train', 'eval'),
                 **kwargs):
        self._unit = unit
        self._unit_dict = create_unit_dict(unit_file=unit_file)
        self._labels_train_record, self._labels_test_record = labels_train_record, labels_test_record
        self._text_dataset = text_dataset
        self._batch_size = batch_size
        self._required_graphs = required_grahps
        if optimiser not in ('Adam', 'AdamW', 'Momentum'):
            raise Exception('Unsupported optimiser, try Adam')                                     # lm.py:453
        if precision != 'float32':
            raise NotImplementedError("precision=%r: the HIP engine computes in float32" % (precision,))
        if cell_type not in ('lstm', 'gru'):
            raise Exception('cell type not supported: {}'.format(cell_type))                      # cells.py:44
        reverse = {v: k for k, v in self._unit_dict.items()}
        common = dict(architecture='lm', video_units=None, audio_units=None, cell_type=cell_type,
                      decoder_units=tuple(decoder_units_per_layer), embedding_size=embedding_size,
                      vocab_size=len(self._unit_dict) - 1, go_id=reverse['GO'], eos_id=reverse['EOS'],
                      recurrent_l2=None if optimiser == 'AdamW' else recurrent_l2_regularisation,   # lm.py:56
                      optimiser=optimiser, weight_decay=kwargs.get('weight_decay', 0.0001), clip_gradients=clip_gradients, max_gradient_norm=max_gradient_norm,
                      learning_rate=learning_rate, warmup_steps=0,                                # lm.py:408: constant learning rate
                      max_label_length={'viseme': 65, 'phoneme': 70, 'character': 100}[unit],
                      decoder_dropout=tuple(decoder_dropout_probability))
        self._shuffle_seed = kwargs.get('shuffle_seed')
        self._model = self._eval_model = None
        if 'train' in required_grahps:
            self._cfg = ModelConfig(use_dropout=use_dropout, sampling_probability=sampling_probability_outputs, **common)
            self._model = Seq2SeqModel(self._cfg, seed=kwargs.get('seed', 0))
            self._trainer = DataParallelTrainer(self._model, None, use_graph=False, check_every_step=True)
        if 'eval' in required_grahps:                                                             # lm.py:371-375: TrainingHelper, mode != 'train'
            self._eval_cfg = ModelConfig(use_dropout=False, sampling_probability=0.0, **common)
            self._eval_model = Seq2SeqModel(self._eval_cfg, seed=kwargs.get('seed', 0))

    # the checkpoint format is the recogniser's (weights + Adam moments + step, TF variable layout)
    save, restore, latest_checkpoint = AVSR.save, AVSR.restore, staticmethod(AVSR.latest_checkpoint)

    def _iterator(self, mode):
        bs = self._batch_size[0 if mode == 'train' else 1]
        if self._text_dataset is not None:
            return make_iterator_from_text_dataset(self._text_dataset, bs, self._unit_dict, shuffle=mode == 'train', bucket_width=30,
                                                   seed=self._shuffle_seed)
        rec = self._labels_train_record if mode == 'train' else self._labels_test_record
        return make_iterator_from_label_record(rec, bs, self._unit_dict, shuffle=mode == 'train', reverse_input=False, bucket_width=30,
                                               seed=self._shuffle_seed)

    @staticmethod
    def _to_batch(bd):
        if (bd.labels < 0).any():
            raise ValueError("the text contains a symbol that is not in the unit list")
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32).cuda()
        return Batch(labels=t(bd.labels), labels_len=t(bd.labels_length))

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
        f = open(logfile, 'a')
        for current_epoch in range(1, num_epochs):
            epoch = last_epoch + current_epoch
            sum_loss, batches = 0.0, 0
            start = time.time()
            for bd in self._iterator('train'):
                loss, _gnorm = self._trainer.train_step(self._to_batch(bd))
                batch_loss = float(loss.item())
                sum_loss += batch_loss
                print('batch: {}, batch loss: {}'.format(batches, batch_loss))
                batches += 1
            print('epoch time: {}'.format(time.time() - start))
            f.write('Average batch_loss as epoch {} is {}\n'.format(epoch, sum_loss / max(1, batches)))
            f.flush()
            self.save(checkpoint_path + '-{}'.format(epoch))                                      # every epoch (lm.py:234)
            kept = sorted(glob.glob(checkpoint_path + '-*.npz'), key=lambda p: int(p[:-4].split('-')[-1]))
            for old in kept[:-5]:                                                                 # Saver(max_to_keep=5)
                os.remove(old)
        f.close()

    def evaluate(self, checkpoint_path, epoch=None):
        """Writes `predictions/<name>/predicted_epoch_<epoch>.mlf`: `<label file name> <average step loss>` per sentence."""
        train_model, self._model = self._model, self._eval_model                                  # restore INTO the evaluate engine
        try:
            self.restore(checkpoint_path)
        finally:
            self._model = train_model
        likelihoods_dict = {}
        n = 0
        for bd in self._iterator('evaluate'):
            vals = self._eval_model.sequence_likelihoods(self._to_batch(bd)).cpu().numpy()
            for element in range(len(vals)):
                name = bd.labels_filenames[element] if bd.labels_filenames is not None else str(n).encode()
                likelihoods_dict[name.decode('utf-8')] = vals[element]
                n += 1
        outdir = path.join('predictions', path.split(path.split(checkpoint_path)[0])[-1])
        makedirs(outdir, exist_ok=True)
        with open(path.join(outdir, 'predicted_epoch_{}.mlf'.format(epoch)), 'w') as f:
            f.write(''.join(['{} {}\n'.format(k, v) for (k, v) in likelihoods_dict.items()]))
        return likelihoods_dict
