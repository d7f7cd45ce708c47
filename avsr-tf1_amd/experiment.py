"""`run_experiment` -- the curriculum driver of avsr/experiment.py:5-136: per noise level, train `iterations[i][0]`
epochs at `learning_rates[i][0]`, then `iterations[i][1]` at `learning_rates[i][1]`, each phase a fresh `AVSR`
object resuming from the checkpoint directory; optional warm-up on short sentences.
`run_experiment_mixedsnrs` (avsr/experiment.py:140-211): the same two-phase schedule over ONE pair of audio records that already mixes
the noise levels; the video front-end follows the architecture (`None` for 'unimodal', 'resnet_cnn' otherwise)."""
from os import path

from .avsr import AVSR


def _phase(logfile, lr, epochs, sep, **kw):
    experiment = AVSR(learning_rate=lr, **kw)
    experiment.train(logfile=logfile, num_epochs=epochs, try_restore_latest_checkpoint=True)
    with open(logfile, 'a') as f:
        f.write(sep * '=' + '\n')
    del experiment


def run_experiment(video_train_record=None, video_test_record=None, labels_train_record=None, labels_test_record=None,
                   audio_train_records=None, audio_test_records=None, unit='character',
                   unit_list_file='./avsr/misc/character_list', iterations=None, learning_rates=None,
                   logfile='tmp_experiment', warmup_epochs=0, warmup_max_len=50, input_modality='audio', **kwargs):
    full_logfile = path.join('./logs', logfile)
    common = dict(unit=unit, unit_file=unit_list_file, video_train_record=video_train_record, video_test_record=video_test_record,
                  labels_train_record=labels_train_record, labels_test_record=labels_test_record, **kwargs)
    if warmup_epochs >= 1:
        with open(full_logfile, 'a') as f:
            f.write('Warm up on short sentences up to {} tokens for {} epochs \n'.format(warmup_max_len, warmup_epochs))
        _phase(full_logfile, learning_rates[0][0], warmup_epochs, 5, max_sentence_length=warmup_max_len,
               audio_train_record=audio_train_records[0] if input_modality != 'video' else None,
               audio_test_record=audio_test_records[0] if input_modality != 'video' else None, **common)
    if input_modality == 'video':
        iters, lr = iterations[0], learning_rates[0]
        _phase(full_logfile, lr[0], iters[0] + 1, 5, **common)
        _phase(full_logfile, lr[1], iters[1] + 1, 20, **common)
        return
    for lr, iters, audio_train, audio_test in zip(learning_rates, iterations, audio_train_records, audio_test_records):
        _phase(full_logfile, lr[0], iters[0] + 1, 5, audio_train_record=audio_train, audio_test_record=audio_test, **common)
        _phase(full_logfile, lr[1], iters[1] + 1, 20, audio_train_record=audio_train, audio_test_record=audio_test, **common)


def run_experiment_mixedsnrs(video_train_record=None, video_test_record=None, labels_train_record=None, labels_test_record=None,
                             audio_train_record=None, audio_test_record=None, unit='character',
                             unit_list_file='./avsr/misc/character_list', iterations=None, learning_rates=None,
                             architecture='unimodal', logfile='tmp_experiment', **kwargs):
    full_logfile = path.join('./logs', logfile)
    common = dict(unit=unit, unit_file=unit_list_file, audio_processing='features', audio_train_record=audio_train_record,
                  audio_test_record=audio_test_record, video_processing=None if architecture == 'unimodal' else 'resnet_cnn',
                  video_train_record=video_train_record, video_test_record=video_test_record, labels_train_record=labels_train_record,
                  labels_test_record=labels_test_record, architecture=architecture, **kwargs)
    for lr, iters in zip(learning_rates, iterations):
        _phase(full_logfile, lr[0], iters[0] + 1, 5, **common)
        _phase(full_logfile, lr[1], iters[1] + 1, 20, **common)
