"""Error-rate metrics and prediction files (same behaviour as avsr/utils.py:4-57 of the reference; own code).

`compute_wer(predictions, ground_truth, split_words)` returns (mean error rate, per-file dict): per file the
Levenshtein distance between the symbol sequences (EOS / END / MASK stripped) divided by the ground-truth
length; with split_words the symbols are joined and split on whitespace first.  Pinned against the real
reference functions by tests/golden/reference_cer_wer.json."""
from os import path

_EXTRA = ("EOS", "END", "MASK")


def _strip_extra_chars(seq):
    return [s for s in seq if s not in _EXTRA]


def levenshtein(ground_truth, prediction):
    """Edit distance (unit costs) with a single rolling row over the shorter sequence."""
    a, b = list(ground_truth), list(prediction)
    if len(a) > len(b):
        a, b = b, a
    row = list(range(len(a) + 1))
    for i, cb in enumerate(b, start=1):
        diag, row[0] = row[0], i
        for j, ca in enumerate(a, start=1):
            cost = diag + (ca != cb)
            diag = row[j]
            row[j] = min(row[j] + 1, row[j - 1] + 1, cost)
    return row[len(a)]


def compute_wer(predictions_dict, ground_truth_dict, split_words=False):
    total, per_file = 0.0, {}
    for fname, pred in predictions_dict.items():
        hyp = _strip_extra_chars(pred)
        ref = _strip_extra_chars(ground_truth_dict[fname])
        if split_words:
            hyp, ref = "".join(hyp).split(), "".join(ref).split()
        err = levenshtein(ref, hyp) / float(len(ref))
        per_file[fname] = err
        total += err
    return total / (float(len(predictions_dict)) or 1), per_file


def write_sequences_to_labelfile(sequence_dict, fname, original_dict, error_dict, sep=""):
    """`<file> <prediction> [<truth>] [<error>]` per line (avsr/utils.py:49-57)."""
    with open(fname, "w") as f:
        for key, seq in sequence_dict.items():
            hyp = sep.join(_strip_extra_chars(seq))
            ref = sep.join(_strip_extra_chars(original_dict[key]))
            f.write("%s %s [%s] [%.3f]\n" % (key, hyp, ref, error_dict[key]))


def get_files(file_list, dataset_dir, remove_sa=True, shuffle_sentences=False):
    with open(file_list, "r") as f:
        names = [path.join(dataset_dir, line.split()[0]) for line in f.read().splitlines()]
    if remove_sa:
        names = [n for n in names if "/sa" not in n]
    if shuffle_sentences:
        from random import shuffle
        shuffle(names)
    return names


# ------------------------------------------------------------------------------------------------
# attention-alignment images (write_attention_alignment=True: avsr/avsr.py:354-366, :404-436)
def alignment_image(alpha):
    """uint8 [T_mem, T_dec] image of one utterance's alignment alpha [T_dec, T_mem], as tf.summary.image renders the
    reference's `1 - transpose(alignment)` float tensor (decoder_unimodal.py:283-288): per image, non-negative values are
    scaled so the largest becomes 255 (0 if it is below 1e-6) and truncated to uint8."""
    import numpy as np
    img = 1.0 - np.asarray(alpha, dtype=np.float32).T
    top = float(img.max()) if img.size else 0.0
    scale = 0.0 if top < 1e-6 else 255.0 / top
    return (img * scale).astype(np.uint8)


def write_png_gray(fname, img):
    """Minimal 8-bit greyscale PNG writer (signature, IHDR, one IDAT of filter-0 scanlines, IEND)."""
    import struct
    import zlib
    import numpy as np
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)
    raw = b"".join(b"\x00" + img[r].tobytes() for r in range(h))
    with open(fname, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
