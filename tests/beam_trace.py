"""The reference's own beam-search trace as a replayable input (test aid for tests/test_beam_trace.py and tests/test_gpu_beam.py).

tests/golden/reference_beam_trace_00025.json holds what TensorFlow's BeamSearchDecoder produced for one utterance
(`BeamSearchDecoderOutput.scores / predicted_ids / parent_ids`, avsr/avsr.py:472-487 -> avsr/visualise/00025.html): 19 decode steps of
10 beams, scores with three decimals.  The model that produced it is gone, but the trace determines the step log-probabilities of the
190 continuations it kept: total(child) - total(parent), with total = score x penalty(length).  `logits_for_step` turns them into a
[K, V] table whose log_softmax reproduces those values and keeps every other continuation below the step's K-th score, so that a
beam-search step fed with the table must select TensorFlow's ids and parents in TensorFlow's order and arrive at TensorFlow's scores --
including what the trace does NOT spell out and the bookkeeping has to get right: the continuation of finished beams (EOS at
log-probability 0 whatever the logits say), their scores (divided by a penalty one position longer from the step after the EOS on) and the end of
the search (all beams finished after step 19).
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
W = 0.6                     # decoder_unimodal.py:261 length_penalty_weight (the trace pins it: see test_trace_pins_the_length_penalty)
EOS_NAME = "EOS"


def load():
    with open(os.path.join(HERE, "golden", "reference_beam_trace_00025.json")) as f:
        tr = json.load(f)
    # ids as the reference's character dictionary numbers them (avsr/misc/character_list after the MASK entry; config.py eos_id / go_id):
    # 0 = MASK, 1 = ' ', 2 = "'", 3..28 = a..z, 29 = EOS, 30 = GO -- vocab_size 31
    vocab = {" ": 1, "'": 2, EOS_NAME: 29}
    vocab.update({chr(ord("a") + i): 3 + i for i in range(26)})
    tr["vocab"], tr["V"], tr["eos"] = vocab, 31, 29
    for st in tr["steps"]:
        st["ids"] = [vocab[n] for n in st["names"]]
        st["score"] = [float(s) for s in st["scores"]]
    return tr


def penalty(length, w=W):
    return ((5.0 + length) / 6.0) ** w


def reconstruct(tr, w=W):
    """Per step: the state TensorFlow's bookkeeping assigns to every kept beam -- total log-probability, length, finished -- and the
    step log-probability of the continuation that produced it (None for the EOS continuation of an already finished beam, whose
    value is the bookkeeping's business, not the model's)."""
    K = tr["beam_width"]
    prev = [dict(total=0.0 if k == 0 else -np.inf, length=0, fin=False) for k in range(K)]
    out = []
    for st in tr["steps"]:
        cur = []
        for name, score, parent in zip(st["names"], st["score"], st["parents"]):
            P = prev[parent]
            if P["fin"]:
                assert name == EOS_NAME, "a finished beam continues with EOS only"
                cur.append(dict(total=P["total"], length=P["length"], fin=True, step_lp=None, used_len=P["length"]))
            else:
                used = P["length"] + (0 if name == EOS_NAME else 1)               # EOS does not count towards the normalising length
                total = score * penalty(used, w)
                cur.append(dict(total=total, length=P["length"] + 1, fin=name == EOS_NAME, step_lp=total - P["total"], used_len=used))
        out.append(cur)
        prev = cur
    return out


def logits_for_step(tr, rec, t, rng=None):
    """[K, V] float64 logits for decode step t (0-based): log_softmax(logits)[parent, id] = the trace's step log-probability for every
    kept continuation of an unfinished beam; the rest of each row's probability mass is spread over its other symbols (all of which
    must score below the step's K-th kept continuation -- asserted).  Rows of finished beams get noise: the step must ignore them."""
    K, V, eos = tr["beam_width"], tr["V"], tr["eos"]
    st, cur = tr["steps"][t], rec[t]
    prev = rec[t - 1] if t else [dict(total=0.0 if k == 0 else -np.inf, length=0, fin=False) for k in range(K)]
    rng = rng or np.random.default_rng(t)
    lg = np.zeros((K, V))
    kth = min(st["score"])
    for k in range(K):
        if prev[k]["fin"]:
            lg[k] = rng.normal(0.0, 3.0, V)
            continue
        kept = {tok: c["step_lp"] for tok, par, c in zip(st["ids"], st["parents"], cur) if par == k}
        mass = sum(np.exp(min(v, 0.0)) for v in kept.values())
        rest = [v for v in range(V) if v not in kept]
        other = np.log(max(1.0 - mass, 1e-9) / len(rest))
        if np.isfinite(prev[k]["total"]):
            # the best the row's other symbols could score: as a non-EOS token, or as EOS (shorter normalising length)
            worst = max((prev[k]["total"] + other) / penalty(prev[k]["length"] + 1), (prev[k]["total"] + other) / penalty(prev[k]["length"]))
            if worst > kth - 0.02:                   # (rows whose kept continuations were rounded to more than the whole mass, or with little kept)
                other = min(other, (kth - 0.05) * penalty(prev[k]["length"]) - prev[k]["total"])
        lg[k, rest] = other
        for tok, v in kept.items():
            lg[k, tok] = v
        lg[k] -= np.log(np.exp(lg[k]).sum())         # exactly normalised: the shift is the rounding of the printed scores (checked by the caller's tolerance)
    return lg
