"""The oracle's beam-search bookkeeping against TensorFlow's own output: the one BeamSearchDecoder trace the reference tree holds
(avsr/visualise/00025.html -> tests/golden/reference_beam_trace_00025.json, see tests/beam_trace.py).  CPU only.
The same replay runs through the HIP beam step in tests/test_gpu_beam.py::test_hip_beam_step_replays_the_reference_trace."""
import numpy as np
import pytest
import torch

import beam_trace as bt
from oracle import avsr_oracle as O

SCORE_TOL = 2.5e-3          # the trace prints 3 decimals: half a unit on the kept score, half on its parent's, x the penalty ratio, + normalisation


def test_trace_is_a_beam_search_decoder_output():
    tr = bt.load()
    K = tr["beam_width"]
    assert K == 10 and len(tr["steps"]) == 19                     # avsr.py:59 beam_width=10
    fin = [False] * K
    for t, st in enumerate(tr["steps"]):
        assert len(st["ids"]) == len(st["parents"]) == K
        assert all(a >= b for a, b in zip(st["score"], st["score"][1:])), "tf.nn.top_k returns the kept beams best first"
        if t == 0:
            assert st["parents"] == [0] * K                       # only beam 0 starts with log-probability 0 (the others -inf)
            assert len(set(st["ids"])) == K
        assert len({(p, i) for p, i in zip(st["parents"], st["ids"])}) == K
        for name, p in zip(st["names"], st["parents"]):
            assert not fin[p] or name == "EOS"                    # _mask_probs: a finished beam's only continuation is EOS
        assert not all(fin), "dynamic_decode stops as soon as every beam has finished"
        fin = [fin[p] or name == "EOS" for name, p in zip(st["names"], st["parents"])]
    assert all(fin)
    best = "".join(n for n in _backtrack(tr, 0) if n != "EOS")
    assert best == "and the next day"                             # beam 0 reads as text (the spoken sentence was tr["transcript"])


def _backtrack(tr, beam):
    out = []
    for st in reversed(tr["steps"]):
        out.append(st["names"][beam])
        beam = st["parents"][beam]
    return out[::-1]


def test_trace_pins_the_length_penalty():
    """What the 3-decimal scores say about `_get_scores`: every beam's score moves ONCE after its EOS, by the factor
    penalty(L) / penalty(L + 1) -- the EOS step is scored with the length before it, later steps with one more -- and that factor
    fixes the exponent to 0.6 +- 0.02 (decoder_unimodal.py:261), excluding the bimodal decoder's 0.5 and an un-normalised score."""
    tr = bt.load()
    rec = bt.reconstruct(tr)
    lo, hi, n = 0.0, 10.0, 0
    for t, st in enumerate(tr["steps"][:-1]):
        nxt = tr["steps"][t + 1]
        for k, name in enumerate(st["names"]):
            if name != "EOS" or rec[t][k]["step_lp"] is None:
                continue                                          # first EOS of a beam only
            kids = [j for j, p in enumerate(nxt["parents"]) if p == k]
            if not kids:
                continue
            L = rec[t][k]["used_len"]
            s_e, s_c = -st["score"][k], -nxt["score"][kids[0]]
            assert s_e - s_c > 4e-3, "the score of a finished beam changes at the step after its EOS"
            r_lo, r_hi = (s_e - 5e-4) / (s_c + 5e-4), (s_e + 5e-4) / (s_c - 5e-4)
            base = np.log((5.0 + L + 1) / (5.0 + L))
            lo, hi, n = max(lo, np.log(r_lo) / base), min(hi, np.log(r_hi) / base), n + 1
            if t + 2 < len(tr["steps"]):                          # ... and never again
                g = [j for j, p in enumerate(tr["steps"][t + 2]["parents"]) if p == kids[0]]
                assert g and abs(tr["steps"][t + 2]["score"][g[0]] - nxt["score"][kids[0]]) < 1e-9
    assert n >= 4 and lo <= 0.6 <= hi and hi - lo < 0.05 and not lo <= 0.5 <= hi, (lo, hi, n)
    # with that exponent no kept continuation has a positive log-probability beyond the rounding of the printed scores
    lps = [c["step_lp"] for cur in rec for c in cur if c["step_lp"] is not None]
    assert len(lps) > 150 and max(lps) < SCORE_TOL
    # ... which an exponent of 0.5 would violate along the best beam (its un-normalised log-probability would RISE by 0.02 over 13 steps)
    rec5 = bt.reconstruct(tr, w=0.5)
    assert rec5[12][0]["total"] - rec5[0][0]["total"] > 0.015


def _replay(tr, candidates, advance):
    """Feeds the trace's step tables to a (candidates, advance) pair; returns (steps with wrong ids / parents, worst score deviation,
    step at which all beams were finished)."""
    K, V, eos = tr["beam_width"], tr["V"], tr["eos"]
    rec = bt.reconstruct(tr)
    logp = torch.full((1, K), -float("inf"), dtype=torch.float64)
    logp[0, 0] = 0.0
    fin, length = torch.zeros(1, K, dtype=torch.bool), torch.zeros(1, K, dtype=torch.int64)
    wrong, dev, done = [], 0.0, None
    for t, st in enumerate(tr["steps"]):
        step_lp = torch.log_softmax(torch.as_tensor(bt.logits_for_step(tr, rec, t)), dim=-1)[None]
        total, scores = candidates(logp, fin, length, step_lp, bt.W, eos)
        order = torch.argsort(scores, dim=1, descending=True, stable=True)[:, :K]
        if (order[0] % V).tolist() != st["ids"] or (order[0] // V).tolist() != st["parents"]:
            wrong.append(t)
            order = torch.as_tensor([[p * V + i for p, i in zip(st["parents"], st["ids"])]])      # carry on along the trace
        dev = max(dev, float((torch.gather(scores, 1, order)[0] - torch.as_tensor(st["score"])).abs().max()))
        logp, fin, length = advance(total, fin, length, order, V, eos)
        if done is None and bool(fin.all()):
            done = t
    return wrong, dev, done


def test_oracle_bookkeeping_replays_the_reference_trace():
    tr = bt.load()
    wrong, dev, done = _replay(tr, O.beam_candidates, O.beam_advance)
    assert wrong == [] and dev < SCORE_TOL and done == len(tr["steps"]) - 1, (wrong, dev, done)


@pytest.mark.parametrize("variant", ["eos_counts_towards_the_length", "length_frozen_at_the_eos_step", "finished_beams_follow_the_logits"])
def test_the_replay_tells_other_bookkeepings_apart(variant):
    """Negative controls: three plausible mis-recollections of `_beam_search_step` do NOT reproduce the trace."""
    tr = bt.load()

    def candidates(logp, fin, length, step_lp, w, eos):
        if variant == "eos_counts_towards_the_length":
            return _cand_eos_counted(logp, fin, length, step_lp, w, eos)
        if variant == "finished_beams_follow_the_logits":
            return O.beam_candidates(logp, torch.zeros_like(fin), length, step_lp, w, eos)
        return O.beam_candidates(logp, fin, length, step_lp, w, eos)

    def advance(total, fin, length, order, V, eos):
        logp, nfin, nlen = O.beam_advance(total, fin, length, order, V, eos)
        if variant == "length_frozen_at_the_eos_step":
            nlen = nlen - (nfin & ~torch.gather(fin, 1, order // V)).to(nlen.dtype)       # the EOS position not counted
        return logp, nfin, nlen

    wrong, dev, _ = _replay(tr, candidates, advance)
    assert wrong or dev > 4 * SCORE_TOL, (variant, wrong, dev)


def _cand_eos_counted(logp, fin, length, step_lp, w, eos):
    V = step_lp.shape[-1]
    fin_row = torch.full((V,), torch.finfo(torch.float32).min, dtype=step_lp.dtype)
    fin_row[eos] = 0.0
    total = logp[:, :, None] + torch.where(fin[:, :, None], fin_row[None, None, :], step_lp)
    new_len = length[:, :, None] + (~fin)[:, :, None].to(torch.int64).expand(-1, -1, V)
    return total, (total / ((5.0 + new_len.to(step_lp.dtype)) / 6.0) ** w).reshape(1, -1)


def test_committed_fixture_is_what_the_generator_extracts_from_the_reference(tmp_path, monkeypatch):
    """Where the reference tree is present (this container, not the GPU box): tests/golden/make_golden.py re-extracts the trace from
    avsr/visualise/00025.html and the result must be the committed fixture, byte for byte."""
    import importlib.util
    import os
    if not os.path.exists("/root/reference/avsr/visualise/00025.html"):
        pytest.skip("reference tree not present")
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    monkeypatch.setattr(mg, "HERE", str(tmp_path))
    mg.reference_beam_trace()
    assert open(tmp_path / "reference_beam_trace_00025.json").read() == open(os.path.join(here, "golden", "reference_beam_trace_00025.json")).read()
