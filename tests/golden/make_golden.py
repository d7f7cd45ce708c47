#!/usr/bin/env python
"""Regenerates the fixtures in this directory.  Run from the repo root: python tests/golden/make_golden.py

1. oracle_restatement_*.npz -- inputs, TF-layout weights and expected outputs (logits, loss, global norm, greedy
   ids, a few gradients) produced BY THE CPU ORACLE (oracle/avsr_oracle.py, fp64).  They pin the oracle against
   silent drift and give the GPU tests fixed vectors; they are NOT TensorFlow outputs ("TF parity unpinned").
2. reference_cer_wer.json -- outputs of the REAL reference functions avsr/utils.py:compute_wer / levenshtein,
   loaded by file path from /root/reference (importing the `avsr` package itself needs TensorFlow).  This is the
   one piece of the reference that can execute here; it pins our CER/WER implementation (avsr_tf1_amd/utils.py).
3. reference_beam_trace_00025.json -- the search tree of the reference's sample beam-search visualisation
   avsr/visualise/00025.html, i.e. `BeamSearchDecoderOutput.scores / predicted_ids / parent_ids` of one utterance as TensorFlow's own
   BeamSearchDecoder produced them (written by avsr/avsr.py:472-487 through avsr/visualise/beam_search.py:create_html), flattened
   to one list per decode step.  The only TensorFlow OUTPUT in the reference tree; it pins the beam bookkeeping (tests/test_beam_trace.py).
"""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CASES = {
    "audio_uni_luong": dict(architecture="unimodal", video_units=None, audio_units=(16, 16), attention_type=(("scaled_luong",), ("scaled_luong",))),
    "audio_bi_bahdanau": dict(architecture="unimodal", encoder_type="bidirectional", video_units=None, audio_units=(16,),
                              attention_type=(("bahdanau",), ("bahdanau",))),
    "bimodal_uni": dict(architecture="bimodal", video_units=(16,), audio_units=(16, 16), regress_aus=True),
    "av_align": dict(architecture="av_align", video_units=(16,), audio_units=(16, 16), regress_aus=True),
    "video_bi_normed": dict(architecture="unimodal", encoder_type="bidirectional", video_units=(16,), audio_units=None,
                            attention_type=(("normed_bahdanau",), ("normed_bahdanau",))),
    # options added later in round 1: multi-layer decoder, residual + instance norm + Dense inputs, weight sharing, focal loss
    "bimodal_dec2": dict(architecture="bimodal", video_units=(16,), audio_units=(16, 16), decoder_units=(16, 16)),
    "audio_residual_instnorm_dense": dict(architecture="unimodal", video_units=None, audio_units=(16, 16, 16), residual_encoder=True,
                                          instance_normalisation=True, input_dense_layers=(16,)),
    "audio_shared_focal": dict(architecture="unimodal", video_units=None, audio_units=(16, 16, 16, 16), encoder_weight_sharing=True,
                               loss_fun="focal_loss", optimiser="Nadam"),
}
COMMON = dict(decoder_units=(16,), embedding_size=8, video_feat=8, audio_feat=12)


def oracle_fixtures():
    from oracle import avsr_oracle as O
    for name, kw in CASES.items():
        cfg = O.OracleConfig(**dict(COMMON, **kw))
        W = O.init_params(cfg, seed=2001)
        rng = np.random.default_rng(11)
        for k in W:
            if k.endswith(("bias", "/b", "beta")):
                W[k] = (rng.standard_normal(W[k].shape) * 0.1).astype(np.float32)
        batch = O.synthetic_batch(cfg, B=3, T_a=11, T_v=6, L=5, ragged=True)
        r = O.train_step(W, None, cfg, batch)
        ids = O.greedy_decode(W, cfg, batch, max_steps=8)
        out = {"cfg_json": np.array(json.dumps(dict(COMMON, **kw)))}
        for k, v in W.items():
            out["w:" + k] = v
        for k in ("audio", "audio_len", "video", "video_len", "aus", "labels", "labels_len"):
            v = getattr(batch, k)
            if v is not None:
                out["in:" + k] = v
        out["out:logits"] = r["logits"].astype(np.float32)
        out["out:loss"] = np.float64(r["loss"])
        out["out:global_norm"] = np.float64(r["global_norm"])
        out["out:greedy_ids"] = ids
        out["out:grad:dec/out/kernel"] = r["grads"]["dec/out/kernel"].astype(np.float32)
        out["out:grad:dec/l0/kernel"] = r["grads"]["dec/l0/kernel"].astype(np.float32)
        np.savez_compressed(os.path.join(HERE, "oracle_restatement_%s.npz" % name), **out)
        print(name, "loss %.6f" % r["loss"], "ids", ids.shape)


def reference_cer():
    path = "/root/reference/avsr/utils.py"
    if not os.path.exists(path):
        print("reference not present; keeping the committed reference_cer_wer.json")
        return
    spec = importlib.util.spec_from_file_location("ref_utils", path)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(3)
    alphabet = list("abcdefghij ") 
    cases = []
    for i in range(12):
        n = int(rng.integers(1, 14))
        truth = [alphabet[j] for j in rng.integers(0, len(alphabet), n)]
        pred = list(truth)
        for _ in range(int(rng.integers(0, 5))):
            op = int(rng.integers(0, 3))
            pos = int(rng.integers(0, max(1, len(pred))))
            if op == 0 and pred:
                pred[pos % len(pred)] = alphabet[int(rng.integers(0, len(alphabet)))]
            elif op == 1:
                pred.insert(pos, alphabet[int(rng.integers(0, len(alphabet)))])
            elif pred:
                pred.pop(pos % len(pred))
        cases.append((truth + ["EOS"], pred + ["EOS", "MASK"]))
    lev = [{"a": "".join(t), "b": "".join(p), "d": ref.levenshtein(t, p)} for t, p in
           [("kitten", "sitting"), ("", "abc"), ("flaw", "lawn"), ("same", "same")]]
    preds = {"f%d" % i: p for i, (t, p) in enumerate(cases)}
    truth = {"f%d" % i: t for i, (t, p) in enumerate(cases)}
    cer, cer_d = ref.compute_wer(preds, truth, split_words=False)
    wer, wer_d = ref.compute_wer(preds, truth, split_words=True)
    with open(os.path.join(HERE, "reference_cer_wer.json"), "w") as f:
        json.dump({"source": "georgesterpu/avsr-tf1 avsr/utils.py compute_wer/levenshtein executed from /root/reference",
                   "levenshtein": lev, "predictions": preds, "truth": truth, "cer": cer, "cer_per_file": cer_d,
                   "wer": wer, "wer_per_file": wer_d}, f, indent=1)
    print("reference CER %.6f WER %.6f" % (cer, wer))


def reference_beam_trace():
    import re
    path = "/root/reference/avsr/visualise/00025.html"
    if not os.path.exists(path):
        print("reference not present; keeping the committed reference_beam_trace_00025.json")
        return
    html = open(path).read()
    tree = json.loads(re.search(r"var treeData = (\{.*\});\s*\n", html).group(1))
    transcript = re.search(r'const transcript = "(.*)";', html).group(1)
    levels = {}

    def walk(node, parent):                        # node id = [decode step (1-based), beam]; the root is START
        lv, beam = node["id"]
        if lv > 0:
            levels.setdefault(lv, {})[beam] = (node["name"], node["score"], parent)
        for ch in node.get("children", []):
            walk(ch, beam)

    walk(tree, None)
    K = len(levels[1])
    steps = []
    for lv in sorted(levels):
        assert sorted(levels[lv]) == list(range(K)), "every step of a BeamSearchDecoder output has beam_width entries"
        steps.append({"names": [levels[lv][i][0] for i in range(K)], "scores": [levels[lv][i][1] for i in range(K)],
                      "parents": [levels[lv][i][2] for i in range(K)]})
    with open(os.path.join(HERE, "reference_beam_trace_00025.json"), "w") as f:
        json.dump({"source": "georgesterpu/avsr-tf1 avsr/visualise/00025.html: BeamSearchDecoderOutput (scores printed with 3 decimals, "
                             "predicted_ids as characters, parent_ids) of one utterance, one entry per decode step",
                   "transcript": transcript, "beam_width": K, "steps": steps}, f, indent=0)
    print("reference beam trace: %d steps x %d beams" % (len(steps), K))


if __name__ == "__main__":
    oracle_fixtures()
    reference_cer()
    reference_beam_trace()
