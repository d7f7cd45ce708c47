"""Fused persistent decode kernel (csrc/dec_persist.hip) against the per-step launch path of the same engine, on the GPU:
identical records / losses / gradients / greedy ids, then per-call timing at the c4 benchmark shape.
Usage: python tools/fused_check.py [--full]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avsr_tf1_amd import ops                                      # noqa: E402
from avsr_tf1_amd.config import ModelConfig                        # noqa: E402
from avsr_tf1_amd.model import Batch, Seq2SeqModel                 # noqa: E402


def synth(cfg, B, Ta, Tv, L, seed=0, ragged=True):
    rng = np.random.default_rng(seed)
    b = Batch()
    if cfg.audio_units is not None:
        b.audio = torch.tensor(rng.standard_normal((B, Ta, cfg.audio_feat)), dtype=torch.float32).cuda()
        al = rng.integers(max(1, Ta // 2), Ta + 1, B) if ragged else np.full(B, Ta)
        al[0] = Ta
        b.audio_len = torch.tensor(al, dtype=torch.int32).cuda()
    if cfg.video_units is not None:
        b.video = torch.tensor(rng.standard_normal((B, Tv, cfg.video_feat)), dtype=torch.float32).cuda()
        vl = rng.integers(max(1, Tv // 2), Tv + 1, B) if ragged else np.full(B, Tv)
        vl[0] = Tv
        b.video_len = torch.tensor(vl, dtype=torch.int32).cuda()
        if cfg.regress_aus:
            b.aus = torch.tensor(rng.uniform(0, 3, (B, Tv, 2)), dtype=torch.float32).cuda()
    ll = rng.integers(max(1, L // 2), L + 1, B) if ragged else np.full(B, L)
    ll[0] = L
    lab = rng.integers(0, cfg.vocab_size - 2, (B, L))
    for i in range(B):
        lab[i, ll[i] - 1] = cfg.eos_id
        lab[i, ll[i]:] = 0
    b.labels = torch.tensor(lab, dtype=torch.int32).cuda()
    b.labels_len = torch.tensor(ll, dtype=torch.int32).cuda()
    return b


def run(cfg, batch, fused, steps=2, greedy_steps=12):
    ops.attn_rnn_set_fused(fused)
    torch.manual_seed(0)
    m = Seq2SeqModel(cfg, seed=3)
    out = {}
    for s in range(steps):
        m.forward_train(batch)
        ws = m._cur[0]
        D = ws["dec"]
        out["elig"] = ops.attn_rnn_fused_eligible(D["desc"])
        out[f"logits{s}"] = D["logits"].clone()
        out[f"fed{s}"] = D["fed"].clone()
        out[f"loss{s}"] = m.loss.clone()
        m.backward()
        out[f"grads{s}"] = m.grads.clone()
        m.apply_update()
        out[f"gnorm{s}"] = m.gnorm.clone()
    ids = m.greedy_decode(batch, max_steps=greedy_steps)
    out["ids"] = ids.clone()
    out["glogits"] = m._last_greedy[0]["dec"]["logits"].clone()
    torch.cuda.synchronize()
    out["err"] = ops.rnn_persistent_error()
    return out


def compare(name, a, b):
    worst = 0.0
    for k in a:
        if k in ("elig", "err"):
            continue
        x, y = a[k].float(), b[k].float()
        if x.shape != y.shape:
            print(f"  {name} {k}: SHAPE {tuple(x.shape)} vs {tuple(y.shape)}")
            worst = 1e9
            continue
        d = (x - y).abs().max().item() if x.numel() else 0.0
        rel = d / max(1e-6, y.abs().max().item())
        tag = "" if rel < 2e-4 else "   <-- MISMATCH"
        if k.startswith(("fed", "ids")):
            tag = "" if d == 0 else "   <-- MISMATCH"
        print(f"  {name} {k:10s} max|d| {d:.3e} rel {rel:.2e}{tag}")
        worst = max(worst, rel if not k.startswith(("fed", "ids")) else d)
    return worst


CASES = {
    "bimodal_small": (dict(architecture="bimodal", video_units=(32,), audio_units=(32, 32), decoder_units=(32,), embedding_size=16,
                           video_feat=12, audio_feat=20, regress_aus=True), 5, 19, 8, 6),
    "bimodal_drop_sample": (dict(architecture="bimodal", video_units=(32,), audio_units=(32, 32), decoder_units=(32,), embedding_size=16,
                                 video_feat=12, audio_feat=20, use_dropout=True, sampling_probability=0.3), 11, 37, 9, 7),
    "unimodal_luong": (dict(architecture="unimodal", video_units=None, audio_units=(48,), decoder_units=(48,), embedding_size=32,
                            audio_feat=20, attention_type=(("luong",), ("luong",)), sampling_probability=0.2), 9, 70, 0, 9),
    "av_align": (dict(architecture="av_align", video_units=(32,), audio_units=(32, 32), decoder_units=(32,), embedding_size=16,
                      video_feat=12, audio_feat=20), 6, 23, 9, 5),
    "c4_width": (dict(architecture="bimodal", video_units=(256,), audio_units=(256,), decoder_units=(256,), embedding_size=128,
                      video_feat=128, audio_feat=80, use_dropout=True, sampling_probability=0.1, regress_aus=True), 64, 60, 20, 10),
}


def main():
    full = "--full" in sys.argv
    bad = 0
    for name, (kw, B, Ta, Tv, L) in CASES.items():
        cfg = ModelConfig(**kw)
        batch = synth(cfg, B, Ta, Tv, L)
        b = run(cfg, batch, 0)
        for mode in (1, 2, 3):
            a = run(cfg, batch, mode)
            print(f"{name}: fused mode {mode} (1 both, 2 fwd only, 3 bwd only) eligible={a['elig']} persistent_error={a['err']}")
            w = compare(name, a, b)
            bad += int(w > 2e-4 or not a["elig"] or a["err"])
    print("RESULT", "FAIL" if bad else "OK", bad)
    if full:
        cfg = ModelConfig(architecture="bimodal", video_units=(256,), audio_units=(256, 256, 256), decoder_units=(256,), embedding_size=128,
                          video_feat=128, audio_feat=80, use_dropout=True, sampling_probability=0.1, regress_aus=True)
        batch = synth(cfg, 64, 500, 75, 40, ragged=False)
        for fused in (1, 2, 0):
            ops.attn_rnn_set_fused(fused)
            m = Seq2SeqModel(cfg, seed=3)
            m.forward_train(batch)
            D = m._cur[0]["dec"]
            d = D["desc"]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for rep in range(3):
                torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    ops.attn_rnn_fwd(d, 0, 40)
                e1.record()
                torch.cuda.synchronize()
                dt = e0.elapsed_time(e1) * 1e-3 / 5
            if fused and os.environ.get("AVSR_HIPCC_FLAGS", "").find("DP_TIMING") >= 0:
                tk = ops._persist_sync[16:40].cpu().numpy()
                names = ["P1 loads+mfma", "P1 epilogue", "publish0", "wait0", "P2 attention", "publish1", "wait1", "P3 attlayer", "publish2",
                         "wait2", "P4 sample", "loop", "p2:scores", "p2:softmax", "p2:ctx", "p3:loads+wgt", "p3:mfma", "p4:logits", "p4:sample", "-", "-", "-", "-", "-"]
                print("   per-step shader-clock ticks of workgroup 0:", ", ".join(f"{n} {int(v)}" for n, v in zip(names, tk) if n != "-"), "sum", int(tk.sum()))
            print(f"c4 decoder forward (40 steps) fused={fused}: {dt * 1e3:.3f} ms = {dt * 1e6 / 40:.2f} us/step; "
                  f"75.37 MB/step -> {75.37e6 / (dt / 40) / 1e12:.2f} TB/s algorithmic; err={ops.rnn_persistent_error()}")
            for rep in range(2):
                t0 = time.perf_counter()
                m.train_step(batch)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            print(f"   eager train step: {dt * 1e3:.2f} ms")
            for rep in range(3):
                m.forward_train(batch)
                torch.cuda.synchronize()
                e0.record()
                m.backward()
                e1.record()
                torch.cuda.synchronize()
            print(f"   backward pass (all of it): {e0.elapsed_time(e1):.3f} ms; err={ops.rnn_persistent_error()}")
            if fused == 1 and os.environ.get("AVSR_HIPCC_FLAGS", "").find("DP_TIMING") >= 0:
                tk = ops._persist_sync[16:28].cpu().numpy()
                names = ["wait2", "A loads+mfma", "A epilogue+stage2", "publish0", "wait0", "B dctx", "B dq", "publish1", "wait1", "C cell", "publish2", "B dalpha"]
                print("   BPTT per-step shader-clock ticks of workgroup 0:", ", ".join(f"{n} {int(v)}" for n, v in zip(names, tk) if n != "-"), "sum", int(tk.sum()))
    return bad


if __name__ == "__main__":
    sys.exit(main())
