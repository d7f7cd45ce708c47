"""bench.py contract on the GPU box: one JSON line on stdout, and the data-parallel code path (RCCL collectives around the
hipGraph replays, sync batch-norm) exercised with ONE rank -- the single-GPU box cannot host two ranks, so the collectives are
forced on at world size 1 and must reproduce the plain single-process loss."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *args):
    env = dict(os.environ, **extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2", "--no-cpu-baseline",
                        "--no-profile", "--batch", "16", *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]                    # exactly one JSON line on stdout
    return json.loads(lines[0])


def test_bench_json_contract_and_forced_collectives():
    plain = _bench({})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in plain, k
    assert plain["n_gpus"] == 1 and plain["steps"] == 20 and plain["value"] > 0 and plain["config"]["launch"] == "hipgraph"
    forced = _bench({"AVSR_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert forced["config"]["launch"] == "hipgraph"                # graphs are replayed around the collectives
    # the strong-scaling section `--gpus N > 1` adds (global batch fixed: here 64 / 8 utterances on the one rank)
    strong = _bench({"AVSR_BENCH_FORCE_DIST": "1", "AVSR_BENCH_FORCE_STRONG": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29535"})
    assert strong["strong_scaling"]["value"] > 0 and strong["strong_scaling"]["utterances_per_gpu"] == 8 and strong["scaling"] == "weak"
    assert not strong["strong_scaling"]["persistent_wait_expired"]
    eager = _bench({"AVSR_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29534", "AVSR_DP_GRAPH": "0"})
    assert eager["config"]["launch"].startswith("eager")           # the escape hatch
    assert abs(eager["final_loss"] - plain["final_loss"]) < 1e-4 * max(1.0, abs(plain["final_loss"]))
    assert abs(forced["final_loss"] - plain["final_loss"]) < 1e-4 * max(1.0, abs(plain["final_loss"]))


@pytest.mark.parametrize("front,fresh", [("features", False), ("resnet_cnn", False), ("resnet_cnn", True)])
def test_queued_graph_replays_equal_eager_steps(front, fresh):
    """Train steps queued back to back (no host sync) through the captured graphs must equal eager steps bit for bit, at the
    benchmark's full batch size.  Regression test for a ROCm 7.0 hazard found with the 75 MB lip-crop batch: a large eager kernel
    or D2D copy between two launches of a captured graph let queued launches overlap (wrong losses, expired persistent-kernel
    waits, GPU memory faults).  `fresh`: a different batch object every step, i.e. the staging copy is exercised."""
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from avsr_tf1_amd import ops
    from avsr_tf1_amd.config import ModelConfig
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    from avsr_tf1_amd.parallel import DataParallelTrainer
    wl = bench.WORKLOADS["c4"]
    cfg = ModelConfig(audio_feat=bench.FA, video_feat=bench.FV, video_processing=front, use_dropout=True, sampling_probability=0.1, **wl["cfg"])
    batch = Batch.from_numpy(bench.NS(bench.synth(cfg, 64, 0)))
    names = ("audio", "audio_len", "video", "video_len", "aus", "labels", "labels_len")
    other = Batch(*[None if getattr(batch, n) is None else getattr(batch, n).clone() for n in names])
    losses = {}
    for mode in ("eager", "graph"):
        m = Seq2SeqModel(cfg, seed=2001)
        t = DataParallelTrainer(m, None, use_graph=(mode == "graph"))
        t.train_step(batch)
        torch.cuda.synchronize()
        for chunk in range(3):
            for i in range(10):
                t.train_step(other if (fresh and i % 2) else batch)
            torch.cuda.synchronize()
        assert mode == "eager" or t.mode == "hipgraph"
        assert not ops.rnn_persistent_error()
        losses[mode] = (float(m.loss.item()), float(m.gnorm.item()))
        del t, m
        torch.cuda.empty_cache()
    assert losses["eager"] == losses["graph"], losses
