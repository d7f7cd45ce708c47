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
    st = forced["collective_selftest"]                             # RCCL pre-flight (known-answer all-reduces before anything is timed)
    assert st["ok"] and st["backend"] == "rccl" and st["ranks"] == 1 and forced["rccl_ranks"] == 1 and st["allreduce_13MB_f32_us"] > 0
    assert plain["collective_selftest"] is None and plain["rccl_ranks"] is None
    assert plain["data"].startswith("synthetic, device-resident")
    # the strong-scaling section `--gpus N > 1` adds (global batch fixed: here 64 / 8 utterances on the one rank)
    strong = _bench({"AVSR_BENCH_FORCE_DIST": "1", "AVSR_BENCH_FORCE_STRONG": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29535"})
    assert strong["strong_scaling"]["value"] > 0 and strong["strong_scaling"]["utterances_per_gpu"] == 8 and strong["scaling"] == "weak"
    assert not strong["strong_scaling"]["persistent_wait_expired"]
    eager = _bench({"AVSR_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29534", "AVSR_DP_GRAPH": "0"})
    assert eager["config"]["launch"].startswith("eager")           # the escape hatch
    assert abs(eager["final_loss"] - plain["final_loss"]) < 1e-4 * max(1.0, abs(plain["final_loss"]))
    assert abs(forced["final_loss"] - plain["final_loss"]) < 1e-4 * max(1.0, abs(plain["final_loss"]))


def _ranks_over_gloo(world, batch, steps):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, AVSR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", "2",
           "--batch", str(batch), "--video-frontend", "features", "--no-cpu-baseline", "--no-profile"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 8])
def test_bench_ranks_over_gloo_with_persistent_kernels_contending(world):
    """`bench.py --gpus N` exactly as the driver launches it (torch.distributed.run, one process per rank; N = 2 and the driver's
    N = 8), except that the transport is gloo and all ranks sit on the box's one GPU: weak-scaling headline + the strong-scaling
    section, the collective self-test, hipGraph replays around the collectives, and -- what a one-rank run can never show -- the
    persistent encoder / decoder kernels of SEVERAL processes contending for the same XCDs.  Whatever the dispatcher does (grids
    resident together, one after the other, or a bounded wait expiring -> flag MAX-reduced over the ranks -> all redo through the
    per-step launches), the run must finish with one JSON line, finite equal losses on the replicas' shared step count, and report
    the flag truthfully."""
    out = _ranks_over_gloo(world, 8, 6 if world == 2 else 3)
    assert out["n_gpus"] == world and out["scaling"] == "weak" and out["config"]["global_batch"] == 8 * world and out["value"] > 0
    assert out["config"]["collectives_per_step"] == 2
    assert out["collective_selftest"]["ok"] and out["rccl_ranks"] == world and out["collective_selftest"]["backend"] == "gloo"
    assert len(out["rank_timing"]["seconds_before_barrier"]) == world
    assert out["final_loss"] == out["final_loss"] and abs(out["final_loss"]) < 100.0            # finite
    ss = out["strong_scaling"]
    assert ss.get("value") and ss["utterances_per_gpu"] == 64 // world and ss["global_batch"] == 64, ss
    assert isinstance(out["persistent_wait_expired"], bool) and isinstance(ss["persistent_wait_expired"], bool)   # reported either way


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


def test_graph_capture_at_the_second_sighting_of_a_batch_shape():
    """AVSR.train's policy (graph_after=2): a batch shape runs eagerly the first time it is seen and is captured the second time, other
    shapes in between stay eager; the parameters after the sequence equal those of eager-only training bit for bit."""
    import sys
    sys.path.insert(0, ROOT)
    import dataclasses
    import numpy as np
    import torch
    from avsr_tf1_amd.config import ModelConfig
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    from avsr_tf1_amd.parallel import DataParallelTrainer
    from oracle import avsr_oracle as O
    ocfg = O.OracleConfig(architecture="bimodal", encoder_type="unidirectional", video_units=(32,), audio_units=(32, 32), decoder_units=(32,),
                          embedding_size=16, video_feat=12, audio_feat=20, regress_aus=True, use_dropout=False, warmup_steps=0)
    mcfg = ModelConfig(**{f.name: getattr(ocfg, f.name) for f in dataclasses.fields(ModelConfig) if hasattr(ocfg, f.name)})
    W = O.init_params(ocfg, seed=3)
    big = Batch.from_numpy(O.synthetic_batch(ocfg, B=4, T_a=17, T_v=7, L=6, ragged=True))
    small = Batch.from_numpy(O.synthetic_batch(ocfg, B=3, T_a=11, T_v=5, L=4, ragged=True, seed=77))
    order = [big, small, big, big, small, big]
    out = {}
    for mode in ("eager", "second"):
        m = Seq2SeqModel(mcfg, weights=W)
        t = DataParallelTrainer(m, None, use_graph=(mode == "second"), graph_after=2)
        modes = []
        for b in order:
            t.train_step(b)
            torch.cuda.synchronize()
            modes.append((t.mode, len(t._graphs)))
        if mode == "second":
            # big: eager, (small: eager), big: captured, big: replayed, small: captured, big: replayed
            assert [n for _, n in modes] == [0, 0, 1, 1, 2, 2] and modes[-1][0] == "hipgraph", modes
        out[mode] = m.export_tf_weights("params")
    for k, v in out["eager"].items():
        assert np.array_equal(v, out["second"][k]), k
