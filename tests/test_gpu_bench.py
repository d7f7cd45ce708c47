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
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
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
    assert plain["n_gpus"] == 1 and plain["steps"] == 3 and plain["value"] > 0 and plain["config"]["launch"] == "hipgraph"
    forced = _bench({"AVSR_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert forced["config"]["launch"] == "hipgraph"
    assert abs(forced["final_loss"] - plain["final_loss"]) < 1e-4 * max(1.0, abs(plain["final_loss"]))
