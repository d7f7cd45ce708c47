import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def cpu_budget():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota (cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


# The fp64 torch-CPU oracle is thousands of tiny [B, H] x [H, 4H] products: torch sizes its intra-op pool from the HOST's core count
# (128 threads on the GPU node) while the container's quota is 16 CPUs, and every op then pays a 128-way fork/join on 16 cores --
# measured on the GPU box (profiles/r06_oracle_threads.txt): the c4 full-length oracle step 11.0 s at the default 128 threads, 0.8 s at
# 8.  The cap is set before torch is imported (children spawned by the multi-process tests inherit it) and again on the live pool.
ORACLE_THREADS = str(min(8, cpu_budget()))
for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, ORACLE_THREADS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        torch.set_num_threads(int(os.environ["OMP_NUM_THREADS"]))
    except Exception:
        pass


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
