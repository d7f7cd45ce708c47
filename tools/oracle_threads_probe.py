"""How fast is the fp64 torch-CPU oracle on this box as a function of the intra-op thread count?
(round 6: the driver's GPU suite spent 1181 s, most of it inside the oracle; this box: 8 cores, 11 s for the case below)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(),
      "interop", torch.get_num_interop_threads())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(p, open(p).read().strip())
    except OSError:
        pass
os.system("lscpu | head -20; cat /proc/loadavg")
import test_gpu_model as T
case, over = T.FULL_LENGTH[0]
for nb, n in ((2, None), (2, 1), (2, 4), (2, 8), (2, 16), (2, 32), (8, 8), (8, 16)):
    if n:
        torch.set_num_threads(n)
    O, ocfg, mcfg, W, batch = T.make(case, B=nb, Ta=500, Tv=75, L=40, ragged=True, **over)
    t = time.time()
    ref = O.train_step(W, None, ocfg, batch)
    print("B", nb, "threads", torch.get_num_threads(), "train_step %.2f s" % (time.time() - t), flush=True)
