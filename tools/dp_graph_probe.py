"""Debug driver for the graphs-around-collectives failure: two engine ranks over gloo with optional syncs
around every all_reduce.  env: AVSR_DP_GRAPH=1, STEPS, NOSYNC=1, WHICH=pre|post|prepost|clone, ENDSYNC=device|stream, STOCH, BPR"""
import os, sys, socket
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import torch.multiprocessing as mp

def worker(rank, world, port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["AVSR_PERSISTENT_RNN"] = "0"
    import torch.distributed as dist
    import bench
    from avsr_tf1_amd.config import ModelConfig
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    from avsr_tf1_amd.parallel import DataParallelTrainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = bench.WORKLOADS["c4"]
    stoch = dict(use_dropout=True, sampling_probability=0.1) if os.environ.get("STOCH", "1") == "1" else {}
    cfg = ModelConfig(audio_feat=bench.FA, video_feat=bench.FV, **wl["cfg"], **stoch)
    m = Seq2SeqModel(cfg, seed=2001)
    t = DataParallelTrainer(m, dist, use_graph=True)
    b = Batch.from_numpy(bench.NS(bench.synth(cfg, int(os.environ.get("BPR", "32")), rank)))
    orig_ar = dist.all_reduce
    WHICH = os.environ.get("WHICH", "")
    def ar3(x, *a, **k):
        y = x.clone()
        r = orig_ar(y, *a, **k)
        x.copy_(y)
        return r
    def ar2(x, *a, **k):
        if "pre" in WHICH:
            torch.cuda.synchronize()
        r = orig_ar(x, *a, **k)
        if "post" in WHICH:
            torch.cuda.synchronize()
        return r
    def ar(x, *a, **k):
        torch.cuda.synchronize()
        pre = (bool(torch.isfinite(x).all()), float(x.abs().max()))
        r = orig_ar(x, *a, **k)
        torch.cuda.synchronize()
        post = (bool(torch.isfinite(x).all()), float(x.abs().max()))
        if rank == 0 and (not pre[0] or not post[0] or x.numel() <= 8):
            print("   all_reduce n=%d pre %s post %s %s" % (x.numel(), pre, post, x[:4].tolist() if x.numel() <= 8 else ""), flush=True)
        return r
    if os.environ.get('NOSYNC') != '1':
        dist.all_reduce = ar
    elif WHICH == 'clone':
        dist.all_reduce = ar3
    elif WHICH:
        dist.all_reduce = ar2
    for i in range(int(os.environ.get("STEPS", "8"))):
        t.train_step(b)
        if os.environ.get("ENDSYNC") == "device":
            torch.cuda.synchronize()
        elif os.environ.get("ENDSYNC") == "stream":
            torch.cuda.current_stream().synchronize()
        if os.environ.get('NOSYNC') == '1' and i + 1 < int(os.environ.get("STEPS", "8")):
            continue
        torch.cuda.synchronize()
        if rank == 0:
            print("step", i, t.mode, "loss %.6f gnorm %.6f" % (float(m.loss.item()), float(m.gnorm.item())), "params finite", bool(torch.isfinite(m.params).all()), flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
