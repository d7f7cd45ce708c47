"""Two REAL engine ranks (both processes on the one GPU of the box, gloo transport, unequal shards, ragged lengths) at the
benchmark size against ONE engine on the whole batch.  python tools/dp_full_check.py   [default: the captured graphs are replayed around the collectives; AVSR_DP_GRAPH=0 launches eagerly -- the graph
around the collectives: that is the configuration that went wrong, DESIGN.md section 5]"""
import os, sys, socket
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import torch.multiprocessing as mp
STEPS = int(os.environ.get("DP_CHECK_STEPS", "12"))

def setup():
    import bench
    from avsr_tf1_amd.config import ModelConfig
    wl = bench.WORKLOADS["c4"]
    cfg = ModelConfig(audio_feat=bench.FA, video_feat=bench.FV, **wl["cfg"])
    full = bench.synth(cfg, 64, 0)
    rng = np.random.default_rng(5)                      # ragged lengths so the shards differ in content
    full["audio_len"] = rng.integers(250, 501, 64).astype(np.int32)
    full["video_len"] = rng.integers(38, 76, 64).astype(np.int32)
    return bench, cfg, full

def shard(full, lo, hi):
    return {k: (None if v is None else np.ascontiguousarray(v[lo:hi])) for k, v in full.items()}

def worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["AVSR_PERSISTENT_RNN"] = "0"
    import torch.distributed as dist
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    from avsr_tf1_amd.parallel import DataParallelTrainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bench, cfg, full = setup()
    m = Seq2SeqModel(cfg, seed=2001)
    t = DataParallelTrainer(m, dist, use_graph=True)
    b = Batch.from_numpy(bench.NS(shard(full, 0, 24) if rank == 0 else shard(full, 24, 64)))
    for i in range(STEPS):
        t.train_step(b)
    torch.cuda.synchronize()
    if rank == 0:
        print("dp mode:", t.mode, "loss share %.7f gnorm %.7f" % (float(m.loss.item()), float(m.gnorm.item())), flush=True)
        np.save(os.path.join(out_dir, "p.npy"), m.params.cpu().numpy())
    dist.destroy_process_group()

if __name__ == "__main__":
    import tempfile
    d = tempfile.mkdtemp()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port, d), nprocs=2, join=True)
    os.environ["AVSR_PERSISTENT_RNN"] = "0"
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    bench, cfg, full = setup()
    m = Seq2SeqModel(cfg, seed=2001)
    b = Batch.from_numpy(bench.NS(full))
    for i in range(STEPS):
        m.train_step(b)
    torch.cuda.synchronize()
    p = np.load(os.path.join(d, "p.npy"))
    q = m.params.cpu().numpy()
    print("single loss %.7f gnorm %.7f" % (float(m.loss.item()), float(m.gnorm.item())))
    print("max |dp - single| = %.3e  (max |param| %.3f, lr 1e-3 x %d steps)" % (np.abs(p - q).max(), np.abs(q).max(), STEPS))
