#!/usr/bin/env python
"""bench.py -- utterances/sec of one full AVSR train step (fwd + BPTT + clip + Adam) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c4|c2|c3|c5] [--no-graph] [--video-frontend resnet_cnn|features]

The headline feeds north_star's synthetic input: T_v = 75 x 36 x 36 x 3 lip crops through the CNN front-end (video.resnet_cnn) for
every workload with a video stream; the same step on pre-computed 128-d lip features is reported next to it as `without_lip_cnn`.

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU, RCCL): utterances are
sharded across ranks (weak scaling: B utterances PER GPU; a `strong_scaling` object reports the same job at the workload's
global B), the collectives on the data path are the gradient all-reduce, a 4-float all-reduce of the loss normalisers and the
two sync-batch-norm reductions of the feature inputs.

Prints ONE JSON line (rank 0).  `value` = utterances processed by all ranks / max-over-ranks wall time,
inputs resident in HBM.  `roofline` = the dominant kernel of the step (by summed time, measured here with
HIP events around every launch in a separate eager pass); `cpu_baseline` = the CPU oracle restatement
(torch-CPU fp32, "port") on a bounded sample of the same workload -- a reported baseline, not a target.
The reference publishes no number for this metric, so `vs_baseline` is null.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # multi-process GPU work on this image needs dmabuf IPC (RCCL)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[3]: AV dual-attention WLAS, reference defaults (uni encoders, scaled_luong), B=64
    "c4": dict(desc="AV dual-attention (bimodal) video 1x256 + audio 3x256 uni-LSTM, scaled_luong, B=64 T_a=500x80 T_v=75 L=40",
               B=64, cfg=dict(architecture="bimodal", encoder_type="unidirectional", video_units=(256,), audio_units=(256, 256, 256),
                              attention_type=(("scaled_luong",), ("scaled_luong",)), regress_aus=True)),
    # configs[1]: audio-only LAS 3 x bi-LSTM-256 + Bahdanau decoder
    "c2": dict(desc="audio-only LAS 3x bi-LSTM-256 + LSTM-256 Bahdanau decoder, B=64 T_a=500x80 L=40",
               B=64, cfg=dict(architecture="unimodal", encoder_type="bidirectional", video_units=None, audio_units=(256, 256, 256),
                              attention_type=(("bahdanau",), ("bahdanau",)))),
    # configs[2]: visual-only 2 x bi-LSTM-256 (lip crops through the CNN front-end by default; --video-frontend features bypasses it)
    "c3": dict(desc="visual-only lip-CNN -> 2x bi-LSTM-256, B=64 T_v=75 L=40",
               B=64, cfg=dict(architecture="unimodal", encoder_type="bidirectional", video_units=(256, 256), audio_units=None,
                              attention_type=(("scaled_luong",), ("scaled_luong",)), regress_aus=True)),
    # configs[4]: AV-Align (uni encoders -- the reference implements AV-Align for unidirectional only), B=128
    "c5": dict(desc="AV-Align video 1x256 + audio 3x256 (top layer attends video), scaled_luong, B=128 T_a=500 T_v=75 L=40",
               B=128, cfg=dict(architecture="av_align", encoder_type="unidirectional", video_units=(256,), audio_units=(256, 256, 256),
                               attention_type=(("scaled_luong",), ("scaled_luong",)), regress_aus=True)),
}
TA, TV, LDEC, FA, FV = 500, 75, 40, 80, 128


def synth(cfg, B, rank):
    """SURVEY 8(d) synthetic inputs, fixed seeds (offset per rank so shards differ)."""
    r = lambda k: np.random.default_rng(1000 + k + 100 * rank)
    d = {}
    if cfg.audio_units is not None:
        d["audio"] = r(1).standard_normal((B, TA, FA)).astype(np.float32)
        d["audio_len"] = np.full(B, TA, np.int32)
    if cfg.video_units is not None:
        if cfg.video_processing == "resnet_cnn":      # lip crops ~ U(-1, 1), i.e. (x - 128) / 128 of dataset_writer.py:537
            d["video"] = r(3).uniform(-1.0, 1.0, (B, TV) + tuple(cfg.video_hw)).astype(np.float32)
        else:
            d["video"] = r(3).standard_normal((B, TV, FV)).astype(np.float32)
        d["video_len"] = np.full(B, TV, np.int32)
        d["aus"] = r(5).uniform(0, 3, (B, TV, 2)).astype(np.float32)
    lab = r(6).integers(1, 29, (B, LDEC)).astype(np.int32)
    lab[:, -1] = 29
    d["labels"], d["labels_len"] = lab, np.full(B, LDEC, np.int32)
    return d


class NS:
    def __init__(self, d):
        self.__dict__.update(d)


def work_model(cfg, B):
    """Algorithmic FLOPs / bytes per launch class for the roofline (fp32)."""
    H = cfg.decoder_units[0]
    mems = cfg.decoder_memories()
    attn_bytes = sum(4 * B * (TA if s == "audio" else TV) * (H + cfg.memory_depth(s)) for s, _ in mems)
    A = H * len(mems)
    # forward LSTM step launches: encoder wavefront (x part of layer 0 is hoisted) + decoder cell (embedding hoisted)
    fl_f = n_f = 0
    fl_b = n_b = 0
    nsteps = 0
    for s in cfg.streams():
        T = TA if s == "audio" else TV
        units = cfg.units(s)
        attentive = cfg.architecture == "av_align" and s == "audio"
        nplain = len(units) - 1 if attentive else len(units)
        nsteps = max(nsteps, T + nplain - 1)
        for _d in cfg.directions():
            for l in range(nplain):
                u = units[l]
                kin = 0 if l == 0 else units[l - 1]
                fl_f += T * 2 * B * (kin + u) * 4 * u
                up = units[l + 1] if l + 1 < nplain else 0
                fl_b += T * 2 * B * (4 * u + 4 * up) * u
        if attentive:
            u = units[-1]
            fl_f += T * 2 * B * (u + u) * 4 * u
            fl_b += T * 2 * B * (4 * u + u) * u
            n_f += T
            n_b += T
    n_f += nsteps + LDEC
    n_b += nsteps + LDEC
    fl_f += LDEC * 2 * B * (A + H) * 4 * H
    fl_b += LDEC * 2 * B * (4 * H + A) * H
    return dict(attn_bytes=attn_bytes, lstm_fwd_flops=fl_f / max(1, n_f), lstm_bwd_flops=fl_b / max(1, n_b))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    try:
        pairs = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        return len(pairs) or None
    except OSError:
        return None


def cpu_budget():
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota (the GPU node shows 256 logical CPUs to a
    container whose cpu.max is 16: a torch pool sized from the host's core count runs 10x slower there, profiles/r06_oracle_threads.txt)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def _cpu_leg(wl, stoch, sample_B, video_frontend, warmups, steps, budget_s, onednn):
    from oracle import avsr_oracle as O
    ncores = cpu_budget()
    torch.set_num_threads(ncores)
    torch.backends.mkldnn.enabled = bool(onednn)
    ocfg = O.OracleConfig(video_processing=video_frontend, **wl["cfg"], **stoch)
    P = O.init_params(ocfg, seed=2001)
    b = O.synthetic_batch(ocfg, B=sample_B, T_a=TA, T_v=TV, L=LDEC)
    t_start = time.perf_counter()
    for _ in range(warmups):
        O.train_step(P, None, ocfg, b, dtype=torch.float32)
    times = []
    while len(times) < steps and (len(times) < 5 or time.perf_counter() - t_start < budget_s):
        t0 = time.perf_counter()
        O.train_step(P, None, ocfg, b, dtype=torch.float32)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    return {"value": round(sample_B / dt, 3), "unit": "utterances/sec", "cores": ncores, "kind": "port", "onednn": bool(onednn),
            "host_cpu": _cpu_model(), "host_physical_cores": _physical_cores(), "host_logical_cpus": os.cpu_count(),
            "container_cpu_quota": ncores, "utterances": sample_B, "video_input": video_frontend,
            "timed_steps": len(times), "warmup_steps": warmups, "step_seconds_median": round(dt, 4),
            "sample": "oracle/avsr_oracle.py train_step (torch-CPU fp32, autograd BPTT, oneDNN %s, %d threads = the CPUs this container may use: "
                      "affinity mask capped by the cgroup quota), B=%d utterances at full T_a=%d T_v=%d L=%d, video input: %s, median of %d timed steps "
                      "after %d warm-ups; TF-1.13.1 reference cannot run here"
                      % ("on" if onednn else "off", ncores, sample_B, TA, TV, LDEC, video_frontend, len(times), warmups)}


def cpu_baseline(wl, stoch, video_frontend="resnet_cnn", onednn=True):
    """the CPU oracle (torch-CPU fp32 restatement, "port") on the SAME workload as the timed GPU step -- the workload's whole
    batch (c4: 64 utterances) from the same input form (lip crops through the CNN front-end), every CPU the container may use, 3 warm-up
    steps, then the median of >= 5 timed steps (more while a 60 s budget lasts).  'TF-1.13.1 CPU number unavailable' -- BASELINE.md section 2."""
    return _cpu_leg(wl, stoch, wl["B"], video_frontend, warmups=3, steps=10, budget_s=60.0, onednn=onednn)


def cpu_baseline_sample(wl, stoch, video_frontend="resnet_cnn"):
    """The bounded sample of earlier rounds (4 utterances at full lengths), kept for continuity with BENCH_r01..r05."""
    return _cpu_leg(wl, stoch, 4, video_frontend, warmups=3, steps=10, budget_s=20.0, onednn=False)


def measure_traffic(args):
    """HBM traffic per dispatch MEASURED IN THIS RUN: two separate `rocprofv3 --pmc <counter> --kernel-trace` passes (FETCH_SIZE, then
    WRITE_SIZE; MI355X_MICROARCH.md: one counter set per pass, kernel trace only) of a short eager run of the same workload in child
    processes, averaged per kernel name.  Returns ({kernel name: {"FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "dispatches"}}, description) or
    (None, reason).  Skipped when rocprofv3 is missing, when this process is itself being profiled, or with AVSR_BENCH_TRAFFIC=0."""
    import glob
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if os.environ.get("AVSR_BENCH_TRAFFIC", "1") == "0":
        return None, "AVSR_BENCH_TRAFFIC=0"
    if any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ):
        return None, "this process is being profiled"
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, "rocprofv3 not found"
    base = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-graph", "--no-cpu-baseline", "--no-profile", "--no-other-workloads",
            "--workload", args.workload, "--video-frontend", args.video_frontend] + (["--no-dropout"] if args.no_dropout else []) + \
           (["--batch", str(args.batch)] if args.batch else [])
    out = {}
    env = dict(os.environ, TMPDIR="/tmp", AVSR_BENCH_TRAFFIC="0")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="avsr_pmc_", dir="/tmp")
        try:
            p = subprocess.run([rp, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--"] + base, cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=400)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if p.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s failed (exit %d)" % (counter, p.returncode)
            cur = sqlite3.connect(dbs[0]).cursor()
            for name, n, val in cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
                                            "group by kernel_name", (counter,)):
                k = re.sub(r"\(.*$", "", name).replace("void ", "")
                out.setdefault(k, {})["dispatches"] = n
                out[k][counter + "_KiB"] = round(float(val), 1)
        except Exception as e:
            return None, "rocprofv3 --pmc %s: %r" % (counter, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out, ("measured in this run: rocprofv3 --pmc FETCH_SIZE --kernel-trace and rocprofv3 --pmc WRITE_SIZE --kernel-trace (separate passes) "
                 "of `bench.py --steps 2 --warmup 1 --no-graph` on the same workload, averages per dispatch")


def collective_selftest(dist, world, rank, backend):
    """Pre-flight for the first contact with N > 1 real devices: before anything is timed, every collective shape the step uses is
    run on buffers with a known answer and checked EXACTLY (small integers: fp32 / fp64 sums are exact) -- the 13 MB fp32 gradient
    all-reduce (SUM), the fp64 batch-norm moments vector (SUM), the fp64 MAX reduction and the all_gather of the timing block.  A
    wrong sum aborts the run (a bench line over a broken transport would be worse than none).  Returns what goes into the JSON line:
    the ranks the transport saw, the RCCL version, and the time of one 13 MB all-reduce after a warm-up."""
    dev = torch.device("cuda", torch.cuda.current_device())
    res = {"backend": "rccl" if backend == "nccl" else backend, "ranks": dist.get_world_size(), "rank0_device": torch.cuda.get_device_name(dev)}
    try:
        res["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
    except Exception:
        res["rccl_version"] = None
    tri = world * (world + 1) // 2
    n = 13 * (1 << 20) // 4
    pat = (torch.arange(n, device=dev, dtype=torch.int64) % 251).to(torch.float32)
    for rep in range(2):                                           # first pass = transport set-up, second pass is the timed one
        g = pat * float(rank + 1)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        dist.all_reduce(g)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if not torch.equal(g, pat * float(tri)):
            raise SystemExit("collective self-test FAILED: 13 MB fp32 all-reduce (SUM) over %d ranks returned a wrong sum on rank %d" % (world, rank))
    res["allreduce_13MB_f32_us"] = round(1e6 * dt, 1)
    m = (torch.arange(4096, device=dev, dtype=torch.float64) % 17) * float(rank + 1)
    dist.all_reduce(m)
    if not torch.equal(m, (torch.arange(4096, device=dev, dtype=torch.float64) % 17) * float(tri)):
        raise SystemExit("collective self-test FAILED: fp64 moments all-reduce (SUM) on rank %d" % rank)
    mx = torch.tensor([float(rank)], device=dev, dtype=torch.float64)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    if float(mx.item()) != float(world - 1):
        raise SystemExit("collective self-test FAILED: fp64 all-reduce (MAX) on rank %d" % rank)
    parts = [torch.zeros(4, device=dev, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(parts, torch.full((4,), float(rank), device=dev, dtype=torch.float64))
    if [float(p[0].item()) for p in parts] != [float(r) for r in range(world)]:
        raise SystemExit("collective self-test FAILED: all_gather on rank %d" % rank)
    seen = sorted({int(p[0].item()) for p in parts})
    res["ranks_seen"] = len(seen)
    res["ok"] = True
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU (default: the workload's B)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-dropout", action="store_true", help="disable DropoutWrapper + scheduled sampling (reference defaults are ON)")
    ap.add_argument("--video-frontend", default="resnet_cnn", choices=["features", "resnet_cnn"],
                    help="resnet_cnn (default): 36x36x3 lip crops through the CNN front-end, north_star's input; features: 128-d lip features")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--section", default=None, choices=[None, "aux_frontend", "cpu_baseline", "cpu_baseline_noonednn", "cpu_baseline_sample"], help="internal: run one auxiliary section in a child process and print its JSON")
    ap.add_argument("--strong", action="store_true", help="N > 1: global batch fixed at the workload's B (B/N utterances per GPU) instead of B per GPU")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--brief", action="store_true", help="timed steps + the per-kernel event pass only (no counters, decode rates, other input form, CPU baseline)")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the c2 / c3 / c5 lines appended to the default c4 run")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): libraries that print banners to fd 1 (RCCL does, through C stdio that is only
    # flushed at exit) are sent to stderr for the whole run; the JSON goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # test hook (tests/test_gpu_bench.py): AVSR_BENCH_BACKEND=gloo runs N ranks on however many GPUs the box has (ranks share a device;
    # RCCL refuses two ranks on one GPU) -- the whole multi-rank path incl. the strong-scaling section, with the persistent kernels of
    # the ranks contending for the same CUs.  Its numbers mean nothing; the driver's runs use RCCL, one rank per GPU.
    backend = os.environ.get("AVSR_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dist = None
    force_dist = world == 1 and os.environ.get("AVSR_BENCH_FORCE_DIST") == "1"   # test hook: RCCL path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    selftest = collective_selftest(dist, world, rank, backend) if dist is not None else None

    trace = (lambda m: (torch.cuda.synchronize(), sys.stderr.write("[bench] %s\n" % m), sys.stderr.flush())) \
        if os.environ.get("AVSR_BENCH_TRACE") else (lambda m: None)
    from avsr_tf1_amd import ops
    from avsr_tf1_amd.config import ModelConfig
    from avsr_tf1_amd.model import Batch, Seq2SeqModel
    from avsr_tf1_amd.parallel import DataParallelTrainer

    wl = WORKLOADS[args.workload]
    B = args.batch or wl["B"]
    stoch = {} if args.no_dropout else dict(use_dropout=True, sampling_probability=0.1)   # avsr/avsr.py:51-56 defaults
    if args.section in ("cpu_baseline", "cpu_baseline_noonednn"):
        os.write(json_fd, (json.dumps(cpu_baseline(wl, stoch, video_frontend=args.video_frontend,
                                                   onednn=args.section == "cpu_baseline")) + "\n").encode())
        return
    if args.section == "cpu_baseline_sample":
        os.write(json_fd, (json.dumps(cpu_baseline_sample(wl, stoch, video_frontend=args.video_frontend)) + "\n").encode())
        return
    if args.section == "aux_frontend":
        cfg2 = ModelConfig(audio_feat=FA, video_feat=FV, video_processing=args.video_frontend, **wl["cfg"], **stoch)
        m2 = Seq2SeqModel(cfg2, seed=2001)
        t2 = DataParallelTrainer(m2, None, use_graph=not args.no_graph, check_every_step=False)
        b2 = t2.static_batch(Batch.from_numpy(NS(synth(cfg2, B, rank))))
        for _ in range(3):
            t2.train_step(b2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            t2.train_step(b2)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / 5
        res = {"value": round(B / dt2, 2), "unit": "utterances/sec", "ms_per_step": round(1e3 * dt2, 4),
               "video_input": ("[B,%d,36,36,3] lip crops through video.resnet_cnn (avsr_tf1_amd/cnn.py)" % TV) if args.video_frontend == "resnet_cnn"
               else "[B,%d,%d] pre-computed lip features (video_processing='features')" % (TV, FV),
               "final_loss": round(float(m2.loss.item()), 5), "launch": t2.mode,
               "persistent_wait_expired": bool(ops.rnn_persistent_error())}
        os.write(json_fd, (json.dumps(res) + "\n").encode())
        return
    if args.strong and world > 1:
        B = max(1, B // world)
    cfg = ModelConfig(audio_feat=FA, video_feat=FV, video_processing=args.video_frontend, **wl["cfg"], **stoch)
    model = Seq2SeqModel(cfg, seed=2001)
    # the timed loop checks the persistent kernels' sticky flag once (first step) and reports it in the JSON line afterwards
    trainer = DataParallelTrainer(model, dist, use_graph=not args.no_graph, force_collectives=force_dist, check_every_step=False)
    batch = trainer.static_batch(Batch.from_numpy(NS(synth(cfg, B, rank))))     # trainer-owned buffers: no staging copy per step

    trace("model built")
    # the first step runs eagerly and captures the graphs, the second is the first replay (one-time upload of the executable
    # graph): both always stay outside the timed region, whatever --warmup says
    for _ in range(max(2, args.warmup) if not args.no_graph else max(1, args.warmup)):
        trainer.train_step(batch)
        trace("warm-up step done")
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # an event behind every step: per-step GPU times of this rank (read after the timed region; an event record costs no host wait)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        trainer.train_step(batch)
        marks[i + 1].record()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    rank_stats = [dt_local, min(step_ms), float(np.median(step_ms)), max(step_ms)]
    per_rank = [rank_stats]
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        allr = [torch.zeros(4, device="cuda", dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allr, torch.tensor(rank_stats, device="cuda", dtype=torch.float64))
        per_rank = [[float(v) for v in a.cpu()] for a in allr]
    loss = float(model.loss.item())
    trace("timed steps done")
    persist_err = bool(ops.rnn_persistent_error())      # sticky flag of the persistent encoder kernels (a bounded device-side wait expired)

    out = {
        "metric": "utterances/sec (train step) at B=64 T_a=500 T_v=75",
        "value": round(B * world * args.steps / dt, 2), "unit": "utterances/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "strong" if (args.strong and world > 1) else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic, device-resident (one batch replayed; no host-to-device copy in the timed region)", "persistent_wait_expired": persist_err,
        "config": {"workload": args.workload + ": " + wl["desc"], "utterances_per_gpu": B, "global_batch": B * world,
                   "T_a": TA, "F_a": FA, "T_v": TV, "F_v": FV, "T_dec": LDEC, "parallelism": "dp%d" % world,
                   "video_frontend": (cfg.video_processing if cfg.video_units is not None else None), "launch": trainer.mode, "dropout": bool(cfg.use_dropout), "dropout_keep": list(cfg.decoder_dropout) if cfg.use_dropout else None,
                   "scheduled_sampling": cfg.sampling_probability,
                   "collectives_per_step": (2 if world > 1 else 0),
                   "dp_batch_norm": ("encoder-input batch norms use the statistics of the GLOBAL batch (fp64 moments in the step's one small "
                                     "all-reduce); the batch norms inside the lip CNN (and the input batch norm of the CNN-fed stream) NORMALISE with PER-RANK "
                                     "statistics -- a documented deviation from one engine on the whole batch; their moving averages are averaged "
                                     "over the ranks in the gradient all-reduce's tail, so replicas stay bit-identical "
                                     "(tests/test_gpu_dp.py::test_two_ranks_with_the_lip_cnn).  Opt-in global statistics for those batch norms: "
                                     "AVSR_DP_SYNC_CNN_BN=1 (16 small collectives inside the step, eager launches; "
                                     "tests/test_gpu_dp.py::test_two_ranks_with_synchronised_cnn_batch_norms_equal_one_engine)") if world > 1 else None,
                   "parity": "vs CPU restatement of TF-1.13.1 semantics; TF parity unpinned (pinned to reference outputs: the beam-search step, CER/WER)"},
        "final_loss": round(loss, 5),
        "rccl_ranks": (selftest["ranks_seen"] if selftest else None), "collective_selftest": selftest,
        # per rank: wall seconds of the timed loop before the closing barrier, and min / median / max GPU milliseconds per step (event
        # pairs; includes the collectives).  A slow rank or a slow step shows here: the first thing to read in a multi-GPU run.
        "rank_timing": {"seconds_before_barrier": [round(r[0], 4) for r in per_rank],
                        "step_ms_min_median_max": [[round(r[1], 3), round(r[2], 3), round(r[3], 3)] for r in per_rank],
                        "slowest_over_fastest_rank": round(max(r[0] for r in per_rank) / max(1e-9, min(r[0] for r in per_rank)), 4)},
    }

    force_strong = force_dist and os.environ.get("AVSR_BENCH_FORCE_STRONG") == "1"      # test hook: this section with one rank
    if (world > 1 or force_strong) and not args.strong and dist is not None:
        # the same job with the GLOBAL batch fixed at the workload's B (strong scaling): B / N utterances per GPU.  The headline above is
        # weak scaling (B per GPU); both figures are labelled, neither is an efficiency.
        Bs = max(1, wl["B"] // (world if world > 1 else 8))
        try:
            m_s = Seq2SeqModel(cfg, seed=2001)
            t_s = DataParallelTrainer(m_s, dist, use_graph=not args.no_graph, check_every_step=False)
            b_s = t_s.static_batch(Batch.from_numpy(NS(synth(cfg, Bs, rank))))
            for _ in range(max(2, args.warmup) if not args.no_graph else max(1, args.warmup)):
                t_s.train_step(b_s)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                t_s.train_step(b_s)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
            out["strong_scaling"] = {"value": round(Bs * world * args.steps / dts, 2), "unit": "utterances/sec", "scaling": "strong",
                                     "global_batch": Bs * world, "utterances_per_gpu": Bs, "ms_per_step": round(1e3 * dts / args.steps, 4),
                                     "launch": t_s.mode, "persistent_wait_expired": bool(ops.rnn_persistent_error())}
            del t_s, m_s
        except Exception as e:      # an auxiliary figure must not take the headline line down
            out["strong_scaling"] = {"value": None, "error": repr(e)}

    if rank == 0 and not args.no_profile:
        # per-kernel timing: one eager step with a HIP-event pair around every engine launch
        # The host enqueues slower than these kernels run, so a busy-wait kernel first holds the stream for ~0.4 s:
        # the whole step is queued behind it and then executes back-to-back, and each event pair brackets
        # (kernel + its dependent-launch boundary) instead of host launch gaps.
        torch.cuda.synchronize()
        ops.prof_begin(1 << 16)
        torch.cuda._sleep(int(0.4 * 2.0e9))
        model.train_step(batch)
        prof = ops.prof_end()
        wm = work_model(cfg, B)
        kinds = {}
        for k, (cnt, ms, fl) in prof.items():
            if cnt:
                kinds[k] = {"launches": cnt, "total_ms": round(ms, 3), "avg_us": round(1e3 * ms / cnt, 3)}
                if fl:
                    kinds[k]["algorithmic_gflop"] = round(fl / 1e9, 3)
        out["kernel_time_events"] = kinds
        # the dominant KERNEL: the event classes gemm / conv_* aggregate many launches of different shapes (their class averages are
        # reported in roofline_other), so the headline roofline is the single kernel with the most time per step
        # (the AV-Align layer's class on c5 is several launches of one kernel: a class of its own since round 3)
        single = [k for k in kinds if not (k == "gemm" or k.startswith("conv_"))] or list(kinds)
        dom = max(single, key=lambda k: kinds[k]["total_ms"])
        HBM_PEAK, MFMA_PEAK = 8000.0, 157.3
        # HBM traffic per launch from the committed PMC passes (profiles/r02_c4_lipcnn_pmc_v5.json: FETCH_SIZE / WRITE_SIZE collected in
        # separate rocprofv3 runs of this same workload, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md); null
        # for workloads / kernels that were not profiled.
        pmc, pmc_file = {}, None
        try:
            if args.workload == "c4" and args.video_frontend == "resnet_cnn":
                # the newest committed counter summary of this workload (profiles/rNN_c4_lipcnn_pmc_vK.json: highest round, then version)
                import glob
                import re
                cands = []
                for path in glob.glob(os.path.join(ROOT, "profiles", "r*_c4_lipcnn_pmc_v*.json")):
                    mm = re.search(r"r(\d+)_c4_lipcnn_pmc_v(\d+)\.json$", path)
                    if mm:
                        cands.append((int(mm.group(1)), int(mm.group(2)), path))
                if cands:
                    path = max(cands)[2]
                    pmc, pmc_file = json.load(open(path))["kernels"], "profiles/" + os.path.basename(path)
        except Exception:
            pmc, pmc_file = {}, None
        # `traffic`: measured in THIS run where rocprofv3 is available (two --pmc passes of a short eager run, child processes: round 5);
        # the committed PMC summary above is the fallback, and the JSON says which one it was
        live, why = (None, "--brief") if args.brief else (measure_traffic(args) if world == 1 else (None, "multi-GPU run"))
        if live:
            pmc, pmc_file = live, why
        else:
            pmc_file = "%s (not measured in this run: %s)" % (pmc_file, why)
        out["traffic_source"] = pmc_file
        # ONE correction rule, calibrated on this GPU with known-bytes kernels in the engine's access forms (tools/pmc_calibrate.py ->
        # profiles/r04_pmc_calibration.json): FETCH_SIZE counts L2-miss read REQUESTS at 64 B apiece, and a request that needs a whole
        # 128-byte line is one request -- contiguous reads of every width (4 / 8 / 16 B per lane, global, buffer and LDS-DMA loads) are
        # reported at exactly half their bytes, half-line reads exactly; WRITE_SIZE is exact; Infinity-Cache hits are counted.
        # traffic = fetch_factor * FETCH_SIZE + write_factor * WRITE_SIZE (every kernel of this engine streams whole lines).
        fetch_factor, write_factor, calib_file = 2.0, 1.0, None
        try:
            cpath = os.path.join(ROOT, "profiles", "r04_pmc_calibration.json")
            ck = json.load(open(cpath))["kernels"]
            fetch_factor = float(ck["calib_rd16b"]["factor_true_over_reported"])
            write_factor = float(ck["calib_wr16"]["factor_true_over_reported"])
            calib_file = "profiles/r04_pmc_calibration.json"
        except Exception:
            pass
        out["traffic_correction"] = {"fetch_factor": fetch_factor, "write_factor": write_factor, "calibration": calib_file,
                                     "rule": "traffic = fetch_factor * FETCH_SIZE + write_factor * WRITE_SIZE per dispatch"}
        # kernel-name prefixes of the PMC summary (template arguments vary with the configuration: the first match is taken)
        pmc_name = {"attn_fwd": ["avsr::attn_fwd_kernel"], "attn_bwd": ["avsr::attn_bwd_kernel"],
                    "dec_persist_fwd": ["avsr::dec_persist_kernel<"], "dec_persist_bwd": ["avsr::dec_persist_bwd_kernel<"],
                    "step_lstm_fwd": ["avsr::step_kernel<1, 1, 1, 4>"], "step_lstm_bwd": ["avsr::step_kernel<2, 1, 1, 8>"],
                    "step_dense": ["avsr::step_kernel<0, 1, 1, 4>"],
                    "rnn_persist_fwd": ["avsr::rnn_persist_fwd_xcd_kernel"],
                    "rnn_persist_bwd": ["avsr::rnn_persist_bwdk_kernel", "avsr::rnn_persist_bwd_kernel"]}

        def traffic(kind):
            for prefix in pmc_name.get(kind, []):
                for name in sorted(pmc):
                    if name.startswith(prefix):
                        e = pmc[name]
                        if "FETCH_SIZE_KiB" in e and "WRITE_SIZE_KiB" in e:
                            return int(1024.0 * (fetch_factor * e["FETCH_SIZE_KiB"] + write_factor * e["WRITE_SIZE_KiB"]))
                        return e["hbm_bytes_per_dispatch_corrected"]
            return None

        def roof(kind):
            us = kinds[kind]["avg_us"]
            if kind in ("dec_persist_fwd", "dec_persist_bwd"):
                # fused persistent decode kernel: ONE launch = all T_dec steps of the (dual-)attention decoder for up to 64 utterances
                # (8 XCDs x 8 rows; a larger batch is consecutive launches over 64-row slices); a step's algorithmic bytes are the
                # keys + values of every memory of the launch's rows (what the per-step attention kernel streamed from HBM); here they
                # are resident in VGPRs / LDS, so `achieved` is the north_star figure "bytes / measured us per decode step" against the
                # 8 TB/s peak, not an HBM counter
                rows = min(B, 64)
                per_step_us = us / LDEC
                byts = wm["attn_bytes"] * rows / B
                ach = byts / (per_step_us * 1e-6) / 1e9
                return {"kernel": kind, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK, 4), "traffic": traffic(kind), "algorithmic_bytes_per_decode_step": int(byts),
                        "rows_per_launch": rows, "decode_steps_per_launch": LDEC, "us_per_decode_step": round(per_step_us, 3), "avg_launch_us": us,
                        "figure": "speed-equivalent, not a bandwidth measurement: the algorithmic bytes a per-step attention kernel would stream, "
                                  "divided by the measured time per decode step; the keys / values are register- and LDS-resident here, so the "
                                  "HBM counters (`traffic`) read a few percent of the algorithmic bytes",
                        "note": "whole decode step (cell + scores/softmax/context + attention layer + output layer + sample) fused; keys/values resident on chip"
                        if kind == "dec_persist_fwd" else
                        "whole BPTT step (attention-layer transpose + attention backward + cell backward) fused; the per-step attention backward "
                        "streamed the same keys + values from HBM every step"}
            if kind in ("align_persist_fwd", "align_persist_bwd"):
                # the AV-Align attentive encoder layer (encoder.py:265-290) through the same fused kernels (16-row groups, 128 utterances
                # per launch): per AUDIO FRAME the layer attends the whole video memory: 4*rows*T_v*(H + D) bytes (19.66 MB at
                # B = 128); one launch walks T_a frames
                rows = min(B, 128)
                byts = 4.0 * rows * TV * (cfg.audio_units[-1] + cfg.memory_depth("video"))
                per_frame_us = us / TA
                ach = byts / (per_frame_us * 1e-6) / 1e9
                return {"kernel": kind, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK, 4), "traffic": None, "algorithmic_bytes_per_audio_frame": int(byts), "rows_per_launch": rows,
                        "audio_frames_per_launch": TA, "us_per_audio_frame": round(per_frame_us, 3), "avg_launch_us": us,
                        "note": "AV-Align attentive layer: LSTM step + attention over the video memory per audio frame, one persistent launch"}
            if kind in ("attn_fwd", "attn_bwd"):
                ach = wm["attn_bytes"] / (us * 1e-6) / 1e9
                return {"kernel": kind, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK, 4), "traffic": traffic(kind),
                        "algorithmic_bytes_per_launch": wm["attn_bytes"], "avg_launch_us": us}
            # fp32 MFMA work: algorithmic FLOPs summed by the launchers (2*M*N*K of every GEMM / step task / persistent sequence)
            fl = prof[kind][2] / max(1, prof[kind][0])
            if not fl:
                return {"kernel": kind, "bound": "mfma", "achieved": None, "peak": MFMA_PEAK, "unit": "TFLOP/s", "frac": None,
                        "traffic": None, "avg_launch_us": us}
            ach = fl / (us * 1e-6) / 1e12
            r = {"kernel": kind, "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK, "unit": "TFLOP/s",
                 "frac": round(ach / MFMA_PEAK, 4), "traffic": traffic(kind), "algorithmic_flops_per_launch": int(fl), "avg_launch_us": us}
            if kind.startswith("conv_"):
                # the lip CNN's layers sit on both sides of the machine balance (24 FLOP/B): next to the class's MFMA fraction, the
                # time of every layer's BINDING bound -- algorithmic bytes (source + destination map, + the residual a forward reads)
                # at 8 TB/s or algorithmic FLOPs at 157.3 TF, whichever is larger -- summed over the class and divided by its time
                try:
                    cnn = model._cur[0]["enc"]["video"]["cnn"]
                    hb = mb = bind = 0.0
                    folded = set(getattr(cnn, "fold_wg", {}).values())      # convolutions whose weight gradient evaluates the BN backward
                    src_of = {op[1]: op[2] for op in cnn.ops if op[0] == "conv"}
                    for name, (n_, h_, w_, ci, co, k_, s_, _pt, _pl, ho, wo) in cnn.mfma.items():
                        if kind == "conv_bwd_data" and ci % 4:
                            continue                                       # no gradient flows into the crops
                        xb, yb = 4.0 * n_ * h_ * w_ * ci, 4.0 * n_ * ho * wo * co
                        # compulsory bytes = every operand map of the launch once: source + destination, plus the maps the fused forms read
                        # where they lie -- the residual a forward adds; the pre-BN map of a data gradient with the fused batch-norm
                        # backward and the earlier contribution it accumulates onto; the y (and the dx it writes) of a weight gradient
                        # that evaluates the batch-norm backward in its loader
                        byt = xb + yb
                        if kind == "conv_fwd" and name in getattr(cnn, "fuse_add", {}):
                            byt += yb
                        if kind == "conv_bwd_data":
                            if name in getattr(cnn, "bnb_conv", {}):
                                byt += xb
                            if src_of.get(name) in getattr(cnn, "acc_ok", ()):
                                byt += xb
                        if kind == "conv_bwd_weight" and name in folded:
                            byt += yb + (yb if src_of.get(name) != "in" else 0.0)
                        t_h, t_m = byt / (HBM_PEAK * 1e3), 2.0 * n_ * ho * wo * k_ * k_ * ci * co / (MFMA_PEAK * 1e6)     # us
                        hb += t_h; mb += t_m; bind += max(t_h, t_m)
                    r.update({"hbm_bound_us": round(hb, 1), "mfma_bound_us": round(mb, 1), "binding_bound_us": round(bind, 1),
                              "frac_of_binding_bound": round(bind / (kinds[kind]["total_ms"] * 1e3), 4)})
                except Exception:
                    pass
            if kind.startswith("rnn_persist"):
                # the recurrent chain is latency-bound -- the meaningful figure is the time per sequential time step
                # (one launch walks the T_a-step layer/time wavefront), next to the ~1.45 us per-launch floor it replaces
                r["sequential_steps"] = TA + 2
                r["us_per_sequential_step"] = round(us / (TA + 2), 3)
                r["note"] = "latency-bound recurrence: one persistent launch for the whole sequence; compare us_per_sequential_step with the ~1.45 us per-launch floor x 3 layers of a launch-per-step design"
            return r

        out["roofline"] = roof(dom)
        out["roofline_other"] = [roof(k) for k in kinds if k != dom and k != "step_dense"]
    if rank == 0 and world == 1 and not args.no_profile and not args.brief:
        # SURVEY 8(d) also asks for the greedy-decode rate: eval graph (no dropout, BN moving statistics), GreedyEmbeddingHelper,
        # T_dec steps per utterance (random weights never emit EOS, so every utterance runs all LDEC steps)
        trace("profile section done")
        try:
            model.greedy_decode(batch, max_steps=LDEC)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                model.greedy_decode(batch, max_steps=LDEC)
            torch.cuda.synchronize()
            dtd = (time.perf_counter() - t0) / 3
            out["greedy_decode"] = {"value": round(B / dtd, 2), "unit": "utterances/sec", "ms_per_batch": round(1e3 * dtd, 3),
                                    "steps": LDEC, "launch": "one launch of all steps where the fused persistent decode kernel takes the block (per-group in-kernel exit once every "
                                              "utterance has emitted EOS); chunks of 8 steps with a host check otherwise"}
        except Exception as e:
            out["greedy_decode"] = {"value": None, "error": repr(e)}
        try:   # the reference's DEFAULT evaluation algorithm: beam search, width 10 (avsr/avsr.py:58-59)
            model.beam_search_decode(batch, beam_width=10, max_steps=LDEC)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                model.beam_search_decode(batch, beam_width=10, max_steps=LDEC)
            torch.cuda.synchronize()
            dtb = (time.perf_counter() - t0) / 3
            out["beam_search_decode"] = {"value": round(B / dtb, 2), "unit": "utterances/sec", "ms_per_batch": round(1e3 * dtb, 3),
                                         "beam_width": 10, "steps": LDEC}
        except Exception as e:
            out["beam_search_decode"] = {"value": None, "error": repr(e)}
    if rank == 0 and world == 1 and not args.no_profile and not args.brief and cfg.video_units is not None:
        # The same workload with the OTHER video input: the headline feeds lip crops through the CNN front-end (north_star's
        # synthetic shape); `without_lip_cnn` is the step on pre-computed 128-d lip features, i.e. the replaced subsystems alone.
        # Runs in a child process: an auxiliary figure must not be able to take the headline line down with it.
        trace("greedy done")
        other = "features" if args.video_frontend == "resnet_cnn" else "resnet_cnn"
        aux_key = "without_lip_cnn" if other == "features" else "with_lip_cnn"
        try:
            import subprocess
            del trainer, model
            torch.cuda.empty_cache()
            cmd = [sys.executable, os.path.abspath(__file__), "--section", "aux_frontend", "--video-frontend", other, "--workload", args.workload,
                   "--batch", str(B)]
            if args.no_graph:
                cmd.append("--no-graph")
            if args.no_dropout:
                cmd.append("--no-dropout")
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            out[aux_key] = json.loads(lines[-1]) if (p.returncode == 0 and lines) else \
                {"value": None, "error": "child exit %d: %s" % (p.returncode, p.stderr.strip().splitlines()[-1][:200] if p.stderr.strip() else "")}
        except Exception as e:
            out[aux_key] = {"value": None, "error": repr(e)}
    if rank == 0 and world == 1 and not args.brief and not args.no_profile and not args.no_other_workloads and args.workload == "c4" and not args.batch:
        # The other BASELINE configs, driver-timed in the same command: the bench line of each (child process, --brief: the timed steps
        # and the per-kernel event pass, nothing else), reduced to its step time, loss and dominant kernel.
        import subprocess
        out["other_workloads"] = {}
        for w in ("c2", "c3", "c5"):
            try:
                cmd = [sys.executable, os.path.abspath(__file__), "--workload", w, "--steps", "10", "--warmup", "3", "--brief"] + \
                      (["--no-graph"] if args.no_graph else []) + (["--no-dropout"] if args.no_dropout else [])
                p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
                if p.returncode != 0 or not lines:
                    raise RuntimeError("child exit %d: %s" % (p.returncode, p.stderr.strip().splitlines()[-1][:200] if p.stderr.strip() else ""))
                r = json.loads(lines[-1])
                rf = r.get("roofline") or {}
                out["other_workloads"][w] = {
                    "workload": r["config"]["workload"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                    "utterances_per_gpu": r["config"]["utterances_per_gpu"], "steps": r["steps"], "launch": r["config"]["launch"],
                    "video_frontend": r["config"]["video_frontend"], "final_loss": r["final_loss"],
                    "persistent_wait_expired": r["persistent_wait_expired"],
                    "dominant_kernel": {k: rf.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us",
                                                                "us_per_sequential_step", "us_per_audio_frame", "us_per_decode_step") if k in rf},
                    "kernel_time_events_ms": {k: v["total_ms"] for k, v in (r.get("kernel_time_events") or {}).items()}}
            except Exception as e:
                out["other_workloads"][w] = {"value": None, "error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.brief:
        # child processes: the CPU oracle is test infrastructure and must not be able to take the headline line down with it; thread
        # count and oneDNN are process-wide, so every leg is its own child.  Main leg = the SAME workload as the timed GPU step (whole
        # batch, same video input); oneDNN on, and once more without it if that child dies (an earlier image's oneDNN convolution
        # backward corrupted the heap on the lip-CNN shapes) -- the JSON says which one ran.
        import subprocess
        env = dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", "0"))

        def leg(section, timeout):
            cmd = [sys.executable, os.path.abspath(__file__), "--section", section, "--video-frontend", args.video_frontend,
                   "--workload", args.workload] + (["--no-dropout"] if args.no_dropout else [])
            try:
                p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
                lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
                if p.returncode == 0 and lines:
                    return json.loads(lines[-1])
                return {"value": None, "error": "child exit %d: %s" % (p.returncode, p.stderr.strip().splitlines()[-1][:200] if p.stderr.strip() else "")}
            except Exception as e:
                return {"value": None, "error": repr(e)}

        main_leg = leg("cpu_baseline", 900)
        if main_leg.get("value") is None:
            first_error = main_leg.get("error")
            main_leg = leg("cpu_baseline_noonednn", 900)
            main_leg["onednn_attempt_error"] = first_error
        out["cpu_baseline"] = main_leg
        out["cpu_baseline"]["four_utterance_sample"] = leg("cpu_baseline_sample", 300)
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        # every rank leaves together: rank 0 alone runs the per-kernel event pass above, and a communicator torn down under a rank that
        # is still working has hung the launcher on some stacks -- one closing barrier, then the teardown (errors there cannot change the
        # line that has already been printed)
        try:
            dist.barrier()
            torch.cuda.synchronize()
            dist.destroy_process_group()
        except Exception as e:      # noqa: BLE001
            sys.stderr.write("[bench] process-group teardown: %r\n" % (e,))


if __name__ == "__main__":
    main()
