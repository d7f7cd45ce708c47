#!/bin/bash
# Every launch mode / opt-in switch of the training step at the FULL benchmark shape (the unit tests run them at small shapes): one line each.
# usage: tools/fullsize_modes.sh > gpurun_out/fullsize_modes.txt   (one GPU; the two-rank runs share it over gloo)
cd "$(dirname "$0")/.."
export AVSR_BENCH_TRAFFIC=0
line() { python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print('%-44s %8.3f ms %9.1f utt/s  %-28s expired=%s loss=%.5f' % (sys.argv[1], d['ms_per_step'], d['value'], d['config']['launch'][:28], d['persistent_wait_expired'], d['final_loss']))
except Exception as e:
    print('%-44s FAILED (%s)' % (sys.argv[1], e))" "$1"; }
one() { label="$1"; shift; timeout 600 python bench.py --steps 6 --warmup 2 --no-profile --no-cpu-baseline "$@" 2>gpurun_out/fm_err.txt | line "$label"; }
two() { label="$1"; shift; AVSR_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 200)) bench.py --gpus 2 --steps 6 --warmup 2 --no-profile --no-cpu-baseline "$@" 2>gpurun_out/fm_err.txt | line "$label"; }
one "c4 default"
one "c4 --no-graph" --no-graph
one "c4 --no-dropout" --no-dropout
one "c4 features front-end" --video-frontend features
one "c4 --batch 128" --batch 128
one "c4 --batch 256" --batch 256
one "c4 --batch 1" --batch 1
one "c4 --batch 7" --batch 7
one "c5 --batch 128" --workload c5 --batch 128
one "c2 --batch 256" --workload c2 --batch 256
one "c3 --batch 256" --workload c3 --batch 256
AVSR_PERSISTENT_RNN=0 one "c4 AVSR_PERSISTENT_RNN=0"
AVSR_CNN_FOLD=0 one "c4 AVSR_CNN_FOLD=0"
AVSR_RNN_BWD_WIDE=1 one "c4 AVSR_RNN_BWD_WIDE=1"
two "2 ranks default"
two "2 ranks --strong" --strong
two "2 ranks --no-graph" --no-graph
AVSR_DP_OVERLAP=1 two "2 ranks AVSR_DP_OVERLAP=1"
AVSR_DP_DRAIN=0 two "2 ranks AVSR_DP_DRAIN=0"
AVSR_DP_OVERLAP=1 AVSR_DP_DRAIN=0 two "2 ranks OVERLAP=1 DRAIN=0"
AVSR_DP_SYNC_CNN_BN=1 two "2 ranks AVSR_DP_SYNC_CNN_BN=1"
AVSR_DP_SYNC_CNN_BN=1 two "2 ranks SYNC_CNN_BN=1 --strong" --strong
two "2 ranks c5" --workload c5
two "2 ranks c2" --workload c2
