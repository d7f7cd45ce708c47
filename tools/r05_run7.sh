#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_time_events']
print('$1', d['ms_per_step'], {n: k[n]['total_ms'] for n in k if n.startswith('conv') or n.startswith('bn')}, d['final_loss'], d['persistent_wait_expired'])"; }
python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -3
for i in 1 2; do AVSR_CNN_FOLD=0 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | line nofold; done
for i in 1 2; do AVSR_CNN_FOLD=1 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | line fold1; done
for i in 1 2; do python bench.py --no-cpu-baseline --steps 20 2>/dev/null | line fold2; done
python -m pytest tests/test_gpu_model.py tests/test_gpu_dp.py -x -q -k "cnn" 2>&1 | tail -3
