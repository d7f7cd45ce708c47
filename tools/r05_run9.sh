#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); rs=[d['roofline']]+d['roofline_other']
print('$1', d['ms_per_step'], [(r['kernel'], r.get('us_per_sequential_step')) for r in rs if r['kernel'].startswith('rnn')], d['final_loss'], d['persistent_wait_expired'])"; }
python -m avsr_tf1_amd.build > /dev/null 2>&1
for i in 1 2 3; do AVSR_BENCH_TRAFFIC=0 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | line base; done
AVSR_HIPCC_FLAGS="-DRNN_ROTATE" python -m avsr_tf1_amd.build > /dev/null 2>&1
for i in 1 2 3; do AVSR_BENCH_TRAFFIC=0 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | line rotate; done
AVSR_BENCH_TRAFFIC=0 python bench.py --workload c2 --no-cpu-baseline --steps 10 2>/dev/null | line rotate_c2
AVSR_BENCH_TRAFFIC=0 python bench.py --workload c5 --no-cpu-baseline --steps 10 2>/dev/null | line rotate_c5
python -m pytest tests/test_gpu_model.py -x -q -k "full_length or stoch or c4" 2>&1 | tail -2
python -m avsr_tf1_amd.build > /dev/null 2>&1
