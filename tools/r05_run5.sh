#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); rs=[d['roofline']]+d['roofline_other']
print('$1', d['ms_per_step'], [(r['kernel'], r.get('us_per_sequential_step') or r.get('us_per_audio_frame') or r.get('us_per_decode_step')) for r in rs if r['kernel'].startswith(('rnn','align'))], d['final_loss'], d['persistent_wait_expired'])"; }
python -m pytest tests/test_gpu_dp.py -x -q -k "synchronised" 2>&1 | tail -3
for i in 1 2; do AVSR_RNN_BWD_WIDE=0 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | line narrow; done
for i in 1 2; do python bench.py --no-cpu-baseline --steps 20 2>/dev/null | line wide; done
AVSR_RNN_BWD_WIDE=0 python bench.py --workload c5 --no-cpu-baseline --steps 10 2>/dev/null | line narrow_c5
python bench.py --workload c5 --no-cpu-baseline --steps 10 2>/dev/null | line wide_c5
AVSR_RNN_BWD_WIDE=0 python bench.py --workload c2 --no-cpu-baseline --steps 10 2>/dev/null | line narrow_c2
python bench.py --workload c2 --no-cpu-baseline --steps 10 2>/dev/null | line wide_c2
python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
