#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); rs=[d['roofline']]+d['roofline_other']
print('$1', d['ms_per_step'], [(r['kernel'], r.get('us_per_sequential_step') or r.get('us_per_audio_frame') or r.get('us_per_decode_step')) for r in rs if r['kernel'].startswith(('rnn','align','dec_persist'))], d['final_loss'], d['persistent_wait_expired'])"; }
python -m avsr_tf1_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_dp.py -x -q -k "synchronised" 2>&1 | tail -40
echo "== RNN_XSKEW=3"
AVSR_HIPCC_FLAGS="-DRNN_XSKEW=3" python -m avsr_tf1_amd.build > /dev/null 2>&1
for i in 1 2; do python bench.py --no-cpu-baseline --steps 20 2>/dev/null | line xskew3; done
python -m avsr_tf1_amd.build > /dev/null 2>&1
