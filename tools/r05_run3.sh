#!/bin/bash
# run 3: pending parity tests on the default build, then A/B of the two chain-shortening switches
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); rs=[d['roofline']]+d['roofline_other']
print('$1', d['ms_per_step'], [(r['kernel'], r.get('us_per_sequential_step') or r.get('us_per_audio_frame') or r.get('us_per_decode_step')) for r in rs if r['kernel'].startswith(('rnn','align','dec_persist'))], d['final_loss'], d['persistent_wait_expired'], d.get('greedy_decode',{}).get('value'), d.get('beam_search_decode',{}).get('value'))"; }
python -m avsr_tf1_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_fused.py tests/test_gpu_dp.py -x -q -k "phoneme or vocab or synchronised or lip_cnn" 2>&1 | tail -4
python -m pytest tests/test_gpu_model.py -x -q -k "greedy or full_length" 2>&1 | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --steps 20 2>/dev/null | line base; done
python bench.py --workload c5 --no-cpu-baseline --steps 10 2>/dev/null | line base_c5
echo "== RNN_XSKEW"
AVSR_HIPCC_FLAGS="-DRNN_XSKEW" python -m avsr_tf1_amd.build > /dev/null 2>&1
for i in 1 2; do python bench.py --no-cpu-baseline --steps 20 2>/dev/null | line xskew; done
python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -3
echo "== DP_HPRE"
AVSR_HIPCC_FLAGS="-DDP_HPRE" python -m avsr_tf1_amd.build > /dev/null 2>&1
for i in 1 2; do python bench.py --no-cpu-baseline --steps 20 2>/dev/null | line hpre; done
python bench.py --workload c5 --no-cpu-baseline --steps 10 2>/dev/null | line hpre_c5
python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3
python -m avsr_tf1_amd.build > /dev/null 2>&1
