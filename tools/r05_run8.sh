#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); rs=[d['roofline']]+d['roofline_other']
print('$1', d['ms_per_step'], [(r['kernel'], r.get('us_per_sequential_step')) for r in rs if r['kernel'].startswith('rnn')], d['final_loss'], d['persistent_wait_expired'])"; }
for p in "" "0,0,2,3" "1,1,2,3" "0,0,3,3" "0,0,1,3" "2,2,2,2" "0,0,0,0"; do
  AVSR_RNN_PRIO=$p python bench.py --no-cpu-baseline --steps 20 2>/dev/null | line "prio[$p]"
done
python bench.py --no-cpu-baseline --steps 20 2>/dev/null | line "prio[] again"
