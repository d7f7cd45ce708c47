#!/bin/bash
# A/B of the weight-gradient kernels' workgroups per CU: AVSR_WG_WPC=2 (the round-4 cap) against the per-form cap (3 / 4 where registers and LDS allow)
cd "$(dirname "$0")/.." || exit 1
export AVSR_BENCH_TRAFFIC=0
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x 2>&1 | tail -2
for P in 2 4; do
  echo "== AVSR_WG_WPC=$P"
  (cd /tmp && export TMPDIR=/tmp && AVSR_WG_WPC=$P rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks$P -o ks -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-profile > /tmp/ks$P.log 2>&1)
  python tools/kstats.py /tmp/ks$P wgrad 6
done
for P in 2 4; do AVSR_WG_WPC=$P python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AVSR_WG_WPC=$P', d['ms_per_step'], d['value'], d['final_loss'])"; done
