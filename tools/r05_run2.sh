#!/bin/bash
# run 2: phase tables (before), hand-off protocol probe, RNN_XPRE variant of the persistent forward
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/handoff_probe.hip -o /tmp/handoff_probe 2>/dev/null && timeout 300 /tmp/handoff_probe > $O/r05_handoff_probe.txt 2>&1
cat $O/r05_handoff_probe.txt
tools/r05_baseline.sh before
echo "== RNN_XPRE"
AVSR_HIPCC_FLAGS="-DRNN_XPRE" python -m avsr_tf1_amd.build > /dev/null 2>&1
for i in 1 2; do python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xpre', d['ms_per_step'], [ (r['kernel'], r.get('us_per_sequential_step')) for r in [d['roofline']]+d['roofline_other'] if r['kernel'].startswith('rnn')], d['final_loss'], d['persistent_wait_expired'])"; done
R05_EXTRA_FLAGS="-DRNN_XPRE" tools/r05_baseline.sh xpre > /dev/null 2>&1; grep -E "^fwd" $O/r05_phase_ticks_xpre.txt
python -m avsr_tf1_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_beam.py -x -q 2>&1 | tail -5
