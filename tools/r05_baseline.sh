#!/bin/bash
# Round-5 "before" evidence: phase tables of the persistent encoder kernels (c4) and of the AV-Align attentive layer (c5), the fused decoder's
# ticks, and the bench lines of the tree the round starts from.  Everything under gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; T=${1:-before}; mkdir -p $O; export TMPDIR=/tmp
python bench.py > $O/r05_bench_default_$T.json 2> $O/r05_bench_default_$T.err
timeout 400 python bench.py --workload c5 --no-cpu-baseline --steps 10 > $O/r05_bench_c5_$T.json 2>/dev/null
export AVSR_HIPCC_FLAGS="-DPERSIST_TIMING -DDP_TIMING $R05_EXTRA_FLAGS"
python -m avsr_tf1_amd.build > $O/build_timing.log 2>&1 || { echo "timing build failed"; tail -5 $O/build_timing.log; }
{ echo "== c4: persistent encoder kernels, ticks per sequential step (shader clock; PERSIST_TIMING build, timing build adds a vmcnt(0) per step)";
  timeout 300 python tools/persist_probe.py 2>&1 | grep -E '^(err|fwd|bwd) ';
  echo "== c5: AV-Align attentive layer (DP_TIMING) + encoder kernels";
  timeout 300 python tools/c5_phase_probe.py 2>&1 | grep -E '^(err|attentive|encoder) ';
  echo "== c4 fused decoder (tools/fused_check.py --full)";
  timeout 300 python tools/fused_check.py --full 2>&1 | grep -E "ticks|us/step|backward pass"; } > $O/r05_phase_ticks_$T.txt 2>&1
unset AVSR_HIPCC_FLAGS
python -m avsr_tf1_amd.build > /dev/null 2>&1      # back to the shipped build
cat $O/r05_phase_ticks_$T.txt
python - <<PY
import json
for w in ("default", "c5"):
    try:
        d = json.loads(open("$O/r05_bench_%s_$T.json" % w).read().strip().splitlines()[-1])
        print(w, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"])
    except Exception as e:
        print(w, "failed", e)
PY
