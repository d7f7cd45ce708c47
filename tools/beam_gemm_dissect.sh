#!/bin/bash
# Dissection of beam_gemm_kernel on the GPU box: rebuilds csrc/beam_gemm.hip with parts taken away and reports the kernels' average
# durations from a rocprofv3 kernel trace of a width-10 beam search over the benchmark batch.   usage: tools/beam_gemm_dissect.sh
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd); C=$ROOT/avsr-tf1_amd/csrc
for V in "" "-DBG_NO_MFMA" "-DBG_NO_LOAD" "-DBG_NO_MFMA -DBG_NO_LOAD"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-pass-failed $V -I $ROOT/include -I $C -c $C/beam_gemm.hip -o $C/beam_gemm.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libavsr_hip.so $C/*.o || exit 1
  D=/tmp/bgd_$RANDOM
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python $ROOT/tools/decode_profile.py beam 3 > /dev/null 2>&1)
  echo "== variant: ${V:-as shipped}"
  python - "$D" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "beam_" in r["Name"] or "attn_fwd_beam" in r["Name"]:
        print("  %-60s calls %4d  avg %7.1f us" % (r["Name"][:60], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
done
# beam_step_kernel cut short after phase 1 (loads) / 2 (log-sum-exp, penalties) / 3 (candidate scores); results are garbage, timing only
for V in 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-pass-failed -I $ROOT/include -I $C -c $C/beam_gemm.hip -o $C/beam_gemm.o || exit 1
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-pass-failed -DBS_STOP=$V -I $ROOT/include -I $C -c $C/attn_rnn.hip -o $C/attn_rnn.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libavsr_hip.so $C/*.o || exit 1
  D=/tmp/bsd_$RANDOM
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python $ROOT/tools/decode_profile.py beam 2 > /dev/null 2>&1)
  echo "== beam_step_kernel stopped after phase $V"
  python - "$D" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "beam_step" in r["Name"]:
        print("  %-60s calls %4d  avg %7.1f us" % (r["Name"][:60], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
done
