#!/bin/bash
# A/B of library variants built by tools/build_variant.py: tools/ab_variants.sh "<tag> <tag> ..." [workload] [rounds]
# every variant is timed `rounds` times, interleaved (box drift affects all alike); prints the per-class microseconds of tools/kind_times.py
cd "$(dirname "$0")/.." || exit 1
TAGS=${1:-"base"}; W=${2:-c4}; R=${3:-2}
for r in $(seq 1 $R); do
  for t in $TAGS; do
    echo "== $t (round $r)"
    AVSR_LIB=$PWD/avsr-tf1_amd/csrc/_probe/libavsr_hip_$t.so python tools/kind_times.py $W 10 2 2>&1 | grep -E "rep|expired"
  done
done
