#!/bin/bash
# End-of-round evidence on the GPU box (everything lands under gpurun_out/; copy what is to be judged into profiles/):
#   bench lines of c4 (headline) / c2 / c3 / c5, rocprofv3 kernel statistics of an eager c4 step, FETCH_SIZE / WRITE_SIZE / MFMA-busy
#   counter passes (each --pmc pass on its own, with --kernel-trace only).   usage: tools/final_profiles.sh <tag, e.g. r04 v2>
cd "$(dirname "$0")/.." || exit 1
R=${1:-r06}; V=${2:-v1}; O=gpurun_out; export TMPDIR=/tmp
mkdir -p $O/ck
python bench.py > $O/${R}_bench_default_${V}.json 2> $O/${R}_bench_default_${V}.err
for w in c2 c3 c5; do timeout 400 python bench.py --workload $w --no-cpu-baseline --steps 10 > $O/${R}_bench_${w}_${V}.json 2>/dev/null; done
python - <<PY
import json
for w in ("default", "c2", "c3", "c5"):
    try:
        d = json.loads(open("$O/${R}_bench_%s_${V}.json" % w).read().strip().splitlines()[-1])
        print(w, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("beam_search_decode", {}).get("value"))
    except Exception as e:
        print(w, "failed", e)
PY
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ck/ks -o ks -- python bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-profile > $O/ck/ks.log 2>&1; echo "kernel stats rc=$?"
cp "$(find $O/ck/ks -name '*kernel_stats.csv' | head -1)" $O/${R}_c4_lipcnn_eager_${V}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d $O/ck/$c -o pmc -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-profile > $O/ck/$c.log 2>&1; echo "$c rc=$?"
done
F=$(find $O/ck/FETCH_SIZE -name "*.db" | head -1); W=$(find $O/ck/WRITE_SIZE -name "*.db" | head -1)
python tools/pmc_json.py "$F" "$W" $O/${R}_c4_lipcnn_pmc_${V}.json "python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-profile ; workload c4 with lip crops (resnet_cnn front-end); MI355X; round ${R} end-of-round tree"
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/ck/mfma -o pmc -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-profile > $O/ck/mfma.log 2>&1; echo "mfma rc=$?"
python tools/pmc_dump.py "$(find $O/ck/mfma -name '*.db' | head -1)" > $O/${R}_c4_mfma_pmc_${V}.txt 2>&1
head -12 $O/${R}_c4_mfma_pmc_${V}.txt
rm -rf $O/ck
