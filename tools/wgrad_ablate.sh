#!/bin/bash
# Weight-gradient ablation on the GPU box: csrc/conv_wgrad.hip (the weight-gradient kernel; conv_mfma.hip before the round-6 split) rebuilt with -DWG_ABLATE=<mask> (there only), the kernel timed per layer.
# usage: tools/wgrad_ablate.sh H,Ci,Co ...
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd); C=$ROOT/avsr-tf1_amd/csrc
run() {
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-pass-failed $1 -I $ROOT/include -I $C -c $C/conv_wgrad.hip -o $C/conv_wgrad.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libavsr_hip.so $C/*.o || exit 1
  shift
  python tools/wgrad_ablate.py "$@" 2>&1 | grep -v amdgpu.ids
}
run "" "as shipped" "$@"
run "-DWG_ABLATE=1" "no dy loads" "$@"
run "-DWG_ABLATE=2" "no LDS operand reads" "$@"
run "-DWG_ABLATE=4" "no MFMAs" "$@"
run "-DWG_ABLATE=16" "no frame staging" "$@"
run "-DWG_ABLATE=3" "no dy loads, no LDS reads" "$@"
run "-DWG_ABLATE=6" "no LDS reads, no MFMAs" "$@"
run "-DWG_ABLATE=17" "no dy loads, no staging" "$@"
run "-DWG_ABLATE=7" "no dy / LDS reads / MFMAs" "$@"
run "-DWG_ABLATE=23" "only the loop skeleton" "$@"
