#!/bin/bash
# Dissection of the deep lip-CNN layers on the GPU box: csrc/conv_mfma.hip rebuilt with -DCONV_DEBUG, tools/conv_dissect.py per layer and
# ablation.   usage: tools/conv_deep_dissect.sh
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd); C=$ROOT/avsr-tf1_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-pass-failed -DCONV_DEBUG -I $ROOT/include -I $C -c $C/conv_mfma.hip -o $C/conv_mfma.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libavsr_hip.so $C/*.o || exit 1
for L in "5,64,64,3,1" "9,32,32,3,1" "9,32,64,3,2" "18,16,16,3,1" "36,8,8,3,1"; do
  for D in 0 8 24; do AVSR_DISSECT_LAYER=$L AVSR_CONV_DBG=$D python tools/conv_dissect.py 2>&1 | grep -v amdgpu.ids; done
done
