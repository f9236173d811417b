#!/bin/bash
# fused position GEMMs + output transform (K = 64 / 128): per layer and in the step, on / off
set -u
mkdir -p gpurun_out; O=gpurun_out/r04f1.txt; : > $O
for f in 0 1; do
  echo "== FSD_WINO_FUSED=$f fwd 104,64,128 / 52,128,256; dgrad 104 128->64" >> $O
  FSD_WINO_FUSED=$f FSD_LB_ONLY=104,64,128 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
  FSD_WINO_FUSED=$f FSD_LB_ONLY=52,128,256 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
  FSD_WINO_FUSED=$f FSD_LB_SWAP=1 FSD_LB_ONLY=104,128,64 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
done
bash tools/gpu_ab.sh "FSD_WINO_FUSED=0" "FSD_WINO_FUSED=1" >> $O 2>&1
cat $O
