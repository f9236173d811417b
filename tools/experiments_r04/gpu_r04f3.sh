#!/bin/bash
set -u
mkdir -p gpurun_out; O=gpurun_out/r04f3.txt; : > $O
(timeout 300 python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -2) >> $O
for w in 4 8; do
  echo "== waves $w: fwd 104,64,128 / 52,128,256 / dgrad 104 128->64" >> $O
  FSD_WINO_FUSED_WAVES=$w FSD_LB_ONLY=104,64,128 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
  FSD_WINO_FUSED_WAVES=$w FSD_LB_ONLY=52,128,256 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
  FSD_WINO_FUSED_WAVES=$w FSD_LB_SWAP=1 FSD_LB_ONLY=104,128,64 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
done
echo "== unfused dgrad 104 128->64" >> $O
FSD_WINO_FUSED=0 FSD_LB_SWAP=1 FSD_LB_ONLY=104,128,64 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
cat $O
