#!/bin/bash
set -u
mkdir -p gpurun_out; O=gpurun_out/r04f2.txt; : > $O
(timeout 300 python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -2) >> $O
for f in 1; do
  echo "== FSD_WINO_FUSED=$f fwd 104,64,128 / 52,128,256" >> $O
  FSD_WINO_FUSED=$f FSD_LB_ONLY=104,64,128 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
  FSD_WINO_FUSED=$f FSD_LB_ONLY=52,128,256 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
done
cat $O
