#!/bin/bash
set -u
mkdir -p gpurun_out; O=gpurun_out/r04f4.txt; : > $O
(timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_kernels.py tests/test_gpu_backward.py -x -q 2>&1 | tail -3) >> $O
for f in 1 0; do
  echo "== fused=$f: fwd 104,64,128 / 52,128,256" >> $O
  FSD_WINO_FUSED=$f FSD_LB_ONLY=104,64,128 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
  FSD_WINO_FUSED=$f FSD_LB_ONLY=52,128,256 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
done
bash tools/gpu_ab.sh "FSD_WINO_FUSED=0" "FSD_WINO_FUSED=1" >> $O 2>&1
cat $O
