#!/bin/bash
# slab-wise Winograd pipeline (V + M of a slab fit the memory-side cache): tests, per layer, train step A/B
set -u
mkdir -p gpurun_out; O=gpurun_out/r04s1.txt; : > $O
(FSD_WINO_SLAB_MB=2 timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_kernels.py tests/test_gpu_backward.py -x -q 2>&1 | tail -3) >> $O
for mb in 0 96 48 160; do
  echo "== slab MB $mb" >> $O
  FSD_WINO_SLAB_MB=$mb FSD_LB_ONLY=104,64,128 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
  FSD_WINO_SLAB_MB=$mb FSD_LB_ONLY=52,128,256 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
  FSD_WINO_SLAB_MB=$mb FSD_LB_ONLY=26,256,512 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
done
bash tools/gpu_ab.sh "FSD_WINO_SLAB_MB=0" "FSD_WINO_SLAB_MB=96" >> $O 2>&1
cat $O
