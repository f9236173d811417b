#!/bin/bash
# row-fused Winograd kernel (fsd_wino_fused_mode 2): contract test, per layer against the three launches
set -u
mkdir -p gpurun_out; O=gpurun_out/r04g1.txt; : > $O
(timeout 300 python -m pytest tests/test_gpu_split.py -x -q -k experimental 2>&1 | tail -12) >> $O
for f in 0 2; do
  echo "== FSD_WINO_FUSED=$f" >> $O
  FSD_WINO_FUSED=$f FSD_LB_ONLY=104,64,128 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
  FSD_WINO_FUSED=$f FSD_LB_ONLY=52,128,256 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
done
cat $O
