#!/bin/bash
set -u
mkdir -p gpurun_out; O=gpurun_out/r04f5.txt; : > $O
for d in 0 1 2 3 4 8 16 12 28 31; do
  echo "dbg=$d" >> $O
  FSD_WINO_FUSED_DBG=$d FSD_LB_ONLY=104,64,128 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
done
cat $O
