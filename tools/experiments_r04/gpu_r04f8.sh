#!/bin/bash
set -u
mkdir -p gpurun_out; O=gpurun_out/r04f8.txt; : > $O
for t in c a; do for ns in 0 1; do
  echo "tile=$t nostore=$ns" >> $O
  FSD_WINO_TILE=$t FSD_GEMM_NOSTORE=$ns FSD_LB_ONLY=104,64,128 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
  FSD_WINO_TILE=$t FSD_GEMM_NOSTORE=$ns FSD_LB_ONLY=52,128,256 timeout 100 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
done; done
cat $O
