#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03c"; mkdir -p "$O"; cd "$R"
python -m pytest tests/test_gpu_bf16.py -x -q -m gpu 2>&1 | tail -4 | tee "$O/tests.log"
for cfg in "0 0" "0 1" "2 1" "6 1" "1 1"; do set -- $cfg
  echo "== tile $1 wide $2" | tee -a "$O/layers.log"
  FSD_CONV_H_WIDE=$2 FSD_CONV_H_TILE=$1 FSD_LB_DTYPE=bf16 python tools/layer_bench.py fwd 2>&1 | grep -v "class_scale\|amdgpu.ids" | tee -a "$O/layers.log"
done
