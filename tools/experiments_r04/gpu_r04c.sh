#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04c"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
timeout 900 python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_timed_config.py -q -k "teacher or wino_gemm" 2>&1 | tail -4
for v in "0 1" "1 0" "1 1"; do set -- $v
  echo "--- fwd SPLIT8=$1 ILV=$2"; FSD_WINO_SPLIT8=$1 FSD_SPLIT8_ILV=$2 timeout 200 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tee "$O/lb_fwd_$1$2.txt" | sed -n '2,8p'
done
echo "--- dgrad SPLIT8=1 ILV=1"; FSD_LB_SWAP=1 timeout 200 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tee "$O/lb_dgrad_11.txt"
