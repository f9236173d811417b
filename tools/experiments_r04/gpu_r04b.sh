#!/bin/bash
# Round 4, second GPU call: the 8-wave 256x128 split GEMM (FSD_WINO_SPLIT8) -- parity tests, then per-layer A/B.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04b"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_timed_config.py -x -q 2>&1 | tail -5
for k in 0 1; do
  echo "--- fwd FSD_WINO_SPLIT8=$k"; FSD_WINO_SPLIT8=$k timeout 200 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tee "$O/lb_fwd_k$k.txt" | sed -n '2,8p'
  echo "--- dgrad FSD_WINO_SPLIT8=$k"; FSD_LB_SWAP=1 FSD_WINO_SPLIT8=$k timeout 200 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | tee "$O/lb_dgrad_k$k.txt"
done
