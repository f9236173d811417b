#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04i"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_timed_config.py -x -q 2>&1 | tail -3
for v in 0 1; do
  echo "--- fwd FSD_SPLIT8_TAIL=$v"; FSD_SPLIT8_TAIL=$v timeout 200 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | sed -n '4,8p'
  echo "--- dgrad FSD_SPLIT8_TAIL=$v"; FSD_LB_SWAP=1 FSD_SPLIT8_TAIL=$v timeout 200 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | sed -n '2,6p'
done
( timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline ) > "$O/bench.out" 2> "$O/bench.err"; tail -c 2500 "$O/bench.out" | cut -c1-330
