#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_timed_config.py -x -q 2>&1 | tail -3
for v in 0 1 0 1; do
  echo "--- fwd FSD_SPLIT8_PERSIST=$v"; FSD_SPLIT8_PERSIST=$v timeout 200 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids | sed -n '5,8p'
done
bash tools/gpu_ab.sh FSD_SPLIT8_PERSIST=1 FSD_SPLIT8_PERSIST=0
