#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05f"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( timeout 900 python -m pytest tests/test_gpu_split.py -m gpu -q -p no:cacheprovider ) > "$O/pytest_new.log" 2>&1
echo "pytest rc=$?"; tail -3 "$O/pytest_new.log"
cd /tmp
export FSD_LB_ONLY=208,32,64
python "$R/tools/layer_bench.py" wgrad 2>&1 | tail -1
python "$R/tools/layer_bench.py" wgrad 2>&1 | tail -1
