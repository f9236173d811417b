#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05d"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k first_layer ) > "$O/pytest_new.log" 2>&1
echo "pytest rc=$?"; tail -4 "$O/pytest_new.log"
cd /tmp
python "$R/tools/experiments_r05/first_layer_probe.py"
python "$R/tools/experiments_r05/first_layer_probe.py"
