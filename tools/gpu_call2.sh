#!/bin/bash
set -u
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 -s ) > gpurun_out/r02/pytest_gpu2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02/pytest_gpu2.log
grep -E "B=64 C2|passed|failed|FAILED|Error|rc=" gpurun_out/r02/pytest_gpu2.log | head -40
tail -25 gpurun_out/r02/pytest_gpu2.log
