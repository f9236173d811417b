#!/bin/bash
set -u
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > gpurun_out/r02/pytest_gpu3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02/pytest_gpu3.log
grep -E "passed|failed|FAILED|rc=" gpurun_out/r02/pytest_gpu3.log | head
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_bf16" -o run -- python "$GRAFT_REPO_ROOT/bench.py" --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_bf16.log" 2>&1
echo "rocprof rc=$?"
