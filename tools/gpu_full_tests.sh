#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -30 $O/pytest_gpu.log
