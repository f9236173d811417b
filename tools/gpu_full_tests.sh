#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
( time timeout 600 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -16 $O/pytest_gpu.log
