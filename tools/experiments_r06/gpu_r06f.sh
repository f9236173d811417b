#!/bin/bash
# round 6: the whole GPU suite as the driver runs it, with durations
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06f; mkdir -p $O
( time timeout 3000 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=15 ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -30 $O/pytest_gpu.log
( time timeout 600 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
