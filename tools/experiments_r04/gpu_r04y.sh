#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_dp.py tests/test_gpu_backward.py tests/test_gpu_drivers.py -x -q 2>&1 | tail -3
bash tools/gpu_ab.sh FSD_PREPACK=1 FSD_PREPACK=0
bash tools/gpu_ab.sh FSD_PREPACK=1 FSD_PREPACK=0
