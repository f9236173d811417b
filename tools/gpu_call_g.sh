#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_first_bwd.py -q -p no:cacheprovider -x 2>&1 | tail -3
python tools/probes/first_bwd_time.py 2>&1 | grep -v amdgpu
