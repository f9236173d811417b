#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; cd /tmp
export FSD_LB_ONLY=208,32,64
for d in 0 8 16; do echo "FSD_WH_DBG=$d"; FSD_WH_DBG=$d python "$R/tools/layer_bench.py" wgrad 2>&1 | grep -v amdgpu | tail -1; done
