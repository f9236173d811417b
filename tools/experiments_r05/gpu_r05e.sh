#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; cd /tmp; export TMPDIR=/tmp
for d in 0 1 2 3; do echo "FSD_FIRST_DBG=$d"; FSD_FIRST_DBG=$d python "$R/tools/experiments_r05/first_layer_probe.py"; done
echo old kernel; FSD_FIRST_SPLIT=0 python "$R/tools/experiments_r05/first_layer_probe.py"
