#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_graphs.py -q -p no:cacheprovider -x 2>&1 | tail -15
timeout 300 python tools/probes/inference_time.py 2>&1 | grep -v "amdgpu.ids\|class_scale"
