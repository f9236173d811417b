#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_inference.py tests/test_gpu_graphs.py tests/test_gpu_kernels.py tests/test_gpu_bf16.py -q -p no:cacheprovider -x 2>&1 | tail -12
timeout 300 python tools/probes/inference_time.py 2>&1 | grep -v "amdgpu.ids\|class_scale"
