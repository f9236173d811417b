#!/bin/bash
# 32x128 tile for <= 32-row position GEMMs: tests, two-image forward on / off, train step on / off
set -u
mkdir -p gpurun_out; O=gpurun_out/r04z5.txt; : > $O
(timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_kernels.py tests/test_gpu_inference.py tests/test_gpu_backward.py tests/test_gpu_timed_config.py -x -q 2>&1 | tail -4) >> $O
for i in 1 2; do
  for t in 0 1; do
    echo "FSD_WINO_TILE32=$t" >> $O
    FSD_WINO_TILE32=$t timeout 200 python tools/probes/inference_time.py 2 2>&1 | grep "B=" >> $O
  done
done
bash tools/gpu_ab.sh "FSD_WINO_TILE32=0" "FSD_WINO_TILE32=1" >> $O 2>&1
cat $O
