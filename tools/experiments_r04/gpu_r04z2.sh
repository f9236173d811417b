#!/bin/bash
# inference at B = 2: F(2x2) for the weight-bound 13x13 layers and 64x64 tiles for the tiny 1x1 grids, on / off; the tests that cover them
set -u
mkdir -p gpurun_out; O=gpurun_out/r04z2.txt; : > $O
for i in 1 2; do
  for f2 in 1 0; do
    echo "FSD_INFER_F2=$f2" >> $O
    FSD_INFER_F2=$f2 timeout 200 python tools/probes/inference_time.py 2 >> $O 2>&1
  done
done
FSD_INFER_F2=1 timeout 200 python tools/probes/inference_time.py 4 >> $O 2>&1
(timeout 600 python -m pytest tests/test_gpu_inference.py tests/test_gpu_split.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -5) >> $O
cat $O
