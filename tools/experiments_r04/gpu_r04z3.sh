#!/bin/bash
# split-K of the short batched position GEMMs: tests, inference B = 2 / 4 with and without, train-step A/B
set -u
mkdir -p gpurun_out; O=gpurun_out/r04z3.txt; : > $O
(timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_kernels.py tests/test_gpu_inference.py tests/test_gpu_backward.py -x -q 2>&1 | tail -4) >> $O
for i in 1 2; do
  for ks in 0 1; do
    if [ $ks = 0 ]; then export FSD_KSPLIT=0; else unset FSD_KSPLIT; fi
    echo "ksplit=$ks" >> $O
    timeout 200 python tools/probes/inference_time.py 2 2>&1 | grep "B=" >> $O
  done
done
unset FSD_KSPLIT
timeout 200 python tools/probes/inference_time.py 4 2>&1 | grep "B=" >> $O
FSD_KSPLIT=0 timeout 200 python tools/probes/inference_time.py 4 2>&1 | grep "B=" >> $O
bash tools/gpu_ab.sh "FSD_KSPLIT=0" "FSD_NOOP=1" >> $O 2>&1
cat $O
