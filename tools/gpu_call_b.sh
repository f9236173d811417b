#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02s; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_streams.py "tests/test_gpu_bf16.py::test_wgrad_bf16_transpose_read_kernel_matches_fp64" -q -p no:cacheprovider ) > $O/pytest_b.log 2>&1
echo "pytest rc=$?"; tail -25 $O/pytest_b.log
for kc in 32 64; do
  FSD_WGRAD_H_KC=$kc FSD_LB_DTYPE=bf16 timeout 300 python tools/layer_bench.py wgrad > $O/lb_wgrad_kc$kc.log 2>&1; echo "kc=$kc"; cat $O/lb_wgrad_kc$kc.log | tail -12
done
