#!/bin/bash
# round 6, second GPU call: the tests that failed / are new, allocation trace after the settle-phase fix, bench lines (f32, bf16)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06b; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_kernels.py tests/test_gpu_dp.py tests/test_gpu_backward.py tests/test_gpu_model.py tests/test_gpu_inference.py tests/test_gpu_graphs.py -m gpu -q -p no:cacheprovider --durations=8 ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
( timeout 600 python tools/alloc_trace.py ) > $O/alloc_trace_f32.txt 2> $O/alloc_trace_f32.err
echo "alloc trace rc=$?"; head -12 $O/alloc_trace_f32.txt | cut -c1-300
( timeout 600 python tools/alloc_trace.py --dtype bf16 ) > $O/alloc_trace_bf16.txt 2> $O/alloc_trace_bf16.err
echo "alloc trace bf16 rc=$?"; head -12 $O/alloc_trace_bf16.txt | cut -c1-300
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 2500 $O/bench.json
