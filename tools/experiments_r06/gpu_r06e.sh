#!/bin/bash
# round 6, fifth GPU call: tape released by the backward pass -> allocation traces; new tests; then the evidence runs
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06e; mkdir -p $O
( timeout 600 python tools/alloc_trace.py ) > $O/alloc_trace_f32.txt 2> $O/alloc_trace_f32.err; echo "alloc f32 rc=$?"; head -3 $O/alloc_trace_f32.txt | cut -c1-250; tail -2 $O/alloc_trace_f32.txt | cut -c1-300
( timeout 600 python tools/alloc_trace.py --dtype bf16 ) > $O/alloc_trace_bf16.txt 2> $O/alloc_trace_bf16.err; echo "alloc bf16 rc=$?"; head -3 $O/alloc_trace_bf16.txt | cut -c1-250; tail -2 $O/alloc_trace_bf16.txt | cut -c1-300
( timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_model.py tests/test_gpu_streams.py tests/test_gpu_launch_configs.py -m gpu -q -p no:cacheprovider -k "tape or timed_region or multiscale or streams or backward" --durations=6 ) > $O/pytest_new.log 2>&1
echo "new tests rc=$?"; tail -14 $O/pytest_new.log
bash tools/gpu_evidence_r06.sh
