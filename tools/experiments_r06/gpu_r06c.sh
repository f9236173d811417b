#!/bin/bash
# round 6, third GPU call: B-direct bf16 kernels (tests + per-layer A/B), allocator policy, bf16 error growth table, 8-rank DP
# tests, Winograd-vs-direct weight-gradient table
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x -p no:cacheprovider -k "b_direct or fused_sgd or one_ulp" ) > $O/pytest_bdir.log 2>&1
echo "bdir tests rc=$?"; tail -5 $O/pytest_bdir.log
( FSD_LB_DTYPE=bf16 FSD_LB_BDIR=1 timeout 600 python tools/layer_bench.py fwd ) > $O/layer_bench_bf16_bdir.txt 2>&1
echo "layer bench bdir rc=$?"; cat $O/layer_bench_bf16_bdir.txt
( timeout 600 python tools/bf16_error_growth.py --out $O/r06_bf16_error_growth ) > $O/error_growth.txt 2>&1
echo "error growth rc=$?"; tail -45 $O/error_growth.txt
( timeout 600 python tools/alloc_trace.py ) > $O/alloc_trace_f32.txt 2> $O/alloc_trace_f32.err
echo "alloc trace rc=$?"; head -4 $O/alloc_trace_f32.txt | cut -c1-300; tail -2 $O/alloc_trace_f32.txt | cut -c1-400
( timeout 600 python tools/alloc_trace.py --dtype bf16 ) > $O/alloc_trace_bf16.txt 2> $O/alloc_trace_bf16.err
echo "alloc trace bf16 rc=$?"; head -4 $O/alloc_trace_bf16.txt | cut -c1-300; tail -2 $O/alloc_trace_bf16.txt | cut -c1-400
( FSD_LB_WGRAD_DIRECT=1 timeout 600 python tools/layer_bench.py wgrad ) > $O/layer_bench_wgrad_direct.txt 2>&1
echo "layer bench wgrad rc=$?"; cat $O/layer_bench_wgrad_direct.txt
( time timeout 1800 python -m pytest tests/test_gpu_dp.py -m gpu -q -p no:cacheprovider -k "eight" --durations=5 ) > $O/pytest_dp8.log 2>&1
echo "dp8 rc=$?"; tail -12 $O/pytest_dp8.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 1800 $O/bench.json
