#!/bin/bash
# round 6, first GPU call: host facts, the new parity tests, the whole GPU suite, the driver's bench command, allocation trace
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06a; mkdir -p $O
( nproc; free -g; rocm-smi --showmeminfo vram | head -8 ) > $O/host.txt 2>&1
( time timeout 1500 python -m pytest tests/test_gpu_launch_configs.py -m gpu -q -x -s -p no:cacheprovider --durations=20 ) > $O/launch_configs.log 2>&1
echo "launch_configs rc=$?"; grep -E "passed|failed|error|Error|assert" $O/launch_configs.log | tail -12
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 --deselect tests/test_gpu_launch_configs.py ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 3000 $O/bench.json
( timeout 600 python tools/alloc_trace.py ) > $O/alloc_trace_f32.txt 2> $O/alloc_trace_f32.err
echo "alloc trace rc=$?"; head -40 $O/alloc_trace_f32.txt
cat $O/host.txt
