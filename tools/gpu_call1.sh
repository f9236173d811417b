#!/bin/bash
# GPU call 1 of round 2: full GPU test-suite, the bench line, and a rocprofv3 kernel-trace of the same command.
set -u
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=15 ) > gpurun_out/r02/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02/pytest_gpu.log
tail -30 gpurun_out/r02/pytest_gpu.log
( time timeout 600 python bench.py ) > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err
echo "bench rc=$?"; cat gpurun_out/r02/bench_default.json | head -c 6000; tail -5 gpurun_out/r02/bench_default.err
( time timeout 300 python bench.py --classes 15 --support 416 --no-cpu-baseline --steps 20 ) > gpurun_out/r02/bench_c2cfg.json 2> gpurun_out/r02/bench_c2cfg.err
echo "bench c2 rc=$?"; head -c 1500 gpurun_out/r02/bench_c2cfg.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_default" -o run -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_default.log" 2>&1
echo "rocprof rc=$?"
cd "$GRAFT_REPO_ROOT"; find gpurun_out/r02/prof_default -name "*kernel_stats.csv" | head; 
f=$(find gpurun_out/r02/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-200
find gpurun_out/r02/prof_default -name "*.db" -size +30M -delete
