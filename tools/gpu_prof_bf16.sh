#!/bin/bash
# rocprofv3 kernel trace of the bf16 training step; prints the per-step kernel table
set -u
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_bf16"
timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_bf16" -o run -- python "$GRAFT_REPO_ROOT/bench.py" --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-parity > "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_bf16.log" 2>&1
echo "rocprof rc=$?"
cd "$GRAFT_REPO_ROOT"
db=$(find gpurun_out/r02/prof_bf16 -name '*.db' | head -1)
python tools/prof_db.py "$db" --steps 8 --csv gpurun_out/r02/bf16_kernels.csv | head -45
