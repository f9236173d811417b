#!/bin/bash
# kernel trace of the train step (streams on): per-kernel table + overlap summary.  $1 = f32|bf16
set -u
R="$GRAFT_REPO_ROOT"; dt=${1:-f32}; O="$R/gpurun_out/r02t_$dt"; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
rm -rf "$O/prof"
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof" -o run -- python "$R/bench.py" ${EXTRA:-} --dtype $dt --steps 8 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity > "$O/prof.log" 2>&1
echo "rocprof rc=$?"; tail -2 "$O/prof.log" | cut -c1-300
cd "$R"; db=$(find "$O/prof" -name '*.db' | head -1)
python tools/prof_db.py "$db" --steps 11 --timeline --csv "$O/kernels.csv" | head -45
find "$O" -name "*.db" -size +40M -delete
