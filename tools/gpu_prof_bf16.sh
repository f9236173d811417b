#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/$1"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_bf16_serial" -o run -- python "$R/bench.py" --dtype bf16 --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity --streams 0 > "$O/stats_bf16_serial.log" 2>&1
echo "rc=$?"
cd "$R"; find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -size +6M -delete; find "$O" -name "*agent_info.csv" -delete
timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | cut -c1-400
