#!/bin/bash
# rocprofv3 kernel stats of the solo first-block backward probe
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r06k"; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp; rm -rf "$O/prof"
timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof" -o run -- python "$R/tools/probes/first_bwd_time.py" > "$O/prof.log" 2>&1
echo "rc=$?"; tail -4 "$O/prof.log"
f=$(find "$O/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -i "first_bwd\|\"Name\"" "$f" < /dev/null | cut -c1-220
find "$O" -name "*.db" -size +20M -delete
