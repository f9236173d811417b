#!/bin/bash
# rocprofv3 kernel-trace stats of the train step (one stream / side streams) -> gpurun_out/$1/
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/$1"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd /tmp
run() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity "$@" > "$O/$name.log" 2>&1
  echo "$name stats rc=$?"
}
run stats_f32_serial --streams 0
run stats_f32_streams --streams 1
cd "$R"
find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -size +6M -delete; find "$O" -name "*agent_info.csv" -delete
du -sh "$O"
