#!/bin/bash
# Round 5, call 1: the whole GPU suite after the prune + the two new kernels, kernel stats of the fp32 and bf16 steps, a bench line.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05a"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 -x ) > "$O/pytest_gpu.log" 2>&1
echo "pytest rc=$?"; tail -12 "$O/pytest_gpu.log"
cd /tmp
run() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity "$@" > "$O/$name.log" 2>&1
  echo "$name stats rc=$?"
}
run stats_f32_serial --streams 0
run stats_bf16_serial --streams 0 --dtype bf16
cd "$R"
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > "$O/bench_f32.json" 2> "$O/bench_f32.err"; echo "bench f32 rc=$?"
cp gpurun_out/bench_full_f32_n1.json "$O/bench_full_f32.json" 2>/dev/null
find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -size +6M -delete; find "$O" -name "*agent_info.csv" -delete
tail -c 2700 "$O/bench_f32.json"
