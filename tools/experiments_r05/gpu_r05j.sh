#!/bin/bash
# Round 5: whole bf16 test file + kernel stats of the bf16 step + bench lines of both modes (after the bf16 halo kernels).
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05j"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_timed_config.py -m gpu -q -p no:cacheprovider -x ) > "$O/pytest.log" 2>&1
echo "pytest rc=$?"; tail -5 "$O/pytest.log"
cd /tmp
run() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity "$@" > "$O/$name.log" 2>&1
  echo "$name stats rc=$?"
}
run stats_bf16_serial --streams 0 --dtype bf16
cd "$R"
( timeout 600 python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline --no-extras ) > "$O/bench_bf16.json" 2> "$O/bench_bf16.err"; echo "bench bf16 rc=$?"
tail -c 1500 "$O/bench_bf16.json"
find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -size +6M -delete; find "$O" -name "*agent_info.csv" -delete
