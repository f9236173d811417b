#!/bin/bash
# Round 5: fp32 64 <-> 128 layers on the halo-staged direct kernel (8 waves, ragged width): tests, kernel stats, bench line.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05n"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( timeout 1200 python -m pytest tests/test_gpu_split.py tests/test_gpu_backward.py tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_timed_config.py -m gpu -q -p no:cacheprovider -x ) > "$O/pytest.log" 2>&1
echo "pytest rc=$?"; tail -5 "$O/pytest.log"
cd /tmp
run() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity "$@" > "$O/$name.log" 2>&1
  echo "$name stats rc=$?"
}
run stats_f32_serial --streams 0
cd "$R"
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras ) > "$O/bench_f32.json" 2> "$O/bench_f32.err"; echo "bench f32 rc=$?"
tail -c 1200 "$O/bench_f32.json"
find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -size +6M -delete; find "$O" -name "*agent_info.csv" -delete
