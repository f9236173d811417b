#!/bin/bash
# Round 5: the backward elementwise kernels with batched loads (act_stats8, act_bwd_pool2, bn_bwd_apply_g8): tests + kernel stats.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05k"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_backward.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x ) > "$O/pytest.log" 2>&1
echo "pytest rc=$?"; tail -5 "$O/pytest.log"
cd /tmp
run() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity "$@" > "$O/$name.log" 2>&1
  echo "$name stats rc=$?"
}
run stats_bf16_serial --streams 0 --dtype bf16
run stats_f32_serial --streams 0
grep -h "act_stats\|act_bwd_pool2\|bn_bwd_apply_g" "$O"/stats_*/run_kernel_stats.csv | cut -c1-70,150-260
find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -size +6M -delete; find "$O" -name "*agent_info.csv" -delete
