#!/bin/bash
# Round 5, call 2: unit tests of the new kernels, PMC of the halo weight gradient on the L2 shape, kernel stats of the step.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05b"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_split.py tests/test_gpu_backward.py tests/test_gpu_first_bwd.py -m gpu -q -p no:cacheprovider ) > "$O/pytest_new.log" 2>&1
echo "pytest rc=$?"; tail -15 "$O/pytest_new.log"
cd /tmp
export FSD_LB_ONLY=208,32,64
python "$R/tools/layer_bench.py" wgrad 2>&1 | tail -3
pmc() { local name=$1; local ctr=$2
  timeout 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/$name" -o run -- python "$R/tools/layer_bench.py" wgrad > "$O/$name.log" 2>&1
  echo "$name rc=$?"; }
pmc p1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
pmc p2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES"
unset FSD_LB_ONLY
python "$R/tools/experiments_r05/pmc_sum.py" wgrad3x3_halo $(find "$O/p1" "$O/p2" -name "*counter_collection.csv")
run() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity "$@" > "$O/$name.log" 2>&1
  echo "$name stats rc=$?"
}
run stats_f32_serial --streams 0
run stats_bf16_serial --streams 0 --dtype bf16
grep -h "conv_first\|first_bwd\|wgrad3x3" "$O"/stats_*/run_kernel_stats.csv | cut -c1-200
cd "$R"; find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -size +6M -delete; find "$O" -name "*agent_info.csv" -delete
