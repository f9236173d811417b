#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r02"; mkdir -p "$O"
export TMPDIR=/tmp; cd /tmp
pmc() { local name=$1; local ctr=$2; shift 2
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$O/$name.log" 2>&1
  echo "$name pmc rc=$?"; }
pmc sq1_f32 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"
pmc sq2_f32 "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"
pmc sq1_bf16 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" --dtype bf16
cd "$R"; find gpurun_out/r02 -name "*.db" -delete; grep -il "error\|invalid" gpurun_out/r02/sq*.log | head
