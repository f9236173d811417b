#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05c"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( timeout 900 python -m pytest tests/test_gpu_split.py -m gpu -q -p no:cacheprovider ) > "$O/pytest_new.log" 2>&1
echo "pytest rc=$?"; tail -4 "$O/pytest_new.log"
cd /tmp
export FSD_LB_ONLY=208,32,64
python "$R/tools/layer_bench.py" wgrad 2>&1 | tail -1
pmc() { local name=$1; local ctr=$2; shift 2
  timeout 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/$name" -o run -- python "$@" > "$O/$name.log" 2>&1
  echo "$name rc=$?"; }
pmc w1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "$R/tools/layer_bench.py" wgrad
unset FSD_LB_ONLY
python "$R/tools/experiments_r05/pmc_sum.py" wgrad3x3_halo $(find "$O/w1" -name "*counter_collection.csv")
python "$R/tools/experiments_r05/first_layer_probe.py"
pmc f1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "$R/tools/experiments_r05/first_layer_probe.py"
pmc f2 "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$R/tools/experiments_r05/first_layer_probe.py"
pmc f3 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" "$R/tools/experiments_r05/first_layer_probe.py"
pmc f4 "TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "$R/tools/experiments_r05/first_layer_probe.py"
for k in "conv_first_split_kernel<3, float>" "conv_first_split_kernel<3, unsigned"; do echo "== $k"; python "$R/tools/experiments_r05/pmc_sum.py" "$k" $(find "$O/f1" "$O/f2" "$O/f3" "$O/f4" -name "*counter_collection.csv"); done
tail -3 "$O/f4.log"
cd "$R"; find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -size +6M -delete; find "$O" -name "*agent_info.csv" -delete
