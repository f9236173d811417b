#!/bin/bash
# PMC on single layers: the 8-wave split GEMM / wgrad kernels vs the 4-wave ones (13x13 1024->1024)
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04e"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd /tmp
pmc() {  # name, counters, env...
  local name=$1; local ctr=$2; shift 2
  env "$@" FSD_LB_ONLY=13,1024,1024 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/$name" -o run -- python "$R/tools/layer_bench.py" all > "$O/$name.log" 2>&1
  echo "$name rc=$?"
}
for v in 1 0; do
  pmc mfma_$v "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" FSD_WINO_SPLIT8=$v FSD_WGRAD_SPLIT8=$v
  pmc wait_$v "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" FSD_WINO_SPLIT8=$v FSD_WGRAD_SPLIT8=$v
  pmc inst_$v "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" FSD_WINO_SPLIT8=$v FSD_WGRAD_SPLIT8=$v
done
cd "$R"
find "$O" -name "*.db" -delete; find "$O" -name "*agent_info.csv" -delete
for v in 1 0; do
python tools/pmc_kernels.py "$O/pmc_$v.csv" 1 a=$O/mfma_$v/run_counter_collection.csv b=$O/wait_$v/run_counter_collection.csv c=$O/inst_$v/run_counter_collection.csv > /dev/null
grep -E "^kernel|conv_gemm|wgrad_kernel|wgrad_split8" "$O/pmc_$v.csv" | cut -c1-1500
done
du -sh "$O"
