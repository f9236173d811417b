#!/bin/bash
# PMC passes over tools/layer_bench.py (per-shape kernel counters).  $1 = fwd|wgrad, env passes through.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r02p"; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
what=${1:-fwd}
pmc() { local name=$1; local ctr=$2
  rm -rf "$O/$name"
  timeout 90 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/$name" -o run -- python "$R/tools/layer_bench.py" $what > "$O/$name.log" 2>&1
  echo "$name rc=$?"; }
pmc p1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
pmc p2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES"
pmc p3 "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
cd "$R"
python tools/pmc_by_shape.py $(find gpurun_out/r02p/p1 gpurun_out/r02p/p2 gpurun_out/r02p/p3 -name "*counter_collection.csv") > gpurun_out/r02p/by_shape_$what.csv
head -30 gpurun_out/r02p/by_shape_$what.csv | cut -c1-400
tail -12 gpurun_out/r02p/p1.log
find gpurun_out/r02p -name "*.db" -delete; find gpurun_out/r02p -name "*kernel_trace.csv" -size +8M -delete
