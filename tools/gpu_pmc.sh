#!/bin/bash
# rocprofv3 evidence of round 2: kernel-trace stats and PMC passes (separate runs per counter group) for the fp32
# headline episode, the configs[1] episode and the bf16 mode.  Everything lands under gpurun_out/r02/.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r02"; mkdir -p "$O"
export TMPDIR=/tmp; cd /tmp
run() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-extras "$@" > "$O/$name.log" 2>&1
  echo "$name stats rc=$?"
}
pmc() {  # name, counters, bench args...
  local name=$1; local ctr=$2; shift 2
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$O/$name.log" 2>&1
  echo "$name pmc rc=$?"
}
run stats_f32
run stats_c2 --classes 15 --support 416
run stats_bf16 --dtype bf16
pmc fetch_c2 FETCH_SIZE --classes 15 --support 416
pmc write_c2 WRITE_SIZE --classes 15 --support 416
pmc mfma_c2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" --classes 15 --support 416
pmc fetch_bf16 FETCH_SIZE --dtype bf16
pmc write_bf16 WRITE_SIZE --dtype bf16
pmc mfma_bf16 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" --dtype bf16
cd "$R"
find gpurun_out/r02 -name "*.csv" | head -40
find gpurun_out/r02 -name "*.db" -delete
du -sh gpurun_out/r02
