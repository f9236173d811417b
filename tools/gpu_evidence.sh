#!/bin/bash
# Round-2 evidence: bench lines, rocprofv3 kernel-trace stats (one stream = kernel durations in isolation, and with the
# side streams), PMC passes (separate runs per counter group).  Everything lands under gpurun_out/r02e/.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r02e"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( time timeout 900 python bench.py ) > "$O/bench_f32.json" 2> "$O/bench_f32.err"; echo "bench f32 rc=$?"
( time timeout 600 python bench.py --dtype bf16 ) > "$O/bench_bf16.json" 2> "$O/bench_bf16.err"; echo "bench bf16 rc=$?"
( time timeout 600 python bench.py --classes 15 --support 416 --no-cpu-baseline --no-extras ) > "$O/bench_c2cfg.json" 2> "$O/bench_c2cfg.err"; echo "bench c2 rc=$?"
cd /tmp
run() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity "$@" > "$O/$name.log" 2>&1
  echo "$name stats rc=$?"
}
pmc() {  # name, counters, bench args...
  local name=$1; local ctr=$2; shift 2
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 3 --warmup 1 --profile-steps 0 --streams 0 --no-cpu-baseline --no-extras --no-parity "$@" > "$O/$name.log" 2>&1
  echo "$name pmc rc=$?"
}
run stats_f32_serial --streams 0
run stats_f32_streams --streams 1
run stats_bf16_serial --streams 0 --dtype bf16
run stats_bf16_streams --streams 1 --dtype bf16
run stats_c2_serial --streams 0 --classes 15 --support 416
pmc fetch_c2 FETCH_SIZE --classes 15 --support 416
pmc write_c2 WRITE_SIZE --classes 15 --support 416
pmc mfma_c2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" --classes 15 --support 416
pmc lds_c2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --classes 15 --support 416
pmc fetch_bf16 FETCH_SIZE --dtype bf16
pmc write_bf16 WRITE_SIZE --dtype bf16
pmc mfma_bf16 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" --dtype bf16
cd "$R"
find "$O" -name "*.db" -delete
find "$O" -name "*kernel_trace.csv" -size +6M -delete
find "$O" -name "*agent_info.csv" -delete
du -sh "$O"; ls "$O"
head -c 1500 "$O/bench_f32.json"
