#!/bin/bash
# Round-4 evidence (gpurun_out/r06z -> profiles/r04_*): the default bench line, rocprofv3 kernel-trace stats (one stream =
# kernel durations in isolation, and with the side streams), PMC passes (separate runs per counter group, never combined with
# trace domains other than --kernel-trace), the bf16 storage mode's kernel stats, a B = 2 inference trace.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r06z"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( timeout 600 python tools/alloc_trace.py ) > "$O/alloc_trace_f32.txt" 2> "$O/alloc_trace_f32.err"; echo "alloc trace f32 rc=$?"; head -3 "$O/alloc_trace_f32.txt" | cut -c1-250
( timeout 600 python tools/alloc_trace.py --dtype bf16 ) > "$O/alloc_trace_bf16.txt" 2> "$O/alloc_trace_bf16.err"; echo "alloc trace bf16 rc=$?"; head -3 "$O/alloc_trace_bf16.txt" | cut -c1-250
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > "$O/bench_f32.json" 2> "$O/bench_f32.err"; echo "bench f32 rc=$?"
cp gpurun_out/bench_full_f32_n1.json "$O/bench_full_f32.json" 2>/dev/null
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
pmc fetch_f32 FETCH_SIZE   # (tools/pmc_traffic.py counts the steps of a pass itself: pass 0 as its step count)
pmc write_f32 WRITE_SIZE
pmc mfma_f32 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
pmc wait_f32 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
pmc fetch_bf16 FETCH_SIZE --dtype bf16
pmc write_bf16 WRITE_SIZE --dtype bf16
pmc mfma_bf16 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" --dtype bf16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_infer_b2" -o run -- python "$R/tools/probes/inference_b2.py" > "$O/stats_infer_b2.log" 2>&1; echo "infer rc=$?"
cd "$R"
find "$O" -name "*.db" -delete
find "$O" -name "*kernel_trace.csv" -size +6M -delete
find "$O" -name "*agent_info.csv" -delete
du -sh "$O"; ls "$O"
tail -c 2700 "$O/bench_f32.json"
