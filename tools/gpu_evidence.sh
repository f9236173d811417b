#!/bin/bash
# Round-3 evidence (second half, gpurun_out/r03m -> profiles/r03c_*): the default bench line, rocprofv3 kernel-trace stats (one stream = kernel durations in isolation, and
# with the side streams), PMC passes (separate runs per counter group, never combined with trace domains other than
# --kernel-trace), a B = 2 inference trace.  Everything lands under gpurun_out/r03m/; the summaries to be judged are then
# copied to profiles/ by hand (tools/pmc_traffic.py, tools/pmc_kernels.py).
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03p"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( time timeout 900 python bench.py ) > "$O/bench_f32.json" 2> "$O/bench_f32.err"; echo "bench f32 rc=$?"
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
# (the fp32 mode's default arithmetic is "split"; the native fp32 MFMA gets its own trace; the bf16 storage mode's kernels are
# unchanged since r03e and keep their r03_* summaries)
run stats_f32_serial --streams 0
run stats_f32_streams --streams 1
run stats_f32_native_serial --streams 0 --f32-gemm native
pmc fetch_f32 FETCH_SIZE
pmc write_f32 WRITE_SIZE
pmc mfma_f32 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
pmc wait_f32 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_infer_b2" -o run -- python "$R/tools/probes/inference_b2.py" > "$O/stats_infer_b2.log" 2>&1; echo "infer rc=$?"
cd "$R"
find "$O" -name "*.db" -delete
find "$O" -name "*kernel_trace.csv" -size +6M -delete
find "$O" -name "*agent_info.csv" -delete
du -sh "$O"; ls "$O"
head -c 600 "$O/bench_f32.json"
