#!/bin/bash
# Round-4 evidence (gpurun_out/r06z -> profiles/r04_*): the default bench line, rocprofv3 kernel-trace stats (one stream =
# kernel durations in isolation, and with the side streams), PMC passes (separate runs per counter group, never combined with
# trace domains other than --kernel-trace), the bf16 storage mode's kernel stats, a B = 2 inference trace.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r06z"; rm -rf "$O"; mkdir -p "$O"; export O_DIR="$O"
export TMPDIR=/tmp; cd "$R"
( timeout 600 python tools/alloc_trace.py ) > "$O/alloc_trace_f32.txt" 2> "$O/alloc_trace_f32.err"; echo "alloc trace f32 rc=$?"; head -3 "$O/alloc_trace_f32.txt" | cut -c1-250
( timeout 600 python tools/alloc_trace.py --dtype bf16 ) > "$O/alloc_trace_bf16.txt" 2> "$O/alloc_trace_bf16.err"; echo "alloc trace bf16 rc=$?"; head -3 "$O/alloc_trace_bf16.txt" | cut -c1-250
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > "$O/bench_f32.json" 2> "$O/bench_f32.err"; echo "bench f32 rc=$?"
cp gpurun_out/bench_full_f32_n1.json "$O/bench_full_f32.json" 2>/dev/null
( time timeout 600 python bench.py --classes 15 --support 416 --no-cpu-baseline --no-extras ) > "$O/bench_c2cfg.json" 2> "$O/bench_c2cfg.err"; echo "bench c2 rc=$?"
cd /tmp
run() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 6 --warmup 2 --profile-steps 0 --no-settle --no-cpu-baseline --no-extras --no-parity "$@" > "$O/$name.log" 2>&1
  echo "$name stats rc=$?"
}
pmc() {  # name, counters, bench args...
  local name=$1; local ctr=$2; shift 2
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 3 --warmup 1 --profile-steps 0 --no-settle --streams 0 --no-cpu-baseline --no-extras --no-parity "$@" > "$O/$name.log" 2>&1
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
# ---- summaries (what goes to profiles/r06_*), then the raw traces go: gpurun copies back at most 64 MiB ----
S="$O/summary"; mkdir -p "$S"
cp "$O/bench_f32.json" "$S/bench_line_f32.json"; cp "$O/bench_full_f32.json" "$S/bench_full_f32.json" 2>/dev/null
cp "$O/bench_c2cfg.json" "$S/bench_line_configs1.json"
for v in f32_serial f32_streams bf16_serial infer_b2; do cp "$O/stats_$v/run_kernel_stats.csv" "$S/stats_${v}_kernel_stats.csv" 2>/dev/null; done
python - <<'PY'
import os, sys
sys.path.insert(0, "tools")
import pmc_traffic
O = os.environ["O_DIR"]
for v in ("f32", "bf16"):
    n = pmc_traffic.steps_in(O + "/fetch_%s/run" % v, "FETCH_SIZE")
    os.system("python tools/pmc_kernels.py %s/summary/pmc_kernels_%s.csv %d a=%s/fetch_%s/run_counter_collection.csv "
              "b=%s/write_%s/run_counter_collection.csv c=%s/mfma_%s/run_counter_collection.csv" % (O, v, n, O, v, O, v, O, v))
    print(v, "steps per pass", n)
os.system("python tools/pmc_kernels.py %s/summary/pmc_wait_f32.csv %d a=%s/wait_f32/run_counter_collection.csv"
          % (O, pmc_traffic.steps_in(O + "/wait_f32/run", "SQ_WAVE_CYCLES"), O))
PY
LAUNCHES=$(python - <<'PY'
import json
d = json.load(open("gpurun_out/r06z/bench_full_f32.json"))
print(int(round(d["roofline"]["algorithmic_speedup"]["launches_per_step"])))
PY
)
python tools/pmc_traffic.py "$O/fetch_f32/run" "$O/write_f32/run" 0 "$LAUNCHES" "$S/conv_traffic.json" metric_string | cut -c1-400
rm -rf "$O"/stats_* "$O"/fetch_* "$O"/write_* "$O"/mfma_* "$O"/wait_*
du -sh "$O"; ls "$S"
tail -c 2700 "$O/bench_f32.json"
