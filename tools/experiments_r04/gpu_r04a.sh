#!/bin/bash
# Round 4, first GPU call: the compact bench line, the power / clock evidence (rocm-smi sampled beside each GEMM regime and
# beside a library bf16 GEMM as a yardstick), and a cache-residency experiment (per-layer times at smaller batches).
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04a"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > "$O/bench_f32.out" 2> "$O/bench_f32.err"; echo "bench rc=$?"
tail -c 3000 "$O/bench_f32.out"; echo
cp gpurun_out/bench_full_f32_n1.json "$O/" 2>/dev/null
sample() {  # name, command...
  local name=$1; shift
  ( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction)" | tr '\n' ';'; echo; sleep 0.4; done ) > "$O/smi_$name.txt" &
  local spid=$!
  "$@" > "$O/run_$name.txt" 2>&1
  kill $spid 2>/dev/null; wait $spid 2>/dev/null
  echo "== $name: $(cat $O/run_$name.txt | tail -1)"
  sed -n '6,12p' "$O/smi_$name.txt" | cut -c1-400
}
sample idle sleep 3
sample native python tools/probes/power_probe.py native 6
sample split python tools/probes/power_probe.py split 6
sample bf16 python tools/probes/power_probe.py bf16 6
sample yard_random python tools/probes/yardstick_gemm.py random 6
sample yard_zeros python tools/probes/yardstick_gemm.py zeros 6
sample yard_random_4096 python tools/probes/yardstick_gemm.py random 4 4096
for b in 64 16 8 4; do
  echo "--- layer_bench fwd batch $b"; FSD_LB_BATCH=$b timeout 200 python tools/layer_bench.py fwd 2>&1 | tee "$O/lb_fwd_b$b.txt" | head -8
done
for b in 64 8; do
  echo "--- layer_bench wgrad batch $b"; FSD_LB_BATCH=$b timeout 200 python tools/layer_bench.py wgrad 2>&1 | tee "$O/lb_wgrad_b$b.txt" | head -8
done
du -sh "$O"
