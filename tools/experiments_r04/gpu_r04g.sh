#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04g"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( time timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) 2>&1
for st in 1 0; do
( timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --streams $st ) > "$O/bench_s$st.out" 2> "$O/bench_s$st.err"; echo "bench streams=$st rc=$?"; python - <<PY
import json
d=json.loads(open("$O/bench_s$st.out").read().strip().splitlines()[-1])
print("ms", d["ms_per_step"], "unprof", d["streams"]["ms_per_step_unprofiled"], "prof", d["streams"]["ms_per_step_profiled"], "gemm", d["roofline"]["kernel_ms_per_step"], "wgrad", d["roofline"]["wgrad_ms_per_step"])
PY
done
FSD_CONV1_SPLIT8=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$O/bench_no1x1.out" 2> "$O/bench_no1x1.err"; tail -c 2500 "$O/bench_no1x1.out" | cut -c1-400
