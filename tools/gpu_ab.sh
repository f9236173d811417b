#!/bin/bash
# Same-box A/B of the train step: tools/gpu_ab.sh "ENV_A=.." "ENV_B=.." ...   (each config twice, interleaved)
set -u
R="$GRAFT_REPO_ROOT"; cd "$R"; export TMPDIR=/tmp
for rep in 1 2; do
  for cfg in "$@"; do
    out=$(env $cfg timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-parity 2>/dev/null | tail -1)
    python - "$cfg" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r = d["roofline"]
print("%-40s ms %.3f unprof %.3f prof %.3f clk %s | gemm %.3f wgrad %.3f xform %.3f hbm %.3f" % (sys.argv[1], d["ms_per_step"], d["streams"]["ms_per_step_unprofiled"],
      d["streams"]["ms_per_step_profiled"], d.get("gpu_clock_mhz"), r["kernel_ms_per_step"], r["wgrad_ms_per_step"], r["wino_transform_ms_per_step"], r["hbm_bound_ms_per_step"]))
PY
  done
done
