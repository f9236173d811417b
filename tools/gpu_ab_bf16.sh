#!/bin/bash
# Same-box A/B of the bf16-mode train step: tools/gpu_ab_bf16.sh "ENV_A=.." "ENV_B=.." ...   (each config twice, interleaved)
set -u
R="$GRAFT_REPO_ROOT"; cd "$R"; export TMPDIR=/tmp
for rep in 1 2; do
  for cfg in "$@"; do
    out=$(env $cfg timeout 600 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-parity 2>/dev/null | tail -1)
    python - "$cfg" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r = d["roofline"]
print("%-40s ms %.3f unprof %.3f prof %.3f | gemm %.3f frac %.3f hbm %.3f" % (sys.argv[1], d["ms_per_step"], d["streams"]["ms_per_step_unprofiled"],
      d["streams"]["ms_per_step_profiled"], r["gemm_kernels_ms_per_step"], r["frac"], r["hbm_bound_ms_per_step"]))
PY
  done
done
