#!/bin/bash
# alternate a GPU test run and a bench run; print per-step GPU times of slow bench runs
set -u
R="$GRAFT_REPO_ROOT"; cd "$R"; export TMPDIR=/tmp
for i in $(seq 1 ${1:-6}); do
  timeout 600 python -m pytest tests/test_gpu_timed_config.py -q -x -k "teacher" > /dev/null 2>&1
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-parity > /tmp/b.out 2>/dev/null
  python - $i <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_full_f32_n1.json"))
ms = d["ms_per_step"]; g = d["step_gpu_ms"]
flag = "  <-- SLOW" if ms > 28.0 else ""
print("run %2s ms %.2f enabled %s tuning %s%s" % (sys.argv[1], ms, d["streams"]["enabled"], [(round(t["streams_ms"],1), round(t["one_stream_ms"],1)) for t in (d["streams"]["tuning"] or {}).get("tries", [])], flag))
if flag: print("    step_gpu_ms", g)
PY
done
