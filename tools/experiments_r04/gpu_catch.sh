#!/bin/bash
# run bench.py N times, print ms and the per-step GPU times of any run slower than 1.15x the fastest so far
set -u
R="$GRAFT_REPO_ROOT"; cd "$R"; export TMPDIR=/tmp
for i in $(seq 1 ${1:-14}); do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-parity > /tmp/b.out 2>/dev/null
  python - $i <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_full_f32_n1.json"))
ms = d["ms_per_step"]; g = d["step_gpu_ms"]
flag = "  <-- SLOW" if ms > 28.5 else ""
print("run %2s ms %.2f tuning %s%s" % (sys.argv[1], ms, [round(t["streams_ms"],1) for t in (d["streams"]["tuning"] or {}).get("tries", [])], flag))
if flag: print("    step_gpu_ms", g)
PY
done
