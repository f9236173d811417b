#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; cd "$R"; export TMPDIR=/tmp
one() {  # label, env...
  local label=$1; shift
  out=$(env "$@" timeout 300 python bench.py --steps 12 --warmup 4 --no-extras --no-cpu-baseline --no-parity --profile-steps 0 2>/dev/null | tail -1)
  python - "$label" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); print("%-28s ms %.2f" % (sys.argv[1], d["ms_per_step"]))
PY
}
for i in 1 2 3 4 5 6 7 8 9 10; do
  one "default#$i" FSD_X=1
  one "hwq8#$i" GPU_MAX_HW_QUEUES=8
done
