#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for dt in f32; do for fu in 0 1 0 1 0 1; do
  FSD_FUSE_FIRST_BWD=$fu timeout 300 python bench.py --dtype $dt --steps 20 --warmup 5 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$dt fuse=$fu', d['ms_per_step'])"
done; done
rocm-smi --showclocks --showtemp 2>/dev/null | head -20
