#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for v in 0 1 0 1 0 1; do FSD_WINO_DY_ON_SIDE=$v timeout 300 python bench.py --steps 20 --warmup 8 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32 dy_on_side=$v', d['ms_per_step'])"; done
