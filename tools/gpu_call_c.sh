#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02s; mkdir -p $O
for dt in f32 bf16; do for pr in 0 1 0 1; do
  FSD_MAIN_PRIO=$pr timeout 300 python bench.py --dtype $dt --steps 20 --warmup 5 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity > $O/bench_${dt}_p$pr.json 2> $O/bench_${dt}_p$pr.err
  echo "$dt prio=$pr rc=$? $(python -c "import json,sys; d=json.load(open('$O/bench_${dt}_p$pr.json')); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done; done
