#!/bin/bash
# A/B of the side streams (FSD_STREAMS=0/1): bit-identity tests, then the train step in both modes.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02s; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_streams.py -x -q -p no:cacheprovider ) > $O/pytest_streams.log 2>&1
echo "pytest rc=$?"; tail -15 $O/pytest_streams.log
for dt in f32 bf16; do for s in 0 1 0 1; do
  FSD_STREAMS=$s timeout 300 python bench.py --dtype $dt --steps 20 --warmup 5 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity > $O/bench_${dt}_s$s.json 2> $O/bench_${dt}_s$s.err
  echo "$dt streams=$s rc=$? $(python -c "import json,sys; d=json.load(open('$O/bench_${dt}_s$s.json')); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done; done
