#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04f"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) 2>&1
for v in 0 1; do
  echo "--- fwd FSD_CONV1_SPLIT8=$v"; FSD_CONV1_SPLIT8=$v timeout 200 python tools/layer_bench.py fwd 2>&1 | grep "k1:" | tee "$O/lb_1x1_$v.txt"
done
( time timeout 600 python bench.py --steps 20 --warmup 5 --no-extras ) > "$O/bench_f32.out" 2> "$O/bench_f32.err"; echo "bench rc=$?"; tail -c 2600 "$O/bench_f32.out" | cut -c1-900
