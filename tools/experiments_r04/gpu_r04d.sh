#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04d"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_timed_config.py -x -q 2>&1 | tail -4
for v in 0 1; do
  echo "--- wgrad FSD_WGRAD_SPLIT8=$v"; FSD_WGRAD_SPLIT8=$v timeout 200 python tools/layer_bench.py wgrad 2>&1 | grep -v amdgpu.ids | tee "$O/lb_wgrad_$v.txt" | sed -n '2,8p'
done
for t in 512 768 1536; do
  echo "--- wgrad target $t"; FSD_WGRAD_S8_TARGET=$t timeout 200 python tools/layer_bench.py wgrad 2>&1 | grep -v amdgpu.ids | sed -n '4,8p'
done
( time timeout 600 python bench.py --steps 20 --warmup 5 --no-extras ) > "$O/bench_f32.out" 2> "$O/bench_f32.err"; echo "bench rc=$?"; tail -c 2600 "$O/bench_f32.out"
