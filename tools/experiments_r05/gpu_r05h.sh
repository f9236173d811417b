#!/bin/bash
# Round 5, call 8: bf16 mode -- the 4-wave hand-pipelined tiles (conv tiles 7 / 8, weight-gradient W4) against the 8-wave ones,
# per layer at B = 64; the bf16 test file; kernel stats of the fp32 and bf16 steps as they stand.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05h"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider -x ) > "$O/pytest_bf16.log" 2>&1
echo "pytest rc=$?"; tail -8 "$O/pytest_bf16.log"
cd /tmp
for t in auto 7 8; do
  echo "== conv tile $t" | tee -a "$O/layers.log"
  if [ "$t" = auto ]; then unset FSD_CONV_H_TILE; else export FSD_CONV_H_TILE=$t; fi
  FSD_LB_DTYPE=bf16 timeout 300 python "$R/tools/layer_bench.py" fwd 2>&1 | grep -v "class_scale\|amdgpu.ids" | tee -a "$O/layers.log"
done
unset FSD_CONV_H_TILE
for w in 0 1; do
  echo "== wgrad W4=$w" | tee -a "$O/layers.log"
  FSD_WGRAD_H_W4=$w FSD_LB_DTYPE=bf16 timeout 300 python "$R/tools/layer_bench.py" wgrad 2>&1 | grep -v "class_scale\|amdgpu.ids" | tee -a "$O/layers.log"
done
run() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o run -- python "$R/bench.py" --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity "$@" > "$O/$name.log" 2>&1
  echo "$name stats rc=$?"
}
run stats_bf16_serial --streams 0 --dtype bf16
run stats_f32_serial --streams 0
cd "$R"; find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -size +6M -delete; find "$O" -name "*agent_info.csv" -delete
tail -2 "$O/stats_bf16_serial.log" | cut -c1-600
