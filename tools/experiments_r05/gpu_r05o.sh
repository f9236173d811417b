#!/bin/bash
# Round 5: the bf16 halo weight gradient: tests + per-layer times against the GEMM kernel.
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05o"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
( timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider -x -k "wgrad" ) > "$O/pytest_bf16.log" 2>&1
echo "pytest rc=$?"; tail -8 "$O/pytest_bf16.log"
cd /tmp
for h in 1 1 0; do
  echo "== FSD_CONV_HALO=$h" | tee -a "$O/layers.log"
  FSD_CONV_HALO=$h FSD_LB_DTYPE=bf16 timeout 300 python "$R/tools/layer_bench.py" wgrad 2>&1 | grep -v "class_scale\|amdgpu.ids" | head -2 | tee -a "$O/layers.log"
done
