#!/bin/bash
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r02p"; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
for b in 0 1; do
  rm -rf "$O/f$b"
  FSD_BATCH_MAJOR=$b timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$O/f$b" -o run -- python "$R/tools/layer_bench.py" fwd > "$O/f$b.log" 2>&1
  echo "batch_major=$b rc=$?"
  python "$R/tools/pmc_by_shape.py" "$O/f$b/run_counter_collection.csv" | grep "conv_gemm" | cut -d, -f1-12
done
