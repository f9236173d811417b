#!/bin/bash
# round 6, fourth GPU call: caching-allocator policies against the timed region's hipMallocs (tools/alloc_trace.py)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06d; mkdir -p $O
i=0
for conf in "roundup_power2_divisions:4,max_split_size_mb:32" "roundup_power2_divisions:4,max_split_size_mb:128" "roundup_power2_divisions:8,max_split_size_mb:64" "expandable_segments:True" "garbage_collection_threshold:0.99" "max_split_size_mb:1000000"; do
  i=$((i+1))
  for dt in f32 bf16; do
    ( PYTORCH_HIP_ALLOC_CONF="$conf" timeout 400 python tools/alloc_trace.py --dtype $dt ) > $O/alloc_${i}_$dt.txt 2> $O/alloc_${i}_$dt.err
    echo "== $conf $dt rc=$?"; head -2 $O/alloc_${i}_$dt.txt | cut -c1-260; tail -2 $O/alloc_${i}_$dt.txt | cut -c1-330
  done
done
( timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider -k "error_growth or one_ulp or fused_sgd or halo_kernel" ) > $O/pytest_bf16.log 2>&1
echo "bf16 tests rc=$?"; tail -5 $O/pytest_bf16.log
