#!/bin/bash
# round 6: CU-mask probe (side streams pinned to CU subsets)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06g; mkdir -p $O
( timeout 500 python tools/experiments_r06/cu_mask_probe.py f32 ) > $O/cu_mask_f32.txt 2> $O/cu_mask_f32.err; echo "f32 rc=$?"; cat $O/cu_mask_f32.txt; tail -3 $O/cu_mask_f32.err
( timeout 500 python tools/experiments_r06/cu_mask_probe.py bf16 ) > $O/cu_mask_bf16.txt 2> $O/cu_mask_bf16.err; echo "bf16 rc=$?"; cat $O/cu_mask_bf16.txt; tail -3 $O/cu_mask_bf16.err
