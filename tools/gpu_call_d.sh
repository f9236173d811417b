#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02s; mkdir -p $O
for bk in 64 32; do
  FSD_CONV_H_BK=$bk FSD_LB_DTYPE=bf16 timeout 300 python tools/layer_bench.py fwd > $O/lb_fwd_bk$bk.log 2>&1; echo "bk=$bk"; tail -10 $O/lb_fwd_bk$bk.log
done
