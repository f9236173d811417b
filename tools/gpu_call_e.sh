#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02s; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_timed_config.py -q -p no:cacheprovider -x -k "variant or dma or plan or tile" ) > $O/pytest_e.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_e.log
timeout 300 python tools/layer_bench.py fwd 2>&1 | tail -8
for b in 0 1; do echo "balance=$b"; FSD_WGRAD_BALANCE=$b timeout 300 python tools/layer_bench.py wgrad 2>&1 | tail -4; done
