#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02s; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_first_bwd.py tests/test_gpu_streams.py tests/test_gpu_model.py tests/test_gpu_backward.py -q -p no:cacheprovider -x ) > $O/pytest_g.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_g.log
for dt in f32 bf16; do for fu in 0 1 0 1; do
  FSD_FUSE_FIRST_BWD=$fu timeout 300 python bench.py --dtype $dt --steps 20 --warmup 5 --profile-steps 0 --no-cpu-baseline --no-extras --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$dt fuse=$fu', d['ms_per_step'])"
done; done
