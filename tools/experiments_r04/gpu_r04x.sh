#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; cd "$R"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_first_bwd.py tests/test_gpu_bf16.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
for v in 0 1; do
FSD_FIRST_WIDE=$v python - <<'PY'
import torch, time, sys, os
sys.path.insert(0, ".")
from fewshot_detection_amd import ops
dev = torch.device("cuda:0")
x = ops.nchw_to_nhwc(torch.rand(64, 3, 416, 416, device=dev))
w = torch.randn(32, 3, 3, 3, device=dev) * 0.2
for dt in (torch.float32, torch.bfloat16):
    f = lambda: ops.conv3x3_c4(x, w, 32, bn_partial=True, out_dtype=dt)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    print("FSD_FIRST_WIDE=%s conv_first 64x416x416 3->32 %s: %.3f ms" % (os.environ["FSD_FIRST_WIDE"], dt, ms))
PY
done
bash tools/gpu_ab.sh FSD_FIRST_WIDE=1 FSD_FIRST_WIDE=0
