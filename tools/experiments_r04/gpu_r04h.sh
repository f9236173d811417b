#!/bin/bash
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04h"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp; cd "$R"
timeout 900 python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -5
for v in 0 1; do
  echo "--- FSD_CONV_HALO=$v"
  FSD_CONV_HALO=$v FSD_LB_ONLY=208,32,64 timeout 200 python tools/layer_bench.py fwd 2>&1 | grep -v amdgpu.ids
  FSD_CONV_HALO=$v FSD_LB_SWAP=0 python - <<'PY'
import torch, time, sys
sys.path.insert(0, ".")
from fewshot_detection_amd import ops
dev = torch.device("cuda:0")
for (B,H,W,ci,co) in [(64,208,208,64,32),(20,112,112,32,64),(20,112,112,64,32)]:
    x = ops.nchw_to_nhwc(torch.randn(B, ci, H, W, device=dev))
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    wp = ops.pack_weight(w)
    f = lambda: ops.conv2d(x, wp, co, 3)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    print("%dx%d %d->%d B=%d: %.3f ms %.1f TF" % (H, W, ci, co, B, ms, 2.0*9*ci*co*B*H*W/ms/1e9))
PY
done
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 ) 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline ) > "$O/bench.out" 2> "$O/bench.err"; tail -c 2500 "$O/bench.out" | cut -c1-330
