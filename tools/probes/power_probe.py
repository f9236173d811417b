"""Probe: run one GEMM-heavy layer in a loop for a few seconds (so that rocm-smi can be sampled beside it)."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from fewshot_detection_amd import ops  # noqa: E402

mode, secs = sys.argv[1], float(sys.argv[2])
dev = torch.device("cuda:0")
if mode in ("native", "split"):
    ops.f32_gemm_mode(mode)
    x = ops.nchw_to_nhwc(torch.randn(64, 1024, 13, 13, device=dev))
    w = torch.randn(1024, 1024, 3, 3, device=dev) * 0.02
    wp = ops.pack_weight_wino(w, 0, 4)
    fn = lambda: ops.conv3x3_wino(x, wp, 1024, tile=4)
else:       # bf16 storage mode kernel
    x = ops.View(torch.randn(64 * 169, 1024, device=dev).to(torch.bfloat16), 64, 13, 13, 1024)
    wp = ops.pack_weight(torch.randn(1024, 1024, 3, 3, device=dev) * 0.02, 0, "bf16")
    fn = lambda: ops.conv2d(x, wp, 1024, 3, bn_partial=True)
fn(); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); n += 20
print(mode, "ms per call %.3f" % ((time.perf_counter() - t0) / n * 1e3), flush=True)
