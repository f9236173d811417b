"""The first conv block's kernels alone on the detector's L0 shape (B x 3 x 416 x 416 -> 32 channels), fp32 and bf16 storage:
forward (conv_first*_kernel) and the fused backward (first_bwd*_kernel).  For rocprofv3 --stats / --pmc runs."""
import sys
import time
import torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
from fewshot_detection_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
x = torch.zeros(B, 4, 416, 416, device=dev)
x[:, :3] = torch.rand(B, 3, 416, 416, device=dev)
xv = ops.nchw_to_nhwc(x)
w = torch.randn(32, 3, 3, 3, device=dev) * 0.2
for dt in (torch.float32, torch.bfloat16):
    for _ in range(3):
        yv, part = ops.conv3x3_c4(xv, w, 32, bn_partial=True, out_dtype=dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        yv, part = ops.conv3x3_c4(xv, w, 32, bn_partial=True, out_dtype=dt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    gb = B * 416 * 416 * (16 + 32 * yv.t.element_size()) / 1e9
    print("conv_first %s: %.3f ms  %.2f TB/s" % (dt, ms, gb / ms))
