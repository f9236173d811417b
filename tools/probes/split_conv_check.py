"""Probe: error of the fp32 conv kernels against a float64 convolution (run once per FSD_F32_SPLIT setting)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from fewshot_detection_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
print("FSD_F32_SPLIT =", os.environ.get("FSD_F32_SPLIT"))
for B, H, W, cin, cout, k in [(4, 52, 52, 128, 256, 3), (4, 26, 26, 256, 512, 3), (8, 13, 13, 1024, 1024, 3), (4, 104, 104, 128, 64, 1),
                              (2, 208, 208, 32, 64, 3), (8, 13, 13, 1024, 30, 1)]:
    xn = torch.randn(B, cin, H, W, device=dev)
    xn = torch.where(xn > 0, xn, 0.1 * xn)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    ref = F.conv2d(xn.double(), w.double(), padding=k // 2)
    x = ops.nchw_to_nhwc(xn)
    tile = ops.wino_tile(cin, cout, k, H, W)
    if tile:
        y = ops.conv3x3_wino(x, ops.pack_weight_wino(w, 0, tile), cout, tile=tile)
    else:
        y = ops.conv2d(x, ops.pack_weight(w), cout, k)
    y = ops.nhwc_to_nchw(y) if hasattr(ops, "nhwc_to_nchw") else y.t.view(B, H, W, -1)[..., :cout].permute(0, 3, 1, 2)
    err = (y.double() - ref)
    print("%3dx%3d %4d->%4d k%d [%s]: rel L2 %.3e  max abs %.3e (ref rms %.3f)" % (H, W, cin, cout, k, tile or "direct",
          float(err.norm() / ref.norm()), float(err.abs().max()), float(ref.pow(2).mean().sqrt())))
