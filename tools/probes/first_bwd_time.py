"""Solo timing of the first block's backward on the L0 shape: one-sweep kernel vs the unfused sequence."""
import sys, time
import torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
from fewshot_detection_amd import ops
dev = torch.device("cuda:0")
B, H, W, cin, cout = 64, 416, 416, 3, 32
if len(sys.argv) > 5:
    B, H, W, cin, cout = (int(v) for v in sys.argv[1:6])
for bf16 in (False, True):
    dt = torch.bfloat16 if bf16 else torch.float32
    x = torch.rand(B, cin, H, W, device=dev)
    xv = ops.nchw_to_nhwc(x)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.3
    bn = torch.nn.BatchNorm2d(cout).to(dev)
    yv, part = ops.conv3x3_c4(xv, w, cout, bn_partial=True, out_dtype=dt)
    scale, shift, mean, invstd = ops.bn_finalize(part, xv.pixels, bn, True)
    dz = ops.View(torch.randn(B * (H // 2) * (W // 2), cout, device=dev).to(dt), B, H // 2, W // 2, cout)
    def unfused():
        d, partial = ops.bn_act_pool_bwd(dz, None, yv, scale, shift, mean, invstd, 0.1, 1)
        _, _, coef = ops.reduce_partials(partial, yv.pixels, cout, scale=scale, want_coef=True)
        return ops.conv3x3_wgrad_c4_bnfused(d, yv, coef, mean, invstd, xv, cin, cout)
    def fused():
        return ops.first_layer_bwd(dz, yv, scale, shift, mean, invstd, 0.1, xv, cin, cout, bn, True)
    for name, fn in (("unfused", unfused), ("fused", fused)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        print("bf16" if bf16 else "f32", name, "%.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
