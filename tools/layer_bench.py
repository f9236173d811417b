"""Per-layer timing of the conv kernels on the shapes of the C2 episode (B=64, 416x416).

    python tools/layer_bench.py [fwd|wgrad|all]

Prints ms and direct-convolution-equivalent TFLOP/s per shape; used for kernel tuning (env knobs
FSD_CONV_TILE / FSD_WGRAD_TILE select kernel variants)."""
import os
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from fewshot_detection_amd import ops  # noqa: E402

SHAPES = [  # B, H, W, cin, cout, k
    (64, 208, 208, 32, 64, 3), (64, 104, 104, 64, 128, 3), (64, 104, 104, 128, 64, 1), (64, 52, 52, 128, 256, 3),
    (64, 26, 26, 256, 512, 3), (64, 13, 13, 512, 1024, 3), (64, 13, 13, 1024, 1024, 3), (64, 13, 13, 1280, 1024, 3),
    (64, 52, 52, 256, 128, 1), (64, 26, 26, 512, 256, 1), (64, 13, 13, 1024, 512, 1), (64, 26, 26, 512, 64, 1),
]


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def timed_classes(fn, iters=5):
    """-> (wall ms, {kernel class: ms}) per call, from the library's per-kernel HIP events."""
    fn()
    torch.cuda.synchronize()
    ops.kernel_profile(True)
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters * 1e3
    ops.kernel_profile(False)
    kp = ops.kernel_profile_collect()
    return wall, {k: v["ms"] / iters for k, v in kp.items() if v["launches"]}


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    dev = torch.device("cuda:0")
    batch = int(os.environ.get("FSD_LB_BATCH", "64"))
    global SHAPES
    SHAPES = [(batch,) + s[1:] for s in SHAPES]
    if os.environ.get("FSD_LB_ONLY"):                 # "H,cin,cout" filters the shape list (PMC runs on one layer)
        h, ci, co = (int(v) for v in os.environ["FSD_LB_ONLY"].split(","))
        SHAPES = [s for s in SHAPES if (s[1], s[3], s[4]) == (h, ci, co)]
    if os.environ.get("FSD_LB_SWAP") == "1":       # the data-gradient shapes: channel counts swapped
        SHAPES = [(b, h, w, co, ci, k) for b, h, w, ci, co, k in SHAPES if k == 3 and ci >= 64]
    if os.environ.get("FSD_WINO4") == "0":
        ops.WINOGRAD4 = False
    if os.environ.get("FSD_WINO4_MIN_CH"):
        ops.WINO4_MIN_CH = int(os.environ["FSD_WINO4_MIN_CH"])
    if os.environ.get("FSD_LB_DTYPE") == "bf16":
        return main_bf16(what, dev)
    for B, H, W, cin, cout, k in SHAPES:
        x = ops.nchw_to_nhwc(torch.randn(B, cin, H, W, device=dev))
        dy = ops.nchw_to_nhwc(torch.randn(B, cout, H, W, device=dev))
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        flops = 2.0 * k * k * cin * cout * B * H * W
        line = "%3dx%3d %4d->%4d k%d:" % (H, W, cin, cout, k)
        if what in ("fwd", "all"):
            tile = ops.wino_tile(cin, cout, k, H, W)
            if tile:
                wp = ops.pack_weight_wino(w, 0, tile)
                ms = timed(lambda: ops.conv3x3_wino(x, wp, cout, tile=tile))
            else:
                wp = ops.pack_weight(w)
                ms = timed(lambda: ops.conv2d(x, wp, cout, k))
            line += "  fwd[%s] %7.3f ms %6.1f TF" % (tile or "d", ms, flops / ms / 1e9)
            if tile:
                _, cl = timed_classes(lambda: ops.conv3x3_wino(x, wp, cout, tile=tile))
                line += " (gemm %.3f xform %.3f)" % (cl.get("gemm_fwd", 0), cl.get("wino_transform", 0))
        if what in ("wgrad", "all"):
            ms = timed(lambda: ops.conv2d_wgrad(dy, cout, x, cin, k))
            line += "  wgrad %7.3f ms %6.1f TF" % (ms, flops / ms / 1e9)
            _, cl = timed_classes(lambda: ops.conv2d_wgrad(dy, cout, x, cin, k))
            line += " (gemm %.3f xform %.3f)" % (cl.get("gemm_wgrad", 0), cl.get("wino_transform", 0))
            if k == 3 and ops.wino_tile(cin, cout, k, H, W) and os.environ.get("FSD_LB_WGRAD_DIRECT") == "1":
                # the same weight gradient in the DIRECT form (split-K reduction over the pixels, 9 taps as shifted reads of x):
                # reads dy and x once, no transforms, but 2.25-4x the MFMA terms (per-layer table)
                ms_d = timed(lambda: ops.conv2d_wgrad(dy, cout, x, cin, k, tile=0))
                line += "  | direct wgrad %7.3f ms %6.1f TF" % (ms_d, flops / ms_d / 1e9)
        print(line, flush=True)


def main_bf16(what, dev):
    """bf16 storage mode: DMA-staged bf16 MFMA forward / data-gradient kernel and the transpose-read weight gradient."""
    for B, H, W, cin, cout, k in SHAPES + [(SHAPES[0][0], 13, 13, 1024, 450, 1), (SHAPES[0][0], 26, 26, 512, 256, 1)]:
        x = ops.View(torch.randn(B * H * W, cin, device=dev).to(torch.bfloat16), B, H, W, cin)
        dy = ops.View(torch.randn(B * H * W, (cout + 7) // 8 * 8, device=dev).to(torch.bfloat16), B, H, W, (cout + 7) // 8 * 8)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        flops = 2.0 * k * k * cin * cout * B * H * W
        line = "%3dx%3d %4d->%4d k%d:" % (H, W, cin, cout, k)
        if what in ("fwd", "all"):
            wp = ops.pack_weight(w, 0, "bf16")
            ms = timed(lambda: ops.conv2d(x, wp, cout, k, bn_partial=True))
            line += "  fwd %7.3f ms %7.1f TF" % (ms, flops / ms / 1e9)
            if cout % 32 == 0:
                wp1 = ops.pack_weight(w, 1, "bf16")
                dyc = ops.View(dy.t, B, H, W, cout)
                ms = timed(lambda: ops.conv2d(dyc, wp1, cin, k))
                line += "  dgrad %7.3f ms %7.1f TF" % (ms, flops / ms / 1e9)
        if what in ("wgrad", "all"):
            ms = timed(lambda: ops.conv2d_wgrad(dy, dy.C, x, cin, k))
            line += "  wgrad %7.3f ms %7.1f TF" % (ms, flops / ms / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
