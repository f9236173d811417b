"""bf16 storage mode (BASELINE configs[2] / [4]) on the MI355X: the DMA-staged bf16 MFMA convolution and the
transpose-read bf16 weight gradient against fp64 convolutions of the SAME bf16-rounded operands (the products are
exact in fp32, so only the accumulation order differs), the bf16 twins of the HBM-bound kernels against their fp32
versions, and the whole model against the CPU oracle run in the same mixed precision."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _bf(t):
    return t.to(BF).float()


def _view_bf16(x_nchw, dev):
    """(B,C,H,W) float (already bf16-representable) -> bf16 NHWC View on the device."""
    from fewshot_detection_amd import ops
    B, C, H, W = x_nchw.shape
    t = x_nchw.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(dev).to(BF)
    return ops.View(t, B, H, W, C)


def _nchw(v):
    return v.t[:, v.c0:v.c0 + v.C].float().reshape(v.B, v.H, v.W, v.C).permute(0, 3, 1, 2).contiguous().cpu()


@pytest.mark.parametrize("B,H,W,cin,cout,k,bias", [
    (2, 13, 13, 64, 128, 3, False),      # 128x128 tile, BK = 64
    (1, 26, 26, 128, 64, 1, True),       # 128x64 tile (4x1 waves), 1x1
    (2, 13, 13, 32, 64, 3, False),       # BK = 32 (Cin = 32), 64-byte LDS rows
    (3, 9, 7, 96, 36, 3, True),          # Cin = 96 -> BK = 32; ragged M (189 rows) and N (36 channels)
    (2, 13, 13, 1280, 1024, 3, False),   # L29: 8 column tiles, K = 11520
    (5, 6, 6, 1024, 1024, 3, False),     # the 6x6 support maps
    (2, 26, 20, 64, 32, 3, False),       # 32 output channels: the 128x32 tile (data gradient of the 32 -> 64 layer)
    (3, 9, 7, 128, 30, 1, True),         # ... ragged: 30 channels, 189 rows, bias
    (1, 16, 16, 64, 32, 3, True),
])
def test_conv_bf16_dma_kernel_matches_fp64_of_rounded_operands(dev, B, H, W, cin, cout, k, bias):
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(cin + cout + H)
    x = _bf(torch.randn(B, cin, H, W, generator=g))
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    ref = F.conv2d(x.double(), _bf(w).double(), None if b is None else b.double(), 1, (k - 1) // 2)
    yv, part = ops.conv2d(_view_bf16(x, dev), ops.pack_weight(w.to(dev), 0, "bf16"), cout, k,
                          bias=None if b is None else b.to(dev), bn_partial=not bias)
    assert yv.bf16
    y = _nchw(yv).double()
    # the accumulator is within fp32 round-off of the fp64 sum; the stored value is its bf16 rounding
    err = (y - ref).abs()
    assert float((err - ref.abs() * 2.0 ** -8).max()) < 1e-3, float(err.max())
    assert float(err.mean()) < 2.0 ** -9 * float(ref.abs().mean()) * 1.2
    if part is not None:                                   # BatchNorm sums come from the fp32 accumulators, not from y
        p = part.double().sum(0).cpu()
        flat = ref.permute(1, 0, 2, 3).reshape(cout, -1)
        assert torch.allclose(p[:, 0], flat.sum(1), rtol=1e-4, atol=1e-2)
        assert torch.allclose(p[:, 1], (flat ** 2).sum(1), rtol=2e-4, atol=1e-2)
    # data gradient = the same kernel on mode-1 weights
    if cout % 32 == 0:
        gy = _bf(torch.randn(B, cout, H, W, generator=g))
        xg = x.double().requires_grad_(True)
        F.conv2d(xg, _bf(w).double(), None, 1, (k - 1) // 2).backward(gy.double())
        dx, _ = ops.conv2d(_view_bf16(gy, dev), ops.pack_weight(w.to(dev), 1, "bf16"), cin, k)
        gref = xg.grad
        e2 = (_nchw(dx).double() - gref).abs()
        assert float((e2 - gref.abs() * 2.0 ** -8).max()) < 1e-3 * max(1.0, float(gref.abs().max()))


@pytest.mark.parametrize("tile,B,H,W,cin,cout,k", [
    (1, 2, 26, 30, 64, 256, 3),          # 256x256, ragged M (1560 rows = 6.1 tiles)
    (1, 1, 13, 13, 1280, 512, 3),        # ... one partial row tile, K = 11520
    (2, 2, 26, 30, 64, 256, 3),          # 192x256 (3 accumulator rows per wave)
    (2, 3, 13, 13, 1024, 1024, 1),       # ... 1x1, four column tiles
    (3, 2, 26, 30, 64, 256, 3),          # 256x128 (4 x 2 waves)
    (3, 1, 20, 20, 128, 128, 3),
    (6, 2, 26, 30, 64, 256, 3),          # 192x128, 4 waves (2 x 2, three accumulator rows), two workgroups per CU
    (6, 3, 13, 13, 1024, 1024, 1),
])
def test_conv_bf16_eight_wave_tiles_match_fp64_of_rounded_operands(dev, monkeypatch, tile, B, H, W, cin, cout, k):
    """The 8-wave tiles of conv_bf16_dma_kernel (one workgroup per CU), forced through FSD_CONV_H_TILE: forward with the
    BatchNorm partial sums (one row per row tile of THAT tile), and the data gradient."""
    from fewshot_detection_amd import ops
    from fewshot_detection_amd._lib import lib
    monkeypatch.setenv("FSD_CONV_H_TILE", str(tile))
    assert lib().fsd_conv2d_h_plan(B * H * W, cin, cout, k, 0, 1) == tile
    bm = {1: 256, 2: 192, 3: 256, 6: 192}[tile]
    assert lib().fsd_conv2d_h_partial_rows(B, H, W, cin, cout, k) == (B * H * W + bm - 1) // bm
    g = torch.Generator().manual_seed(tile * 100 + cin)
    x = _bf(torch.randn(B, cin, H, W, generator=g))
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ref = F.conv2d(x.double(), _bf(w).double(), None, 1, (k - 1) // 2)
    yv, part = ops.conv2d(_view_bf16(x, dev), ops.pack_weight(w.to(dev), 0, "bf16"), cout, k, bn_partial=True)
    assert part.shape[0] == (B * H * W + bm - 1) // bm
    err = (_nchw(yv).double() - ref).abs()
    assert float((err - ref.abs() * 2.0 ** -8).max()) < 1e-3, float(err.max())
    assert float(err.mean()) < 2.0 ** -9 * float(ref.abs().mean()) * 1.2
    p = part.double().sum(0).cpu()
    flat = ref.permute(1, 0, 2, 3).reshape(cout, -1)
    assert torch.allclose(p[:, 0], flat.sum(1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(p[:, 1], (flat ** 2).sum(1), rtol=2e-4, atol=1e-2)
    # with a bias and the leaky epilogue (inference form), no statistics
    b = torch.randn(cout, generator=g)
    y2, _ = ops.conv2d(_view_bf16(x, dev), ops.pack_weight(w.to(dev), 0, "bf16"), cout, k, bias=b.to(dev), slope=0.1)
    ref2 = F.leaky_relu(ref + b.double().view(1, -1, 1, 1), 0.1)
    e2 = (_nchw(y2).double() - ref2).abs()
    assert float((e2 - ref2.abs() * 2.0 ** -8).max()) < 1e-3
    # data gradient: the same kernel on mode-1 weights (N = Cin: only tiles whose width divides it are taken)
    gy = _bf(torch.randn(B, cout, H, W, generator=g))
    xg = x.double().requires_grad_(True)
    F.conv2d(xg, _bf(w).double(), None, 1, (k - 1) // 2).backward(gy.double())
    dx, _ = ops.conv2d(_view_bf16(gy, dev), ops.pack_weight(w.to(dev), 1, "bf16"), cin, k)
    e3 = (_nchw(dx).double() - xg.grad).abs()
    assert float((e3 - xg.grad.abs() * 2.0 ** -8).max()) < 1e-3 * max(1.0, float(xg.grad.abs().max()))


@pytest.mark.parametrize("B,H,W,cin,cout", [
    (3, 24, 48, 32, 64),        # 3 x 3 blocks per image: border blocks on every side and an interior one
    (2, 8, 16, 32, 64),         # one block per image (every halo pixel outside the image)
    (10, 64, 128, 32, 64),      # 640 blocks on 512 persistent workgroups: runs of one and two blocks
    (3, 24, 48, 64, 32),        # the data-gradient form: 128-byte patch rows, lane-pair stores
    (10, 64, 128, 64, 32),
    (2, 16, 24, 32, 64),        # W % 16 == 8: NOT on the halo kernel for 32 -> 64 / 64 -> 32 (their ragged epilogues
    (2, 16, 40, 64, 32),        # skip whole store instructions behind a counted vmcnt) -- the GEMM kernel, same results
    (3, 24, 48, 64, 128),       # 8 waves (2 pixel groups x 4 channel groups), one workgroup per CU
    (2, 16, 40, 64, 128),       # ... ragged width (every store under a per-lane predicate: the store count is constant)
    (5, 104, 104, 64, 128),     # the timed map size (B = 5): 455 blocks on 256 workgroups
    (24, 104, 104, 64, 128),    # ragged AND 8.5 blocks per workgroup (> D = 3 patches in flight): the counted wait under load
    (1, 304, 304, 32, 64),      # the 608 x 608 episodes of configs[4]: layer 2 ...
    (1, 152, 152, 64, 128),     # ... and layers 4 / 6 (152 = 9.5 blocks wide)
])
def test_conv_bf16_halo_kernel_matches_fp64_of_rounded_operands(dev, B, H, W, cin, cout):
    """conv3x3_halo_h_kernel (persistent workgroups, weights in registers, one DMA-staged halo patch per 8 x 16 block): forward
    with BatchNorm partial sums (one row per workgroup), with bias + leaky, and through the data gradient of the twin shape."""
    from fewshot_detection_amd import ops
    from fewshot_detection_amd._lib import lib
    blocks = B * (H // 8) * ((W + 15) // 16)
    rows = min(blocks, 256 if cout == 128 else 512)
    if W % 16 and cout != 128:                      # whole blocks only for 32 -> 64 / 64 -> 32: one row per 128-pixel GEMM tile
        rows = (B * H * W + 127) // 128
    assert lib().fsd_conv2d_h_partial_rows(B, H, W, cin, cout, 3) == rows
    g = torch.Generator().manual_seed(B * 1000 + cin)
    x = _bf(torch.randn(B, cin, H, W, generator=g))
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    ref = F.conv2d(x.double(), _bf(w).double(), None, 1, 1)
    yv, part = ops.conv2d(_view_bf16(x, dev), ops.pack_weight(w.to(dev), 0, "bf16"), cout, 3, bn_partial=True)
    assert part.shape[0] == rows
    err = (_nchw(yv).double() - ref).abs()
    assert float((err - ref.abs() * 2.0 ** -8).max()) < 1e-3, float(err.max())
    assert float(err.mean()) < 2.0 ** -9 * float(ref.abs().mean()) * 1.2
    p = part.double().sum(0).cpu()
    flat = ref.permute(1, 0, 2, 3).reshape(cout, -1)
    assert torch.allclose(p[:, 0], flat.sum(1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(p[:, 1], (flat ** 2).sum(1), rtol=2e-4, atol=1e-2)
    b = torch.randn(cout, generator=g)
    y2, _ = ops.conv2d(_view_bf16(x, dev), ops.pack_weight(w.to(dev), 0, "bf16"), cout, 3, bias=b.to(dev), slope=0.1)
    ref2 = F.leaky_relu(ref + b.double().view(1, -1, 1, 1), 0.1)
    e2 = (_nchw(y2).double() - ref2).abs()
    assert float((e2 - ref2.abs() * 2.0 ** -8).max()) < 1e-3
    # the data gradient of this layer is the twin shape (cout -> cin) on mode-1 weights
    gy = _bf(torch.randn(B, cout, H, W, generator=g))
    xg = x.double().requires_grad_(True)
    F.conv2d(xg, _bf(w).double(), None, 1, 1).backward(gy.double())
    dx, _ = ops.conv2d(_view_bf16(gy, dev), ops.pack_weight(w.to(dev), 1, "bf16"), cin, 3)
    e3 = (_nchw(dx).double() - xg.grad).abs()
    assert float((e3 - xg.grad.abs() * 2.0 ** -8).max()) < 1e-3 * max(1.0, float(xg.grad.abs().max()))
    # an output view with a wider pixel stride (a slice of a route buffer)
    wide = ops.View(torch.zeros(B * H * W, cout + 16, device=dev, dtype=BF), B, H, W, cout, c0=8)
    y3, _ = ops.conv2d(_view_bf16(x, dev), ops.pack_weight(w.to(dev), 0, "bf16"), cout, 3, out=wide)
    assert torch.equal(_nchw(y3), _nchw(yv))
    assert float(wide.t[:, :8].abs().max()) == 0.0 and float(wide.t[:, 8 + cout:].abs().max()) == 0.0
    # a pixel stride the halo kernel's 16-byte stores cannot take (y_ld % 8 != 0) is not a hard failure -- the plan
    # query for THESE operands reports the GEMM kernel's row tiles and the launch follows it, BatchNorm sums included
    odd = ops.View(torch.zeros(B * H * W, cout + 6, device=dev, dtype=BF), B, H, W, cout, c0=2)
    xv = _view_bf16(x, dev)
    if cout != 32:                                           # (64 -> 32 stores 4 bytes per lane: it takes this stride too)
        bm = {0: 128, 1: 256, 2: 192, 3: 256, 4: 128, 5: 128, 6: 192}[lib().fsd_conv2d_h_plan(B * H * W, cin, cout, 3, 0, 1)]
        assert lib().fsd_conv2d_h_partial_rows_at(B, H, W, cin, cout, 3, xv.ptr, xv.ld, odd.ptr, odd.ld) == (B * H * W + bm - 1) // bm
    y4, part4 = ops.conv2d(xv, ops.pack_weight(w.to(dev), 0, "bf16"), cout, 3, out=odd, bn_partial=True)
    err4 = (_nchw(y4).double() - ref).abs()                 # (another kernel, another summation order: the fp64 bound again)
    assert float((err4 - ref.abs() * 2.0 ** -8).max()) < 1e-3, float(err4.max())
    assert float(odd.t[:, :2].abs().max()) == 0.0 and float(odd.t[:, 2 + cout:].abs().max()) == 0.0
    p4 = part4.double().sum(0).cpu()
    assert torch.allclose(p4[:, 0], flat.sum(1), rtol=1e-4, atol=1e-2) and torch.allclose(p4[:, 1], (flat ** 2).sum(1), rtol=2e-4, atol=1e-2)


def test_timed_bf16_shapes_take_the_large_tiles(dev):
    """Plan query on the B = 64, 416x416 layer shapes (no launches): the 13x13 layers run on 192x256 (one round of 228
    workgroups), the data gradient of L29 (1280 outputs) on 256x256, every other >= 128-channel 3x3 layer on 192x128;
    64-channel outputs, the Cin = 32 layer and the float-NCHW head keep the small 4-wave tiles."""
    from fewshot_detection_amd._lib import lib
    plan = lib().fsd_conv2d_h_plan
    for pixels, cin, cout, k, want in [(64 * 169, 512, 1024, 3, 2), (64 * 169, 1024, 1024, 3, 2), (64 * 169, 1280, 1024, 3, 2),
                                       (64 * 169, 1024, 1280, 3, 1), (64 * 169, 1024, 512, 3, 0), (64 * 676, 256, 512, 3, 6),
                                       (64 * 676, 512, 256, 3, 6), (64 * 2704, 128, 256, 3, 6), (64 * 10816, 64, 128, 3, 6)]:
        assert plan(pixels, cin, cout, k, 0, 1) == want, (pixels, cin, cout, plan(pixels, cin, cout, k, 0, 1))
    assert plan(64 * 10816, 128, 64, 1, 0, 1) == 4 and plan(64 * 43264, 64, 32, 3, 0, 0) == 5
    assert plan(64 * 43264, 32, 64, 3, 0, 1) == 4 and plan(64 * 169, 1024, 450, 1, 1, 0) == 0


@pytest.mark.parametrize("cout,cin,k", [(64, 32, 3), (128, 64, 1), (1024, 512, 3), (30, 1024, 1), (72, 200, 3), (1024, 1280, 3)])
def test_single_pass_weight_pair_equals_the_two_single_packings(dev, cout, cin, k):
    """fsd_pack_conv_weight_bf16_pair (one read of W) == mode 0 and mode 1 of fsd_pack_conv_weight_bf16, bit for bit,
    and re-packing into the kept buffers leaves the zero padding intact."""
    from fewshot_detection_amd import ops
    torch.manual_seed(cout + cin)
    w = torch.randn(cout, cin, k, k).to(dev)
    pair = ops.pack_weight_bf16_pair(w)
    for mode in (0, 1):
        want = ops.pack_weight(w, mode, "bf16")
        assert torch.equal(pair[mode].view(torch.int16), want.view(torch.int16)), mode
    w2 = -2.0 * w
    pair2 = ops.pack_weight_bf16_pair(w2, pair)
    assert pair2[0].data_ptr() == pair[0].data_ptr()
    for mode in (0, 1):
        assert torch.equal(pair2[mode].view(torch.int16), ops.pack_weight(w2, mode, "bf16").view(torch.int16)), mode


def test_conv_bf16_head_nchw_float_output(dev):
    """The fused reweighting (x) head GEMM: bf16 operands, float NCHW output (the loss input), 450 ragged channels."""
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, W, cin, cout = 3, 13, 13, 1024, 450
    x = _bf(torch.randn(B, cin, H, W, generator=g))
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), _bf(w).double(), b.double())
    y, _ = ops.conv2d(_view_bf16(x, dev), ops.pack_weight(w.to(dev), 0, "bf16"), cout, 1, bias=b.to(dev), nchw_out=True)
    assert y.dtype == torch.float32 and y.shape == (B, cout, H, W)
    assert float((y.cpu().double() - ref).abs().max()) < 2e-5 * float(ref.abs().max()) + 1e-5


@pytest.mark.parametrize("B,H,W,cin,cout,k", [
    (2, 13, 13, 64, 128, 3),        # 128 x 64 tile
    (1, 26, 26, 128, 64, 1),        # 64 x 128 tile, 1x1
    (4, 26, 26, 32, 64, 3),         # Cin = 32: the 32-wide x tile (4 x 1 waves)
    (3, 9, 7, 96, 40, 3),           # ragged channels both sides, odd extents, 189 pixels
    (2, 13, 13, 1280, 1024, 3),     # L29
    (2, 13, 13, 1024, 512, 1),      # head shape (rows padded to 512)
    (16, 52, 52, 128, 256, 3),      # many pixels: several splits, image rows wrap inside a 32-pixel chunk
    (20, 3, 3, 1024, 1024, 3),      # last reweighting-net layer of the metric-string episode (224x224 supports -> 3x3 maps):
                                    # a chunk spans several whole images
    (20, 3, 3, 64, 64, 3),          # the same on the 64 x 64 tile (32-pixel chunks)
    (9, 2, 2, 32, 64, 3),           # 2x2 and 1x1 maps: every tap but the centre is padding for most pixels
    (70, 1, 1, 128, 128, 3),
    (3, 5, 4, 128, 128, 3),         # non-square map smaller than a chunk
    (8, 26, 26, 256, 512, 3),       # 8-wave 256 x 256 tile: 2 x 9 tiles, several splits, 64-pixel chunks that wrap image rows
    (2, 13, 13, 264, 328, 3),       # ... ragged in both dimensions (328 = 256 + 72 rows, 2376 = 9 x 256 + 72 columns)
    (3, 7, 5, 512, 256, 1),         # ... 1x1, 105 pixels: one partial chunk
])
def test_wgrad_bf16_transpose_read_kernel_matches_fp64(dev, B, H, W, cin, cout, k):
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(cin + cout + B)
    x = _bf(torch.randn(B, cin, H, W, generator=g))
    gy = _bf(torch.randn(B, cout, H, W, generator=g))
    w = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, 1, (k - 1) // 2).backward(gy.double())
    dw = ops.conv2d_wgrad(_view_bf16(gy, dev), cout, _view_bf16(x, dev), cin, k)
    assert dw.dtype == torch.float32 and dw.shape == (cout, cin, k, k)
    ref = w.grad
    err = float((dw.cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-5, err


def _wgrad_fp64_on_device(x, gy, k):
    """dW of a stride-1 'same' convolution in float64 on the device (one matrix product per tap): the large shapes that reach
    the 256 x 256 weight-gradient tile would take minutes through CPU autograd."""
    B, cin, H, W = x.shape
    cout, pad = gy.shape[1], (k - 1) // 2
    xp = F.pad(x.double(), (pad, pad, pad, pad))
    g2 = gy.double().permute(1, 0, 2, 3).reshape(cout, -1)
    dw = torch.empty(cout, cin, k, k, dtype=torch.float64, device=x.device)
    for ky in range(k):
        for kx in range(k):
            xs = xp[:, :, ky:ky + H, kx:kx + W].permute(1, 0, 2, 3).reshape(cin, -1)
            dw[:, :, ky, kx] = g2 @ xs.t()
    return dw


@pytest.mark.parametrize("B,H,W,cin,cout,k", [
    (16, 26, 26, 512, 512, 3),      # 2 x 18 tiles of 256 x 256, 6 splits, 64-pixel chunks that wrap image rows
    (19, 26, 26, 264, 520, 3),      # ragged in both dimensions (520 = 2 x 256 + 8 rows, 2376 = 9 x 256 + 72 columns), 7 splits
    (49, 26, 26, 1024, 1024, 1),    # 1x1: 16 tiles, 16 splits
    (64, 13, 13, 512, 1024, 3),     # a timed shape (L18): image rows shorter than a chunk
])
def test_wgrad_bf16_256x256_tile_matches_fp64(dev, B, H, W, cin, cout, k):
    """The 8-wave 256 x 256 weight-gradient tile at shapes that really reach it (asserted through the plan query: it needs
    tiles x splits >= 192, i.e. >= 10 k pixels -- the shapes of the generic test above all fall back to 128 x 128)."""
    from fewshot_detection_amd import ops
    from fewshot_detection_amd._lib import lib
    assert lib().fsd_conv2d_wgrad_h_plan(B * H * W, cin, cout, k) == 256256
    g = torch.Generator().manual_seed(cin + cout + B)
    x = _bf(torch.randn(B, cin, H, W, generator=g))
    gy = _bf(torch.randn(B, cout, H, W, generator=g))
    ref = _wgrad_fp64_on_device(x.to(dev), gy.to(dev), k)
    dw = ops.conv2d_wgrad(_view_bf16(gy, dev), cout, _view_bf16(x, dev), cin, k)
    err = float((dw.double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-5, err


@pytest.mark.parametrize("B,H,W,cin,cout", [
    (3, 24, 48, 32, 64),        # 3 x 3 blocks per image: border blocks on every side and an interior one
    (2, 8, 16, 32, 64),         # one block per image
    (10, 64, 128, 32, 64),      # 640 blocks on 512 persistent workgroups
    (3, 24, 48, 64, 128),       # 64 -> 128: eight waves of nine accumulators
    (9, 64, 64, 64, 128),       # 288 blocks on 256 workgroups
    (2, 16, 24, 32, 64),        # W % 16 == 8: half of the last block of a row is outside the image
    (3, 104, 104, 64, 128),     # the timed map size
    (1, 304, 304, 32, 64),      # the 608 x 608 episodes of configs[4]
    (1, 152, 152, 64, 128),
])
def test_wgrad_bf16_halo_kernel_matches_fp64(dev, B, H, W, cin, cout):
    """wgrad3x3_halo_h_kernel (persistent workgroups, dy tile + x halo patch staged once per 8 x 16 block, transposing fragment
    reads at the taps' pixel offsets, one partial per workgroup folded in a fixed order)."""
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(cin + cout + B)
    x = _bf(torch.randn(B, cin, H, W, generator=g))
    gy = _bf(torch.randn(B, cout, H, W, generator=g))
    ref = _wgrad_fp64_on_device(x.to(dev), gy.to(dev), 3)
    dw = ops.conv2d_wgrad(_view_bf16(gy, dev), cout, _view_bf16(x, dev), cin, 3)
    assert dw.dtype == torch.float32 and dw.shape == (cout, cin, 3, 3)
    err = float((dw.double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-5, err
    # operands that are channel slices of wider buffers (route buffers): leading dimensions > channel counts
    xw = ops.View(torch.randn(B * H * W, cin + 16, generator=g).to(dev).to(BF), B, H, W, cin, c0=8)
    xw.t[:, 8:8 + cin] = _view_bf16(x, dev).t
    gw = ops.View(torch.randn(B * H * W, cout + 8, generator=g).to(dev).to(BF), B, H, W, cout, c0=8)
    gw.t[:, 8:8 + cout] = _view_bf16(gy, dev).t
    dw2 = ops.conv2d_wgrad(gw, cout, xw, cin, 3)
    assert torch.equal(dw2, dw)


# ---- the HBM-bound kernels: one template, two storage types ------------------------------------------------------

def _pair(dev, B, H, W, C, seed):
    """The same bf16-representable NHWC tensor as a float view and as a bf16 view."""
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(seed)
    t = _bf(torch.randn(B * H * W, C, generator=g)).to(dev)
    return ops.View(t.clone(), B, H, W, C), ops.View(t.to(BF), B, H, W, C)


@pytest.mark.parametrize("pool", [0, 1, 2])
def test_bn_act_pool_forward_and_backward_twins(dev, pool):
    from fewshot_detection_amd import ops
    B, H, W, C = 3, 13, 13, 24
    yf, yh = _pair(dev, B, H, W, C, 1)
    g = torch.Generator().manual_seed(2)
    scale, shift = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    mean, invstd = torch.randn(C, generator=g).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev)
    zf, zh = ops.bn_act_pool(yf, scale, shift, 0.1, pool), ops.bn_act_pool(yh, scale, shift, 0.1, pool)
    assert zh.bf16 and torch.equal(zh.t.float(), _bf(zf.t.cpu()).to(dev))          # same float math, rounded once at the store
    OH, OW = zf.H, zf.W
    dzf, dzh = _pair(dev, B, OH, OW, C, 3)
    dtf, pf = ops.bn_act_pool_bwd(dzf, None, yf, scale, shift, mean, invstd, 0.1, pool)
    dth, ph = ops.bn_act_pool_bwd(dzh, None, yh, scale, shift, mean, invstd, 0.1, pool)
    assert dth.bf16 and torch.equal(dth.t.float(), _bf(dtf.t.cpu()).to(dev))
    assert torch.allclose(ph.sum(0), pf.sum(0), rtol=1e-5, atol=1e-4)              # partial sums from the float values
    coef = (torch.rand(3, C, generator=g) + 0.5).to(dev)
    df = ops.View(dtf.t.clone(), B, H, W, C)
    dh = ops.View(_bf(dtf.t.cpu()).to(dev).to(BF), B, H, W, C)
    ref_in = ops.View(dh.t.float(), B, H, W, C)
    ops.bn_bwd_apply(ref_in, yf, coef, mean, invstd)
    ops.bn_bwd_apply(dh, yh, coef, mean, invstd)
    assert torch.equal(dh.t.float(), _bf(ref_in.t.cpu()).to(dev))
    del df


def test_layout_and_scatter_twins(dev):
    from fewshot_detection_amd import ops
    xf, xh = _pair(dev, 2, 8, 6, 16, 5)
    assert torch.equal(ops.reorg(xh, 2).t.float(), ops.reorg(xf, 2).t)
    assert torch.equal(ops.nhwc_to_nchw(xh), ops.nhwc_to_nchw(xf))
    nchw = ops.nhwc_to_nchw(xf)
    back = ops.nchw_to_nhwc(nchw, pad_to=64, dtype=BF)
    assert back.bf16 and back.C == 64 and torch.equal(back.t[:, :16].float(), xf.t) and float(back.t[:, 16:].float().abs().max()) == 0
    sf, sh = _pair(dev, 3, 6, 6, 8, 6)
    vf, af = ops.global_maxpool(sf, True)
    vh, ah = ops.global_maxpool(sh, True)
    assert torch.equal(vf, vh) and torch.equal(af, ah)
    gout = torch.randn(3, 8, device=dev)
    assert torch.equal(ops.global_maxpool_bwd(gout, ah, sh).t.float(), _bf(ops.global_maxpool_bwd(gout, af, sf).t.cpu()).to(dev))
    g1f, g1h = _pair(dev, 2, 4, 3, 64, 7)          # gradient of reorg(x, 2): (2, 4, 3, 64)
    assert torch.equal(ops.reorg_bwd(g1h, xh, 2).t.float(), ops.reorg_bwd(g1f, xf, 2).t)
    a_f, a_h = _pair(dev, 2, 5, 5, 12, 8)
    b_f, b_h = _pair(dev, 2, 5, 5, 12, 9)
    ops.add_inplace(a_f, b_f)
    ops.add_inplace(a_h, b_h)
    assert torch.equal(a_h.t.float(), _bf(a_f.t.cpu()).to(dev))
    cs_f, cs_h = _pair(dev, 4, 13, 13, 30, 10)
    assert torch.allclose(ops.colsum(cs_h, 30), ops.colsum(cs_f, 30), rtol=1e-6, atol=1e-5)


# ---- the whole model ---------------------------------------------------------------------------------------------

def _targets(rng, bs, cs):
    tgt = np.zeros((bs, cs, 250), np.float64)
    fill = np.zeros((bs, cs), np.int64)
    for b in range(bs):
        for _ in range(rng.randint(1, 4)):
            n = rng.randint(0, cs)
            w, h = rng.uniform(0.05, 0.5, 2)
            cx = float(np.clip(rng.uniform(0.1, 0.9), w / 2, 0.999 - w / 2))
            cy = float(np.clip(rng.uniform(0.1, 0.9), h / 2, 0.999 - h / 2))
            t = fill[b, n]
            tgt[b, n, 5 * t:5 * t + 5] = [n, cx, cy, w, h]
            fill[b, n] += 1
    return torch.from_numpy(tgt)


def test_full_architecture_bf16_mode_vs_oracle_emulation(dev, tmp_path):
    """darknet_dynamic.cfg + reweighting_net.cfg (66.3 M parameters) in bf16 storage mode against the oracle's
    restatement of that mode (oracle/net.py _walk_bf16: bf16-rounded conv outputs, activations and packed weights, fp32
    accumulation, BatchNorm statistics from the unrounded conv output).  Agreement is at bf16 resolution: an accumulator
    that sits on a rounding boundary may round the other way after a 1e-7 difference in summation order."""
    from fewshot_detection_amd import cfgs
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.darknet_meta import Darknet
    from oracle.net import OracleDarknet
    from oracle.region import region_loss_v2
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    torch.manual_seed(31)
    ora = OracleDarknet(dyn_cfg, rw_cfg).train()
    net = Darknet(dyn_cfg, rw_cfg)
    net.load_state_dict(ora.state_dict())
    net = net.to(dev).train().set_compute_dtype("bf16")
    B, N, S = 4, 3, 416
    g = torch.Generator().manual_seed(32)
    x, metax = torch.rand(B, 3, S, S, generator=g), torch.rand(N, 3, S, S, generator=g)
    mask = torch.zeros(N, 1, S, S)
    mask[:, :, 100:300, 50:250] = 1
    tgt = _targets(np.random.RandomState(33), B, N)
    cfg.neg_ratio = "full"
    region = net.models[len(net.models) - 1]
    region.verbose = False
    region.seen = 0
    out = net(x.to(dev), metax.to(dev), mask.to(dev))
    assert net._det.fallback_convs == 0 and net._meta.fallback_convs == 0          # every layer on the bf16 kernels
    loss = region(out, tgt)
    loss.backward()
    ref, dyn_ref = ora.forward_bf16(x, metax, mask)
    r = region_loss_v2(ref, tgt, ora.region.anchors, seen=0)
    r["loss"].backward()
    o, rf = out.detach().cpu(), ref.detach()
    rel = float((o - rf).norm() / rf.norm())
    print("bf16 mode forward: relative L2 %.3e, max|diff| %.3e (max|out| %.2f); loss %.2f vs %.2f"
          % (rel, float((o - rf).abs().max()), float(rf.abs().max()), float(loss.detach()), float(r["loss"].detach())))
    # Forward agreement: the two runs share every rounding point and differ only where an fp32 accumulator sits within
    # ~1e-7 of a bf16 rounding boundary (2.7e-5 relative after layer 0).  This 30-layer, randomly initialised, batch-
    # normalised net amplifies any perturbation ~1.35x per layer (the fp32 path shows the same factor: 1e-7 -> 3.8e-4,
    # test_gpu_timed_config.py), so 2.7e-5 becomes ~8e-2 at the head; tools/probes/bf16_layers_debug.py prints the curve.
    # The layer-by-layer test below (identical inputs per layer) is the tight one.
    assert out.shape == ref.shape and rel < 0.2
    # the loss is a sum over the same outputs: a 0.1 relative-L2 forward drift moves it by a few 1e-3 (3.4e-3 measured after
    # the first layer moved to the split arithmetic, 2.5e-4 before: which side of a rounding boundary, not accuracy)
    # (back to the measured level -- 3.4e-3 here, 2e-4 ... 6e-4 at the timed shapes of test_gpu_launch_configs.py)
    assert abs(float(loss.detach()) - float(r["loss"].detach())) < 5e-3 * abs(float(r["loss"].detach()))
    named, mine = dict(ora.named_parameters()), dict(net.named_parameters())
    cos = []
    for name, p in mine.items():
        gm, gr = p.grad.cpu().double().flatten(), named[name].grad.double().flatten()
        cos.append((float(torch.dot(gm, gr) / (gm.norm() * gr.norm() + 1e-30)), name))
    cos.sort()
    print("bf16 mode gradients: worst cosines", cos[:4], "median %.4f" % cos[len(cos) // 2][0])
    # gradients inherit the forward divergence (the activations they are evaluated at differ by ~10 %): directions agree
    assert cos[0][0] > 0.6 and cos[len(cos) // 2][0] > 0.8
    gm, gr = mine["models.31.conv24.bias"].grad.cpu(), named["models.31.conv24.bias"].grad
    assert float((gm - gr).norm() / gr.norm()) < 0.05           # next to the loss: depends on the head output only


def test_bf16_blocks_layer_by_layer_on_identical_inputs(dev, tmp_path):
    """Every conv + BatchNorm + leaky (+ maxpool) block of darknet_dynamic.cfg in bf16 storage mode, each fed the
    ORACLE's (bf16-valued) input of that block: conv kernel + statistics from the fp32 accumulators + affine/activation/
    pool pass == oracle/net.py _conv_block_bf16, up to the rare accumulator that rounds the other way (one bf16 ulp)."""
    from fewshot_detection_amd import cfgs, ops
    from fewshot_detection_amd.darknet_meta import Darknet
    from oracle import net as onet
    from oracle.net import OracleDarknet
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    torch.manual_seed(41)
    ora = OracleDarknet(dyn_cfg, rw_cfg).train()
    net = Darknet(dyn_cfg, rw_cfg)
    net.load_state_dict(ora.state_dict())
    net = net.to(dev).train().set_compute_dtype("bf16")
    eng = net._det
    eng._record = False
    x = torch.rand(2, 3, 416, 416, generator=torch.Generator().manual_seed(42))
    outs, xx = {}, x
    blocks, mods = ora.blocks, ora.models
    worst = 0.0
    with torch.no_grad():
        for idx, blk in enumerate(blocks[1:]):
            kind = blk["type"]
            if kind == "route":
                src = [int(v) if int(v) > 0 else int(v) + idx for v in blk["layers"].split(",")]
                xx = outs[src[0]] if len(src) == 1 else torch.cat([outs[s] for s in src], 1)
            elif kind == "convolutional" and onet.is_dynamic(blk):
                break
            elif kind == "convolutional":
                ref = onet._conv_block_bf16(mods[idx], xx, True)
                # the same block on the device, from the oracle's input
                if xx.shape[1] <= 4:
                    xin = ops.nchw_to_nhwc(xx.to(dev))
                else:
                    xin = _view_bf16(xx, dev)
                z, _ = eng._conv(idx, blk, xin, True, 0, {}, [])
                got = _nchw(z)
                d = (got - ref).abs()
                rel = float((got - ref).norm() / ref.norm())
                # a flipped rounding of one conv output moves z by one bf16 ulp of y times the BatchNorm scale
                assert float(d.max()) <= 2.0 ** -7 * float(ref.abs().max()) * 1.5, (idx, float(d.max()), float(ref.abs().max()))
                assert rel < 3e-4, (idx, rel)
                worst = max(worst, rel)
                xx = ref
            else:
                xx = mods[idx](xx)
            outs[idx] = xx
    print("bf16 blocks on identical inputs: worst relative L2 %.2e" % worst)


def test_bf16_mode_trains_on_the_reduced_width_net(dev):
    """mini_dynamic / mini_reweight (4 ... 64 channels): channel counts below the bf16 kernels' granularity go through the
    documented fp32-kernel fallback on bf16-valued operands; the step still follows the fp32 model."""
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.darknet_meta import Darknet
    torch.manual_seed(22)
    cfgs_ = (os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    a = Darknet(*cfgs_)
    b = Darknet(*cfgs_)
    b.load_state_dict(a.state_dict())
    a, b = a.to(dev).train(), b.to(dev).train().set_compute_dtype("bf16")
    x, metax = torch.rand(2, 3, 96, 96, device=dev), torch.rand(3, 3, 96, 96, device=dev)
    mask = (torch.rand(3, 1, 96, 96, device=dev) > 0.5).float()
    tgt = torch.zeros(2, 3, 250, dtype=torch.float64)
    tgt[0, 1, :5] = torch.tensor([1, 0.5, 0.5, 0.4, 0.3])
    tgt[1, 2, :5] = torch.tensor([2, 0.3, 0.6, 0.2, 0.5])
    cfg.neg_ratio = "full"
    losses, grads = [], []
    for net in (a, b):
        region = net.models[len(net.models) - 1]
        region.verbose = False
        loss = region(net(x, metax, mask), tgt)
        loss.backward()
        losses.append(float(loss.detach()))
        grads.append(torch.cat([p.grad.flatten() for p in net.parameters()]))
    assert b._det.fallback_convs > 0
    assert abs(losses[0] - losses[1]) < 0.05 * abs(losses[0])
    cos = float(torch.dot(grads[0], grads[1]) / (grads[0].norm() * grads[1].norm()))
    assert cos > 0.8, cos          # bf16 noise on a tiny random net with 2x2 .. 6x6 feature maps


def test_bf16_mode_loss_trajectory_follows_fp32_over_120_sgd_steps_on_the_full_width_nets(dev, tmp_path):
    """darknet_dynamic.cfg + reweighting_net.cfg (66.3 M parameters), the same initial state, the same fixed episode, 120
    SGD(momentum) steps in each storage mode through EpisodeTrainer (the bench's step).  bf16 storage perturbs every layer at
    2^-9 relative, so the two runs are different samples of the same optimisation, not the same numbers: what must hold is that
    both descend, at the same pace -- the running loss of the bf16 run stays within 15 % of the fp32 run's at every tenth of
    the way, and both end far below where they started (measured: 1087 -> 1.5 and 1093 -> 2.1; tenths 458/458, 41/42, 15/17,
    9/10, 5/6, 4/4, ...).  The learning rate is small enough for a smooth descent: at 5x this rate both runs overshoot to 6-7 k
    in the first steps before they recover, and the transients of two chaotic runs are not comparable."""
    from fewshot_detection_amd import cfgs
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.darknet_meta import Darknet
    from fewshot_detection_amd.dp import EpisodeTrainer
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    torch.manual_seed(41)
    ref = Darknet(dyn_cfg, rw_cfg)
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    B, N, S, steps = 4, 5, 160, 120
    g = torch.Generator().manual_seed(42)
    x, metax = torch.rand(B, 3, S, S, generator=g).to(dev), torch.rand(N, 3, S, S, generator=g).to(dev)
    mask = torch.zeros(N, 1, S, S)
    mask[:, :, 40:120, 30:110] = 1
    mask = mask.to(dev)
    tgt = _targets(np.random.RandomState(43), B, N)
    cfg.neg_ratio = "full"
    curves = {}
    for mode in ("f32", "bf16"):
        net = Darknet(dyn_cfg, rw_cfg)
        net.load_state_dict(state)
        net = net.to(dev).train().set_compute_dtype(mode)
        region = net.models[len(net.models) - 1]
        region.verbose = False
        region.seen = 0
        trainer = EpisodeTrainer(net, lr=2e-5 / B, momentum=0.9, weight_decay=5e-4 * B,
                                 grad_dtype=torch.bfloat16 if mode == "bf16" else torch.float32)
        losses = []
        for _ in range(steps):
            loss = region(net(x, metax, mask), tgt)
            losses.append(loss.detach())
            trainer.backward_and_step(loss)
        curves[mode] = torch.stack(losses).float().cpu()
        assert torch.isfinite(curves[mode]).all()
        if mode == "bf16":
            assert net._det.fallback_convs == 0 and net._meta.fallback_convs == 0
        del net, trainer
    f, h = curves["f32"], curves["bf16"]
    print("loss f32 %.1f -> %.1f, bf16 %.1f -> %.1f" % (float(f[0]), float(f[-1]), float(h[0]), float(h[-1])))
    assert float(f[-10:].mean()) < 0.5 * float(f[:10].mean()) and float(h[-10:].mean()) < 0.5 * float(h[:10].mean())
    tenths = [(float(f[lo:lo + steps // 10].mean()), float(h[lo:lo + steps // 10].mean())) for lo in range(0, steps, steps // 10)]
    print("tenths (f32, bf16): " + "  ".join("%.1f/%.1f" % t for t in tenths))
    print("first steps f32 " + " ".join("%.0f" % v for v in f[:12]) + " | bf16 " + " ".join("%.0f" % v for v in h[:12]))
    for a, b in tenths:
        assert abs(a - b) < 0.15 * a + 1.0, tenths          # (+ 1.0: a thousandth of the starting loss, for the flat tail)


def test_fused_sgd_and_bf16_repack_equals_the_two_pass_form(dev, tmp_path):
    """fsd_sgd_step_multi (VERDICT r5 #2d): the optimizer step of the bf16 storage mode updates every tensor of the flat
    buffer in one launch per bucket and writes the bf16 forward / data-gradient operands of the NEW conv weights in the same
    pass.  Against the two-pass form it replaces (fsd_sgd_step per bucket, then 20 fsd_pack_conv_weight_bf16_pair launches at
    the next forward): same parameters and momentum after 3 steps (fp32 round-off of two compilations of the same
    expression), the kept operand copies BIT-equal to a fresh packing of the updated weights, the same losses, and no pack
    launch left in the steps after the first (the weight cache sees fresh copies)."""
    from fewshot_detection_amd import cfgs, ops
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.darknet_meta import Darknet
    from fewshot_detection_amd.dp import EpisodeTrainer
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    torch.manual_seed(51)
    ref = Darknet(dyn_cfg, rw_cfg)
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    B, N, S = 4, 5, 160
    g = torch.Generator().manual_seed(52)
    x, metax = torch.rand(B, 3, S, S, generator=g).to(dev), torch.rand(N, 3, S, S, generator=g).to(dev)
    mask = torch.zeros(N, 1, S, S)
    mask[:, :, 40:120, 30:110] = 1
    mask = mask.to(dev)
    tgt = _targets(np.random.RandomState(53), B, N)
    cfg.neg_ratio = "full"
    res = {}
    for fused in (True, False):
        net = Darknet(dyn_cfg, rw_cfg)
        net.load_state_dict(state)
        net = net.to(dev).train().set_compute_dtype("bf16")
        region = net.models[len(net.models) - 1]
        region.verbose = False
        region.seen = 0
        trainer = EpisodeTrainer(net, lr=2e-5 / B, momentum=0.9, weight_decay=5e-4 * B, grad_dtype=torch.bfloat16)
        trainer.FUSE_PACK = fused
        losses, packs = [], []
        for step in range(3):
            ops.launch_count(reset=True)
            packed_before = _pack_calls[0]
            loss = region(net(x, metax, mask), tgt)
            losses.append(float(loss.detach()))
            trainer.backward_and_step(loss)
            packs.append(_pack_calls[0] - packed_before)
        if fused:
            assert trainer.__dict__.get("_multi_built") is not None            # the tables were built and used
            assert packs[0] > 0 and packs[1] == 0 and packs[2] == 0, packs     # first forward packs; afterwards nobody does
            # the kept operand copies == a fresh packing of the weights as they are now, bit for bit
            n_checked = 0
            for eng in (net._det, net._meta):
                for p in eng.models.parameters():
                    pr = eng.cache.bf16_pair(p) if p.dim() == 4 else None
                    if pr is not None:
                        fresh = _real_pack(p.detach())
                        assert torch.equal(pr[0], fresh[0]) and torch.equal(pr[1], fresh[1])
                        n_checked += 1
            assert n_checked >= 25, n_checked
        else:
            assert packs[1] > 0 and packs[2] > 0, packs
        res[fused] = (losses, trainer.flat.clone(), trainer.mom.clone())
        del net, trainer
    (la, wa, ma), (lb, wb, mb) = res[True], res[False]
    assert la[0] == lb[0]                                                       # identical first step
    for a, b in zip(la, lb):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (la, lb)
    assert float((wa - wb).abs().max()) <= 1e-6 * float(wb.abs().max())
    assert float((ma - mb).abs().max()) <= 1e-5 * float(mb.abs().max())


# (count the calls of the pair-packing op without touching the product: a wrapper installed for this module's tests)
_pack_calls = [0]


def _install_pack_counter():
    from fewshot_detection_amd import ops
    real = ops.pack_weight_bf16_pair

    def counted(w, out=None):
        _pack_calls[0] += 1
        return real(w, out)
    ops.pack_weight_bf16_pair = counted
    return real


_real_pack = _install_pack_counter()


def test_bf16_error_growth_stays_within_the_committed_table(dev):
    """the bf16 mode's end-to-end deviation (rel-L2 ~0.23 vs the fp32 oracle at random init) is pinned layer by
    layer.  profiles/r06_bf16_error_growth.{md,json} (tools/bf16_error_growth.py, B = 8, train-mode BatchNorm) holds, after
    every block, hip-vs-fp32, walk-vs-fp32 (the storage mode's own cost, no kernel involved) and hip-vs-walk; the same
    measurement is repeated here and every entry must stay within 1.5x of the committed value (+ 2e-5: the first layers sit
    at single rounding flips), and the product must never be further from fp32 than 1.5x what the mode's definition is."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location(
        "bf16_error_growth", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bf16_error_growth.py"))
    eg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(eg)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_bf16_error_growth.json")
    committed = {r["layer"]: r for r in json.load(open(path))["rows"]}
    rows = eg.measure(dev)
    assert [r[0] for r in rows] == list(committed)
    for name, hip_f32, walk_f32, hip_walk in rows:
        c = committed[name]
        assert hip_walk <= 1.5 * c["hip_vs_walk"] + 2e-5, (name, hip_walk, c["hip_vs_walk"])
        assert hip_f32 <= 1.5 * c["hip_vs_fp32"] + 2e-5, (name, hip_f32, c["hip_vs_fp32"])
        assert hip_f32 <= 1.5 * walk_f32 + 2e-5, (name, hip_f32, walk_f32)      # the kernels add nothing to the mode's own error
    print("bf16 error growth: head (end to end) hip/fp32 %.3f, walk/fp32 %.3f, hip/walk %.3f" % rows[-1][1:])


def _blocks_on_identical_inputs(dev, tmp_path, B, perturb=None, bound=1.5e-4):
    """The teacher-forced per-block check of test_bf16_blocks_layer_by_layer_on_identical_inputs with a hook that can damage
    the kernel's output (to show the check would notice)."""
    from fewshot_detection_amd import cfgs, ops
    from fewshot_detection_amd.darknet_meta import Darknet
    from oracle import net as onet
    from oracle.net import OracleDarknet
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    torch.manual_seed(41)
    ora = OracleDarknet(dyn_cfg, rw_cfg).train()
    net = Darknet(dyn_cfg, rw_cfg)
    net.load_state_dict(ora.state_dict())
    net = net.to(dev).train().set_compute_dtype("bf16")
    eng = net._det
    eng._record = False
    xx = torch.rand(B, 3, 416, 416, generator=torch.Generator().manual_seed(42))
    outs, worst = {}, 0.0
    with torch.no_grad():
        for idx, blk in enumerate(ora.blocks[1:]):
            kind = blk["type"]
            if kind == "route":
                src = [int(v) if int(v) > 0 else int(v) + idx for v in blk["layers"].split(",")]
                xx = outs[src[0]] if len(src) == 1 else torch.cat([outs[s] for s in src], 1)
            elif kind == "convolutional" and onet.is_dynamic(blk):
                break
            elif kind == "convolutional":
                ref = onet._conv_block_bf16(ora.models[idx], xx, True)
                xin = ops.nchw_to_nhwc(xx.to(dev)) if xx.shape[1] <= 4 else _view_bf16(xx, dev)
                z, _ = eng._conv(idx, blk, xin, True, 0, {}, [])
                if perturb is not None:
                    perturb(idx, z)
                got = _nchw(z)
                rel = float((got - ref).norm() / ref.norm())
                assert float((got - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max()) * 1.5, (idx, "max")
                assert rel < bound, (idx, rel)
                worst = max(worst, rel)
                xx = ref
            else:
                xx = ora.models[idx](xx)
            outs[idx] = xx
    return worst


def test_a_one_ulp_rounding_bug_in_a_bf16_kernel_fails_the_per_block_check(dev, tmp_path):
    """The per-block bound is 1.5e-4 relative L2 on identical inputs (measured 7-8e-5 at every batch size: rare accumulators
    that sit on a rounding boundary).  Demonstration that it is tight enough (VERDICT r5 #3 'done' criterion): the same check
    with one bf16 ulp added to every 16th stored element of ONE layer's output -- what a wrong rounding mode in 6 % of the
    lanes of one kernel would do -- fails; undamaged it passes."""
    assert _blocks_on_identical_inputs(dev, tmp_path, 2) < 1.5e-4

    def one_ulp_in_layer_12(idx, z):
        if idx == 12:
            bits = z.t.view(torch.int16)
            bits[:, ::16] += 1                                  # next representable bf16 (sign-magnitude: away from zero)
    with pytest.raises(AssertionError):
        _blocks_on_identical_inputs(dev, tmp_path, 2, perturb=one_ulp_in_layer_12)
