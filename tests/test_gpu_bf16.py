"""bf16 storage mode (BASELINE configs[2] / [4]) on the MI355X: the DMA-staged bf16 MFMA convolution and the
transpose-read bf16 weight gradient against fp64 convolutions of the SAME bf16-rounded operands (the products are
exact in fp32, so only the accumulation order differs), the bf16 twins of the HBM-bound kernels against their fp32
versions, and the whole model against the CPU oracle run in the same mixed precision."""
import os

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
