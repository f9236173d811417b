"""One-sweep backward of a first conv block (csrc/first_bwd.hip: dW = c1 (S1 - c2 S2 - c3 S3), dt never written) against
the unfused kernel sequence it replaces and against fp64 autograd of the same block."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _block_inputs(dev, B, H, W, cin, cout, seed, bf16):
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.3
    gamma = torch.rand(cout, generator=g) + 0.5
    gamma[::5] *= -1.0                                         # negative scales flip the pool winner
    beta = torch.randn(cout, generator=g) * 0.2
    dz = torch.randn(B, cout, H // 2, W // 2, generator=g)
    if bf16:
        dz = dz.to(torch.bfloat16).float()
    xv = ops.nchw_to_nhwc(x.to(dev))                           # NHWC4 (channel 3 zero for cin = 3)
    return x, w, gamma, beta, dz, xv


def _unfused(ops, dzv, yv, scale, shift, mean, invstd, slope, xv, cin, cout, bn):
    dt, partial = ops.bn_act_pool_bwd(dzv, None, yv, scale, shift, mean, invstd, slope, 1)
    dbeta, dgamma, coef = ops.reduce_partials(partial, yv.pixels, cout, scale=scale, want_coef=True)
    dw = ops.conv3x3_wgrad_c4_bnfused(dt, yv, coef, mean, invstd, xv, cin, cout)
    return dw, dbeta, dgamma


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 32, 48, 3, 32), (3, 20, 12, 4, 64), (5, 6, 2, 3, 32), (1, 2, 2, 4, 32),
                                            (4, 64, 64, 3, 32)])
@pytest.mark.parametrize("bf16", [False, True, "native"])
def test_first_block_backward_one_sweep(dev, B, H, W, cin, cout, bf16):
    """bf16 = False: fp32 storage, the default arithmetic (six bf16-MFMA terms of split operands); "native": fp32 storage with
    fsd_f32_gemm_mode(0), the sums on v_mfma_f32_32x32x2_f32; True: bf16 storage (bf16 MFMA)."""
    from fewshot_detection_amd import ops
    if bf16 == "native":
        prev = ops.f32_gemm_mode("native")
        try:
            return _one_sweep(dev, B, H, W, cin, cout, False)
        finally:
            ops.f32_gemm_mode(prev)
    return _one_sweep(dev, B, H, W, cin, cout, bf16)


def _one_sweep(dev, B, H, W, cin, cout, bf16):
    from fewshot_detection_amd import ops
    x, w, gamma, beta, dz, xv = _block_inputs(dev, B, H, W, cin, cout, 11 + H + cin, bf16)
    bn = torch.nn.BatchNorm2d(cout).to(dev)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    dt = torch.bfloat16 if bf16 else torch.float32
    yv, part = ops.conv3x3_c4(xv, w.to(dev), cout, bn_partial=True, out_dtype=dt)
    scale, shift, mean, invstd = ops.bn_finalize(part, xv.pixels, bn, True)
    dzv = ops.nchw_to_nhwc(dz.to(dev), pad_to=4, dtype=dt)
    dw_u, dbeta_u, dgamma_u = _unfused(ops, dzv, yv, scale, shift, mean, invstd, 0.1, xv, cin, cout, bn)
    dw_f, dbeta_f, dgamma_f = ops.first_layer_bwd(dzv, yv, scale, shift, mean, invstd, 0.1, xv, cin, cout, bn, True)
    torch.cuda.synchronize()
    # same arithmetic up to summation order (and, for dW, the algebraic regrouping): compare relative to each tensor's max
    for a, b, tol, name in ((dbeta_f, dbeta_u, 2e-5, "dbeta"), (dgamma_f, dgamma_u, 2e-5, "dgamma"),
                            (dw_f, dw_u, 2e-4 if not bf16 else 2e-3, "dW")):
        err = float((a - b).abs().max()) / max(1e-12, float(b.abs().max()))
        assert err < tol, (name, err)
    if not bf16:
        # fp64 autograd of conv -> train-mode BN -> leaky -> maxpool with the same operands
        x64 = x.double()
        w64 = w.double().requires_grad_(True)
        g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
        y = F.conv2d(x64, w64, None, 1, 1)
        z = F.max_pool2d(F.leaky_relu(F.batch_norm(y, None, None, g64, b64, True, 0.1, 1e-5), 0.1), 2, 2)
        z.backward(dz.double())
        for a, ref, name in ((dw_f, w64.grad, "dW"), (dgamma_f, g64.grad, "dgamma"), (dbeta_f, b64.grad, "dbeta")):
            err = float((a.cpu().double() - ref).abs().max()) / max(1e-12, float(ref.abs().max()))
            assert err < 5e-4, (name, err)


def test_first_block_backward_frozen_statistics(dev):
    """eval-mode BatchNorm (running statistics): dy = scale * dt, the coefficient rows c2 / c3 are zero."""
    from fewshot_detection_amd import ops
    B, H, W, cin, cout = 2, 16, 16, 3, 32
    x, w, gamma, beta, dz, xv = _block_inputs(dev, B, H, W, cin, cout, 5, False)
    bn = torch.nn.BatchNorm2d(cout).to(dev).eval()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
        bn.running_mean.uniform_(-0.2, 0.2)
        bn.running_var.uniform_(0.5, 1.5)
    yv, _ = ops.conv3x3_c4(xv, w.to(dev), cout)
    scale, shift, mean, invstd = ops.bn_finalize(None, xv.pixels, bn, False)
    dzv = ops.nchw_to_nhwc(dz.to(dev), pad_to=4)
    dw_f, _, _ = ops.first_layer_bwd(dzv, yv, scale, shift, mean, invstd, 0.1, xv, cin, cout, bn, False)
    x64, w64 = x.double(), w.double().requires_grad_(True)
    y = F.conv2d(x64, w64, None, 1, 1)
    t = F.batch_norm(y, bn.running_mean.cpu().double(), bn.running_var.cpu().double(), gamma.double(), beta.double(),
                     False, 0.1, 1e-5)
    F.max_pool2d(F.leaky_relu(t, 0.1), 2, 2).backward(dz.double())
    err = float((dw_f.cpu().double() - w64.grad).abs().max()) / float(w64.grad.abs().max())
    assert err < 5e-4, err
