"""Backward-pass parity on the MI355X: kernel level (wgrad, BN/leaky/pool backward) against fp64
autograd, model level against gradients minted from the reference and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,H,W,cin,cout,k", [
    (2, 13, 13, 64, 128, 3), (3, 9, 7, 3, 32, 3), (2, 13, 13, 256, 30, 1), (1, 6, 6, 1280, 64, 3), (4, 26, 26, 32, 64, 3),
    (2, 13, 13, 1280, 1024, 3), (3, 104, 104, 64, 128, 3), (2, 26, 26, 512, 64, 1),
    # Winograd F(3x3,2x2) weight-gradient shapes (>=128 channels both sides), incl. odd extents
    (3, 13, 11, 128, 256, 3), (2, 52, 52, 128, 256, 3), (1, 7, 9, 256, 128, 3), (5, 1, 1, 128, 128, 3)])
def test_wgrad_matches_fp64_autograd(dev, B, H, W, cin, cout, k):
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(B, cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(cout, cin, k, k, generator=g, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(B, cout, H, W, generator=g, dtype=torch.float64)
    F.conv2d(x, w, None, 1, (k - 1) // 2).backward(gy)
    xv = ops.nchw_to_nhwc(x.float().to(dev))
    gv = ops.nchw_to_nhwc(gy.float().to(dev))
    dw = ops.conv2d_wgrad(gv, cout, xv, cin, k).cpu()
    ref = w.grad.float()
    assert torch.allclose(dw, ref, rtol=2e-4, atol=2e-4 * float(ref.abs().max())), float((dw - ref).abs().max())


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 13, 13, 3, 32), (3, 7, 2, 4, 64), (5, 64, 48, 3, 32), (1, 1, 2, 3, 32)])
def test_first_layer_fused_wgrad_matches_unfused_path(dev, B, H, W, cin, cout):
    """fsd_conv3x3_wgrad_c4_bnfused == fsd_bn_bwd_apply + fsd_conv2d_wgrad (and fp64 autograd of dy (*) x)."""
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.zeros(B, 4, H, W)
    x[:, :cin] = torch.randn(B, cin, H, W, generator=g)
    xv = ops.nchw_to_nhwc(x.to(dev))
    yv = ops.nchw_to_nhwc(torch.randn(B, cout, H, W, generator=g).to(dev))
    dt = ops.nchw_to_nhwc(torch.randn(B, cout, H, W, generator=g).to(dev))
    coef = (torch.rand(3, cout, generator=g) + 0.5).to(dev)
    mean, invstd = torch.randn(cout, generator=g).to(dev), (torch.rand(cout, generator=g) + 0.5).to(dev)
    assert ops.c4_bnfused_eligible(xv, cout, 3)
    dw = ops.conv3x3_wgrad_c4_bnfused(dt, yv, coef, mean, invstd, xv, cin, cout)
    dy = ops.View(dt.t.clone(), B, H, W, cout)
    ops.bn_bwd_apply(dy, yv, coef, mean, invstd)
    ref = ops.conv2d_wgrad(dy, cout, xv, cin, 3)
    assert dw.shape == ref.shape == (cout, cin, 3, 3)
    assert torch.allclose(dw, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max())), float((dw - ref).abs().max())
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    dy64 = dy.t.view(B, H, W, cout).permute(0, 3, 1, 2).double().cpu()
    F.conv2d(x[:, :cin].double(), w, None, 1, 1).backward(dy64)
    assert torch.allclose(dw.cpu(), w.grad.float(), rtol=2e-4, atol=2e-4 * float(w.grad.abs().max()))


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 13, 13, 1280, 1024), (3, 13, 11, 128, 256), (2, 52, 52, 128, 256),
                                             (1, 7, 9, 256, 128), (5, 1, 1, 128, 128), (2, 26, 26, 64, 128)])
def test_wgrad_winograd_4x4_matches_fp64_autograd(dev, B, H, W, cin, cout):
    """F(3x3, 4x4) weight gradient (36 batched reduction GEMMs over 4x4 tiles) against fp64 autograd; also with the
    transformed input kept by a tile-4 forward pass."""
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(B, cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(B, cout, H, W, generator=g, dtype=torch.float64)
    F.conv2d(x, w, None, 1, 1).backward(gy)
    xv = ops.nchw_to_nhwc(x.float().to(dev))
    gv = ops.nchw_to_nhwc(gy.float().to(dev))
    dw = ops.conv2d_wgrad(gv, cout, xv, cin, 3, tile=4)
    ref = w.grad.float()
    err = float((dw.cpu() - ref).abs().max())
    assert err < 1e-4 * float(ref.abs().max()), err
    kept = []
    ops.conv3x3_wino(xv, ops.pack_weight_wino(w.detach().float().to(dev), 0, 4), cout, keep_v=kept, tile=4)
    assert torch.equal(ops.conv2d_wgrad(gv, cout, xv, cin, 3, wino_v=kept[0], tile=4), dw)


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 13, 13, 128, 256), (1, 26, 22, 64, 128), (3, 5, 7, 128, 128)])
def test_fused_winograd_gradient_transforms_match_separate_path(dev, B, H, W, cin, cout):
    """fsd_wino_grad_transforms + (v_in, wt_in) == fsd_bn_bwd_apply + the separate dy transforms, bit for bit."""
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(H * W + cin)
    xv = ops.nchw_to_nhwc(torch.randn(B, cin, H, W, generator=g).to(dev))
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dev)
    yv = ops.nchw_to_nhwc(torch.randn(B, cout, H, W, generator=g).to(dev))
    dt = ops.nchw_to_nhwc(torch.randn(B, cout, H, W, generator=g).to(dev))
    coef = (torch.rand(3, cout, generator=g) + 0.5).to(dev)
    mean, invstd = torch.randn(cout, generator=g).to(dev), (torch.rand(cout, generator=g) + 0.5).to(dev)
    kept = []
    ops.conv3x3_wino(xv, ops.pack_weight_wino(w, 0, 4), cout, keep_v=kept, tile=4)
    u1 = ops.pack_weight_wino(w, 1, 4)
    vd, wt = ops.wino_grad_transforms(dt, yv, coef, mean, invstd)
    dw_f = ops.conv2d_wgrad(dt, cout, xv, cin, 3, wino_v=kept[0], tile=4, wt_in=wt)
    dx_f, _ = ops.conv3x3_wino(dt, u1, cin, tile=4, v_in=vd)
    dy = ops.View(dt.t.clone(), B, H, W, cout)
    ops.bn_bwd_apply(dy, yv, coef, mean, invstd)
    dw_s = ops.conv2d_wgrad(dy, cout, xv, cin, 3, wino_v=kept[0], tile=4)
    dx_s, _ = ops.conv3x3_wino(dy, u1, cin, tile=4)
    assert torch.equal(dw_f, dw_s)
    assert torch.equal(dx_f.t, dx_s.t)
    # the default path: BN backward fused into the weight-gradient transform only, dt -> dy in place
    dt2 = ops.View(dt.t.clone(), B, H, W, cout)
    wt2 = ops.wino_dy_bn_transform(dt2, yv, coef, mean, invstd)
    assert torch.equal(dt2.t, dy.t)                                   # exactly what bn_bwd_apply leaves behind
    assert torch.equal(ops.conv2d_wgrad(dt2, cout, xv, cin, 3, wino_v=kept[0], tile=4, wt_in=wt2), dw_s)


def test_wgrad_winograd_agrees_with_direct(dev, monkeypatch):
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(5)
    xv = ops.nchw_to_nhwc(torch.randn(4, 256, 26, 26, generator=g).to(dev))
    gv = ops.nchw_to_nhwc(torch.randn(4, 512, 26, 26, generator=g).to(dev))
    assert ops.wino_eligible(256, 512, 3)
    a = ops.conv2d_wgrad(gv, 512, xv, 256, 3, tile=2)
    monkeypatch.setattr(ops, "WINOGRAD", False)
    b = ops.conv2d_wgrad(gv, 512, xv, 256, 3)
    assert not torch.equal(a, b)                      # really two different code paths
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))
    # the transformed input kept by the forward pass gives the same weight gradient bit for bit
    monkeypatch.setattr(ops, "WINOGRAD", True)
    kept = []
    ops.conv3x3_wino(xv, ops.pack_weight_wino(torch.randn(512, 256, 3, 3, generator=g).to(dev)), 512, keep_v=kept)
    c = ops.conv2d_wgrad(gv, 512, xv, 256, 3, wino_v=kept[0], tile=2)
    assert torch.equal(a, c)


@pytest.mark.parametrize("pool,B,H,W,cin,cout", [(0, 2, 13, 13, 8, 16), (1, 2, 13, 13, 8, 16), (2, 2, 13, 13, 8, 16),
                                                  (0, 2, 13, 13, 1280, 1024), (1, 3, 104, 104, 64, 128),
                                                  (1, 2, 26, 26, 256, 512)])
def test_conv_bn_leaky_pool_block_backward(dev, pool, B, H, W, cin, cout):
    """Whole fused block: dgamma, dbeta, dW and dx against torch autograd (fp64), incl. full-size layer shapes."""
    from fewshot_detection_amd import ops
    torch.manual_seed(10 + pool)
    x = torch.randn(B, cin, H, W, dtype=torch.float64, requires_grad=True)
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1, bias=False).double()
    bn = torch.nn.BatchNorm2d(cout).double()
    bn.weight.data.uniform_(-1, 1)
    bn.bias.data.uniform_(-0.5, 0.5)
    z = F.leaky_relu(bn(conv(x)), 0.1)
    if pool == 1:
        z = F.max_pool2d(z, 2, 2)
    elif pool == 2:
        z = F.max_pool2d(F.pad(z, (0, 1, 0, 1), mode="replicate"), 2, stride=1)
    gz = torch.randn_like(z)
    z.backward(gz)

    bn32 = torch.nn.BatchNorm2d(cout).to(dev)
    bn32.weight.data.copy_(bn.weight.data.float())
    bn32.bias.data.copy_(bn.bias.data.float())
    w32 = conv.weight.data.float().to(dev)
    xv = ops.nchw_to_nhwc(x.detach().float().to(dev))
    yv, part = ops.conv2d(xv, ops.pack_weight(w32), cout, 3, bn_partial=True)
    scale, shift, mean, invstd = ops.bn_finalize(part, xv.pixels, bn32, True)
    gzv = ops.nchw_to_nhwc(gz.float().to(dev))
    dt, partial = ops.bn_act_pool_bwd(gzv, None, yv, scale, shift, mean, invstd, 0.1, pool)
    dbeta, dgamma, coef = ops.reduce_partials(partial, yv.pixels, cout, scale=scale, want_coef=True)
    ops.bn_bwd_apply(dt, yv, coef, mean, invstd)
    dw = ops.conv2d_wgrad(dt, cout, xv, cin, 3)
    dx, _ = ops.conv2d(dt, ops.pack_weight(w32, mode=1), cin, 3)
    tol = dict(rtol=1e-3, atol=1e-4)
    assert torch.allclose(dbeta.cpu(), bn.bias.grad.float(), **tol)
    assert torch.allclose(dgamma.cpu(), bn.weight.grad.float(), **tol)
    # dW of the wide layers comes from the Winograd F(3x3,4x4) reduction: round-off ~1.5e-5 of the tensor's magnitude
    wref = conv.weight.grad.float()
    assert float((dw.cpu() - wref).abs().max()) < 1e-4 * float(wref.abs().max())
    assert torch.allclose(ops.nhwc_to_nchw(dx).cpu(), x.grad.float(), **tol)


@pytest.mark.parametrize("pool,full,B,H,W,C", [(0, False, 2, 13, 13, 1024), (1, False, 3, 104, 104, 128), (1, True, 2, 26, 26, 512),
                                               (0, True, 2, 26, 26, 64), (1, False, 2, 13, 13, 32), (1, False, 1, 7, 9, 16)])
def test_statistics_only_first_pass_and_recomputing_second_pass_are_bit_identical(dev, pool, full, B, H, W, C):
    """fp32 BatchNorm backward without a materialised dt (ops.DEFER_DT): bn_act_pool_bwd(want_dt=False) returns the same
    partial sums, and bn_bwd_apply_g / wino_dy_bn_transform_g write the same dy (and Winograd operand) as the two-pass form,
    bit for bit -- with and without the 2x2 / stride-2 pool (odd sizes: border cells without a pooling window), with and
    without a gradient on the un-pooled tap."""
    from fewshot_detection_amd import ops
    torch.manual_seed(pool * 7 + C)
    yv = ops.nchw_to_nhwc(torch.randn(B, C, H, W, device=dev))
    OH, OW = (H // 2, W // 2) if pool else (H, W)
    gz = ops.nchw_to_nhwc(torch.randn(B, C, OH, OW, device=dev))
    gzf = ops.nchw_to_nhwc(torch.randn(B, C, H, W, device=dev)) if full else None
    scale = torch.rand(C, device=dev) + 0.5
    shift = torch.randn(C, device=dev) * 0.3
    mean = torch.randn(C, device=dev) * 0.1
    invstd = torch.rand(C, device=dev) + 0.5
    dt, partial = ops.bn_act_pool_bwd(gz, gzf, yv, scale, shift, mean, invstd, 0.1, pool)
    none, partial2 = ops.bn_act_pool_bwd(gz, gzf, yv, scale, shift, mean, invstd, 0.1, pool, want_dt=False)
    assert none is None and torch.equal(partial, partial2)
    _, _, coef = ops.reduce_partials(partial, yv.pixels, C, scale=scale, want_coef=True)
    dy_g = ops.bn_bwd_apply_g(gz, gzf, yv, scale, shift, 0.1, pool, coef, mean, invstd)
    dy_w, wt_g = ops.wino_dy_bn_transform_g(gz, gzf, yv, scale, shift, 0.1, pool, coef, mean, invstd)
    wt = ops.wino_dy_bn_transform(dt, yv, coef, mean, invstd)            # dt -> dy in place
    assert torch.equal(dy_g.t, dt.t) and torch.equal(dy_w.t, dt.t) and torch.equal(wt_g, wt)
    dt2, _ = ops.bn_act_pool_bwd(gz, gzf, yv, scale, shift, mean, invstd, 0.1, pool)
    ops.bn_bwd_apply(dt2, yv, coef, mean, invstd)
    assert torch.equal(dt2.t, dy_g.t)


@pytest.mark.parametrize("gemm", ["split", "native"])
def test_activation_formed_on_load_is_bit_identical_to_the_materialised_one(dev, gemm):
    """Deferred activations (ops.View.lazy): conv2d (1x1), conv3x3_wino (F(4x4)) and the 1x1 weight gradient form
    leaky(y * scale + shift) in their staging registers; results equal those on the materialised activation bit for bit."""
    from fewshot_detection_amd import ops
    before = ops.f32_gemm_mode(gemm)
    try:
        torch.manual_seed(21)
        B, H, W, C, cout = 3, 13, 13, 256, 128
        yv = ops.nchw_to_nhwc(torch.randn(B, C, H, W, device=dev))
        scale = torch.rand(C, device=dev) + 0.5
        shift = torch.randn(C, device=dev) * 0.3
        lazy = ops.View(yv.t, B, H, W, C, 0, lazy=(scale, shift, 0.1))
        act = ops.materialise(lazy)
        assert act.lazy is None and act.t.data_ptr() != yv.t.data_ptr()
        w1 = torch.randn(cout, C, 1, 1, device=dev) * 0.05
        w3 = torch.randn(cout, C, 3, 3, device=dev) * 0.05
        a, pa = ops.conv2d(act, ops.pack_weight(w1), cout, 1, bn_partial=True)
        b, pb = ops.conv2d(lazy, ops.pack_weight(w1), cout, 1, bn_partial=True)
        assert torch.equal(a.t, b.t) and torch.equal(pa, pb)
        u = ops.pack_weight_wino(w3, 0, 4)
        ka, kb = [], []
        a, pa = ops.conv3x3_wino(act, u, cout, bn_partial=True, keep_v=ka, tile=4)
        b, pb = ops.conv3x3_wino(lazy, u, cout, bn_partial=True, keep_v=kb, tile=4)
        assert torch.equal(a.t, b.t) and torch.equal(pa, pb) and torch.equal(ka[0], kb[0])
        dy = ops.nchw_to_nhwc(torch.randn(B, cout, H, W, device=dev))
        assert torch.equal(ops.conv2d_wgrad(dy, cout, act, C, 1), ops.conv2d_wgrad(dy, cout, lazy, C, 1))
        assert torch.equal(ops.conv2d_wgrad(dy, cout, act, C, 3, wino_v=ka[0], tile=4),
                           ops.conv2d_wgrad(dy, cout, lazy, C, 3, wino_v=kb[0], tile=4))
        with pytest.raises(ValueError):
            ops.conv2d_wgrad(dy, cout, lazy, C, 3, tile=0)          # a direct 3x3 weight gradient reads a materialised x
    finally:
        ops.f32_gemm_mode(before)


@pytest.mark.parametrize("shape,slope", [
    ((64, 13, 13, 1024, 512), 0.1),      # L19 / L21 of the detector at B = 64: 42 x 4 = 168 tiles of 256 x 128
    ((24, 26, 26, 512, 512), 0.1),       # 63 x 4 = 252 tiles, 96 rows left over for the 64-row tail launch
    ((24, 26, 26, 512, 512), 1.7),       # a slope outside [0, 1]: the ACT = 2 form (select instead of max)
    ((64, 13, 13, 1024, 512), -0.3),
])
def test_activation_on_load_in_the_8_wave_1x1_kernel_is_bit_identical(dev, shape, slope):
    """the 1x1 layers that go to conv_gemm_split8_kernel (>= 128 tiles of 256 x 128: split8_1x1 in csrc/conv.hip)
    form leaky(y * scale + shift) in their hand-scheduled staging pipeline (ACT = 1: 0 <= slope <= 1 as max(t, slope * t);
    ACT = 2: any slope).  Output and BatchNorm partial sums equal those on the materialised activation bit for bit -- the
    small case above (507 pixels) never reaches this kernel."""
    from fewshot_detection_amd import ops
    before = ops.f32_gemm_mode("split")
    try:
        B, H, W, C, cout = shape
        assert (B * H * W // 256) * (cout // 128) >= 128         # the launcher's condition for the 8-wave kernel
        torch.manual_seed(sum(shape))
        yv = ops.nchw_to_nhwc(torch.randn(B, C, H, W, device=dev))
        scale = torch.rand(C, device=dev) + 0.5
        shift = torch.randn(C, device=dev) * 0.3
        lazy = ops.View(yv.t, B, H, W, C, 0, lazy=(scale, shift, slope))
        act = ops.materialise(lazy)
        w1 = torch.randn(cout, C, 1, 1, device=dev) * 0.05
        pw = ops.pack_weight(w1)
        a, pa = ops.conv2d(act, pw, cout, 1, bn_partial=True)
        b, pb = ops.conv2d(lazy, pw, cout, 1, bn_partial=True)
        assert torch.equal(a.t, b.t) and torch.equal(pa, pb)
        ref = F.conv2d(F.leaky_relu(ops.nhwc_to_nchw(yv).double() * scale.double().view(1, -1, 1, 1)
                                    + shift.double().view(1, -1, 1, 1), slope), w1.double())
        got = ops.nhwc_to_nchw(b).double()
        assert float((got - ref).norm() / ref.norm()) < 5e-6
    finally:
        ops.f32_gemm_mode(before)


def test_training_step_with_deferred_activations_equals_the_materialised_step(dev, tmp_path, monkeypatch):
    """engine.DEFER_ACTIVATION: the step that hands raw conv outputs to their single consumer (1x1 convs and F(4x4) layers
    of the standard detector) computes, bit for bit, the output and every gradient of the step that runs each BatchNorm +
    leaky pass -- with fewer passes."""
    from fewshot_detection_amd import cfgs, engine, ops
    from fewshot_detection_amd.darknet_meta import Darknet
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    calls = []
    real = ops.bn_act_pool
    monkeypatch.setattr(ops, "bn_act_pool", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    results = []
    for on in (False, True):
        monkeypatch.setattr(engine, "DEFER_ACTIVATION", on)
        torch.manual_seed(17)
        net = Darknet(dyn_cfg, rw_cfg).to(dev).train()
        x = torch.rand(2, 3, 160, 160, device=dev)
        metax = torch.rand(4, 3, 160, 160, device=dev)
        mask = (torch.rand(4, 1, 160, 160, device=dev) > 0.5).float()
        del calls[:]
        out = net(x, metax, mask)
        n_passes = len(calls)
        out.float().pow(2).sum().backward()
        results.append((out.detach().clone(), [p.grad.clone() for p in net.parameters() if p.grad is not None], n_passes))
    (o0, g0, n0), (o1, g1, n1) = results
    assert n1 <= n0 - 8, (n0, n1)                       # the 1x1 layers and the F(4x4) layers behind no-pool convs
    assert torch.equal(o0, o1) and len(g0) == len(g1)
    for a_, b_ in zip(g0, g1):
        assert torch.equal(a_, b_)


def _load_pair(dev, seed=0, randomize_bn=True):
    from fewshot_detection_amd.darknet_meta import Darknet
    from oracle.net import OracleDarknet
    cfgs = (os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    torch.manual_seed(seed)
    ora = OracleDarknet(*cfgs)
    if randomize_bn:
        for m in ora.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.uniform_(-0.2, 0.2)
    net = Darknet(*cfgs)
    net.load_state_dict(ora.state_dict())
    return ora.train(), net.to(dev).train()


def _rel_err(a, b):
    return float((a - b).abs().max()) / max(1e-6, float(b.abs().max()))


def test_model_gradients_vs_reference_golden(dev):
    """Gradients of the reference's own autograd (tests/golden/mini_forward.npz) for the same upstream grad."""
    from fewshot_detection_amd.darknet_meta import Darknet
    d = np.load(os.path.join(GOLD, "mini_forward.npz"))
    net = Darknet(os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    net.load_weights(os.path.join(GOLD, "mini.weights"))
    net = net.to(dev).train()
    x, metax, mask = (torch.from_numpy(d[k]).to(dev) for k in ("x", "metax", "mask"))
    out = net(x, metax, mask)
    out.backward(torch.from_numpy(d["grad_out"]).to(dev))
    named = dict(net.named_parameters())
    checked = 0
    for k in d.files:
        if k.startswith("grad:"):
            g = named[k[5:]].grad
            assert g is not None, k
            assert _rel_err(g.cpu(), torch.from_numpy(d[k])) < 2e-3, (k, _rel_err(g.cpu(), torch.from_numpy(d[k])))
            checked += 1
    assert checked == 9


def test_full_episode_gradients_vs_oracle(dev):
    """forward + RegionLossV2 + backward: every parameter gradient against the CPU oracle (autograd)."""
    from fewshot_detection_amd.cfg import cfg
    from oracle.region import region_loss_v2
    ora, net = _load_pair(dev, seed=5)
    B, N, S = 3, 4, 128
    x, metax = torch.rand(B, 3, S, S), torch.rand(N, 3, S, S)
    mask = (torch.rand(N, 1, S, S) > 0.6).float()
    tgt = torch.zeros(B, N, 250, dtype=torch.float64)
    tgt[0, 1, :5] = torch.tensor([1, 0.52, 0.43, 0.4, 0.3])
    tgt[1, 2, :10] = torch.tensor([2, 0.3, 0.6, 0.2, 0.5, 2, 0.7, 0.2, 0.25, 0.2])
    tgt[2, 0, :5] = torch.tensor([0, 0.8, 0.8, 0.3, 0.3])
    cfg.neg_ratio = "full"
    region = net.models[len(net.models) - 1]
    region.verbose = False
    region.seen = 20000
    out = net(x.to(dev), metax.to(dev), mask.to(dev))
    loss = region(out, tgt)
    loss.backward()
    ref_out = ora(x, metax, mask)
    r = region_loss_v2(ref_out, tgt, ora.region.anchors, seen=20000)
    r["loss"].backward()
    assert abs(float(loss) - float(r["loss"])) < 1e-3 * max(1.0, abs(float(r["loss"])))
    ref = dict(ora.named_parameters())
    worst = 0.0
    for name, p in net.named_parameters():
        assert p.grad is not None, name
        e = _rel_err(p.grad.cpu(), ref[name].grad)
        worst = max(worst, e)
        assert e < 5e-3, (name, e)
    print("worst relative gradient error", worst)


def test_sgd_step_matches_torch_optim(dev):
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.dp import EpisodeTrainer
    from oracle.region import region_loss_v2
    ora, net = _load_pair(dev, seed=7)
    B, N, S = 2, 3, 64
    x, metax = torch.rand(B, 3, S, S), torch.rand(N, 3, S, S)
    mask = (torch.rand(N, 1, S, S) > 0.5).float()
    tgt = torch.zeros(B, N, 250, dtype=torch.float64)
    tgt[0, 1, :5] = torch.tensor([1, 0.5, 0.5, 0.4, 0.3])
    tgt[1, 0, :5] = torch.tensor([0, 0.3, 0.6, 0.2, 0.5])
    cfg.neg_ratio = "full"
    lr, mom, wd = 1e-5, 0.9, 5e-3      # small step: compares the optimizer arithmetic, not chaotic divergence
    region = net.models[len(net.models) - 1]
    region.verbose = False
    trainer = EpisodeTrainer(net, lr=lr, momentum=mom, weight_decay=wd)
    opt = torch.optim.SGD(ora.parameters(), lr=lr, momentum=mom, weight_decay=wd)
    for step in range(2):
        loss = region(net(x.to(dev), metax.to(dev), mask.to(dev)), tgt)
        trainer.backward_and_step(loss)
        opt.zero_grad()
        r = region_loss_v2(ora(x, metax, mask), tgt, ora.region.anchors, seen=0)
        r["loss"].backward()
        opt.step()
        assert abs(float(loss) - float(r["loss"])) < 2e-3 * max(1.0, abs(float(r["loss"]))), step
    ref = dict(ora.named_parameters())
    for name, p in net.named_parameters():
        assert _rel_err(p.detach().cpu(), ref[name].detach()) < 1e-3, name


def test_plain_yolo_backward_vs_oracle(dev):
    """tiny-yolo style net (MaxPoolStride1, RegionLoss v1, 40-channel biased head) end to end."""
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.darknet import Darknet
    from oracle.net import OracleYolo
    from oracle.region import region_loss_v1
    torch.manual_seed(11)
    c = os.path.join(GOLD, "mini_tiny_yolo.cfg")
    ora = OracleYolo(c).train()
    net = Darknet(c)
    net.load_state_dict(ora.state_dict())
    net = net.to(dev).train()
    x = torch.rand(2, 3, 64, 64)
    tgt = torch.zeros(2, 250, dtype=torch.float64)
    tgt[0, :5] = torch.tensor([2, 0.5, 0.5, 0.4, 0.3])
    tgt[1, :10] = torch.tensor([0, 0.3, 0.6, 0.2, 0.5, 1, 0.7, 0.2, 0.25, 0.2])
    cfg.neg_ratio, cfg.metayolo = "full", False
    try:
        region = net.models[len(net.models) - 1]
        region.verbose = False
        loss = region(net(x.to(dev)), tgt)
        loss.backward()
        r = region_loss_v1(ora(x), tgt, ora.region.anchors, 5, 3)
        r["loss"].backward()
        assert abs(float(loss) - float(r["loss"])) < 1e-3 * max(1.0, abs(float(r["loss"])))
        ref = dict(ora.named_parameters())
        for name, p in net.named_parameters():
            assert p.grad is not None, name
            assert _rel_err(p.grad.cpu(), ref[name].grad) < 5e-3, name
    finally:
        cfg.metayolo = True


def _wide_yolo_cfg(path):
    """A plain detector wide enough for every conv form: first-layer kernel (3->32), direct 3x3 (32->64), Winograd
    F(4x4) (64->128, 128->128 @16x16; 128->256 @8x8), 1x1, a biased head, RegionLoss v1."""
    def conv(f, k, bn=1, act="leaky"):
        return "[convolutional]\n%sfilters=%d\nsize=%d\nstride=1\npad=1\nactivation=%s\n\n" % (
            "batch_normalize=1\n" if bn else "", f, k, act)
    pool = "[maxpool]\nsize=2\nstride=2\n\n"
    txt = ("[net]\nbatch=2\nwidth=64\nheight=64\nchannels=3\n\n" + conv(32, 3) + pool + conv(64, 3) + pool + conv(128, 3)
           + conv(128, 3) + pool + conv(256, 3) + conv(128, 1) + conv(256, 3) + conv(40, 1, bn=0, act="linear")
           + "[region]\nanchors = 1.08,1.19,  3.42,4.41,  6.63,11.38,  9.42,5.11,  16.62,10.52\nbias_match=1\nclasses=3\n"
             "coords=4\nnum=5\nsoftmax=1\njitter=.2\nrescore=1\nobject_scale=5\nnoobject_scale=1\nclass_scale=1\n"
             "coord_scale=1\nabsolute=1\nthresh = .6\nrandom=1\n")
    open(path, "w").write(txt)
    return path


def test_wide_detector_three_sgd_steps_follow_the_oracle(dev, tmp_path):
    """Trajectory test over every conv form (incl. Winograd with kept V, the fused first-layer weight gradient and the
    gradient sink): three SGD steps on the HIP path == oracle + torch.optim.SGD on the CPU."""
    from fewshot_detection_amd import ops
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.darknet import Darknet
    from fewshot_detection_amd.dp import EpisodeTrainer
    from oracle.net import OracleYolo
    from oracle.region import region_loss_v1
    c = _wide_yolo_cfg(os.path.join(str(tmp_path), "wide.cfg"))
    assert ops.wino_tile(64, 128, 3, 16, 16) == 4 and ops.wino_tile(128, 256, 3, 8, 8) == 4
    torch.manual_seed(13)
    ora = OracleYolo(c).train()
    net = Darknet(c)
    net.load_state_dict(ora.state_dict())
    net = net.to(dev).train()
    x = torch.rand(4, 3, 64, 64)
    tgt = torch.zeros(4, 250, dtype=torch.float64)
    for b in range(4):
        tgt[b, :5] = torch.tensor([b % 3, 0.3 + 0.1 * b, 0.5, 0.3, 0.4])
    cfg.neg_ratio, cfg.metayolo = "full", False
    try:
        region = net.models[len(net.models) - 1]
        region.verbose = False
        lr, mom, wd = 2e-5, 0.9, 5e-3
        trainer = EpisodeTrainer(net, lr=lr, momentum=mom, weight_decay=wd)
        opt = torch.optim.SGD(ora.parameters(), lr=lr, momentum=mom, weight_decay=wd)
        for step in range(3):
            loss = region(net(x.to(dev)), tgt)
            trainer.backward_and_step(loss)
            opt.zero_grad()
            r = region_loss_v1(ora(x), tgt, ora.region.anchors, 5, 3)
            r["loss"].backward()
            opt.step()
            assert abs(float(loss.detach()) - float(r["loss"])) < 2e-3 * max(1.0, abs(float(r["loss"]))), step
        ref = dict(ora.named_parameters())
        for name, p in net.named_parameters():
            # BN biases start at 0 and have moved by ~1e-4 after three steps: scale the tolerance by at least 1e-3
            r_ = ref[name].detach()
            err = float((p.detach().cpu() - r_).abs().max())
            assert err < 2e-3 * max(1e-3, float(r_.abs().max())), (name, err)
    finally:
        cfg.metayolo = True


def test_tape_is_released_by_the_backward_pass(dev):
    """Round 6: a kept `loss` must not keep the previous step's activations alive.  After backward() the
    network's autograd node has dropped its tape (as autograd drops saved tensors without retain_graph): device memory in use
    is back at the parameters' level while `loss` and the output are still referenced, and a second backward through the same
    forward raises instead of silently replaying freed buffers."""
    from fewshot_detection_amd.cfg import cfg
    ora, net = _load_pair(dev, seed=9)
    B, N, S = 4, 3, 160
    x, metax = torch.rand(B, 3, S, S).to(dev), torch.rand(N, 3, S, S).to(dev)
    mask = (torch.rand(N, 1, S, S) > 0.5).float().to(dev)
    tgt = torch.zeros(B, N, 250, dtype=torch.float64)
    tgt[0, 1, :5] = torch.tensor([1, 0.5, 0.5, 0.4, 0.3])
    cfg.neg_ratio = "full"
    region = net.models[len(net.models) - 1]
    region.verbose = False
    for _ in range(2):                       # packed weights, workspaces: reach the steady state first
        region(net(x, metax, mask), tgt).backward()
    net.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated(dev)
    out = net(x, metax, mask)
    loss = region(out, tgt)
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated(dev) - base
    loss.backward()
    torch.cuda.synchronize()
    torch.empty(1, device=dev)                             # (lets the allocator retire the blocks that crossed a stream)
    after = torch.cuda.memory_allocated(dev) - base
    grads = sum(p.grad.numel() * 4 for p in net.parameters() if p.grad is not None)
    assert held > 20 * out.numel() * 4                     # the tape held the activations of every layer ...
    assert after - grads < 0.25 * held, (held, after, grads)  # ... and they are gone although `loss` / `out` are alive
    with pytest.raises(RuntimeError, match="second time|retain_graph|freed|already been freed"):
        region(out, tgt).backward()
