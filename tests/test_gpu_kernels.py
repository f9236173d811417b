"""Kernel-level parity (MI355X): each HIP op of include/fsdet.h against PyTorch-CPU fp32 / the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _conv_case(dev, B, H, W, cin, cout, k, bias, seed):
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), 1, (k - 1) // 2).float()
    xv = ops.nchw_to_nhwc(x.to(dev))
    wp = ops.pack_weight(w.to(dev))
    yv, part = ops.conv2d(xv, wp, cout, k, bias=None if b is None else b.to(dev), bn_partial=not bias)
    y = ops.nhwc_to_nchw(yv).cpu()
    return ref, y, part, yv


@pytest.mark.parametrize("B,H,W,cin,cout,k,bias", [
    (2, 13, 13, 64, 128, 3, False),     # 128x128 tile config, ragged M (338 rows)
    (1, 26, 26, 32, 64, 3, False),      # 256x64 config
    (2, 20, 24, 3, 32, 3, False),       # first layer: Cin=3 padded to 4, K=36 (K tail), 256x32 config
    (2, 13, 13, 256, 30, 1, True),      # 1x1 + bias, Cout tail
    (1, 7, 9, 1280, 200, 3, False),     # Cin/4 not a power of two, Cout tail across 2 tiles
    (3, 6, 6, 4, 8, 3, False),          # tiny everything
    (2, 13, 13, 452, 1024, 1, False),   # generic (Cin % 32 != 0) path with K > 320 -> 64x64 tiles (head data gradient shape)
    (1, 9, 9, 40, 64, 3, False),        # generic path, 3x3, K = 360
    (2, 13, 13, 1024, 1280, 3, False),  # L29 data-gradient shape
    (3, 52, 52, 128, 64, 3, False),     # reweighting-net conv3 data-gradient shape
])
def test_conv_forward_matches_fp64_reference(dev, B, H, W, cin, cout, k, bias):
    ref, y, part, _ = _conv_case(dev, B, H, W, cin, cout, k, bias, seed=B * 1000 + cin)
    # exact-fp32 MFMA: error is fp32 round-off of a K-term dot product
    assert torch.allclose(y, ref, rtol=1e-4, atol=2e-5), float((y - ref).abs().max())
    if part is not None:
        p = part.double().sum(0).cpu()
        flat = ref.double().permute(1, 0, 2, 3).reshape(cout, -1)
        assert torch.allclose(p[:, 0], flat.sum(1), rtol=1e-4, atol=1e-3)
        assert torch.allclose(p[:, 1], (flat ** 2).sum(1), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("B,H,W,cin,cout,bias", [(2, 20, 24, 3, 32, False), (1, 13, 13, 4, 64, True), (3, 416, 64, 3, 32, False),
                                                  (2, 1, 2, 3, 32, False), (5, 7, 5, 4, 32, False)])
def test_first_layer_conv_matches_fp64_reference(dev, B, H, W, cin, cout, bias):
    """fsd_conv3x3_c4_fwd (direct-operand MFMA, OIHW weights) incl. its BatchNorm partial sums."""
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(B * 7 + H)
    x = torch.zeros(B, 4, H, W)
    x[:, :cin] = torch.randn(B, cin, H, W, generator=g)
    if cin == 3:
        x[:, 3] = float("nan")                             # the padding channel must not leak into the result (not even as 0 * NaN)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    ref = F.conv2d(x[:, :cin].double(), w.double(), None if b is None else b.double(), 1, 1).float()
    xv = ops.nchw_to_nhwc(x.to(dev))
    yv, part = ops.conv3x3_c4(xv, w.to(dev), cout, bias=None if b is None else b.to(dev), bn_partial=not bias)
    y = ops.nhwc_to_nchw(yv).cpu()
    assert torch.allclose(y, ref, rtol=1e-4, atol=2e-5), float((y - ref).abs().max())
    if part is not None:
        p = part.double().sum(0).cpu()
        flat = ref.double().permute(1, 0, 2, 3).reshape(cout, -1)
        assert torch.allclose(p[:, 0], flat.sum(1), rtol=1e-4, atol=1e-3)
        assert torch.allclose(p[:, 1], (flat ** 2).sum(1), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("B,H,W,cin,cout,bias", [(2, 20, 32, 3, 32, False), (1, 13, 96, 4, 64, True), (2, 5, 64, 4, 32, False),
                                                  (1, 3, 32, 1, 32, False), (3, 64, 160, 3, 64, True), (2, 33, 224, 2, 32, False)])
@pytest.mark.parametrize("out_dtype", ["f32", "bf16"])
def test_first_layer_split_kernel_matches_fp64_reference(dev, B, H, W, cin, cout, bias, out_dtype):
    """conv_first_split_kernel (round 5): widths that are multiples of 32 take the bf16-MFMA kernel on three-way split operands
    under the split arithmetic -- swapped operands, 16-byte stores from registers, per-lane BatchNorm sums.  Same contract as the
    fp32-MFMA kernel (which the native arithmetic keeps): output incl. bias, image borders, the padding channel, partial sums
    of the fp32 accumulators; the bf16 store of the bf16 mode is the rounded fp32 result."""
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(B * 7 + H + W)
    x = torch.zeros(B, 4, H, W)
    x[:, :cin] = torch.rand(B, cin, H, W, generator=g) * 2 - 0.5
    if cin < 4:
        x[:, cin:] = float("nan")                          # padding channels must not leak into the result (not even as 0 * NaN)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    ref = F.conv2d(x[:, :cin].double(), w.double(), None if b is None else b.double(), 1, 1)
    xv = ops.nchw_to_nhwc(x.to(dev))
    res = {}
    before = ops.f32_gemm_mode()
    try:
        for mode in ("split", "native"):
            ops.f32_gemm_mode(mode)
            yv, part = ops.conv3x3_c4(xv, w.to(dev), cout, bias=None if b is None else b.to(dev), bn_partial=not bias,
                                      out_dtype=torch.bfloat16 if out_dtype == "bf16" else torch.float32)
            res[mode] = (ops.nhwc_to_nchw(yv).double().cpu(), None if part is None else part.double().sum(0).cpu())
    finally:
        ops.f32_gemm_mode(before)
    for mode, (y, p) in res.items():
        err = float((y - ref).norm() / ref.norm())
        assert err < (3e-3 if out_dtype == "bf16" else 2e-6), (mode, err)
        if out_dtype == "f32":
            assert torch.allclose(y, ref, rtol=1e-5, atol=2e-6), (mode, float((y - ref).abs().max()))
        else:                                              # the stored value is the bf16 rounding of an fp32-accurate result
            assert float((y - ref).abs().max()) <= float(ref.abs().max()) * 2.0 ** -8
        if p is not None:
            flat = ref.permute(1, 0, 2, 3).reshape(cout, -1)
            assert torch.allclose(p[:, 0], flat.sum(1), rtol=1e-5, atol=1e-3), mode
            assert torch.allclose(p[:, 1], (flat ** 2).sum(1), rtol=1e-5, atol=1e-3), mode
    if out_dtype == "f32":                                 # two fp32-accurate kernels: they agree to round-off
        assert float((res["split"][0] - res["native"][0]).abs().max()) < 4e-6 * float(ref.abs().max())


def test_conv_nchw_store_and_asymmetric_weights(dev):
    """Transposed-accumulator epilogue: identity-like input with an ASYMMETRIC weight catches row/col swaps."""
    from fewshot_detection_amd import ops
    B, H, W, cin, cout = 2, 5, 7, 8, 45
    x = torch.randn(B, cin, H, W)
    w = torch.arange(cout * cin, dtype=torch.float32).view(cout, cin, 1, 1) / 100.0
    b = torch.arange(cout, dtype=torch.float32)
    ref = F.conv2d(x, w, b)
    xv = ops.nchw_to_nhwc(x.to(dev))
    y, _ = ops.conv2d(xv, ops.pack_weight(w.to(dev)), cout, 1, bias=b.to(dev), nchw_out=True)
    assert torch.allclose(y.cpu(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 9, 9, 8, 12), (2, 13, 13, 1280, 1024), (2, 13, 13, 1024, 450)])
def test_data_gradient_packing(dev, B, H, W, cin, cout):
    """mode-1 packing turns the forward kernel into dL/dx (incl. the L29 and fused-head shapes)."""
    from fewshot_detection_amd import ops
    x = torch.randn(B, cin, H, W, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cout, cin, 3, 3, dtype=torch.float64)
    gy = torch.randn(B, cout, H, W, dtype=torch.float64)
    F.conv2d(x, w, None, 1, 1).backward(gy)
    gv = ops.nchw_to_nhwc(gy.float().to(dev))
    dx, _ = ops.conv2d(gv, ops.pack_weight(w.float().to(dev), mode=1), cin, 3)
    ref = x.grad.float()
    assert torch.allclose(ops.nhwc_to_nchw(dx).cpu(), ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))


@pytest.mark.parametrize("pool", [0, 1, 2])
def test_bn_leaky_pool_matches_torch(dev, pool):
    from fewshot_detection_amd import ops
    torch.manual_seed(pool)
    B, H, W, cin, cout = 3, 13, 13, 8, 16          # odd size: floor pooling 13 -> 6
    x = torch.randn(B, cin, H, W)
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1, bias=False)
    bn = torch.nn.BatchNorm2d(cout)
    bn.weight.data.uniform_(-1, 1)                 # negative gammas: affine must precede the max
    bn.bias.data.uniform_(-0.5, 0.5)
    ref_bn = torch.nn.BatchNorm2d(cout)
    ref_bn.load_state_dict(bn.state_dict())
    z = F.leaky_relu(ref_bn(conv(x)), 0.1)
    if pool == 1:
        z = F.max_pool2d(z, 2, 2)
    elif pool == 2:
        z = F.max_pool2d(F.pad(z, (0, 1, 0, 1), mode="replicate"), 2, stride=1)
    bn = bn.to(dev)
    xv = ops.nchw_to_nhwc(x.to(dev))
    yv, part = ops.conv2d(xv, ops.pack_weight(conv.weight.data.to(dev)), cout, 3, bn_partial=True)
    scale, shift, mean, invstd = ops.bn_finalize(part, xv.pixels, bn, True)
    out = ops.nhwc_to_nchw(ops.bn_act_pool(yv, scale, shift, 0.1, pool)).cpu()
    assert torch.allclose(out, z.detach(), rtol=1e-4, atol=1e-5), float((out - z).abs().max())
    assert torch.allclose(bn.running_mean.cpu(), ref_bn.running_mean, atol=1e-6)
    assert torch.allclose(bn.running_var.cpu(), ref_bn.running_var, rtol=1e-5, atol=1e-6)
    # eval mode: running statistics
    ref_bn.eval()
    z2 = F.leaky_relu(ref_bn(conv(x)), 0.1)
    s2, h2, _, _ = ops.bn_finalize(None, xv.pixels, bn, False)
    out2 = ops.nhwc_to_nchw(ops.bn_act_pool(yv, s2, h2, 0.1, 0)).cpu()
    assert torch.allclose(out2, z2.detach(), rtol=1e-4, atol=1e-5)


def test_reorg_globalmax_dynamic_conv(dev):
    from fewshot_detection_amd import ops
    from oracle.net import reorg, reweight
    x = torch.randn(2, 8, 6, 6)
    xv = ops.nchw_to_nhwc(x.to(dev))
    assert torch.equal(ops.nhwc_to_nchw(ops.reorg(xv, 2)).cpu(), reorg(x, 2))
    gm, arg = ops.global_maxpool(xv, want_argmax=True)
    assert torch.equal(gm.cpu(), x.amax(dim=(2, 3)))
    assert torch.equal(arg.cpu().long(), x.flatten(2).argmax(2))
    d = np.load(os.path.join(GOLD, "dconv.npz"))          # reference dynamic_conv2d output
    out = ops.dynamic_conv(torch.from_numpy(d["x"]).to(dev), torch.from_numpy(d["w"]).to(dev))
    assert torch.equal(out.cpu(), torch.from_numpy(d["out"]))
    assert torch.equal(out.cpu(), reweight(torch.from_numpy(d["x"]), torch.from_numpy(d["w"])))


def test_global_avg_pool_module_and_cfg_block_forward_backward(dev, tmp_path):
    """pooling.GlobalAvgPool2d (pooling.py:29-45, F.adaptive_avg_pool2d(x, 1)) and the [globalavg] cfg block the reference's
    reweighting_net.cfg keeps as the commented-out alternative of [globalmax]: module form, and a reweighting net that ends in
    [globalavg] through meta_forward + backward against the oracle (fp32) -- VERDICT r5 #8 (was NotImplementedError)."""
    from fewshot_detection_amd import ops
    from fewshot_detection_amd.darknet_meta import Darknet
    from fewshot_detection_amd.pooling import GlobalAvgPool2d
    from oracle.net import OracleDarknet
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 10, 7, 7, generator=g)
    ref = F.adaptive_avg_pool2d(x.double(), 1)
    got = GlobalAvgPool2d()(x.to(dev)).cpu()
    assert got.shape == (3, 10, 1, 1) and float((got.double() - ref).abs().max()) < 1e-7
    xb = x.to(torch.bfloat16)
    vb = ops.View(xb.permute(0, 2, 3, 1).reshape(3 * 49, 10).contiguous().to(dev), 3, 7, 7, 10)
    assert float((ops.global_avgpool(vb).cpu().double() - xb.double().mean(dim=(2, 3))).abs().max()) < 1e-6
    # a reweighting net ending in [globalavg]
    rw = open(os.path.join(GOLD, "mini_reweight.cfg")).read()
    assert "[globalmax]" in rw
    path = os.path.join(str(tmp_path), "mini_reweight_avg.cfg")
    open(path, "w").write(rw.replace("[globalmax]", "[globalavg]"))
    dyn_cfg = os.path.join(GOLD, "mini_dynamic.cfg")
    torch.manual_seed(6)
    ora = OracleDarknet(dyn_cfg, path).train()
    net = Darknet(dyn_cfg, path)
    net.load_state_dict(ora.state_dict())
    net = net.to(dev).train()
    metax, mask = torch.rand(3, 3, 160, 160, generator=g), (torch.rand(3, 1, 160, 160, generator=g) > 0.5).float()      # 5x5 / 2x2 final map
    out = net.meta_forward(metax.to(dev), mask.to(dev))[0]
    want = ora.meta_forward(metax, mask)[0]
    assert out.shape == want.shape and float((out.detach().cpu() - want.detach()).abs().max()) < 1e-4
    go = torch.randn(want.shape, generator=g)
    out.backward(go.to(dev))
    want.backward(go)
    named = dict(ora.named_parameters())
    for name, p in net.named_parameters():
        if name.startswith("learnet_models"):
            gr = named[name].grad
            assert float((p.grad.cpu() - gr).abs().max()) <= 2e-4 * max(1e-3, float(gr.abs().max())), name


def test_fused_reweight_head_equals_materialised_path(dev):
    from fewshot_detection_amd import ops
    from oracle.net import reweight
    torch.manual_seed(9)
    B, C, G, N, O = 3, 64, 5, 4, 30
    x = torch.randn(B, C, G, G)
    dyn = torch.randn(N, C, 1, 1)
    hw_, hb = torch.randn(O, C, 1, 1) / 8, torch.randn(O)
    ref = F.conv2d(reweight(x, dyn), hw_, hb)                      # (B*N, 30, G, G)
    w_eff, b_eff, _ = ops.fold_reweight_head(hw_.to(dev), hb.to(dev), dyn.to(dev))
    y, _ = ops.conv2d(ops.nchw_to_nhwc(x.to(dev)), w_eff, N * O, 1, bias=b_eff, nchw_out=True)
    assert torch.allclose(y.view(B * N, O, G, G).cpu(), ref, rtol=1e-4, atol=1e-4)


MASKS = ["coord_mask", "conf_mask", "cls_mask", "tx", "ty", "tw", "th", "tconf", "tcls"]
ANCH = [1.3221, 1.73145, 3.19275, 4.00944, 5.05587, 8.09892, 9.47112, 4.84053, 11.2364, 10.0071]
ANCH_V1 = [1.08, 1.19, 3.42, 4.41, 6.63, 11.38, 9.42, 5.11, 16.62, 10.52]


def _check_loss(mod, d, dev):
    out = torch.from_numpy(d["output"]).to(dev).requires_grad_(True)
    mod.debug_targets = True
    loss = mod(out, torch.from_numpy(d["target"]))
    loss.backward()
    s = mod.stats()
    assert s["nGT"] == int(d["nGT"]) and s["nCorrect"] == int(d["nCorrect"])
    got = mod.last_targets.cpu().numpy()
    for i, k in enumerate(MASKS):
        if k in ("coord_mask", "conf_mask", "cls_mask", "tcls", "tx", "ty"):
            assert np.array_equal(got[i], d[k]), k            # assignment: bit-exact
        else:
            assert np.allclose(got[i], d[k], rtol=1e-5, atol=1e-6), k
    assert abs(float(loss) - float(d["loss"])) <= 1e-3, (float(loss), float(d["loss"]))
    assert np.allclose(out.grad.cpu().numpy(), d["grad"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", ["full_seen0", "full_seen20000", "neg0_seen20000", "neg1_seen20000"])
def test_region_loss_v2_vs_reference_golden(dev, case):
    import random
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.region_loss import RegionLossV2
    d = np.load(os.path.join(GOLD, "region_v2_%s.npz" % case))
    neg = str(d["neg_ratio"])
    cfg.neg_ratio = neg if neg == "full" else int(neg)
    random.seed(int(d["py_seed"]))
    try:
        mod = RegionLossV2(1, ANCH, 5)
        mod.seen = int(d["seen"])
        _check_loss(mod, d, dev)
    finally:
        cfg.neg_ratio = "full"


@pytest.mark.parametrize("case", ["seen0", "seen20000", "metayolo"])
def test_region_loss_v1_vs_reference_golden(dev, case):
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.region_loss import RegionLoss
    d = np.load(os.path.join(GOLD, "region_v1_%s.npz" % case))
    cfg.neg_ratio = "full"
    cfg.metayolo = bool(d["metayolo"])
    try:
        mod = RegionLoss(3, ANCH_V1, 5)
        mod.seen = int(d["seen"])
        _check_loss(mod, d, dev)
    finally:
        cfg.metayolo = True


@pytest.mark.parametrize("only_obj", [1, 0])
def test_region_decode_vs_reference_golden(dev, only_obj):
    """utils.get_region_boxes_v2 on the device == the reference's python triple loop (tests/golden/decode_v2.npz)."""
    from fewshot_detection_amd import utils
    d = np.load(os.path.join(GOLD, "decode_v2.npz"))
    out = torch.from_numpy(d["output"]).to(dev)
    cs = int(d["n_models"])
    got = utils.get_region_boxes_v2(out, cs, float(d["conf_thresh"]), 1, ANCH, 5, only_objectness=only_obj)
    if only_obj:
        flat = np.array([[r] + b for r, bl in enumerate(got) for b in bl])
        ref = d["boxes"]
        assert flat.shape == ref.shape
        assert np.array_equal(flat[:, 0], ref[:, 0]) and np.array_equal(flat[:, 7], ref[:, 7])
        assert np.allclose(flat[:, 1:7], ref[:, 1:7], rtol=1e-5, atol=1e-6)
        kept = [[r] + b for r, bl in enumerate(got) for b in utils.nms(bl, float(d["nms_thresh"]))]
        assert np.allclose(np.array(kept), d["kept"], rtol=1e-5, atol=1e-6)
    else:
        # det_conf * softmax-over-classes confidence: compare against the oracle formula
        o = torch.from_numpy(d["output"])
        rows, _, H, W = o.shape
        o5 = o.view(rows // cs, cs, 5, 6, H, W)
        prob = torch.softmax(o5[:, :, :, 5], dim=1)
        det = torch.sigmoid(o5[:, :, :, 4])
        n_ref = int(((det * prob) > float(d["conf_thresh"])).sum())
        assert sum(len(bl) for bl in got) == n_ref


@pytest.mark.parametrize("tile", [2, 4])
@pytest.mark.parametrize("B,H,W,cin,cout,bias", [
    (2, 13, 13, 64, 128, False),     # odd size: last tile row/column partly outside the image
    (1, 26, 26, 128, 64, True),
    (2, 6, 6, 1280, 1024, False),    # L29 channel widths
    (3, 8, 10, 96, 36, False),       # Cout not a multiple of 64
    (1, 2, 2, 64, 32, True),         # a single tile per image
    (2, 5, 3, 32, 32, False),        # smaller than one 4x4 tile in one direction
])
def test_winograd_conv_matches_direct_reference(dev, B, H, W, cin, cout, bias, tile):
    """F(2x2,3x3) to ~1e-6, F(4x4,3x3) to ~1.5e-5 of the output magnitude (fp32 round-off of the transforms)."""
    from fewshot_detection_amd import ops
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), 1, 1).float()
    xv = ops.nchw_to_nhwc(x.to(dev))
    yv, part = ops.conv3x3_wino(xv, ops.pack_weight_wino(w.to(dev), 0, tile), cout,
                                bias=None if b is None else b.to(dev), bn_partial=not bias, tile=tile)
    y = ops.nhwc_to_nchw(yv).cpu()
    tol = 1e-5 if tile == 2 else 6e-5                      # relative to the output magnitude
    assert float((y - ref).abs().max()) < tol * float(ref.abs().max()), float((y - ref).abs().max())
    if part is not None:
        p = part.double().sum(0).cpu()
        flat = ref.double().permute(1, 0, 2, 3).reshape(cout, -1)
        assert torch.allclose(p[:, 0], flat.sum(1), rtol=1e-4, atol=1e-3)
        assert torch.allclose(p[:, 1], (flat ** 2).sum(1), rtol=1e-4, atol=1e-3)
    if cout % 32:                                # the data gradient swaps the roles of cin / cout
        return
    # data gradient through the same pipeline (mode-1 weights)
    xg = x.double().requires_grad_(True)
    gy = torch.randn(B, cout, H, W, generator=g)
    F.conv2d(xg, w.double(), None, 1, 1).backward(gy.double())
    dx, _ = ops.conv3x3_wino(ops.nchw_to_nhwc(gy.to(dev)), ops.pack_weight_wino(w.to(dev), 1, tile), cin, tile=tile)
    gref = xg.grad.float()
    err = float((ops.nhwc_to_nchw(dx).cpu() - gref).abs().max())
    assert err < tol * float(gref.abs().max()), err


def _nasty_targets(rng, bs, cs, grid):
    """Targets that sit on the awkward spots of build_targets: boxes on cell borders and image edges, boxes sharing a
    cell (later box wins), tiny and huge boxes, rows with up to 50 boxes, empty rows and empty images."""
    tgt = np.zeros((bs, cs, 250), np.float64)
    for b in range(bs):
        if rng.rand() < 0.15:
            continue                                             # an image without any box
        for n in range(cs):
            if rng.rand() < 0.5:
                continue
            k = int(rng.choice([1, 1, 2, 3, 7, 50]))
            for t in range(k):
                mode = rng.randint(0, 5)
                if mode == 0:                                    # exactly on a cell border
                    cx, cy = rng.randint(1, grid) / grid, rng.randint(1, grid) / grid
                elif mode == 1:                                  # image edge
                    cx, cy = rng.choice([0.001, 0.998]), rng.uniform(0.05, 0.95)
                elif mode == 2 and t > 0:                        # same cell as the previous box
                    cx, cy = tgt[b, n, 5 * (t - 1) + 1] + 1e-4, tgt[b, n, 5 * (t - 1) + 2] + 1e-4
                else:
                    cx, cy = rng.uniform(0.02, 0.97, 2)
                w, h = rng.choice([0.002, 0.03, 0.2, 0.6, 0.98]), rng.choice([0.002, 0.05, 0.3, 0.7, 0.98])
                tgt[b, n, 5 * t:5 * t + 5] = [n, cx, cy, w, h]
    return torch.from_numpy(tgt)


@pytest.mark.parametrize("seed", range(12))
def test_region_loss_v2_random_sweep_vs_oracle(dev, seed):
    """GPU RegionLossV2 against the oracle (itself pinned to the reference goldens) on randomised shapes and nasty
    targets: row selection and every assignment mask bit-exact, loss / gradient / statistics within fp32 tolerance."""
    import random
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.region_loss import RegionLossV2
    from oracle.region import region_loss_v2
    rng = np.random.RandomState(100 + seed)
    bs, cs = int(rng.randint(1, 5)), int(rng.choice([1, 2, 5, 15, 20]))
    grid = int(rng.choice([7, 13, 19]))
    seen = int(rng.choice([0, 12799, 12800, 20000]))
    neg = ["full", 0, 1, 5][seed % 4]
    tgt = _nasty_targets(rng, bs, cs, grid)
    out_cpu = torch.from_numpy(rng.randn(bs * cs, 30, grid, grid).astype(np.float32) * 1.5)
    cfg.neg_ratio = neg
    try:
        random.seed(seed)
        ref_in = out_cpu.clone().requires_grad_(True)
        r = region_loss_v2(ref_in, tgt, ANCH, seen=seen, neg_ratio=neg)
        r["loss"].backward()
        random.seed(seed)
        mod = RegionLossV2(1, ANCH, 5)
        mod.verbose = False
        mod.seen = seen
        mod.debug_targets = True
        out = out_cpu.to(dev).requires_grad_(True)
        loss = mod(out, tgt)
        loss.backward()
        assert list(mod.last_keep) == list(r["keep"])                       # same rows survive neg_filter
        s = mod.stats()
        assert (s["nGT"], s["nCorrect"], s["nProposals"]) == (r["nGT"], r["nCorrect"], r["nProposals"])
        got = mod.last_targets.cpu().numpy()
        for i, k in enumerate(MASKS):
            want = r["targets"][k]
            if k in ("coord_mask", "conf_mask", "cls_mask", "tcls", "tx", "ty"):
                assert np.array_equal(got[i], want), k
            else:
                assert np.allclose(got[i], want, rtol=1e-5, atol=1e-6), k
        ref_loss = float(r["loss"].detach())
        assert abs(float(loss.detach()) - ref_loss) <= 1e-3 * max(1.0, abs(ref_loss)) * 0.1
        g_ref = ref_in.grad.numpy()
        assert np.allclose(out.grad.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(g_ref).max())))
    finally:
        cfg.neg_ratio = "full"


@pytest.mark.parametrize("seed", range(6))
def test_region_loss_v1_random_sweep_vs_oracle(dev, seed):
    """Same sweep for the classic per-cell softmax loss (C1 path), incl. cfg.metayolo zeroing the class targets."""
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.region_loss import RegionLoss
    from oracle.region import region_loss_v1
    rng = np.random.RandomState(300 + seed)
    bs, nc = int(rng.randint(1, 5)), int(rng.choice([1, 3, 20]))
    grid = int(rng.choice([7, 13, 19]))
    seen = int(rng.choice([0, 12800, 20000]))
    meta = bool(seed % 2)
    tgt3 = _nasty_targets(rng, bs, nc, grid)                     # (bs, nc, 250): reuse, then flatten per image
    tgt = torch.zeros(bs, 250, dtype=torch.float64)
    for b in range(bs):
        rows = tgt3[b].reshape(-1, 5)
        rows = rows[rows[:, 3] > 0][:50]
        tgt[b, :rows.numel()] = rows.reshape(-1)
    out_cpu = torch.from_numpy(rng.randn(bs, 5 * (5 + nc), grid, grid).astype(np.float32) * 1.5)
    cfg.neg_ratio, cfg.metayolo = "full", meta
    try:
        ref_in = out_cpu.clone().requires_grad_(True)
        r = region_loss_v1(ref_in, tgt, ANCH_V1, 5, nc, seen=seen, metayolo=meta)
        r["loss"].backward()
        mod = RegionLoss(nc, ANCH_V1, 5)
        mod.verbose = False
        mod.seen = seen
        mod.debug_targets = True
        out = out_cpu.to(dev).requires_grad_(True)
        loss = mod(out, tgt)
        loss.backward()
        s = mod.stats()
        assert (s["nGT"], s["nCorrect"], s["nProposals"]) == (r["nGT"], r["nCorrect"], r["nProposals"])
        got = mod.last_targets.cpu().numpy()
        for i, k in enumerate(MASKS):
            want = r["targets"][k]
            if k in ("coord_mask", "conf_mask", "cls_mask", "tcls", "tx", "ty"):
                assert np.array_equal(got[i], want), k
            else:
                assert np.allclose(got[i], want, rtol=1e-5, atol=1e-6), k
        ref_loss = float(r["loss"].detach())
        assert abs(float(loss.detach()) - ref_loss) <= 1e-4 * max(1.0, abs(ref_loss))
        g_ref = ref_in.grad.numpy()
        assert np.allclose(out.grad.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(g_ref).max())))
    finally:
        cfg.metayolo = True


@pytest.mark.parametrize("k", [0, 1, 2])
def test_meta_decode_more_shapes_vs_reference_golden(dev, k):
    """get_region_boxes_v2 + nms at 1 / 5 / 20 class rows and 7 / 19 / 13 grids (tests/golden/decode_extra.npz)."""
    from fewshot_detection_amd import utils
    d = np.load(os.path.join(GOLD, "decode_extra.npz"))
    bs, cs, g, th = d["m%d_cfg" % k]
    got = utils.get_region_boxes_v2(torch.from_numpy(d["m%d_output" % k]).to(dev), int(cs), float(th), 1, ANCH, 5)
    flat = np.array([[r] + b for r, bl in enumerate(got) for b in bl], np.float64).reshape(-1, 8)
    ref = d["m%d_boxes" % k]
    assert flat.shape == ref.shape
    assert np.array_equal(flat[:, 0], ref[:, 0]) and np.array_equal(flat[:, 7], ref[:, 7])      # row and class id
    assert np.allclose(flat[:, 1:7], ref[:, 1:7], rtol=1e-5, atol=1e-6)
    kept = np.array([[r] + b for r, bl in enumerate(got) for b in utils.nms(bl, 0.45)], np.float64).reshape(-1, 8)
    assert np.allclose(kept, d["m%d_kept" % k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("k", [0, 1])
def test_plain_decode_vs_reference_golden(dev, k):
    """get_region_boxes (per-cell softmax over 3 / 20 classes, reference utils.py:112-193)."""
    from fewshot_detection_amd import utils
    d = np.load(os.path.join(GOLD, "decode_extra.npz"))
    bs, nc, g, th = d["y%d_cfg" % k]
    got = utils.get_region_boxes(torch.from_numpy(d["y%d_output" % k]).to(dev), float(th), int(nc), ANCH_V1, 5)
    flat = np.array([[r] + b for r, bl in enumerate(got) for b in bl], np.float64).reshape(-1, 8)
    ref = d["y%d_boxes" % k]
    assert flat.shape == ref.shape
    assert np.array_equal(flat[:, 0], ref[:, 0]) and np.array_equal(flat[:, 7], ref[:, 7])
    assert np.allclose(flat[:, 1:7], ref[:, 1:7], rtol=1e-5, atol=1e-6)


def test_clock_probe_reports_a_plausible_shader_clock(dev):
    """fsd_clock_probe: a dependent fp32-MFMA chain of known length (64 cycles per instruction) timed with HIP events."""
    from fewshot_detection_amd import ops
    mhz = sorted(ops.clock_probe_mhz(dev) for _ in range(7))
    # (the boxes are shared: a single reading can catch the chip throttled or time-sliced -- 138 MHz was seen once among
    # 2208 / 2224; bench.py waits such a phase out.  The median and its neighbours are the property.)
    assert 500.0 < mhz[3] < 3000.0, mhz
    assert mhz[4] / mhz[2] < 1.5, mhz
