"""Arithmetic of the fp32 GEMM kernels (include/fsdet.h, fsd_f32_gemm_mode): the default "split" mode -- six bf16 MFMA terms
of three-way split operands, fp32 accumulate -- has to be as accurate as the native fp32 matrix instruction it replaces.
The referee is a float64 convolution; both modes are run on the same inputs, forward / data gradient / weight gradient,
Winograd and direct forms."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def restore_mode():
    from fewshot_detection_amd import ops
    before = ops.f32_gemm_mode()
    yield
    ops.f32_gemm_mode(before)


def _rel(a, ref):
    return float((a.double() - ref).norm() / ref.norm())


SHAPES = [  # B, H, W, cin, cout, k
    (4, 52, 52, 128, 256, 3),      # Winograd F(4x4)
    (8, 13, 13, 1024, 1024, 3),    # Winograd F(4x4), long reduction
    (2, 40, 40, 64, 128, 3),       # F(4x4), 64 / 128-channel reductions
    (5, 100, 96, 64, 128, 3),      # ... 3000 tiles (BatchNorm partial rows of 16 tiles)
    (2, 104, 104, 32, 64, 3),      # direct 3x3
    (2, 112, 112, 32, 64, 3),      # direct 3x3, halo-staged under split (H % 8 == 0, W % 16 == 0): forward 32->64, data gradient 64->32
    (3, 16, 32, 64, 64, 3),        # halo-staged, two 32-channel slices each way
    (3, 8, 8, 32, 64, 3),          # halo-staged weight gradient (wgrad_halo.hip: 32 -> 64, H % 8 == W % 8 == 0): one block per image
    (1, 24, 40, 32, 64, 3),        # ... 3 x 5 blocks, one workgroup walks all of them
    (6, 64, 48, 32, 64, 3),        # ... 288 blocks on 36 workgroups of 8
    (4, 26, 26, 512, 256, 1),      # 1x1 (20 tiles of 256x128: too few for the 8-wave kernel, 64x64 tiles)
    (24, 26, 26, 512, 512, 1),     # 1x1 on the 8-wave kernel: 63 whole 256-row tiles x 4 + a 96-row tail on 64x64 tiles
    (3, 7, 7, 256, 512, 3),        # small map of the reweighting net
    (2, 13, 13, 512, 1024, 3),     # two images (valid_ensemble.py's batch): 32 rows per position (FSD_KSPLIT=a: K cut into 2 / 4 slices)
]


@pytest.mark.parametrize("shape", SHAPES)
def test_split_is_as_accurate_as_the_native_fp32_mfma(dev, shape):
    from fewshot_detection_amd import ops
    B, H, W, cin, cout, k = shape
    torch.manual_seed(sum(shape))
    xn = torch.randn(B, cin, H, W, device=dev)
    xn = torch.where(xn > 0, xn, 0.1 * xn)                       # activations as the network sees them (leaky)
    w = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
    dyn = torch.randn(B, cout, H, W, device=dev)
    xd = xn.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    ref_y = F.conv2d(xd, wd, padding=k // 2)
    ref_dx, ref_dw = torch.autograd.grad(ref_y, (xd, wd), dyn.double())
    ref_y = ref_y.detach()

    x = ops.nchw_to_nhwc(xn)
    dy = ops.nchw_to_nhwc(dyn)
    tile = ops.wino_tile(cin, cout, k, H, W)
    errs = {}
    outs = {}
    for mode in ("native", "split"):
        ops.f32_gemm_mode(mode)
        assert ops.f32_gemm_mode() == mode
        if tile:
            y, _ = ops.conv3x3_wino(x, ops.pack_weight_wino(w, 0, tile), cout, tile=tile)
            dx, _ = ops.conv3x3_wino(dy, ops.pack_weight_wino(w, 1, tile), cin, tile=tile)
        else:
            y, _ = ops.conv2d(x, ops.pack_weight(w), cout, k)
            dx, _ = ops.conv2d(dy, ops.pack_weight(w, 1), cin, k)
        dw = ops.conv2d_wgrad(dy, cout, x, cin, k)
        outs[mode] = (ops.nhwc_to_nchw(y), ops.nhwc_to_nchw(dx), dw)
        errs[mode] = (_rel(outs[mode][0], ref_y), _rel(outs[mode][1], ref_dx), _rel(dw, ref_dw))
    for what, en, es in zip(("forward", "data gradient", "weight gradient"), errs["native"], errs["split"]):
        # fp32 round-off level for both (the F(4x4) transforms contribute ~1e-5, see winograd.hip) ...
        assert en < 5e-5 and es < 5e-5, (what, en, es)
        # ... and the split arithmetic is not the less accurate of the two (1.25 = run-to-run slack between two roundings)
        assert es <= 1.25 * en + 2e-7, (what, en, es)
    # the two modes are different arithmetic: close, not identical
    assert not torch.equal(outs["native"][0], outs["split"][0])


@pytest.mark.parametrize("scale_x,scale_w", [(1e-18, 1e18), (1e15, 1e-3), (1e-30, 1.0), (3e4, 3e4)])
def test_split_keeps_fp32_range(dev, scale_x, scale_w):
    """bfloat16 has fp32's exponent range: operands far from 1 (tiny gradients, large activations, products near the ends of
    the fp32 range) go through the three-way split with the same relative accuracy, no overflow, no flush to zero."""
    from fewshot_detection_amd import ops
    B, H, W, cin, cout = 2, 13, 13, 256, 128
    torch.manual_seed(5)
    xn = torch.randn(B, cin, H, W, device=dev) * scale_x
    w = torch.randn(cout, cin, 1, 1, device=dev) * scale_w / cin ** 0.5
    ref = F.conv2d(xn.double(), w.double())
    x = ops.nchw_to_nhwc(xn)
    errs = {}
    for mode in ("native", "split"):
        ops.f32_gemm_mode(mode)
        y, _ = ops.conv2d(x, ops.pack_weight(w), cout, 1)
        y = ops.nhwc_to_nchw(y)
        assert torch.isfinite(y).all()
        errs[mode] = _rel(y, ref)
    assert errs["split"] < 2e-6 and errs["split"] <= 1.25 * errs["native"] + 1e-7, errs


def test_mode_switch_reports_the_previous_mode(dev):
    from fewshot_detection_amd import ops
    first = ops.f32_gemm_mode()
    assert first in ("native", "split")
    assert ops.f32_gemm_mode("native") == first
    assert ops.f32_gemm_mode("split") == "native"
    assert ops.f32_gemm_mode() == "split"


def test_training_step_in_both_modes_agrees_to_round_off(dev, tmp_path):
    """One episode step of the standard pair of networks per mode from the same initial state.  The loss agrees to fp32
    round-off; the gradients agree as well as two fp32 evaluations of this network can: a different rounding flips a few
    leaky-ReLU / max-pool winners near ties, and a fraction f of flipped winners costs ~sqrt(f) in relative L2 (the
    teacher-forced per-block test of tests/test_gpu_timed_config.py removes exactly that and holds 1e-4; the replay of the
    reference's train_meta.py in tests/test_gpu_drivers.py measures 9.5e-3 per update against the reference itself)."""
    from fewshot_detection_amd import cfgs, ops
    from fewshot_detection_amd.darknet_meta import Darknet
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    results = {}
    for mode in ("native", "split"):
        ops.f32_gemm_mode(mode)
        torch.manual_seed(11)
        net = Darknet(dyn_cfg, rw_cfg).to(dev).train()
        n_cls = 5
        x = torch.rand(2, 3, 160, 160, device=dev)
        metax = torch.rand(n_cls, 3, 160, 160, device=dev)
        mask = (torch.rand(n_cls, 1, 160, 160, device=dev) > 0.5).float()
        out = net(x, metax, mask)
        loss = out.float().pow(2).mean()
        loss.backward()
        grads = torch.cat([p.grad.flatten() for p in net.parameters() if p.grad is not None])
        results[mode] = (float(loss.detach()), grads.double().clone())
    (ln, gn), (ls, gs) = results["native"], results["split"]
    assert abs(ln - ls) <= 1e-5 * abs(ln)
    assert float((gn - gs).norm() / gn.norm()) < 3e-2


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 16, 32, 64), (3, 24, 48, 64, 32), (1, 208, 208, 32, 64)])
def test_halo_staged_narrow_conv_epilogue(dev, B, H, W, cin, cout):
    """csrc/conv_halo.hip (split arithmetic, 32 / 64 channels, H % 8 == 0, W % 16 == 0): output, BatchNorm partial sums (one
    row per 128 pixels), bias and the inference-form leaky epilogue against float64."""
    from fewshot_detection_amd import ops
    ops.f32_gemm_mode("split")
    g = torch.Generator().manual_seed(B * H + cin)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    xv = ops.nchw_to_nhwc(x.to(dev))
    wp = ops.pack_weight(w.to(dev))
    y, part = ops.conv2d(xv, wp, cout, 3, bn_partial=True)
    assert part.shape[0] == B * H * W // 128
    out = ops.nhwc_to_nchw(y).cpu().double()
    assert float((out - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    p = part.double().sum(0).cpu()
    flat = ref.permute(1, 0, 2, 3).reshape(cout, -1)
    assert torch.allclose(p[:, 0], flat.sum(1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(p[:, 1], (flat ** 2).sum(1), rtol=2e-4, atol=1e-2)
    # bias + leaky epilogue (inference form), into a channel slice of a wider buffer
    wide = ops.new_view(B, H, W, cout + 32, dev)
    wide.t.fill_(7.0)
    dst = ops.View(wide.t, B, H, W, cout, 32)
    ops.conv2d(xv, wp, cout, 3, bias=bias.to(dev), out=dst, slope=0.1)
    refb = F.leaky_relu(ref + bias.double().view(1, -1, 1, 1), 0.1)
    got = ops.nhwc_to_nchw(dst).cpu().double()
    assert float((got - refb).abs().max()) < 2e-5 * float(refb.abs().max())
    assert float((wide.t[:, :32] - 7.0).abs().max()) == 0.0          # the neighbouring channels are untouched


def test_positions_of_at_most_32_rows_take_the_32x128_tile(dev):
    """Two images at 13x13 (valid_ensemble.py's batch) and the reweighting net's 3x3 maps: 32 / 20 tile rows per Winograd
    position.  Under the split arithmetic such a launch runs 32x128 tiles (half of a 64x64 tile would be padding that is
    split and multiplied like data); 33+ rows, narrow outputs and the native arithmetic keep 64x64."""
    import ctypes as C
    from fewshot_detection_amd import ops
    L = ops.lib()

    def plan(B, H, W, cin, cout):
        a = (C.c_int * 4)()
        assert L.fsd_wino_fwd_plan(B, H, W, cin, cout, 4, a) == 0
        return tuple(a)[:2] + (a[3],)

    ops.f32_gemm_mode("split")
    assert plan(2, 13, 13, 1024, 1024) == (32, 128, 1)
    assert plan(20, 3, 3, 512, 1024) == (32, 128, 1)
    assert plan(3, 13, 13, 1024, 1024) == (64, 64, 1)          # 48 rows
    assert plan(2, 13, 13, 1024, 64) == (64, 64, 1)
    ops.f32_gemm_mode("native")
    assert plan(2, 13, 13, 1024, 1024) == (64, 64, 1)
