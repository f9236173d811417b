"""Pin the CPU oracle against golden vectors minted from the reference itself
(tests/golden/make_golden.py) and, when /root/reference is present, against the live reference.
CPU only."""
import os
import random
import sys

import numpy as np
import pytest
import torch

from oracle import boxes, region
from oracle.cfgparse import parse_cfg
from oracle.net import OracleDarknet, OracleYolo, reweight

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import ref_shim  # noqa: E402

ANCH = [1.3221, 1.73145, 3.19275, 4.00944, 5.05587, 8.09892, 9.47112, 4.84053, 11.2364, 10.0071]
ANCH_V1 = [1.08, 1.19, 3.42, 4.41, 6.63, 11.38, 9.42, 5.11, 16.62, 10.52]
needs_ref = pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")


def test_iou_matches_reference_goldens():
    d = np.load(os.path.join(GOLD, "boxes.npz"))
    sc = np.array([boxes.iou_scalar(list(map(float, a)), list(map(float, b))) for a, b in zip(d["a"], d["b"])])
    assert np.array_equal(sc, d["scalar"])                       # python doubles: bit-exact
    vec = boxes.iou_vector(d["a"].astype(np.float32).T, d["b"].astype(np.float32).T)
    assert np.array_equal(vec, d["vector"])                      # float32 op order preserved
    assert sc[0] == 0.0 and abs(sc[1] - 1.0) < 1e-12 and abs(sc[2] - 1.0 / 3.0) < 1e-12 and sc[3] == 0.0


def test_reweight_matches_reference_dynamic_conv():
    d = np.load(os.path.join(GOLD, "dconv.npz"))
    out = reweight(torch.from_numpy(d["x"]), torch.from_numpy(d["w"]))
    assert torch.equal(out, torch.from_numpy(d["out"]))


def test_meta_detector_forward_matches_reference():
    d = np.load(os.path.join(GOLD, "mini_forward.npz"))
    net = OracleDarknet(os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    net.load_weights(os.path.join(GOLD, "mini.weights"))
    assert net.seen == 4242
    x, metax, mask = (torch.from_numpy(d[k]) for k in ("x", "metax", "mask"))
    net.eval()
    with torch.no_grad():
        dyn = net.meta_forward(metax, mask)
        out = net.detect_forward(x, dyn)
    assert torch.allclose(dyn[0], torch.from_numpy(d["dyn_eval"]), atol=1e-6)
    assert torch.allclose(out, torch.from_numpy(d["out_eval"]), atol=1e-5)
    net.train()
    dyn = net.meta_forward(metax, mask)
    out = net.detect_forward(x, dyn)
    assert torch.allclose(dyn[0], torch.from_numpy(d["dyn_train"]), atol=1e-6)
    assert torch.allclose(out, torch.from_numpy(d["out_train"]), atol=1e-5)
    out.backward(torch.from_numpy(d["grad_out"]))
    sd = net.state_dict()
    assert np.allclose(sd["models.0.bn1.running_mean"].numpy(), d["bn1_mean_after"], atol=1e-6)
    assert np.allclose(sd["models.0.bn1.running_var"].numpy(), d["bn1_var_after"], atol=1e-6)
    assert np.allclose(sd["learnet_models.10.bn6.running_var"].numpy(), d["lbn6_var_after"], atol=1e-6)
    named = dict(net.named_parameters())
    for k in d.files:
        if k.startswith("grad:"):
            g = named[k[5:]].grad.numpy()
            assert np.allclose(g, d[k], rtol=1e-4, atol=1e-5), k


def test_weight_stream_roundtrip_is_byte_exact(tmp_path):
    net = OracleDarknet(os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    net.load_weights(os.path.join(GOLD, "mini.weights"))
    p = tmp_path / "rt.weights"
    net.save_weights(str(p))
    assert open(str(p), "rb").read() == open(os.path.join(GOLD, "mini.weights"), "rb").read()


def test_plain_yolo_forward_matches_reference():
    d = np.load(os.path.join(GOLD, "mini_yolo_forward.npz"))
    net = OracleYolo(os.path.join(GOLD, "mini_tiny_yolo.cfg"))
    net.load_weights(os.path.join(GOLD, "mini_yolo.weights"))
    x = torch.from_numpy(d["x"])
    net.eval()
    with torch.no_grad():
        assert torch.allclose(net(x), torch.from_numpy(d["out_eval"]), atol=1e-5)
    net.train()
    assert torch.allclose(net(x), torch.from_numpy(d["out_train"]), atol=1e-5)


MASKS = ["coord_mask", "conf_mask", "cls_mask", "tx", "ty", "tw", "th", "tconf", "tcls"]


@pytest.mark.parametrize("case", ["full_seen0", "full_seen20000", "neg0_seen20000", "neg1_seen20000"])
def test_region_loss_v2_matches_reference(case):
    d = np.load(os.path.join(GOLD, "region_v2_%s.npz" % case))
    neg = str(d["neg_ratio"])
    neg = neg if neg == "full" else int(neg)
    random.seed(int(d["py_seed"]))
    out = torch.from_numpy(d["output"]).clone().requires_grad_(True)
    r = region.region_loss_v2(out, torch.from_numpy(d["target"]), ANCH, seen=int(d["seen"]), neg_ratio=neg)
    r["loss"].backward()
    assert r["nGT"] == int(d["nGT"]) and r["nCorrect"] == int(d["nCorrect"])
    for k in MASKS:                                   # anchor assignment + targets: bit-exact
        assert np.array_equal(r["targets"][k], d[k]), k
    assert abs(r["loss"].item() - float(d["loss"])) <= 1e-4 * abs(float(d["loss"]))
    assert np.allclose(out.grad.numpy(), d["grad"], rtol=1e-5, atol=1e-6)
    if case.startswith("neg"):
        assert len(r["keep"]) < d["output"].shape[0]


@pytest.mark.parametrize("case", ["seen0", "seen20000", "metayolo"])
def test_region_loss_v1_matches_reference(case):
    d = np.load(os.path.join(GOLD, "region_v1_%s.npz" % case))
    out = torch.from_numpy(d["output"]).clone().requires_grad_(True)
    r = region.region_loss_v1(out, torch.from_numpy(d["target"]), ANCH_V1, 5, 3, seen=int(d["seen"]),
                              metayolo=bool(d["metayolo"]))
    r["loss"].backward()
    assert r["nGT"] == int(d["nGT"]) and r["nCorrect"] == int(d["nCorrect"])
    for k in MASKS:
        assert np.array_equal(r["targets"][k], d[k]), k
    assert abs(r["loss"].item() - float(d["loss"])) <= 1e-4 * abs(float(d["loss"]))
    assert np.allclose(out.grad.numpy(), d["grad"], rtol=1e-5, atol=1e-6)


def test_hand_derived_single_box():
    """One GT centred in cell (6,6) with exactly anchor-2's shape: closed-form targets."""
    tgt = torch.zeros(1, 1, 250, dtype=torch.float64)
    tgt[0, 0, :5] = torch.tensor([0, 6.5 / 13, 6.5 / 13, ANCH[4] / 13, ANCH[5] / 13])
    out = torch.zeros(1, 30, 13, 13)
    r = region.region_loss_v2(out, tgt, ANCH, seen=20000)
    (row, t, best_n, gj, gi), = r["targets"]["matches"]
    assert (best_n, gj, gi) == (2, 6, 6)
    tg = r["targets"]
    assert tg["tx"][0, 2, 6, 6] == np.float32(0.5) and tg["ty"][0, 2, 6, 6] == np.float32(0.5)
    assert abs(tg["tw"][0, 2, 6, 6]) < 1e-6 and abs(tg["th"][0, 2, 6, 6]) < 1e-6
    assert abs(tg["tconf"][0, 2, 6, 6] - 1.0) < 1e-6          # zero logits decode to the anchor box itself
    assert tg["conf_mask"][0, 2, 6, 6] == 5.0 and tg["coord_mask"].sum() == 1 and r["nCorrect"] == 1
    # sigmoid(0)=.5 -> x,y terms vanish; w,h terms vanish; conf term = .5*5*(.5-1)^2 + noobj .5*(.5)^2 elsewhere
    silenced = int((tg["conf_mask"] == 0).sum())
    expect_conf = 0.5 * 5 * 0.25 + 0.5 * 0.25 * (5 * 169 - 1 - silenced)
    assert abs(r["parts"][4].item() - expect_conf) < 1e-3
    assert abs(r["parts"][5].item() - 0.0) < 1e-6              # one class -> softmax over N=1 is certain


@needs_ref
def test_parse_cfg_matches_live_reference():
    ref = ref_shim.load("cfg")
    for name in ("darknet_dynamic.cfg", "reweighting_net.cfg", "tiny-yolo-voc.cfg", "yolo-voc.cfg"):
        p = os.path.join(ref_shim.REF, "cfg", name)
        assert parse_cfg(p) == ref.parse_cfg(p)


@needs_ref
def test_full_size_construction_matches_live_reference():
    ref = ref_shim.load("darknet_meta")
    d = os.path.join(ref_shim.REF, "cfg")
    r = ref.Darknet(os.path.join(d, "darknet_dynamic.cfg"), os.path.join(d, "reweighting_net.cfg"))
    o = OracleDarknet(os.path.join(d, "darknet_dynamic.cfg"), os.path.join(d, "reweighting_net.cfg"))
    rs, os_ = r.state_dict(), o.state_dict()
    assert list(rs.keys()) == list(os_.keys())
    assert all(rs[k].shape == os_[k].shape for k in rs)
    assert sum(p.numel() for p in o.parameters()) == 66287742


def _nasty_targets(rng, bs, cs, grid):
    """Boxes on cell borders / image edges, boxes sharing a cell, tiny and huge boxes, long rows, empty rows/images."""
    tgt = np.zeros((bs, cs, 250), np.float64)
    for b in range(bs):
        if rng.rand() < 0.15:
            continue
        for n in range(cs):
            if rng.rand() < 0.5:
                continue
            for t in range(int(rng.choice([1, 1, 2, 3, 7, 50]))):
                mode = rng.randint(0, 5)
                if mode == 0:
                    cx, cy = rng.randint(1, grid) / grid, rng.randint(1, grid) / grid
                elif mode == 1:
                    cx, cy = rng.choice([0.001, 0.998]), rng.uniform(0.05, 0.95)
                elif mode == 2 and t > 0:
                    cx, cy = tgt[b, n, 5 * (t - 1) + 1] + 1e-4, tgt[b, n, 5 * (t - 1) + 2] + 1e-4
                else:
                    cx, cy = rng.uniform(0.02, 0.97, 2)
                w, h = rng.choice([0.002, 0.03, 0.2, 0.6, 0.98]), rng.choice([0.002, 0.05, 0.3, 0.7, 0.98])
                tgt[b, n, 5 * t:5 * t + 5] = [n, cx, cy, w, h]
    return tgt


@pytest.mark.parametrize("seed", range(16))
def test_region_loss_v2_live_sweep(seed):
    """The oracle against the LIVE reference RegionLossV2 (py2->py3 shim) on randomised nasty targets: every mask of
    build_targets bit-exact, loss and gradient equal.  Skipped where /root/reference is absent (GPU box)."""
    import random
    import ref_shim
    if not ref_shim.available():
        pytest.skip("reference tree not present")
    import make_golden
    from oracle.region import region_loss_v2
    cfgmod, rl = ref_shim.load("cfg"), ref_shim.load("region_loss")
    rng = np.random.RandomState(500 + seed)
    bs, cs, grid = int(rng.randint(1, 4)), int(rng.choice([1, 3, 5])), int(rng.choice([5, 7, 13]))
    seen = int(rng.choice([0, 12800, 20000]))
    neg = ["full", 0, 1][seed % 3]
    tgt = _nasty_targets(rng, bs, cs, grid)
    out = torch.from_numpy(rng.randn(bs * cs, 30, grid, grid).astype(np.float32) * 1.5)
    mod = rl.RegionLossV2()
    mod.anchors, mod.num_anchors, mod.num_classes, mod.anchor_step = make_golden.ANCH, 5, 1, 2
    mod.seen = seen
    cfgmod.cfg.neg_ratio = neg
    try:
        if not (tgt.reshape(bs * cs, -1).sum(1) != 0).any() and neg != "full":
            pytest.skip("no positive row: the reference divides by zero here")
        n_pos = int((tgt.reshape(bs * cs, -1).sum(1) != 0).sum())
        if neg == 0 and n_pos == 1:
            # reference quirk (SURVEY 8a'): np.argwhere(flags).squeeze() is 0-d for a single surviving row, pred[inds]
            # loses its batch dimension and the reference raises a shape error; the oracle / kernels keep the row
            pytest.skip("single surviving row: the reference itself raises here")
        random.seed(seed)
        ref = make_golden._run_loss(rl, mod, out, torch.from_numpy(tgt))
        random.seed(seed)
        o = out.clone().requires_grad_(True)
        r = region_loss_v2(o, torch.from_numpy(tgt), make_golden.ANCH, seen=seen, neg_ratio=neg)
        r["loss"].backward()
    finally:
        cfgmod.cfg.neg_ratio = "full"
    assert (r["nGT"], r["nCorrect"]) == (int(ref["nGT"]), int(ref["nCorrect"]))
    for k in ("coord_mask", "conf_mask", "cls_mask", "tx", "ty", "tw", "th", "tconf", "tcls"):
        assert np.array_equal(r["targets"][k], ref[k]), k
    assert abs(float(r["loss"].detach()) - float(ref["loss"])) <= 1e-4 * max(1.0, abs(float(ref["loss"])))
    assert np.allclose(o.grad.numpy(), ref["grad"], rtol=1e-5, atol=1e-6)


@needs_ref
def test_global_pools_match_live_reference_pooling():
    """pooling.py:8-45: the oracle's [globalmax] / [globalavg] layers are the reference's GlobalMaxPool2d / GlobalAvgPool2d."""
    from oracle.net import _GlobalAvg, _GlobalMax
    ref = ref_shim.load("pooling")
    x = torch.randn(3, 16, 6, 6, generator=torch.Generator().manual_seed(3))
    assert torch.equal(_GlobalMax()(x), ref.GlobalMaxPool2d()(x))
    assert torch.equal(_GlobalAvg()(x), ref.GlobalAvgPool2d()(x))
