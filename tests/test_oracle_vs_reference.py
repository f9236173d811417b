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
