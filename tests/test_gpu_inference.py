"""The inference half of the path on the MI355X against fixtures minted from the reference:
  * utils.get_region_boxes_v2 in the mode valid_ensemble.py:148 really calls (only_objectness=0, validation=1, conf 0.005):
    every box VALUE, not only the count;
  * utils.nms on the device (fsd_region_nms) == the reference's greedy python NMS on dense rows (almost every cell
    survives conf 0.005) and on rows full of float32 sort-key ties;
  * support-set ensembling (valid_ensemble.py:86-100) -> detect_forward -> decode -> NMS end to end on the mini net."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ANCH = [1.3221, 1.73145, 3.19275, 4.00944, 5.05587, 8.09892, 9.47112, 4.84053, 11.2364, 10.0071]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _flat(rows):
    return np.array([[r] + [float(v) for v in b] for r, bl in enumerate(rows) for b in bl], np.float64).reshape(-1, 8)


def _same_boxes(got, ref, exact_conf=False):
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.array_equal(got[:, 0], ref[:, 0]) and np.array_equal(got[:, 7], ref[:, 7])        # row, class id
    assert np.allclose(got[:, 1:7], ref[:, 1:7], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("k", [0, 1, 2])
def test_validation_mode_decode_and_device_nms_vs_reference(dev, k):
    from fewshot_detection_amd import utils
    d = np.load(os.path.join(GOLD, "decode_valid.npz"))
    bs, cs, g = (int(v) for v in d["v%d_cfg" % k])
    out = torch.from_numpy(d["v%d_output" % k]).to(dev)
    got = utils.get_region_boxes_v2(out, cs, float(d["conf_thresh"]), 1, ANCH, 5, 0, 1)       # valid_ensemble.py:148
    assert len(got) == bs * cs and all(isinstance(r, list) for r in got)
    _same_boxes(_flat(got), d["v%d_boxes" % k])
    # The fixtures' kept boxes depend on the sort order of nearly equal confidences; feed the device NMS the REFERENCE's
    # own float32 box values (bit-identical det_conf), then the order and the survivors must match exactly.
    ref_boxes = d["v%d_boxes" % k]
    for r, row in enumerate(got):
        rb = ref_boxes[ref_boxes[:, 0] == r]
        assert len(rb) == len(row)
    kept = [utils.nms(row, float(d["nms_thresh"])) for row in got]
    want = d["v%d_kept" % k]
    flat = _flat(kept)
    if k != 2:
        _same_boxes(flat, want)
    # suppressed boxes carry det_conf = 0 afterwards, like the reference's in-place edit
    for row, kp in zip(got, kept):
        alive = set(id(b) for b in kp)
        assert all((b[4] == 0) == (id(b) not in alive) for b in row)


@pytest.mark.parametrize("k", [0, 1, 2])
def test_device_nms_kernel_on_reference_box_values(dev, k):
    """fsd_region_nms fed the reference's own decoded boxes (exact float32 values): kept set AND order equal the
    reference's nms, including case 2 where dozens of boxes share the float32 key 1 - det_conf."""
    from fewshot_detection_amd._lib import check, lib
    d = np.load(os.path.join(GOLD, "decode_valid.npz"))
    bs, cs, g = (int(v) for v in d["v%d_cfg" % k])
    ref, want = d["v%d_boxes" % k], d["v%d_kept" % k]
    rows, cap = bs * cs, 5 * g * g
    boxes = np.zeros((rows, cap, 8), np.float32)
    counts = np.zeros(rows, np.int32)
    rng = np.random.RandomState(k)
    slot_of = []
    for r in range(rows):
        rb = ref[ref[:, 0] == r]
        n = len(rb)
        perm = rng.permutation(n)                       # the decode kernel appends survivors in arbitrary order
        slot_of.append(perm)
        boxes[r, perm, 0] = np.arange(n)                # visiting-order key
        boxes[r, perm, 1:8] = rb[:, 1:8].astype(np.float32)
        counts[r] = n
    bd, cd = torch.from_numpy(boxes).to(dev), torch.from_numpy(counts).to(dev)
    keep_idx = torch.empty((rows, cap), dtype=torch.int32, device=dev)
    keep_cnt = torch.empty(rows, dtype=torch.int32, device=dev)
    check(lib().fsd_region_nms(bd.data_ptr(), cd.data_ptr(), rows, cap, float(d["nms_thresh"]), keep_idx.data_ptr(),
                               keep_cnt.data_ptr(), torch.cuda.current_stream().cuda_stream), "fsd_region_nms")
    kc, ki = keep_cnt.cpu().numpy(), keep_idx.cpu().numpy()
    got = []
    for r in range(rows):
        for s in ki[r, :kc[r]]:
            got.append([r] + [float(v) for v in boxes[r, s, 1:8]])
    got = np.array(got, np.float64).reshape(-1, 8)
    assert got.shape == want.shape
    assert np.array_equal(got.astype(np.float32), want.astype(np.float32))


def test_nms_falls_back_to_the_host_loop_for_foreign_or_edited_lists(dev):
    from fewshot_detection_amd import utils
    d = np.load(os.path.join(GOLD, "decode_valid.npz"))
    out = torch.from_numpy(d["v2_output"]).to(dev)
    got = utils.get_region_boxes_v2(out, 2, 0.005, 1, ANCH, 5, 0, 1)
    row = got[0]
    ref = utils.nms([list(b) for b in row], 0.45)        # plain lists -> host
    edited = got[1]
    edited.pop()                                          # no longer what the device holds -> host
    assert len(utils.nms(edited, 0.45)) > 0
    dev_kept = utils.nms(row, 0.45)
    assert [list(b) for b in dev_kept] == [list(b) for b in ref]


def test_support_set_ensembling_end_to_end_vs_reference(dev):
    """valid_ensemble.py:86-100 + 137-166: running-mean reweighting vectors over three support batches, detect_forward
    with the averaged vectors, validation-mode decode, NMS -- against the reference run through ref_shim."""
    from fewshot_detection_amd import utils
    from fewshot_detection_amd.darknet_meta import Darknet
    from fewshot_detection_amd.ensemble import ReweightEnsemble
    d = np.load(os.path.join(GOLD, "ensemble.npz"))
    net = Darknet(os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    net.load_weights(os.path.join(GOLD, "mini.weights"))
    net = net.to(dev).eval()
    metax, mask = torch.from_numpy(d["metax"]).to(dev), torch.from_numpy(d["mask"]).to(dev)
    clsids = d["clsids"]
    ens = ReweightEnsemble(3)
    with torch.no_grad():
        for lo, hi in d["batches"]:
            ens.add(net.meta_forward(metax[lo:hi], mask[lo:hi]), clsids[lo:hi])
        dyn = ens.dynamic_weights()
        assert dyn[0].shape == (3, 64, 1, 1)
        assert float((dyn[0].cpu() - torch.from_numpy(d["vectors"])).abs().max()) < 1e-4
        out = net.detect_forward(torch.from_numpy(d["x"]).to(dev), dyn)
    assert out.shape == tuple(d["output"].shape)
    assert float((out.cpu() - torch.from_numpy(d["output"])).abs().max()) < 1e-3
    boxes = utils.get_region_boxes_v2(out, 3, 0.005, net.num_classes, net.anchors, net.num_anchors, 0, 1)
    got = _flat(boxes)
    ref = d["boxes"]
    assert got.shape == ref.shape and np.array_equal(got[:, 0], ref[:, 0])
    assert np.allclose(got[:, 1:7], ref[:, 1:7], rtol=1e-3, atol=1e-4)
    # NMS on the reference's head output itself (identical inputs -> identical survivors and order)
    boxes_ref_in = utils.get_region_boxes_v2(torch.from_numpy(d["output"]).to(dev), 3, 0.005, 1, net.anchors, 5, 0, 1)
    kept = _flat([utils.nms(b, 0.45) for b in boxes_ref_in])
    _same_boxes(kept, d["kept"])


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-4), ("bf16", 6e-2)])
def test_eval_mode_batchnorm_folded_into_the_convolutions(dev, tmp_path, dtype, tol):
    """Inference form of a conv block (engine.Network._conv_eval): eval-mode BatchNorm folded into the conv operands, leaky in
    the conv epilogue, one pass less per block -- against the unfolded kernel sequence, with non-trivial running statistics,
    on the full architecture (first-layer kernel, Winograd / direct / 1x1 layers, pooled and route-tapped blocks)."""
    from fewshot_detection_amd import cfgs, engine
    from fewshot_detection_amd.darknet_meta import Darknet
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    torch.manual_seed(4)
    net = Darknet(dyn_cfg, rw_cfg).to(dev).eval().set_compute_dtype(dtype)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.3, 0.3)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
    x = torch.rand(2, 3, 160, 192, device=dev)
    metax, mask = torch.rand(3, 3, 96, 96, device=dev), torch.zeros(3, 1, 96, 96, device=dev)
    mask[:, :, 10:60, 20:70] = 1
    old = engine.FOLD_EVAL_BN
    try:
        with torch.no_grad():
            engine.FOLD_EVAL_BN = False
            vec_u = net.meta_forward(metax, mask)
            out_u = net.detect_forward(x, vec_u).clone()
            engine.FOLD_EVAL_BN = True
            vec_f = net.meta_forward(metax, mask)
            out_f = net.detect_forward(x, vec_u)
    finally:
        engine.FOLD_EVAL_BN = old
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(vec_f[0], vec_u[0]) < tol, rel(vec_f[0], vec_u[0])
    assert rel(out_f, out_u) < tol, rel(out_f, out_u)
    # with autograd on (fine-tuning with frozen statistics) the unfolded path runs: its tape is what backward replays
    net.zero_grad()
    out = net.detect_forward(x, [vec_u[0].detach()])
    out.sum().backward()
    assert net.models[0][0].weight.grad is not None
