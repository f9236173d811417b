"""CPU-only checks: the C-ABI library loads and exports everything include/fsdet.h declares, host logic
(neg_filter, cfg plumbing, weight stream, box helpers) matches the oracle / reference, the product never
falls back to the CPU."""
import os
import random
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLD)
import ref_shim  # noqa: E402


def test_library_exports_every_declared_symbol():
    from fewshot_detection_amd import _lib
    header = open(os.path.join(ROOT, "include", "fsdet.h")).read()
    assert "#ifdef" not in header.replace("#ifdef __cplusplus", "")      # one ABI: no conditionally declared entry points
    declared = set(re.findall(r"\b(fsd_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 29
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    lib = _lib.lib()                                   # loads without a GPU (no compute calls here)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fsd_version().startswith(b"fsdet-hip")
    assert lib.fsd_conv_row_tiles(1, 25, 40, 256, 64, 3) == 16 and lib.fsd_packed_weight_elems(30, 1024, 1) == 128 * 1024


def test_no_cpu_fallback():
    from fewshot_detection_amd import ops
    from fewshot_detection_amd.darknet_meta import Darknet
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.require_device(torch.zeros(1))
    net = Darknet(os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.rand(1, 3, 64, 64), torch.rand(2, 3, 64, 64), torch.rand(2, 1, 64, 64))
    # the product never imports the oracle
    for mod in list(sys.modules):
        if mod.startswith("fewshot_detection_amd"):
            src = getattr(sys.modules[mod], "__file__", None)
            if src and src.endswith(".py"):
                assert "import oracle" not in open(src).read() and "from oracle" not in open(src).read(), mod


def test_neg_filter_matches_oracle_and_consumes_rng_identically():
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.region_loss import neg_filter_indices
    from oracle.region import select_rows
    rng = np.random.RandomState(0)
    rows = np.zeros((40, 250))
    rows[rng.choice(40, 9, replace=False), 1] = 0.5
    try:
        for neg in ("full", 0, 1, 2, 5):
            cfg.neg_ratio = neg
            random.seed(3)
            a = neg_filter_indices(rows)
            after_a = random.random()
            random.seed(3)
            b = select_rows(rows, neg, random.random)
            after_b = random.random()
            assert a == b and after_a == after_b, neg
        cfg.neg_ratio = 0
        assert neg_filter_indices(rows) == sorted(np.nonzero(rows[:, 1])[0].tolist())
    finally:
        cfg.neg_ratio = "full"


def test_weight_stream_roundtrip_and_partial_file(tmp_path):
    from fewshot_detection_amd.darknet_meta import Darknet
    cfgs = (os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    net = Darknet(*cfgs)
    net.load_weights(os.path.join(GOLD, "mini.weights"))          # written by the reference's save_weights
    assert int(net.seen) == 4242
    out = str(tmp_path / "a.weights")
    net.save_weights(out)
    blob = open(out, "rb").read()
    assert blob == open(os.path.join(GOLD, "mini.weights"), "rb").read()
    # a truncated file (backbone only, like darknet19_448.conv.23) initialises a prefix and stops quietly
    part = str(tmp_path / "part.weights")
    n_first = sum(t.numel() for t in (net.models[0][0].weight, net.models[0][1].weight, net.models[0][1].bias,
                                      net.models[0][1].running_mean, net.models[0][1].running_var))
    open(part, "wb").write(blob[:16 + 4 * n_first])              # ends on a layer boundary
    other = Darknet(*cfgs)
    before = other.learnet_models[0][0].weight.detach().clone()
    other.load_weights(part)
    assert torch.equal(other.models[0][0].weight, net.models[0][0].weight)
    assert torch.equal(other.learnet_models[0][0].weight, before)
    # cutoff: only the first detector block
    cut = str(tmp_path / "cut.weights")
    net.save_weights(cut, cutoff=1)
    n0 = sum(t.numel() for t in (net.models[0][0].weight, net.models[0][1].weight, net.models[0][1].bias,
                                 net.models[0][1].running_mean, net.models[0][1].running_var))
    assert os.path.getsize(cut) == 16 + 4 * n0


def test_state_dict_keys_match_oracle_and_reference_layout():
    from fewshot_detection_amd.darknet_meta import Darknet
    from oracle.net import OracleDarknet
    cfgs = (os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    a, b = Darknet(*cfgs).state_dict(), OracleDarknet(*cfgs).state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(a[k].shape == b[k].shape for k in a)


def test_box_helpers_match_reference_goldens():
    from fewshot_detection_amd import utils
    d = np.load(os.path.join(GOLD, "boxes.npz"))
    sc = np.array([utils.bbox_iou(list(map(float, a)), list(map(float, b)), x1y1x2y2=False) for a, b in zip(d["a"], d["b"])])
    assert np.array_equal(sc, d["scalar"])
    vec = utils.bbox_ious(torch.from_numpy(d["a"].astype(np.float32)).t().contiguous(),
                          torch.from_numpy(d["b"].astype(np.float32)).t().contiguous(), x1y1x2y2=False)
    assert np.array_equal(vec.numpy(), d["vector"])


def test_nms_matches_reference_golden():
    from fewshot_detection_amd import utils
    d = np.load(os.path.join(GOLD, "decode_v2.npz"))
    boxes, kept = d["boxes"], d["kept"]
    mine = []
    for r in sorted(set(boxes[:, 0].astype(int))):
        lst = [list(b[1:]) for b in boxes if int(b[0]) == r]
        mine += [[r] + b for b in utils.nms(lst, float(d["nms_thresh"]))]
    assert np.allclose(np.array(mine), kept)


def test_cfg_data_plumbing(tmp_path):
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.utils import read_data_cfg
    p = tmp_path / "m.data"
    p.write_text("metayolo=1\nmetain_type=2\ndata=voc\nneg = 1\nrand = 0\nnovel = bird,bus,cow,motorbike,sofa\n"
                 "novelid = 0\nmeta = data/voc_traindict_full.txt\nbackup = backup/metayolo\ngpus=1,2,3,4\n")
    opt = read_data_cfg(str(p))
    assert opt["num_workers"] == "10" and opt["gpus"] == "1,2,3,4"
    saved = dict(cfg)
    try:
        cfg.config_data(opt)
        assert cfg.neg_ratio == 1 and cfg.num_gpus == 4 and len(cfg.base_classes) == 15 and cfg.metayolo is True
        assert cfg.backup == "backup/metayolo_novel0_neg1"
        meta = {"height": "416", "width": "416", "feat_layer": "0"}
        cfg.config_meta(meta)
        assert meta["channels"] == 4 and cfg.mask_height == 416
        cfg.config_net({"height": "416", "width": "416", "batch": "64"})
        assert cfg.batch_size == 64
    finally:
        cfg.clear()
        cfg.update(saved)


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")
def test_cfg_data_plumbing_matches_live_reference(tmp_path):
    from fewshot_detection_amd.cfg import cfg
    ref = ref_shim.load("cfg").cfg
    opt = {"metayolo": "1", "metain_type": "2", "data": "voc", "neg": "1", "rand": "0", "novel": "bird,bus,cow",
           "novelid": "0", "meta": "x", "backup": "backup/metayolo", "gpus": "1,2"}
    saved = dict(cfg)
    try:
        cfg.config_data(dict(opt))
        ref.config_data(dict(opt))
        for k in ("neg_ratio", "num_gpus", "base_classes", "base_ids", "novel_ids", "backup", "metayolo", "tuning"):
            assert cfg[k] == ref[k], k
    finally:
        cfg.clear()
        cfg.update(saved)


def test_compat_aliases_expose_reference_names():
    sys.path.insert(0, os.path.join(ROOT, "fewshot_detection_amd", "compat"))
    try:
        for m in ("cfg", "darknet_meta", "region_loss", "dynamic_conv", "pooling", "utils", "darknet"):
            sys.modules.pop(m, None)
        import cfg as c
        import darknet_meta as dm
        import region_loss as rl
        assert hasattr(c, "parse_cfg") and hasattr(c, "cfg") and hasattr(dm, "Darknet")
        assert hasattr(rl, "RegionLossV2") and hasattr(rl, "RegionLoss")
    finally:
        sys.path.pop(0)
        for m in ("cfg", "darknet_meta", "region_loss", "dynamic_conv", "pooling", "utils", "darknet"):
            sys.modules.pop(m, None)


def test_region_targets_with_bad_class_ids_are_refused_on_the_host():
    """a class id outside the episode makes the reference's CrossEntropyLoss raise
    (region_loss.py:349-352); with verbose=False nobody reads the device-side bad-target counter, so the host refuses."""
    import pytest
    from fewshot_detection_amd.region_loss import _validate_targets
    rows = np.zeros((6, 250))
    rows[1, :5] = [2, 0.5, 0.5, 0.2, 0.2]
    rows[4, :10] = [0, 0.3, 0.3, 0.1, 0.1, 1, 0.6, 0.6, 0.2, 0.3]
    _validate_targets(rows, 3)
    _validate_targets(rows, None)
    bad = rows.copy()
    bad[1, 0] = 3                                          # == number of labels
    with pytest.raises(ValueError):
        _validate_targets(bad, 3)
    bad = rows.copy()
    bad[4, 5] = -1
    with pytest.raises(ValueError):
        _validate_targets(bad, 3)
    dead = rows.copy()
    dead[0, 5:10] = [99, 0.0, 0.5, 0.1, 0.1]               # behind the zero terminator: never read, never refused
    _validate_targets(dead, 3)


def test_host_nms_matches_reference_on_dense_rows_and_key_ties():
    """utils.nms on plain lists == the reference's (tests/golden/decode_valid.npz: conf 0.005 -> ~all cells survive;
    case v2 has many boxes sharing the float32 sort key 1 - det_conf, which only a stable sort on THAT key orders
    like the reference)."""
    from fewshot_detection_amd import utils
    d = np.load(os.path.join(GOLD, "decode_valid.npz"))
    for k in (2, 0):
        boxes, kept = d["v%d_boxes" % k], d["v%d_kept" % k]
        rows = sorted(set(boxes[:, 0].astype(int)))[:2]               # the O(n^2) python loop: two rows are enough
        mine = []
        for r in rows:
            lst = [[float(v) for v in b[1:]] for b in boxes if int(b[0]) == r]
            mine += [[r] + b for b in utils.nms(lst, float(d["nms_thresh"]))]
        want = kept[np.isin(kept[:, 0].astype(int), rows)]
        assert np.array_equal(np.array(mine), want), k


def test_reweight_ensemble_running_mean_is_the_reference_expression():
    from fewshot_detection_amd.ensemble import ReweightEnsemble, mean_by_group
    g = torch.Generator().manual_seed(0)
    clsids = [0, 2, 1, 0, 2, 2, 1, 0, 0]
    dw = torch.randn(len(clsids), 8, 1, 1, generator=g)
    ens = ReweightEnsemble(3)
    enews, cnt = [0.0] * 3, [0.0] * 3
    for lo, hi in ((0, 4), (4, 8), (8, 9)):
        ens.add([dw[lo:hi]], torch.tensor(clsids[lo:hi]))
        for ci, c in enumerate(clsids[lo:hi]):                             # valid_ensemble.py:96-98
            enews[c] = enews[c] * cnt[c] / (cnt[c] + 1) + dw[lo:hi][ci] / (cnt[c] + 1)
            cnt[c] += 1
    out = ens.dynamic_weights()
    assert len(out) == 1 and out[0].shape == (3, 8, 1, 1) and torch.equal(out[0], torch.stack(enews))
    assert ens.counts == [4.0, 2.0, 3.0]
    for c in range(3):
        rows = [i for i, k in enumerate(clsids) if k == c]
        assert torch.allclose(out[0][c], dw[rows].mean(0), atol=1e-6)
    m = mean_by_group([dw], [4, 2, 3])
    assert torch.allclose(m[0][1], dw[4:6].mean(0))
    import pytest
    with pytest.raises(ValueError):
        ReweightEnsemble(4).dynamic_weights()


def test_deferred_views_are_refused_by_operations_that_read_the_activation_itself():
    """ops.View.lazy: the buffer holds a convolution's RAW output; anything that is not one of the consumers that form the
    activation on load must refuse the view instead of silently reading the wrong tensor."""
    import pytest
    import torch
    from fewshot_detection_amd import ops
    t = torch.zeros(2 * 4 * 4, 32)
    lazy = ops.View(t, 2, 4, 4, 32, 0, lazy=(torch.ones(32), torch.zeros(32), 0.1))
    plain = ops.View(t, 2, 4, 4, 32)
    assert plain.lazy is None and lazy.lazy is not None
    for call in (lambda: ops.nhwc_to_nchw(lazy), lambda: ops.reorg(lazy, 2), lambda: ops.global_maxpool(lazy),
                 lambda: ops.bn_act_pool(lazy, None, None, 1.0, 1), lambda: ops.cast_view(lazy, torch.bfloat16),
                 lambda: ops.conv3x3_c4(lazy, torch.zeros(32, 4, 3, 3), 32)):
        with pytest.raises(ValueError, match="deferred"):
            call()


def test_which_layers_defer_their_activation_to_the_consumer(tmp_path):
    """engine.Network._defers_to_consumer on darknet_dynamic.cfg at 416x416 (fp32 training): the 14 no-pool conv + BatchNorm
    layers whose only reader is a 1x1 convolution or an F(4x4) Winograd layer; pooled layers, route sources (16, 24), the
    layer feeding the reorg branch's concat and the head's input stay materialised; nothing defers in the bf16 mode, in
    plain inference, or in the reweighting net (every layer there is pooled)."""
    import torch
    from fewshot_detection_amd import cfgs, ops
    from fewshot_detection_amd.darknet_meta import Darknet
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    net = Darknet(dyn_cfg, rw_cfg)
    det = net._det

    def deferring(network, size, training=True, record=True):
        network._record = record
        out, side = [], size
        for ind, blk in enumerate(network.layers):
            if blk["type"] == "maxpool" and int(blk["stride"]) == 2:
                side //= 2
            if blk["type"] == "reorg":
                side //= int(blk["stride"])
            if blk["type"] == "route":
                side = 13 if size == 416 else side
            if blk["type"] != "convolutional" or not int(blk.get("batch_normalize", 0)):
                continue
            nxt = network.layers[ind + 1] if ind + 1 < len(network.layers) else None
            if nxt is not None and nxt["type"] == "maxpool":
                continue
            cout = int(blk["filters"])
            side_here = 26 if ind == 26 else side            # layer 26 reads the 26x26 map of the route before it
            y = ops.View(torch.zeros(1, cout), 2, side_here, side_here, cout)
            if network._defers_to_consumer(ind, y, cout, training):
                out.append(ind)
        return out

    assert deferring(det, 416) == [4, 5, 8, 9, 12, 13, 14, 15, 18, 19, 20, 21, 22, 23]
    assert deferring(det, 416, training=False, record=False) == []
    det.compute_dtype = "bf16"
    assert deferring(det, 416) == []
    det.compute_dtype = "f32"
    assert deferring(net._meta, 224) == []


def test_batchnorm_batch_counters_follow_state_dict_and_load_state_dict():
    """the host-side BatchNorm batch counts are flushed by per-module hooks (submodule state_dict() calls and
    torch.save(model.models.state_dict()) see them) and dropped when a state is loaded (no pre-load batches on top)."""
    import torch
    from fewshot_detection_amd import ops
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4))
    ops.install_bn_counter_hooks(m)
    ops.install_bn_counter_hooks(m)                         # idempotent
    m[1]._fsd_pending_batches = 3
    assert int(m[1].state_dict()["num_batches_tracked"]) == 3 and m[1]._fsd_pending_batches == 0     # the submodule alone
    m[1]._fsd_pending_batches = 2
    saved = {k: v.clone() for k, v in m.state_dict().items()}                                         # through the parent
    assert int(saved["1.num_batches_tracked"]) == 5
    m[1]._fsd_pending_batches = 7                            # batches run before the load do not belong to the loaded state
    m.load_state_dict(saved)
    assert int(m[1].num_batches_tracked) == 5 and m[1]._fsd_pending_batches == 0
    m[1]._fsd_pending_batches = 1
    ops.flush_bn_counters(m)
    assert int(m[1].num_batches_tracked) == 6


def test_inference_can_pick_the_smaller_winograd_form_when_the_weights_are_the_traffic(monkeypatch):
    """Opt-in rule (FSD_INFER_F2=1; measured slower, see ops.wino_tile_inference): valid_ensemble.py's two images at 13x13 are
    32 tiles per position -> F(2x2) (16 weight planes instead of 36); the training batch and the wider feature maps keep
    wino_tile's answer; narrow layers too (their weights are small either way).  Off: wino_tile's answer everywhere."""
    from fewshot_detection_amd import ops
    assert ops.wino_tile(1024, 1024, 3, 13, 13) == 4
    monkeypatch.setattr(ops, "SMALL_BATCH_F2", False)
    assert ops.wino_tile_inference(1024, 1024, 3, 13, 13, 2) == 4
    monkeypatch.setattr(ops, "SMALL_BATCH_F2", True)
    assert ops.wino_tile_inference(1024, 1024, 3, 13, 13, 2) == 2
    assert ops.wino_tile_inference(512, 1024, 3, 13, 13, 2) == 2
    assert ops.wino_tile_inference(1024, 1024, 3, 13, 13, 64) == 4
    assert ops.wino_tile_inference(256, 512, 3, 26, 26, 2) == 4        # 98 tiles per position
    assert ops.wino_tile_inference(128, 256, 3, 13, 13, 2) == 4        # 1.2 MB of transformed weights
    assert ops.wino_tile_inference(1024, 1024, 1, 13, 13, 2) == 0


def test_winograd_workspace_is_the_three_launch_pipelines_in_both_arithmetics():
    """fsd_wino_workspace_bytes is host logic (no GPU): V + M of the transform -> position GEMMs -> transform pipeline, the same
    size under both arithmetics of the fp32 GEMMs.  (The fused-pipeline experiment of round 4 and its switch left the tree:
    tools/experiments_r04/wino_fused.patch.)"""
    from fewshot_detection_amd import _lib
    lib = _lib.lib()
    B, H, W, cin, cout = 64, 104, 104, 64, 128
    T = B * 26 * 26
    plain = 36 * T * (cin + cout) * 4
    split_before = lib.fsd_f32_gemm_mode(-1)
    try:
        for mode in (0, 1):
            lib.fsd_f32_gemm_mode(mode)
            assert lib.fsd_wino_workspace_bytes(B, H, W, cin, cout, 4) == plain
            assert lib.fsd_wino_workspace_bytes(B, 13, 13, 1024, 1024, 4) == 36 * B * 16 * 2048 * 4
    finally:
        lib.fsd_f32_gemm_mode(split_before)
    assert not hasattr(lib, "fsd_wino_fused_mode")


def test_a_constructed_model_pickles(tmp_path):
    """the BatchNorm counter hooks were local closures (and the reference's DynamicConv2d is a class defined inside
    its factory): torch.save(model) / handing the module to a spawned process failed.  Both are module-level now."""
    import io
    from fewshot_detection_amd.darknet_meta import Darknet
    net = Darknet(os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    buf = io.BytesIO()
    torch.save(net, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    assert type(back).__name__ == "Darknet"
    for (ka, va), (kb, vb) in zip(net.state_dict().items(), back.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


def test_every_environment_switch_of_the_default_build_is_documented():
    """at most 15 FSD_* environment switches in the default library + package, each with a row in
    INTEGRATION.md's table; every other knob is FSD_TUNE(...) -- compiled in only with -DFSD_EXPERIMENTS."""
    import glob
    pkg = os.path.join(ROOT, "fewshot_detection_amd")
    read = set()
    for path in glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.hpp")) + \
            glob.glob(os.path.join(pkg, "csrc", "*.inc")):
        read |= set(re.findall(r'[^_A-Za-z]getenv\("(FSD_[A-Z0-9_]+)"\)', open(path).read()))
    for path in glob.glob(os.path.join(pkg, "*.py")) + [os.path.join(ROOT, "bench.py")]:
        read |= set(re.findall(r'environ(?:\.get)?[\(\[]\s*"(FSD_[A-Z0-9_]+)"', open(path).read()))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = set(re.findall(r"^\| `(FSD_[A-Z0-9_]+)` \|", doc, flags=re.M))
    assert read == table, (read - table, table - read)
    assert len(read) <= 15, sorted(read)
