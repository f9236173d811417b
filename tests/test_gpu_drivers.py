"""The reference's two drivers against the MI355X modules (executed proof of "drops in").

tests/golden/drivers.npz was minted by exec'ing the reference's OWN train_meta.py (whole script) and
valid_ensemble.valid() on a synthetic on-disk dataset (tests/golden/drivers_golden.py; PyTorch-CPU, the reference's
dataset.py / image.py loaders and augmentation).  Here the py3 spelling of the same two loops (tests/drivers_py3.py,
bare-name imports resolved through fewshot_detection_amd/compat exactly as INTEGRATION.md §1 prescribes) runs on the
GPU on the batches the reference's loaders produced, and has to reproduce the reference run: per-step losses, LR
schedule, parameters after four SGD steps, and the comp4_det_test_<class>.txt result files of valid().

(The reference sources cannot be exec'd here: /root/reference does not exist on the GPU box, and the build container
that has it has no GPU.  tests/test_dropin_cpu.py covers the import side against the real files.)
"""
import os
import random
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
COMPAT = os.path.join(ROOT, "fewshot_detection_amd", "compat")
BARE = ("cfg", "utils", "darknet_meta", "darknet", "region_loss", "dynamic_conv", "pooling", "drivers_py3")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture()
def drivers():
    """Import tests/drivers_py3.py the way a maintainer's driver would run: compat/ in front on sys.path."""
    from fewshot_detection_amd.cfg import cfg
    saved_cfg = dict(cfg)
    saved_mods = {m: sys.modules.pop(m, None) for m in BARE}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, COMPAT)
    try:
        import drivers_py3
        assert "compat" in sys.modules["darknet_meta"].__file__ and "compat" in sys.modules["utils"].__file__
        yield drivers_py3
    finally:
        sys.path.remove(COMPAT)
        sys.path.remove(os.path.join(ROOT, "tests"))
        for m, v in saved_mods.items():
            sys.modules.pop(m, None)
            if v is not None:
                sys.modules[m] = v
        cfg.clear()
        cfg.update(saved_cfg)


def _write_inputs(d, tmp, tuning):
    tmp = str(tmp)
    with open(os.path.join(tmp, "net.cfg"), "wb") as fh:
        fh.write(d["net_cfg"].tobytes())
    with open(os.path.join(tmp, "novels.txt"), "w") as fh:
        fh.write("bird,bus,cow,motorbike,sofa\n")
    os.makedirs(os.path.join(tmp, "backup", "w"), exist_ok=True)
    with open(os.path.join(GOLD, "mini.weights"), "rb") as src, open(os.path.join(tmp, "backup", "w", "mini.weights"), "wb") as dst:
        dst.write(src.read())
    os.makedirs(os.path.join(tmp, "VOC", "images"), exist_ok=True)
    lines = []
    for name in d["valid_image_files"]:
        p = os.path.join(tmp, "VOC", "images", str(name))
        with open(p, "wb") as fh:
            fh.write(d["valid_png/" + str(name)].tobytes())
        lines.append(p)
    with open(os.path.join(tmp, "valid.txt"), "w") as fh:
        fh.write("".join(p + "\n" for p in lines))
    common = ("metayolo=1\nmetain_type=2\ndata=voc\nrand = 0\nnovel = %s\nnovelid = 0\nscale = 0\nnum_workers = 0\n"
              "meta = %s\ntrain = %s\nvalid = %s\ngpus=0\n" % (os.path.join(tmp, "novels.txt"), os.path.join(tmp, "metadict_2shot.txt"),
                                                              os.path.join(tmp, "train.txt"), os.path.join(tmp, "valid.txt")))
    path = os.path.join(tmp, "tune.data" if tuning else "base.data")
    with open(path, "w") as fh:
        fh.write(common + ("neg = 0\ntuning = 1\nmax_epoch = 2\nrepeat = 1\ndynamic = 0\nbackup = backup/metatune\n" if tuning
                           else "neg = 1\nbackup = backup/metayolo\n"))
    return path, lines


def _f(a):
    return torch.from_numpy(a.astype(np.float32) / np.float32(255.0))


class _Replay(list):
    """A loader that hands out recorded batches; `.dataset` only has a length (train_meta.py:226 reads it)."""

    def __init__(self, batches, n):
        super(_Replay, self).__init__(batches)
        self.dataset = range(n)


def test_train_meta_loop_reproduces_the_reference_run(dev, drivers, tmp_path):
    """train_meta.py on the base-training data cfg (neg = 1, 15 base classes, batch 4, lr steps -1,1061,1063 after the 1060 batches the weight file's `seen` stands for): four steps =
    two epochs of two batches, torch.optim.SGD stepping the parameters through plain autograd `.grad`s."""
    from fewshot_detection_amd.region_loss import RegionLossV2
    d = np.load(os.path.join(GOLD, "drivers.npz"))
    datacfg, _ = _write_inputs(d, tmp_path, tuning=False)
    n = int(d["train_n"])
    steps = [(_f(d["train%d_data" % i]), torch.from_numpy(d["train%d_target" % i]), _f(d["train%d_metax" % i]),
              _f(d["train%d_mask" % i])) for i in range(n)]

    def make_loaders(epoch, model):
        if epoch >= 2:
            return None
        part = steps[2 * epoch:2 * epoch + 2]
        return _Replay([(s[0], s[1]) for s in part], 8), [(s[2], s[3]) for s in part]

    calls = {"n": 0, "outputs": []}
    orig = RegionLossV2.forward

    def seeded(self, output, target):
        calls["n"] += 1
        calls["outputs"].append(output.detach().cpu())
        random.seed(9000 + calls["n"])                       # the stream the reference run's neg_filter drew from
        return orig(self, output, target)

    RegionLossV2.forward = seeded
    try:
        cwd = os.getcwd()
        os.chdir(str(tmp_path))
        r = drivers.train(datacfg, os.path.join(str(tmp_path), "net.cfg"), os.path.join(GOLD, "mini_reweight.cfg"),
                          "backup/w/mini.weights", make_loaders, dev)
    finally:
        os.chdir(cwd)
        RegionLossV2.forward = orig
    from fewshot_detection_amd.cfg import cfg
    assert [str(cfg.neg_ratio), str(cfg.backup)] == d["train_cfg"].tolist()
    assert len(cfg.base_classes) == 15 and steps[0][1].shape == (4, 15, 250)
    hp = d["train_hparams"]
    g = r["optimizer"].param_groups[0]
    assert np.allclose([g["lr"], g["momentum"], g["weight_decay"]], hp[:3], rtol=1e-12)
    assert np.allclose([r["adjust_learning_rate"](r["optimizer"], b) for b in range(1059, 1066)], d["train_lrs"], rtol=1e-12)
    assert r["processed_batches"] == int(d["train_processed_batches"]) and int(r["region_loss"].seen) == int(d["train3_seen"])
    # head output of the first step: same weights, same batch -> north-star tolerance
    assert float((calls["outputs"][0] - torch.from_numpy(d["train0_output"])).abs().max()) < 1e-3
    ref_losses = np.array([float(d["train%d_loss" % i]) for i in range(n)])
    assert np.allclose(r["losses"], ref_losses, rtol=1e-3), (r["losses"], ref_losses)
    # parameters after the four steps: the UPDATE (final - initial) of every tensor agrees with the reference's
    init = drivers.Darknet(os.path.join(str(tmp_path), "net.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    init.load_weights(os.path.join(GOLD, "mini.weights"))
    init_sd, got_sd = init.state_dict(), r["model"].state_dict()
    worst = 0.0
    for k, v0 in init_sd.items():
        if k.endswith("num_batches_tracked"):
            assert int(got_sd[k]) == int(d["train_final/" + k])
            continue
        ref, got = torch.from_numpy(d["train_final/" + k]), got_sd[k].cpu()
        upd = (ref - v0).norm()
        assert upd > 0, k
        err = float((got - ref).norm() / upd)
        worst = max(worst, err)
        assert err < 2e-2, (k, err)
    print("worst relative error of a parameter update after 4 steps: %.2e" % worst)


class _ReplayData(object):
    """Stands in for the reference's dataset.py: the items its listDataset / MetaDataset produced for drivers.npz."""

    def __init__(self, d, lines, classes):
        self.d, self.lines_, self.classes_ = d, lines, classes

    def listDataset(self, root, shape=None, shuffle=True, transform=None, train=False, **kw):
        d, outer = self.d, self
        assert not shuffle and not train and tuple(shape) == (64, 64)
        imgs = torch.cat([_f(d["valid_data%d" % i]) for i in range(int(d["valid_n_data"]))])

        class DS(torch.utils.data.Dataset):
            lines = [ln + "\n" for ln in outer.lines_]

            def __len__(self):
                return imgs.shape[0]

            def __getitem__(self, i):
                return imgs[i], torch.zeros(1)
        return DS()

    def MetaDataset(self, metafiles, train=False, ensemble=False, with_ids=False, **kw):
        d, outer = self.d, self
        assert ensemble and with_ids and not train
        k = int(d["valid_n_meta"])
        mx = torch.cat([_f(d["valid_metax%d" % i]) for i in range(k)])
        mk = torch.cat([_f(d["valid_mask%d" % i]) for i in range(k)])
        ids = np.concatenate([d["valid_clsids%d" % i] for i in range(k)])

        class MS(torch.utils.data.Dataset):
            classes = outer.classes_

            def __len__(self):
                return mx.shape[0]

            def __getitem__(self, i):
                return mx[i], mk[i], int(ids[i])
        return MS()


def _match(ref_rows, got_rows):
    """Greedy one-to-one matching of result lines (prob, x1, y1, x2, y2) within print precision + fp32 forward noise."""
    used, hit = set(), 0
    for r in ref_rows:
        for j, g in enumerate(got_rows):
            if j in used:
                continue
            if abs(g[0] - r[0]) <= 2e-4 + 2e-3 * abs(r[0]) and np.all(np.abs(g[1:] - r[1:]) <= 0.05 + 1e-3 * np.abs(r[1:])):
                used.add(j)
                hit += 1
                break
    return hit


def test_valid_ensemble_writes_the_reference_result_files(dev, drivers, tmp_path):
    """valid_ensemble.valid(): 40 supports (2 per class, 20 classes) averaged by class id, five query images in batches
    of 2, decode at conf 0.005 with the class softmax over the 20 rows, NMS 0.45, one result file per class."""
    from fewshot_detection_amd.cfg import cfg, parse_cfg
    from fewshot_detection_amd.utils import read_data_cfg
    d = np.load(os.path.join(GOLD, "drivers.npz"))
    datacfg, lines = _write_inputs(d, tmp_path, tuning=True)
    net_cfg, rw_cfg = os.path.join(str(tmp_path), "net.cfg"), os.path.join(GOLD, "mini_reweight.cfg")
    darknet, learnet = parse_cfg(net_cfg), parse_cfg(rw_cfg)
    opts = read_data_cfg(datacfg)                              # valid_ensemble.py:196-208
    opts["gpus"] = "0"
    cfg.config_data(opts)
    cfg.config_meta(learnet[0])
    cfg.config_net(darknet[0])
    assert len(cfg.classes) == 20 and cfg.tuning
    assert [tuple(s) for s in d["valid_sizes"]] == [drivers.get_image_size(p) for p in lines]
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        prefix, dyn = drivers.valid(datacfg, darknet, learnet, "backup/w/mini.weights", "comp4_det_test_",
                                    _ReplayData(d, lines, cfg.classes), dev)
    finally:
        os.chdir(cwd)
    assert prefix == os.path.join("results", "w/enemini")
    assert float((dyn[0].cpu() - torch.from_numpy(d["valid_vectors"])).abs().max()) < 1e-4
    total = matched = got_total = 0
    for c in cfg.classes:
        path = os.path.join(str(tmp_path), prefix, "comp4_det_test_%s.txt" % c)
        rows = [ln.split() for ln in open(path)]
        ids = [r[0] for r in rows]
        vals = np.array([[float(v) for v in r[1:]] for r in rows], np.float64).reshape(-1, 5)
        ref_ids, ref_vals = d["valid_ids/" + c].tolist(), d["valid_rows/" + c]
        assert sorted(set(ids)) == sorted(set(ref_ids)), c
        # image ids appear in loader order in both files
        assert [i for k, i in enumerate(ids) if k == 0 or ids[k - 1] != i] == [i for k, i in enumerate(ref_ids)
                                                                               if k == 0 or ref_ids[k - 1] != i], c
        for img in sorted(set(ref_ids)):
            r = ref_vals[[k for k, i in enumerate(ref_ids) if i == img]]
            g = vals[[k for k, i in enumerate(ids) if i == img]]
            total += len(r)
            got_total += len(g)
            matched += _match(r, g)
    print("result lines: reference %d, here %d, matched %d" % (total, got_total, matched))
    assert total > 500
    # a box whose objectness sits at the 0.005 threshold or whose IoU sits at 0.45 may flip under fp32 forward noise
    assert matched >= 0.99 * total and abs(got_total - total) <= 0.01 * total


def test_build_targets_function_matches_the_reference(dev):
    """region_loss.build_targets called on its own with DECODED boxes (the reference's public function,
    region_loss.py:37-132): assignment tensors bit-exact, nGT / nCorrect equal."""
    from fewshot_detection_amd.region_loss import build_targets
    d = np.load(os.path.join(GOLD, "region_fns.npz"))
    anch = [1.3221, 1.73145, 3.19275, 4.00944, 5.05587, 8.09892, 9.47112, 4.84053, 11.2364, 10.0071]
    names = ["nGT", "nCorrect", "coord_mask", "conf_mask", "cls_mask", "tx", "ty", "tw", "th", "tconf", "tcls"]
    for k in range(int(d["bt_n"])):
        rows, g, seen = [int(v) for v in d["bt%d_cfg" % k]]
        res = build_targets(torch.from_numpy(d["bt%d_pred" % k]).to(dev), torch.from_numpy(d["bt%d_target" % k]), anch, 5, 1,
                            g, g, 1, 5, 0.6, seen)
        assert len(res) == 11
        for n, v in zip(names, res):
            want = d["bt%d_%s" % (k, n)]
            if n in ("nGT", "nCorrect"):
                assert int(v) == int(want), (k, n, v, want)
            elif n in ("tw", "th", "tconf"):
                assert np.allclose(v.cpu().numpy(), want, rtol=1e-5, atol=1e-6), (k, n)
            else:
                assert np.array_equal(v.cpu().numpy(), want), (k, n)
        assert int(d["bt%d_nCorrect" % k]) > 0


def test_do_detect_runs_the_plain_detector_end_to_end(dev):
    """utils.do_detect (utils.py:413-458) on a PIL image through the non-meta Darknet: forward, decode and NMS on the
    device; equals the three calls made by hand."""
    from PIL import Image
    from fewshot_detection_amd import utils
    from fewshot_detection_amd.darknet import Darknet
    net = Darknet(os.path.join(GOLD, "mini_tiny_yolo.cfg"))
    net.load_weights(os.path.join(GOLD, "mini_yolo.weights"))
    net = net.to(dev)
    rng = np.random.RandomState(0)
    img = Image.fromarray(rng.randint(0, 256, (net.height, net.width, 3)).astype(np.uint8))
    boxes = utils.do_detect(net, img, 0.05, 0.4)
    with torch.no_grad():
        out = net(utils.image2torch(img).to(dev))
    want = utils.nms(utils.get_region_boxes(out, 0.05, net.num_classes, net.anchors, net.num_anchors)[0], 0.4)
    assert len(boxes) == len(want) and len(boxes) > 0
    assert np.allclose(np.array(boxes, np.float64), np.array(want, np.float64))
    assert not net.training
