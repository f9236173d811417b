"""CPU-only checks of the drop-in boundary: with fewshot_detection_amd/compat in front
of the reference tree on PYTHONPATH, every public name of the reference's hot-path modules resolves here, the
reference's own dataset.py imports, and the host helpers of utils.py return what the reference's return
(tests/golden/utils_host.npz, region_fns.npz -- minted from the reference's source by tests/golden/make_golden.py)."""
import ast
import os
import random
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
COMPAT = os.path.join(ROOT, "fewshot_detection_amd", "compat")
sys.path.insert(0, GOLD)
import ref_shim  # noqa: E402

HOT_MODULES = ("utils", "cfg", "darknet_meta", "darknet", "region_loss", "dynamic_conv", "pooling")


def _public_names(path):
    """Top-level functions, classes and plain assignments of a python-2 era source file (parsed, not imported)."""
    tree = ast.parse(open(path).read())
    names, methods = [], {}
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            names.append(node.name)
            if isinstance(node, ast.ClassDef):
                methods[node.name] = [b.name for b in node.body if isinstance(b, ast.FunctionDef)]
        elif isinstance(node, ast.Assign):
            names += [t.id for t in node.targets if isinstance(t, ast.Name)]
    return names, methods


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")
def test_every_public_name_of_the_reference_modules_resolves_under_the_integration_recipe():
    """INTEGRATION.md §1 verbatim: PYTHONPATH = repo : repo/fewshot_detection_amd/compat : reference.  The name lists
    are derived from the reference files with `ast`; dataset.py / image.py are the REFERENCE's own files and must import
    against the compat modules (torchvision is not installed in this image: a two-class stand-in is injected)."""
    want = {}
    for mod in HOT_MODULES:
        names, methods = _public_names(os.path.join(ref_shim.REF, mod + ".py"))
        want[mod] = (names, methods)
    script = textwrap.dedent("""
        import importlib, json, sys, types
        try:
            import torchvision  # noqa
        except ImportError:
            tv = types.ModuleType("torchvision"); tr = types.ModuleType("torchvision.transforms")
            class Compose(object):
                def __init__(self, ts): self.ts = ts
                def __call__(self, x):
                    for t in self.ts: x = t(x)
                    return x
            class ToTensor(object):
                def __call__(self, img):
                    import numpy as np, torch
                    return torch.from_numpy(np.asarray(img).transpose(2, 0, 1).copy()).float().div(255.0)
            tr.Compose, tr.ToTensor = Compose, ToTensor
            tv.transforms = tr; tv.datasets = types.ModuleType("torchvision.datasets")
            sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.datasets": tv.datasets})
        want = json.loads(sys.argv[1])
        missing = []
        for mod, (names, methods) in want.items():
            m = importlib.import_module(mod)
            assert "fewshot_detection_amd" in m.__file__, (mod, m.__file__)
            missing += [mod + "." + n for n in names if not hasattr(m, n)]
            for cls, ms in methods.items():
                if hasattr(m, cls):
                    missing += ["%s.%s.%s" % (mod, cls, x) for x in ms if not hasattr(getattr(m, cls), x)]
        import dataset, image                                    # the reference's own files
        assert "fewshot_detection_amd" not in dataset.__file__
        ns = {}
        exec("from utils import *", ns)
        for n in ("read_data_cfg", "get_region_boxes_v2", "nms", "get_image_size", "bbox_iou", "logging", "file_lines"):
            assert n in ns, n
        from darknet_meta import Darknet                         # noqa
        from cfg import parse_cfg, cfg                           # noqa
        assert dataset.read_truths_args.__module__.startswith("fewshot_detection_amd")
        print(json.dumps(missing))
    """)
    import json
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, COMPAT, ref_shim.REF]))
    out = subprocess.run([sys.executable, "-c", script, json.dumps(want)], env=env, capture_output=True, text=True,
                         cwd=str(ROOT), timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    missing = json.loads(out.stdout.strip().splitlines()[-1])
    assert missing == [], missing


def test_compat_modules_carry_the_names_the_drivers_import():
    """Same check without the reference tree (runs on the GPU box too): the names train_meta.py / valid_ensemble.py /
    dataset.py take from the hot-path modules (train_meta.py:17-24,85,123-163,201-226,283-300;
    valid_ensemble.py:1-8,14,31-35,89-100,142-166; dataset.py:11-13)."""
    need = {"utils": ["read_data_cfg", "get_region_boxes", "get_region_boxes_v2", "nms", "bbox_iou", "bbox_ious", "logging",
                      "get_image_size", "read_truths", "read_truths_args", "is_dict", "file_lines", "load_class_names",
                      "image2torch", "do_detect", "scale_bboxes", "softmax", "sigmoid", "convert2cpu", "convert2cpu_long",
                      "plot_boxes", "plot_boxes_cv2", "_file_lines", "all_file_lines"],
            "cfg": ["cfg", "parse_cfg", "print_cfg", "load_conv", "load_conv_bn", "save_conv", "save_conv_bn", "load_fc",
                    "save_fc", "load_convfromcoco", "load_classes", "get_ids", "get_novels", "add_backup", "__C"],
            "darknet_meta": ["Darknet", "MaxPoolStride1", "Reorg", "EmptyModule", "Reshape", "maybe_repeat"],
            "darknet": ["Darknet", "MaxPoolStride1", "Reorg", "GlobalAvgPool2d", "EmptyModule"],
            "region_loss": ["RegionLoss", "RegionLossV2", "neg_filter", "build_targets", "select_classes"],
            "dynamic_conv": ["dynamic_conv2d", "_ConvNd"],
            "pooling": ["GlobalMaxPool2d", "GlobalAvgPool2d", "Split"]}
    saved = {m: sys.modules.pop(m, None) for m in need}
    sys.path.insert(0, COMPAT)
    try:
        import importlib
        for mod, names in need.items():
            m = importlib.import_module(mod)
            assert "compat" in m.__file__
            for n in names:
                assert hasattr(m, n), (mod, n)
        import cfg as c
        for k in ("config_data", "config_meta", "config_net", "neg_ratio", "metayolo", "metain_type", "max_boxes", "tuning",
                  "repeat", "save_interval", "multiscale", "voc_classes", "coco_classes"):
            assert k in c.cfg, k
    finally:
        sys.path.remove(COMPAT)
        for m, v in saved.items():
            sys.modules.pop(m, None)
            if v is not None:
                sys.modules[m] = v


def _restore_files(d, tmp):
    for k in d.files:
        if k.startswith("file_"):
            data = d[k].tobytes()
            if k == "file_dict.txt":
                data = data.replace(b"@TMP@", str(tmp).encode())
            with open(os.path.join(str(tmp), k[5:]), "wb") as fh:
                fh.write(data)


def test_utils_host_helpers_match_the_reference(tmp_path):
    from fewshot_detection_amd import utils as u
    from PIL import Image
    d = np.load(os.path.join(GOLD, "utils_host.npz"))
    _restore_files(d, tmp_path)
    p = lambda n: os.path.join(str(tmp_path), n)                  # noqa: E731
    for name in ("empty.txt", "one.txt", "many.txt"):
        got, want = np.asarray(u.read_truths(p(name)), np.float64), d["truths_" + name]
        assert got.shape == want.shape and np.array_equal(got, want), name
        got, want = np.asarray(u.read_truths_args(p(name), 0.05), np.float64), d["truths_args_" + name]
        assert got.shape == want.shape and np.array_equal(got, want), name
    assert np.asarray(u.read_truths(p("nope.txt"))).shape == d["truths_missing"].shape
    assert list(u.load_class_names(p("x.names"))) == list(d["names"])
    assert [u.is_dict(p("list1.txt")), u.is_dict(p("dict.txt"))] == d["is_dict"].tolist()
    assert [u.file_lines(p("list1.txt")), u.file_lines(p("list2.txt")), u.file_lines(p("dict.txt")),
            u._file_lines(p("dict.txt"))] == d["file_lines"].tolist()
    sizes = [u.get_image_size(p(n)) for n in ("a.png", "b.gif", "c.jpg", "d.jpg")]
    assert np.array_equal(np.array(sizes), d["image_sizes"])
    assert u.get_image_size(p("short.png")) is None and u.get_image_size(p("text.jpg")) is None
    assert d["image_size_short"].all()
    t = u.image2torch(Image.fromarray(d["image2torch_in"]))
    assert t.shape == d["image2torch"].shape and np.array_equal(t.numpy(), d["image2torch"])
    boxes = d["scale_bboxes_in"].tolist()
    assert np.array_equal(np.array(u.scale_bboxes(boxes, 640, 480)), d["scale_bboxes"])
    assert boxes == d["scale_bboxes_in"].tolist()                  # the input is not edited
    assert np.allclose(u.softmax(torch.from_numpy(d["softmax_in"])).numpy(), d["softmax"], rtol=1e-6, atol=0)


def test_neg_filter_function_matches_the_reference():
    """region_loss.neg_filter (the reference's public function, region_loss.py:15-34): kept rows, filtered tensors and
    the number of random() draws."""
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.region_loss import neg_filter
    d = np.load(os.path.join(GOLD, "region_fns.npz"))
    saved = cfg.neg_ratio
    try:
        for k in range(int(d["nf_n"])):
            neg = str(d["nf%d_neg" % k])
            cfg.neg_ratio = neg if neg == "full" else int(neg)
            tgt = torch.from_numpy(d["nf%d_target" % k])
            pred = torch.arange(tgt.shape[0] * 4, dtype=torch.float32).view(-1, 4)
            random.seed(500 + k)
            p2, t2, inds = neg_filter(pred, tgt, withids=True)
            assert random.random() == float(d["nf%d_next_random" % k]), k
            assert list(np.asarray(inds).reshape(-1)) == d["nf%d_inds" % k].tolist(), k
            assert np.array_equal(p2.numpy(), d["nf%d_pred" % k]) and t2.shape[0] == len(inds)
            p3, t3 = neg_filter(pred, tgt)
            assert p3.shape[1] == 4
    finally:
        cfg.neg_ratio = saved


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")
def test_reference_drivers_only_use_names_the_boundary_has():
    """Static cross-check of the two drivers the north star names: every attribute they read from the model / loss /
    cfg objects and every bare function they call from `utils` exists on this side (train_meta.py, valid_ensemble.py)."""
    from fewshot_detection_amd import cfg as cfgmod
    from fewshot_detection_amd import utils as u
    from fewshot_detection_amd.darknet_meta import Darknet
    from fewshot_detection_amd.region_loss import RegionLossV2
    net = Darknet(os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    ref_utils_names, _ = _public_names(os.path.join(ref_shim.REF, "utils.py"))
    objs = {"model": net, "cur_model": net, "m": net, "region_loss": RegionLossV2(), "cfg": cfgmod.cfg}
    cfg_keys_set_later = {"backup", "base_classes", "classes", "data", "novel_classes", "num_gpus", "batch_size", "max_epoch",
                          "novelid", "_real_base_ids", "tuning", "repeat", "save_interval", "neg_ratio"}
    for drv in ("train_meta.py", "valid_ensemble.py"):
        tree = ast.parse(open(os.path.join(ref_shim.REF, drv)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in objs:
                if node.value.id == "cfg" and node.attr in cfg_keys_set_later:
                    continue
                if node.attr == "module":                      # nn.DataParallel wrapper, replaced by one process per GPU
                    continue
                assert hasattr(objs[node.value.id], node.attr), (drv, node.value.id, node.attr)
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in ref_utils_names:
                assert hasattr(u, node.func.id), (drv, node.func.id)
