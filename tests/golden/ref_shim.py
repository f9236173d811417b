"""Load the reference (/root/reference, Python 2.7 + torch 0.3.1 era) under py3 / torch 2.x.

TEST INFRASTRUCTURE ONLY.  Nothing here is shipped or imported by the product.
The reference sources are read *where they lie* (never copied into this repo) and
exec'd after a short, explicit list of textual py2->py3 / torch-0.3->2.x compat
substitutions.  Each substitution is semantics-preserving for the reference's
own (python-2, CUDA) execution:

  * `xrange` -> `range`
  * py2 integer `/` on ints -> `//` at the handful of call sites that feed
    sizes/indices (pad, anchor_step, Reorg views)
  * `torch.cuda.XTensor` / `.cuda()` -> CPU equivalents (there is no GPU here)
  * `.data[0]` on 0-d tensors -> `.item()`
  * `conf_mask[b][flat_mask] = 0` (0.3.1 allowed a flat 845-mask on a (A,H,W)
    tensor) -> `.view(-1)[flat_mask] = 0`
  * `x.data + grid_x` on equal-numel, different-shape tensors (0.3.1 semantics:
    element-wise by linear index) -> explicit `.view(-1)`
  * `pred_boxes[i]` row -> `.tolist()` (0.3.1 element access gave python doubles)
  * `size_average=False/True` -> `reduction='sum'/'mean'`
  * `easydict` (absent) -> a 10-line attribute dict
  * image.py: `resize(shape)` -> `resize(shape, Image.NEAREST)` and `point(f)` -> `point(int(f))`: the defaults of the
    reference's Pillow (< 7.0: NEAREST resize; <= 8: C-int truncation of point() tables), which newer Pillow changed
  * utils.read_truths `truths.size/5` -> `//`; utils._file_lines `buffer.count('\\n')` on a binary file -> `b'\\n'`
  * dataset.py: `np.int` -> `int` (alias removed from numpy); `torchvision.transforms` (not installed) -> a stand-in with
    Compose and ToTensor (HWC uint8 -> CHW float / 255)
  * `torch.sort(det_confs)` in utils.nms -> `stable=True` (tie order of equal float32 keys is unspecified in torch and
    version dependent; pinned to the visiting order)

It is used by tests/golden/make_golden.py to mint fixtures from the reference
itself, and by `-m "not gpu"` tests (skipped when /root/reference is absent).
"""
import os
import re
import sys
import types

REF = os.environ.get("FSDET_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF, "region_loss.py"))


class _EasyDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


_SUBS_COMMON = [
    ("xrange(", "range("),
    (".cuda()", ""),
    ("torch.cuda.FloatTensor", "torch.FloatTensor"),
    ("torch.cuda.LongTensor", "torch.LongTensor"),
    ("size_average=False", "reduction='sum'"),
    ("size_average=True", "reduction='mean'"),
    ("F.sigmoid", "torch.sigmoid"),
]

_SUBS = {
    "region_loss": [
        ("anchor_step = len(anchors)/num_anchors", "anchor_step = len(anchors)//num_anchors"),
        ("self.anchor_step = len(anchors)/num_anchors", "self.anchor_step = len(anchors)//num_anchors"),
        ("conf_mask[b][cur_ious>sil_thresh] = 0", "conf_mask[b].view(-1)[cur_ious>sil_thresh] = 0"),
        (".data[0]", ".item()"),
        # 0.3.1 added equal-numel tensors of different shape element-wise (no broadcasting)
        ("x.data + grid_x", "x.data.view(-1) + grid_x"),
        ("y.data + grid_y", "y.data.view(-1) + grid_y"),
        ("torch.exp(w.data) * anchor_w", "torch.exp(w.data).view(-1) * anchor_w"),
        ("torch.exp(h.data) * anchor_h", "torch.exp(h.data).view(-1) * anchor_h"),
        # python-2 `target[b][i]` on a DoubleTensor yields a python float; keep that
        ("build_targets(pred_boxes, target.data,", "build_targets(pred_boxes, _PyFloatRows(target.data),"),
        # 0.3.1: element access on a 1-D FloatTensor returned a python float (double), so the
        # tconf IoU ran in double arithmetic; modern torch would keep float32 0-d tensors
        ("pred_box = pred_boxes[b*nAnchors+best_n*nPixels+gj*nW+gi]",
         "pred_box = pred_boxes[b*nAnchors+best_n*nPixels+gj*nW+gi].tolist()"),
        # 0.3.1 boolean-mask indexing used ByteTensor; modern torch wants bool (same selection)
        ("cls[Variable(cls_mask.view(-1, 1).repeat(1,cs))]", "cls[cls_mask.view(-1, 1).repeat(1,cs).bool()]"),
        ("cls        = cls[cls_mask].view(-1, nC)", "cls        = cls[cls_mask.bool()].view(-1, nC)"),
        ("tcls.view(-1)[cls_mask]", "tcls.view(-1)[cls_mask.view(-1).bool()]"),
    ],
    "darknet_meta": [
        ("pad = (kernel_size-1)/2 if is_pad else 0", "pad = (kernel_size-1)//2 if is_pad else 0"),
        ("loss.anchor_step = len(loss.anchors)/loss.num_anchors", "loss.anchor_step = len(loss.anchors)//loss.num_anchors"),
        ("H/hs", "H//hs"), ("W/ws", "W//ws"),
        ("import pdb", "pdb = None"),
    ],
    "darknet": [
        ("pad = (kernel_size-1)/2 if is_pad else 0", "pad = (kernel_size-1)//2 if is_pad else 0"),
        ("loss.anchor_step = len(loss.anchors)/loss.num_anchors", "loss.anchor_step = len(loss.anchors)//loss.num_anchors"),
        ("H/hs", "H//hs"), ("W/ws", "W//ws"),
    ],
    "utils": [
        ("anchor_step = len(anchors)/num_anchors", "anchor_step = len(anchors)//num_anchors"),
        # utils.nms sorts the FLOAT32 keys 1 - det_conf; torch.sort leaves the order of EQUAL keys unspecified (and the
        # CPU implementation differs between torch versions).  The fixtures pin ties to the boxes' visiting order, the
        # only order that is a property of the algorithm rather than of one library build.
        ("_,sortIds = torch.sort(det_confs)", "_,sortIds = torch.sort(det_confs, stable=True)"),
        # py2 integer division / py2 str-is-bytes in the file helpers
        ("truths.reshape(truths.size/5, 5)", "truths.reshape(truths.size//5, 5)"),
        ("count += buffer.count('\\n')", "count += buffer.count(b'\\n')"),
    ],
    "cfg": [],
    "image": [
        # Pillow < 7.0 (the reference's era) resized with NEAREST by default; >= 7.0 with BICUBIC
        ("cropped.resize(shape)", "cropped.resize(shape, Image.NEAREST)"),
        ("img = img.resize(shape)", "img = img.resize(shape, Image.NEAREST)"),
        # Pillow <= 8 converted point() table entries with C (int) truncation; >= 9 rounds half-to-even
        ("cs[1].point(lambda i: i * sat)", "cs[1].point(lambda i: int(i * sat))"),
        ("cs[2].point(lambda i: i * val)", "cs[2].point(lambda i: int(i * val))"),
        ("cs[0] = cs[0].point(change_hue)", "cs[0] = cs[0].point(lambda i: int(change_hue(i)))"),
    ],
    "dynamic_conv": [("import pdb", "pdb = None")],
    # numpy >= 1.24 dropped the `np.int` alias of the builtin
    "dataset": [(".astype(np.int)", ".astype(int)")],
    "pooling": [],
}

_PRELUDE = {
    "region_loss": (
        "class _PyFloatRows(object):\n"
        "    '''rows of python floats, i.e. what torch-0.3.1 DoubleTensor[b][i] returned'''\n"
        "    def __init__(self, t):\n"
        "        self._t = t; self._rows = t.double().tolist()\n"
        "    def size(self, d=None):\n"
        "        return self._t.size() if d is None else self._t.size(d)\n"
        "    def __getitem__(self, b):\n"
        "        return self._rows[b]\n"
    ),
}

_loaded = {}


def _torchvision_stub():
    """dataset.py imports `torchvision.transforms` for Compose / ToTensor only; torchvision is not installed here."""
    if "torchvision" in sys.modules:
        return
    import numpy as np
    import torch
    tv, tr = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")

    class Compose(object):
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToTensor(object):          # torchvision's: HWC uint8 -> CHW float / 255
        def __call__(self, img):
            return torch.from_numpy(np.asarray(img).transpose(2, 0, 1).copy()).float().div(255)

    tr.Compose, tr.ToTensor = Compose, ToTensor
    tv.transforms, tv.datasets = tr, types.ModuleType("torchvision.datasets")
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.datasets": tv.datasets})


def load(name):
    """Return the reference module `name` (cfg, utils, region_loss, darknet_meta, ...)."""
    if name in _loaded:
        return _loaded[name]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    sys.modules.setdefault("easydict", types.SimpleNamespace(EasyDict=_EasyDict))
    src = open(os.path.join(REF, name + ".py")).read()
    for a, b in _SUBS_COMMON + _SUBS.get(name, []):
        src = src.replace(a, b)
    src = _PRELUDE.get(name, "") + src
    mod = types.ModuleType("ref_" + name)
    mod.__file__ = os.path.join(REF, name + ".py")
    # the reference modules import each other by bare name; serve the shimmed ones
    saved = {}
    deps = {"cfg": ["utils"], "region_loss": ["utils", "cfg"],
            "darknet_meta": ["utils", "cfg", "region_loss", "dynamic_conv", "pooling"],
            "darknet": ["utils", "cfg", "region_loss"], "image": ["cfg"],
            "dataset": ["utils", "image", "cfg"]}.get(name, [])
    if name == "dataset":
        _torchvision_stub()
    for d in deps:
        saved[d] = sys.modules.get(d)
        sys.modules[d] = load(d)
    cwd = os.getcwd()
    try:
        os.chdir(REF)  # cfg.py opens data/coco.names relative to its own dir anyway
        exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    finally:
        os.chdir(cwd)
        for d in deps:
            if saved[d] is None:
                sys.modules.pop(d, None)
            else:
                sys.modules[d] = saved[d]
    _loaded[name] = mod
    return mod
