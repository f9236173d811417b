"""Configuration: the process-global `cfg`, darknet .cfg parsing and the weight-stream helpers.

Mirrors the public names of reference cfg.py (cfg, parse_cfg :198-228, print_cfg :230-409,
load_conv/save_conv/load_conv_bn/save_conv_bn/load_fc/save_fc :411-481, cfg.config_data /
config_meta / config_net :70-195) so callers written against the reference keep working.
"""
from os import path

import numpy as np
import torch


class AttrDict(dict):
    """Minimal stand-in for easydict.EasyDict (not installed here): attribute access to keys."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = value


cfg = AttrDict()

cfg.voc_classes = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow",
                   "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa",
                   "train", "tvmonitor"]
cfg.coco_classes = [
    "person", "bicycle", "car", "motorbike", "aeroplane", "bus", "train", "truck", "boat", "traffic light",
    "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow",
    "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee",
    "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard", "surfboard",
    "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple",
    "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "sofa",
    "pottedplant", "bed", "diningtable", "toilet", "tvmonitor", "laptop", "mouse", "remote", "keyboard",
    "cell phone", "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors",
    "teddy bear", "hair drier", "toothbrush"]
cfg.vocids_in_coco = [cfg.coco_classes.index(c) for c in cfg.voc_classes]
cfg.cocoonly_ids = [i for i in range(len(cfg.coco_classes)) if i not in cfg.vocids_in_coco]

# read by the loss and the model at run time (reference cfg.py:27-39)
cfg.max_boxes = 50          # boxes per (image, class) row
cfg.neg_ratio = "full"      # 'full' or a number: negatives kept per positive row
cfg.tuning = False
cfg.metayolo = True
cfg.repeat = 1
cfg.save_interval = 10
cfg.multiscale = True
cfg.metain_type = 2         # 1 image only, 2 image + mask, 3 image + mask + cropped object


def _novel_classes(spec, idx):
    if spec.endswith("txt"):
        if idx == "None":
            return []
        with open(spec, "r") as fh:
            return fh.readlines()[int(idx)].strip().split(",")
    return spec.split(",")


def _few_shot_ids(listfile):
    with open(listfile, "r") as fh:
        rows = [ln.rstrip().split() for ln in fh]
    lines = []
    for row in rows:
        if row[0] in cfg.base_classes:
            with open(row[-1], "r") as fh:
                lines.extend(fh.readlines())
    return [ln.split("/")[-1].split(".")[0] for ln in sorted(set(lines))]


def _suffix_first(name, addon):
    parts = name.split("_")
    parts[0] += addon
    return "_".join(parts)


def _configure_data(opt):
    """`.data` key/value options -> global cfg (reference cfg.py:70-150)."""
    cfg.data = opt["data"]
    if opt["data"] == "voc":
        cfg.classes = cfg.voc_classes
    elif opt["data"] == "coco":
        cfg.classes = cfg.coco_classes
        cfg.save_interval = 2
    if "scale" in opt:
        cfg.multiscale = int(opt["scale"])
    if "metain_type" in opt:
        cfg.metain_type = int(opt["metain_type"])
    if "tuning" in opt:
        cfg.tuning = bool(int(opt["tuning"]))
        cfg.max_epoch = int(opt.get("max_epoch", 500))
        cfg.repeat = int(opt.get("repeat", 100))
        span = cfg.max_epoch / cfg.repeat
        cfg.save_interval = 1 if span <= 20 else 2 if span <= 50 else 5 if span <= 100 else 10
        if cfg.data == "coco":
            cfg.save_interval = 2
        cfg.shot = int(opt["meta"].split(".")[0].split("_")[-1].replace("shot", ""))
    cfg.novelid = opt.get("novelid", "None")
    cfg.novel_classes = _novel_classes(opt["novel"], cfg.novelid)
    if cfg.tuning:
        if opt["data"] not in ("coco", "voc"):
            raise NotImplementedError("Data type {} not found".format(opt["data"]))
        cfg.base_classes = cfg.classes
    else:
        cfg.base_classes = [c for c in cfg.classes if c not in cfg.novel_classes]
    cfg.base_ids = [cfg.classes.index(c) for c in cfg.base_classes]
    cfg.novel_ids = [cfg.classes.index(c) for c in cfg.novel_classes]
    cfg._real_base_ids = [i for i in range(len(cfg.classes)) if i not in cfg.novel_ids]
    cfg.num_gpus = len(opt["gpus"].split(","))
    cfg.neg_ratio = opt.get("neg", cfg.neg_ratio)
    cfg.randmeta = bool(int(opt["rand"])) if "rand" in opt else False
    cfg.metayolo = bool(int(opt["metayolo"]))
    if isinstance(cfg.neg_ratio, str) and cfg.neg_ratio.isdigit():
        v = float(cfg.neg_ratio)
        cfg.neg_ratio = int(v) if v.is_integer() else v
    backup = opt["backup"]
    if not cfg.multiscale:
        backup += "fix"
    if cfg.metain_type != 2:
        backup = _suffix_first(backup, "in{}".format(cfg.metain_type))
    backup += "_novel{}".format(cfg.novelid)
    if cfg.metayolo:
        backup += "_neg{}".format(cfg.neg_ratio)
    if cfg.randmeta:
        backup += "_rand"
    cfg.yolo_joint = int(opt["joint"]) if "joint" in opt else False
    if cfg.yolo_joint:
        cfg.metaids = _few_shot_ids(opt["meta"])
        backup += "_joint{}".format(int(opt["meta"].split(".")[0].split("_")[-1].replace("shot", "")))
    cfg.backup = backup


def _configure_net(opt):
    cfg.height = int(opt["height"])
    cfg.width = int(opt["width"])
    cfg.batch_size = int(opt["batch"])


_META_CHANNELS = {0: {1: 3, 2: 4, 3: 7, 4: 6}, 4: {1: 64, 2: 65, 3: 129, 4: 128}}


def _configure_meta(opt):
    """[learnet] options -> mask size and the reweighting net's input channels (cfg.py:152-190)."""
    cfg.meta_height = int(opt["height"])
    cfg.meta_width = int(opt["width"])
    factor = int(opt["feat_layer"])
    if factor not in _META_CHANNELS:
        raise NotImplementedError("Feat layer not found{}".format(factor))
    div = factor if factor else 1
    cfg.mask_height = cfg.meta_height // div
    cfg.mask_width = cfg.meta_width // div
    if cfg.metain_type not in _META_CHANNELS[factor]:
        raise NotImplementedError("Meta input type not found: {}".format(cfg.metain_type))
    opt["channels"] = _META_CHANNELS[factor][cfg.metain_type]


cfg.config_data = _configure_data
cfg.config_meta = _configure_meta
cfg.config_net = _configure_net


def load_classes(data="voc"):
    """Class names of a dataset (reference cfg.py:11-17 reads data/<name>.names next to the module; the two lists the
    reference ships are built in here, any other name is looked up as a .names file in the working directory's data/)."""
    if data in ("voc", "coco"):
        return list(cfg[data + "_classes"])
    with open(path.join("data", "{}.names".format(data))) as fh:
        return [ln.strip() for ln in fh.readlines()]


# the reference's module-level helper names (cfg.py:7, 41-68)
__C = cfg
get_novels = _novel_classes
get_ids = _few_shot_ids
add_backup = _suffix_first


def parse_cfg(cfgfile):
    """Darknet .cfg -> list of blocks (dicts of strings); see reference cfg.py:198-228."""
    blocks, cur = [], None
    with open(cfgfile, "r") as fh:
        for raw in fh:
            line = raw.rstrip()
            if line == "" or line[0] == "#":
                continue
            if line[0] == "[":
                if cur:
                    blocks.append(cur)
                cur = {"type": line.lstrip("[").rstrip("]")}
                if cur["type"] == "convolutional":
                    cur["batch_normalize"] = 0
            else:
                key, value = line.split("=")
                key = key.strip()
                cur["_type" if key == "type" else key] = value.strip()
    if cur:
        blocks.append(cur)
    return blocks


def print_cfg(blocks):
    """One line per layer with input/output shapes (same information as reference cfg.py:230-409)."""
    print("layer     filters    size              input                output")
    w = h = 416
    ch = 3
    shapes = []
    idx = -2
    for blk in blocks:
        idx += 1
        kind = blk["type"]
        if kind in ("net", "learnet"):
            w, h, ch = int(blk["width"]), int(blk["height"]), int(blk.get("channels", ch))
            continue
        iw, ih, ic = w, h, ch
        detail = ""
        if kind == "convolutional":
            k, s = int(blk["size"]), int(blk["stride"])
            pad = (k - 1) // 2 if int(blk["pad"]) else 0
            w, h, ch = (iw + 2 * pad - k) // s + 1, (ih + 2 * pad - k) // s + 1, int(blk["filters"])
            detail = "%4d  %d x %d / %d" % (ch, k, k, s)
        elif kind == "maxpool":
            k, s = int(blk["size"]), int(blk["stride"])
            if s > 1:
                w, h = iw // s, ih // s
            detail = "      %d x %d / %d" % (k, k, s)
        elif kind in ("avgpool", "globalavg", "globalmax"):
            w = h = 1
        elif kind == "reorg":
            s = int(blk["stride"])
            w, h, ch = iw // s, ih // s, ic * s * s
            detail = "            / %d" % s
        elif kind == "route":
            src = [int(v) if int(v) > 0 else int(v) + idx for v in blk["layers"].split(",")]
            w, h = shapes[src[0]][0], shapes[src[0]][1]
            ch = sum(shapes[s][2] for s in src)
            detail = " ".join(str(s) for s in src)
        elif kind == "shortcut":
            w, h, ch = shapes[idx - 1]
        elif kind == "connected":
            ch = int(blk["output"])
        elif kind == "split":
            ch = [int(v) for v in blk["splits"].split(",")][-1]
        elif kind in ("region", "cost", "softmax"):
            pass
        else:
            print("unknown type %s" % kind)
        print("%5d %-12s %-18s %3d x %3d x%4d   ->  %3d x %3d x%4d" % (idx, kind[:12], detail, iw, ih, ic, w, h, ch))
        shapes.append((w, h, ch))


# ---- darknet float32 weight stream ------------------------------------------------------------

def _pull(buf, start, dst):
    n = dst.numel()
    dst.data.copy_(torch.from_numpy(buf[start:start + n]).view_as(dst))
    # `.data.copy_` does not move the parameter's autograd version counter, which is what the engine's packed-weight
    # cache keys on: invalidate the cache explicitly so a model that already ran a forward pass sees the new weights
    from .engine import bump_weight_epoch
    bump_weight_epoch()
    return start + n


def _push(fp, t):
    t.detach().to("cpu", torch.float32).contiguous().numpy().tofile(fp)


def load_conv(buf, start, conv_model):
    if conv_model.bias is not None:
        start = _pull(buf, start, conv_model.bias)
    return _pull(buf, start, conv_model.weight)


def load_convfromcoco(buf, start, conv_model):
    """Initialise a 20-class VOC head (5 anchors x 25) from the 80-class COCO head stored in the stream (5 x 85 rows
    of 1024 weights): the box/objectness rows and the VOC classes' rows of every anchor (reference cfg.py:419-435)."""
    rows = np.concatenate([np.arange(5), np.asarray(cfg.vocids_in_coco) + 5])
    rows = np.concatenate([rows + a * 85 for a in range(5)])
    if conv_model.bias is not None:
        conv_model.bias.data.copy_(torch.from_numpy(buf[start:start + 425][rows].copy()))
        start += 425
    w = buf[start:start + 425 * 1024].reshape(425, 1024, 1, 1)[rows]
    conv_model.weight.data.copy_(torch.from_numpy(w.copy()))
    from .engine import bump_weight_epoch
    bump_weight_epoch()
    return start + 425 * 1024


def save_conv(fp, conv_model):
    if conv_model.bias is not None:
        _push(fp, conv_model.bias)
    _push(fp, conv_model.weight)


def load_conv_bn(buf, start, conv_model, bn_model):
    for t in (bn_model.bias, bn_model.weight, bn_model.running_mean, bn_model.running_var, conv_model.weight):
        start = _pull(buf, start, t)
    return start


def save_conv_bn(fp, conv_model, bn_model):
    for t in (bn_model.bias, bn_model.weight, bn_model.running_mean, bn_model.running_var, conv_model.weight):
        _push(fp, t)


def load_fc(buf, start, fc_model):
    start = _pull(buf, start, fc_model.bias)
    return _pull(buf, start, fc_model.weight)


def save_fc(fp, fc_model):
    _push(fp, fc_model.bias)
    _push(fp, fc_model.weight)
