"""Host-side episode bookkeeping around the hot path (label side): the per-class target rows RegionLossV2
consumes, the support mask, and the learning-rate schedule of train_meta.py.  Plain numpy / python, same arithmetic and
the same order-dependent quirks as the reference; pinned by tests/golden/episode.npz (minted from the reference's own
functions).  The image side of the input pipeline (PIL crop / resize / HSV jitter) is NOT here.

    fill_truth_detection_meta  <- image.py:144-192       fill_truth_detection <- image.py:90-141
    support_mask               <- dataset.py:378-398     lr_factor / adjust_learning_rate <- train_meta.py:123-163
"""
import os

import numpy as np

from .cfg import cfg


def _warp(row, flip, dx, dy, sx, sy):
    """Box (cls, cx, cy, w, h) through the crop/scale jitter and flip; None if it degenerates (image.py:116-136)."""
    x1 = row[1] - row[3] / 2
    y1 = row[2] - row[4] / 2
    x2 = row[1] + row[3] / 2
    y2 = row[2] + row[4] / 2
    x1 = min(0.999, max(0, x1 * sx - dx))
    y1 = min(0.999, max(0, y1 * sy - dy))
    x2 = min(0.999, max(0, x2 * sx - dx))
    y2 = min(0.999, max(0, y2 * sy - dy))
    out = np.array([row[0], (x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], dtype=np.float64)
    if flip:
        out[1] = 0.999 - out[1]
    if out[3] < 0.001 or out[4] < 0.001:
        return None
    return out


def _rows(labels):
    """Label source -> (k, 5) float64: a darknet label file path, an array, or nothing."""
    if labels is None:
        return np.zeros((0, 5))
    if isinstance(labels, (str, bytes, os.PathLike)):
        if not (os.path.exists(labels) and os.path.getsize(labels)):
            return np.zeros((0, 5))
        labels = np.loadtxt(labels)
    return np.reshape(np.asarray(labels, dtype=np.float64), (-1, 5))


def fill_truth_detection_meta(labels, w, h, flip, dx, dy, sx, sy, base_ids=None, n_cls=None, max_boxes=None):
    """(n_cls, max_boxes*5) targets, one row per base class, boxes of class `base_ids[n]` in arrival order with their
    class field rewritten to n; boxes of other classes are skipped; stops once 50 boxes are placed in total (the
    reference hard-codes 50, image.py:187-188).  `w`, `h` are unused, as in the reference."""
    base_ids = list(cfg.base_ids if base_ids is None else base_ids)
    n_cls = len(cfg.base_classes) if n_cls is None else n_cls
    max_boxes = cfg.max_boxes if max_boxes is None else max_boxes
    label = np.zeros((n_cls, max_boxes, 5))
    counts = [0] * n_cls
    for row in _rows(labels):
        clsid = int(row[0])
        if clsid not in base_ids:
            continue
        box = _warp(row, flip, dx, dy, sx, sy)
        if box is None:
            continue
        ind = base_ids.index(clsid)
        if ind >= n_cls or counts[ind] >= max_boxes:
            raise ValueError("class slot overflow (the reference drops into pdb here, image.py:181-182)")
        box[0] = ind
        label[ind][counts[ind]] = box
        counts[ind] += 1
        if sum(counts) >= 50:
            break
    return np.reshape(label, (n_cls, -1))


def fill_truth_detection(labels, w, h, flip, dx, dy, sx, sy, base_ids=None, max_boxes=None, keep_all=False):
    """(max_boxes*5,) targets of the plain detector (image.py:90-141).  `keep_all` stands for the reference's
    `cfg.yolo_joint and imgid in cfg.metaids` escape (boxes of non-base classes are kept for those images)."""
    base_ids = list(cfg.base_ids if base_ids is None else base_ids)
    max_boxes = cfg.max_boxes if max_boxes is None else max_boxes
    label = np.zeros((max_boxes, 5))
    cc = 0
    for row in _rows(labels):
        if not (int(row[0]) in base_ids or keep_all):
            continue
        box = _warp(row, flip, dx, dy, sx, sy)
        if box is None:
            continue
        label[cc] = box
        cc += 1
        if cc >= 50:
            break
    return np.reshape(label, (-1))


def support_mask(box, w, h):
    """Pixel rectangle (x1, y1, x2, y2) and the (1, h, w) binary mask of one support box (cx, cy, bw, bh in [0,1]);
    mask is None when the rectangle is empty (dataset.py:378-398; python-3 `round` = banker's rounding, like there)."""
    x1 = int(max(0, round((box[0] - box[2] / 2) * w)))
    y1 = int(max(0, round((box[1] - box[3] / 2) * h)))
    x2 = int(min(w, round((box[0] + box[2] / 2) * w)))
    y2 = int(min(h, round((box[1] + box[3] / 2) * h)))
    if x1 == x2 or y1 == y2:
        return (x1, y1, x2, y2), None
    mask = np.zeros((1, h, w), dtype=np.float32)
    mask[:, y1:y2, x1:x2] = 1
    return (x1, y1, x2, y2), mask


def lr_factor(neg_ratio, n_test_classes):
    """train_meta.py:123-135: the divisor applied to the cfg learning rate (and multiplied into the weight decay)."""
    if neg_ratio == "full":
        return 15.0
    if neg_ratio == 1:
        return 3.0
    if neg_ratio == 0:
        return 1.5
    if neg_ratio == 5:
        return 8.0
    return n_test_classes


def adjust_learning_rate(batch, learning_rate, steps, scales, batch_size):
    """train_meta.py:150-163 -> (lr, lr per image = the value the optimizer gets).  `learning_rate` is the cfg value
    already divided by lr_factor."""
    lr = learning_rate
    for i in range(len(steps)):
        scale = scales[i] if i < len(scales) else 1
        if batch >= steps[i]:
            lr = lr * scale
            if batch == steps[i]:
                break
        else:
            break
    return lr, lr / batch_size
