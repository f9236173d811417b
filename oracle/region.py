"""Oracle: YOLOv2 region losses of the meta detector.  TEST INFRASTRUCTURE.

Restates, on the CPU, reference region_loss.py:
  * `select_rows`      <- neg_filter            (region_loss.py:15-34)
  * `assign_targets`   <- build_targets         (region_loss.py:37-132)
  * `region_loss_v2`   <- RegionLossV2.forward  (region_loss.py:252-366)  softmax over the N
                                                 episode classes at each (image, anchor, cell)
  * `region_loss_v1`   <- RegionLoss.forward    (region_loss.py:148-232)  classic per-cell softmax

Numeric types follow the reference exactly: ground truth is python-double, the IoU
"silence" test is float32 (utils.bbox_ious), anchor matching and tconf are python-double
on float32-valued predictions, and every stored target is float32.

The loss itself is written with torch ops so that autograd yields the reference gradients
(all masks/targets are constants, as in the reference which builds them from `.data`).
"""
import math
from numbers import Number
from random import random as _py_random

import numpy as np
import torch

from .boxes import iou_scalar, iou_vector


def select_rows(target_rows, neg_ratio, rand=_py_random):
    """Indices of the (image, class) rows that enter the box/objectness loss.

    target_rows: (R, L) float64 array.  neg_ratio: 'full' or a number.
    `rand` is called once per *negative* row, in row order (short-circuit `and`),
    exactly like reference region_loss.py:26.
    """
    n_rows = target_rows.shape[0]
    if neg_ratio == "full":
        return list(range(n_rows))
    if not isinstance(neg_ratio, Number):
        raise NotImplementedError("neg_ratio not recognized")
    pos = [bool(v != 0) for v in np.asarray(target_rows, dtype=np.float64).sum(axis=1)]
    n_pos = sum(pos)
    if n_pos == n_rows:          # reference divides by zero here; keep everything
        return list(range(n_rows))
    ratio = neg_ratio * n_pos * 1.0 / (n_rows - n_pos)
    if ratio >= 1:
        return list(range(n_rows))
    keep = [True if p else (not (rand() > ratio)) for p in pos]
    return [i for i, k in enumerate(keep) if k]


def assign_targets(pred_boxes, target_rows, anchors, n_anchors, n_h, n_w,
                   noobject_scale, object_scale, sil_thresh, seen, max_boxes=50):
    """pred_boxes: (R*A*H*W, 4) float32 (cx, cy, w, h in cell units), row-major over
    (row, anchor, j, i).  target_rows: (R, L) float64, [cls, cx, cy, w, h] x 50.
    Returns a dict of float32 arrays shaped (R, A, H, W) plus counters and the match list.
    """
    f32 = np.float32
    pred_boxes = np.asarray(pred_boxes, dtype=f32)
    tgt = [[float(v) for v in row] for row in np.asarray(target_rows, dtype=np.float64)]
    n_rows = len(tgt)
    step = len(anchors) // n_anchors
    assert step == 2, "only (w, h) anchors are used by the shipped cfgs"
    shape = (n_rows, n_anchors, n_h, n_w)
    conf_mask = np.full(shape, noobject_scale, dtype=f32)
    coord_mask = np.zeros(shape, f32)
    cls_mask = np.zeros(shape, f32)
    tx = np.zeros(shape, f32)
    ty = np.zeros(shape, f32)
    tw = np.zeros(shape, f32)
    th = np.zeros(shape, f32)
    tconf = np.zeros(shape, f32)
    tcls = np.zeros(shape, f32)
    per_row = n_anchors * n_h * n_w
    per_plane = n_h * n_w

    # (i) silence: predictions that already overlap some ground truth by > sil_thresh
    for r in range(n_rows):
        cur = pred_boxes[r * per_row:(r + 1) * per_row].T        # (4, per_row)
        best = np.zeros(per_row, f32)
        for t in range(max_boxes):
            if tgt[r][t * 5 + 1] == 0:
                break
            g = np.array([tgt[r][t * 5 + 1] * n_w, tgt[r][t * 5 + 2] * n_h,
                          tgt[r][t * 5 + 3] * n_w, tgt[r][t * 5 + 4] * n_h]).astype(f32)
            # torch.max(a, b) propagates NaN from either side
            iou = iou_vector(cur, g.reshape(4, 1))
            best = np.where(np.isnan(iou) | np.isnan(best), f32(np.nan), np.maximum(best, iou))
        conf_mask[r].reshape(-1)[best > sil_thresh] = 0

    # (ii) early training: pull every prediction towards its cell centre / anchor shape
    if seen < 12800:
        tx.fill(0.5)
        ty.fill(0.5)
        coord_mask.fill(1)

    # (iii) one responsible (anchor, cell) per ground-truth box; later boxes overwrite
    n_gt = 0
    n_correct = 0
    matches = []
    for r in range(n_rows):
        for t in range(50):
            if tgt[r][t * 5 + 1] == 0:
                break
            n_gt += 1
            gx = tgt[r][t * 5 + 1] * n_w
            gy = tgt[r][t * 5 + 2] * n_h
            gw = tgt[r][t * 5 + 3] * n_w
            gh = tgt[r][t * 5 + 4] * n_h
            gi, gj = int(gx), int(gy)
            best_iou, best_n = 0.0, -1
            for n in range(n_anchors):
                v = iou_scalar([0, 0, anchors[step * n], anchors[step * n + 1]], [0, 0, gw, gh])
                if v > best_iou:
                    best_iou, best_n = v, n
            pb = pred_boxes[r * per_row + best_n * per_plane + gj * n_w + gi]
            pb = [float(pb[0]), float(pb[1]), float(pb[2]), float(pb[3])]
            coord_mask[r, best_n, gj, gi] = 1
            cls_mask[r, best_n, gj, gi] = 1
            conf_mask[r, best_n, gj, gi] = object_scale
            tx[r, best_n, gj, gi] = gx - gi
            ty[r, best_n, gj, gi] = gy - gj
            tw[r, best_n, gj, gi] = math.log(gw / anchors[step * best_n])
            th[r, best_n, gj, gi] = math.log(gh / anchors[step * best_n + 1])
            v = iou_scalar([gx, gy, gw, gh], pb)
            tconf[r, best_n, gj, gi] = v
            tcls[r, best_n, gj, gi] = tgt[r][t * 5]
            if v > 0.5:
                n_correct += 1
            matches.append((r, t, best_n, gj, gi))
    return dict(nGT=n_gt, nCorrect=n_correct, coord_mask=coord_mask, conf_mask=conf_mask,
                cls_mask=cls_mask, tx=tx, ty=ty, tw=tw, th=th, tconf=tconf, tcls=tcls,
                matches=matches)


def _decode(out5, anchors, n_anchors):
    """out5: (R, A, 5+C, H, W) fp32.  Returns x, y, w, h, conf and pred boxes (R*A*H*W, 4)."""
    n_rows, _, _, n_h, n_w = out5.shape
    x = torch.sigmoid(out5[:, :, 0])
    y = torch.sigmoid(out5[:, :, 1])
    w = out5[:, :, 2]
    h = out5[:, :, 3]
    conf = torch.sigmoid(out5[:, :, 4])
    step = len(anchors) // n_anchors
    aw = torch.tensor([anchors[step * a] for a in range(n_anchors)], dtype=torch.float32)
    ah = torch.tensor([anchors[step * a + 1] for a in range(n_anchors)], dtype=torch.float32)
    gx = torch.arange(n_w, dtype=torch.float32).view(1, 1, 1, n_w)
    gy = torch.arange(n_h, dtype=torch.float32).view(1, 1, n_h, 1)
    with torch.no_grad():
        boxes = torch.stack([
            (x + gx).reshape(-1), (y + gy).reshape(-1),
            (torch.exp(w) * aw.view(1, -1, 1, 1)).reshape(-1),
            (torch.exp(h) * ah.view(1, -1, 1, 1)).reshape(-1)], dim=1)
    return x, y, w, h, conf, boxes.numpy()


def _box_terms(x, y, w, h, conf, tg, coord_scale):
    as_t = lambda a: torch.from_numpy(a)
    cm = as_t(tg["coord_mask"])
    sq = as_t(np.sqrt(tg["conf_mask"]))

    def half_sse(a, b):
        return ((a - b) ** 2).sum() / 2.0

    lx = coord_scale * half_sse(x * cm, as_t(tg["tx"]) * cm)
    ly = coord_scale * half_sse(y * cm, as_t(tg["ty"]) * cm)
    lw = coord_scale * half_sse(w * cm, as_t(tg["tw"]) * cm)
    lh = coord_scale * half_sse(h * cm, as_t(tg["th"]) * cm)
    lc = half_sse(conf * sq, as_t(tg["tconf"]) * sq)
    return lx, ly, lw, lh, lc


def region_loss_v2(output, target, anchors, n_anchors=5, n_classes=1, coord_scale=1.0,
                   noobject_scale=1.0, object_scale=5.0, class_scale=1.0, thresh=0.6, seen=0,
                   neg_ratio="full", max_boxes=50, rand=_py_random):
    """output: (B*N, A*(5+nC), H, W) fp32 (rows ordered b*N+n); target: (B, N, L) float64."""
    assert n_classes == 1, "the meta detector uses classes=1 (cfg/darknet_dynamic.cfg:263)"
    bs, cs = target.shape[0], target.shape[1]
    n_h, n_w = output.shape[2], output.shape[3]
    all5 = output.view(output.shape[0], n_anchors, 5 + n_classes, n_h, n_w)
    # logits[(b, a, j, i), n] = output[b*N+n, a*6+5, j, i]
    logits = all5[:, :, 5].reshape(bs, cs, n_anchors * n_h * n_w).transpose(1, 2).reshape(-1, cs)

    rows = np.asarray(target.reshape(-1, target.shape[-1]).double().numpy())
    keep = select_rows(rows, neg_ratio, rand)
    counts, _ = np.histogram(keep, bins=bs, range=(0, bs * cs))
    keep_t = torch.as_tensor(keep, dtype=torch.long)
    out5 = all5.index_select(0, keep_t)
    x, y, w, h, conf, boxes = _decode(out5, anchors, n_anchors)
    tg = assign_targets(boxes, rows[keep], anchors, n_anchors, n_h, n_w, noobject_scale,
                        object_scale, thresh, seen, max_boxes)

    # a cell trains the class softmax only if exactly one kept row of the image claims it
    img_mask = np.zeros((bs, n_anchors, n_h, n_w), np.float32)
    img_cls = np.zeros((bs, n_anchors, n_h, n_w), np.float32)
    start = 0
    for b in range(bs):
        if counts[b]:
            img_mask[b] = tg["cls_mask"][start:start + counts[b]].sum(axis=0)
            img_cls[b] = tg["tcls"][start:start + counts[b]].sum(axis=0)
        start += counts[b]
    sel = torch.from_numpy(img_mask == 1).view(-1)
    labels = torch.from_numpy(img_cls).view(-1)[sel].long()
    n_prop = int((conf > 0.25).sum().item())

    lx, ly, lw, lh, lc = _box_terms(x, y, w, h, conf, tg, coord_scale)
    if int(sel.sum()) > 0:
        lcls = class_scale * torch.nn.functional.cross_entropy(logits[sel], labels, reduction="sum")
    else:
        lcls = logits.sum() * 0.0
    total = lx + ly + lw + lh + lc + lcls
    return dict(loss=total, parts=(lx, ly, lw, lh, lc, lcls), nGT=tg["nGT"],
                nCorrect=tg["nCorrect"], nProposals=n_prop, keep=keep, targets=tg,
                img_cls_mask=(img_mask == 1), img_tcls=img_cls)


def region_loss_v1(output, target, anchors, n_anchors, n_classes, coord_scale=1.0,
                   noobject_scale=1.0, object_scale=5.0, class_scale=1.0, thresh=0.6, seen=0,
                   neg_ratio="full", max_boxes=50, metayolo=False, rand=_py_random):
    """output: (B, A*(5+nC), H, W); target: (B, L) or (B, N, L) float64 (flattened to rows)."""
    if target.dim() == 3:
        target = target.reshape(-1, target.shape[-1])
    rows = np.asarray(target.double().numpy())
    keep = select_rows(rows, neg_ratio, rand)
    n_h, n_w = output.shape[2], output.shape[3]
    out5 = output.view(output.shape[0], n_anchors, 5 + n_classes, n_h, n_w)
    out5 = out5.index_select(0, torch.as_tensor(keep, dtype=torch.long))
    x, y, w, h, conf, boxes = _decode(out5, anchors, n_anchors)
    # logits[(r, a, j, i), c] = output[r, a*(5+nC)+5+c, j, i]
    logits = out5[:, :, 5:].permute(0, 1, 3, 4, 2).reshape(-1, n_classes)
    tg = assign_targets(boxes, rows[keep], anchors, n_anchors, n_h, n_w, noobject_scale,
                        object_scale, thresh, seen, max_boxes)
    sel = torch.from_numpy(tg["cls_mask"] == 1).view(-1)
    if metayolo:                      # region_loss.py:198-199 zeroes the class targets in place
        tg["tcls"][:] = 0
    labels = torch.from_numpy(tg["tcls"]).view(-1)[sel].long()
    n_prop = int((conf > 0.25).sum().item())
    lx, ly, lw, lh, lc = _box_terms(x, y, w, h, conf, tg, coord_scale)
    if int(sel.sum()) > 0:
        lcls = class_scale * torch.nn.functional.cross_entropy(logits[sel], labels, reduction="sum")
    else:
        lcls = logits.sum() * 0.0
    total = lx + ly + lw + lh + lc + lcls
    return dict(loss=total, parts=(lx, ly, lw, lh, lc, lcls), nGT=tg["nGT"],
                nCorrect=tg["nCorrect"], nProposals=n_prop, keep=keep, targets=tg)
