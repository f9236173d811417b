"""Box math and small helpers with the reference's names (utils.py).

bbox_iou / bbox_ious / nms / read_data_cfg / convert2cpu / logging keep the reference's semantics
(utils.py:21-104, 460-475, 571-572).  get_region_boxes_v2 (utils.py:195-290) -- the inference-side
decode of the meta detector -- runs its softmax-over-classes + box decode on the device and only
ships the surviving boxes to the host.
"""
import math
import time

import torch


def sigmoid(x):
    return 1.0 / (math.exp(-x) + 1.0)


def bbox_iou(box1, box2, x1y1x2y2=True):
    """IoU of two boxes given as python numbers (corner or centre format); reference utils.py:21-52."""
    if x1y1x2y2:
        mx, Mx = min(box1[0], box2[0]), max(box1[2], box2[2])
        my, My = min(box1[1], box2[1]), max(box1[3], box2[3])
        w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
        w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
    else:
        mx = min(box1[0] - box1[2] / 2.0, box2[0] - box2[2] / 2.0)
        Mx = max(box1[0] + box1[2] / 2.0, box2[0] + box2[2] / 2.0)
        my = min(box1[1] - box1[3] / 2.0, box2[1] - box2[3] / 2.0)
        My = max(box1[1] + box1[3] / 2.0, box2[1] + box2[3] / 2.0)
        w1, h1, w2, h2 = box1[2], box1[3], box2[2], box2[3]
    cw = w1 + w2 - (Mx - mx)
    ch = h1 + h2 - (My - my)
    if cw <= 0 or ch <= 0:
        return 0.0
    carea = cw * ch
    return carea / (w1 * h1 + w2 * h2 - carea)


def bbox_ious(boxes1, boxes2, x1y1x2y2=True):
    """Vectorised IoU of (4, n) tensors; reference utils.py:54-83."""
    if x1y1x2y2:
        mx, Mx = torch.min(boxes1[0], boxes2[0]), torch.max(boxes1[2], boxes2[2])
        my, My = torch.min(boxes1[1], boxes2[1]), torch.max(boxes1[3], boxes2[3])
        w1, h1 = boxes1[2] - boxes1[0], boxes1[3] - boxes1[1]
        w2, h2 = boxes2[2] - boxes2[0], boxes2[3] - boxes2[1]
    else:
        mx = torch.min(boxes1[0] - boxes1[2] / 2.0, boxes2[0] - boxes2[2] / 2.0)
        Mx = torch.max(boxes1[0] + boxes1[2] / 2.0, boxes2[0] + boxes2[2] / 2.0)
        my = torch.min(boxes1[1] - boxes1[3] / 2.0, boxes2[1] - boxes2[3] / 2.0)
        My = torch.max(boxes1[1] + boxes1[3] / 2.0, boxes2[1] + boxes2[3] / 2.0)
        w1, h1, w2, h2 = boxes1[2], boxes1[3], boxes2[2], boxes2[3]
    cw = w1 + w2 - (Mx - mx)
    ch = h1 + h2 - (My - my)
    carea = cw * ch
    carea = torch.where((cw <= 0) | (ch <= 0), torch.zeros_like(carea), carea)
    return carea / (w1 * h1 + w2 * h2 - carea)


def _nms_host(boxes, nms_thresh):
    """utils.nms on plain python lists, statement for statement (the float32 sort key 1 - det_conf included)."""
    import numpy as np
    if len(boxes) == 0:
        return boxes
    keys = [float(np.float32(1.0) - np.float32(b[4])) for b in boxes]
    order = sorted(range(len(boxes)), key=lambda i: keys[i])
    out = []
    for pos, i in enumerate(order):
        bi = boxes[i]
        if bi[4] > 0:
            out.append(bi)
            for j in order[pos + 1:]:
                bj = boxes[j]
                if bbox_iou(bi, bj, x1y1x2y2=False) > nms_thresh:
                    bj[4] = 0
    return out


class _DecodedBatch(object):
    """Device-resident result of one fsd_region_decode call, shared by the per-row box lists it produced, so that the
    first `nms(row, thresh)` on any of them runs fsd_region_nms for ALL rows of the batch in one launch."""

    def __init__(self, boxes_dev, counts_dev, counts_host, positions):
        self.boxes_dev, self.counts_dev = boxes_dev, counts_dev
        self.counts = counts_host            # survivors per row
        self.positions = positions           # per row: slot -> position in the reference-ordered python list
        self._nms = {}

    def kept_positions(self, thresh):
        import numpy as np

        from ._lib import check, lib
        key = float(thresh)
        if key not in self._nms:
            rows, cap = self.boxes_dev.shape[0], self.boxes_dev.shape[1]
            keep_idx = torch.empty((rows, cap), dtype=torch.int32, device=self.boxes_dev.device)
            keep_cnt = torch.empty(rows, dtype=torch.int32, device=self.boxes_dev.device)
            check(lib().fsd_region_nms(self.boxes_dev.data_ptr(), self.counts_dev.data_ptr(), rows, cap, key,
                                       keep_idx.data_ptr(), keep_cnt.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "fsd_region_nms")
            cnt = keep_cnt.cpu().numpy()
            top = int(cnt.max()) if rows else 0
            idx = keep_idx[:, :max(top, 1)].cpu().numpy()
            self._nms[key] = [self.positions[r][idx[r, :cnt[r]]] if cnt[r] else np.zeros(0, np.int64)
                              for r in range(rows)]
        return self._nms[key]


class BoxList(list):
    """A row of get_region_boxes[_v2]: a plain list of [cx, cy, w, h, det_conf, cls_conf, cls_id] (the reference's
    return type) that remembers where its boxes live on the device."""
    __slots__ = ("_batch", "_row", "_n")


NMS_MAX_DEVICE_ROW = 2048      # fsd_region_nms holds one row in LDS


def nms(boxes, nms_thresh):
    """Greedy NMS over box lists [cx, cy, w, h, det_conf, ...]; reference utils.py:85-104.  Suppressed boxes get
    det_conf = 0 in place, like the reference.  Rows that came out of get_region_boxes[_v2] unmodified are suppressed
    on the device (fsd_region_nms, all rows of their batch in one launch); anything else takes the host loop."""
    if len(boxes) == 0:
        return boxes
    batch = getattr(boxes, "_batch", None)
    if batch is None or boxes._n != len(boxes) or batch.boxes_dev.shape[1] > NMS_MAX_DEVICE_ROW:
        return _nms_host(boxes, nms_thresh)
    import numpy as np
    kept = batch.kept_positions(nms_thresh)[boxes._row]
    dead = np.ones(len(boxes), bool)
    dead[kept] = False
    for pos in np.flatnonzero(dead).tolist():
        boxes[pos][4] = 0                   # the reference's in-place side effect on suppressed boxes
    return [boxes[p] for p in kept.tolist()]


def convert2cpu(gpu_matrix):
    return torch.FloatTensor(gpu_matrix.size()).copy_(gpu_matrix)


def convert2cpu_long(gpu_matrix):
    return torch.LongTensor(gpu_matrix.size()).copy_(gpu_matrix)


def read_data_cfg(datacfg):
    """key = value file -> dict, with the reference's defaults (utils.py:460-475)."""
    options = {"gpus": "0,1,2,3", "num_workers": "10"}
    with open(datacfg, "r") as fp:
        for line in fp:
            line = line.strip()
            if line == "" or "=" not in line or line.startswith("#"):
                continue
            key, value = line.split("=", 1)
            options[key.strip()] = value.strip()
    return options


def logging(message):
    print("%s %s" % (time.strftime("%Y-%m-%d %H:%M:%S", time.localtime()), message))


def _decode(output, rows_per_image, conf_thresh, num_classes, anchors, num_anchors, only_objectness, over_rows):
    import numpy as np

    from ._lib import check, lib
    from .ops import require_device
    if output.dim() == 3:
        output = output.unsqueeze(0)
    require_device(output)
    out = output.detach().contiguous().float()
    rows, chans, h, w = out.shape
    assert chans == (5 + num_classes) * num_anchors
    cap = num_anchors * h * w
    boxes = torch.empty((rows, cap, 8), dtype=torch.float32, device=out.device)
    counts = torch.empty(rows, dtype=torch.int32, device=out.device)
    anc = np.ctypeslib.as_ctypes(np.asarray(anchors, dtype=np.float64)[:2 * num_anchors].copy())
    check(lib().fsd_region_decode(out.data_ptr(), boxes.data_ptr(), counts.data_ptr(), rows, rows_per_image,
                                  num_anchors, num_classes, h, w, anc, float(conf_thresh), int(bool(only_objectness)),
                                  int(over_rows), cap, torch.cuda.current_stream().cuda_stream), "fsd_region_decode")
    n = counts.cpu().numpy()                              # tiny; the survivors follow in one more copy
    top = int(n.max()) if rows else 0
    host = boxes[:, :max(top, 1)].cpu().numpy()
    all_boxes = []
    positions = []
    for r in range(rows):
        b = host[r, :n[r]]
        order = np.argsort(b[:, 0], kind="stable")        # restore the reference's (cy, cx, anchor) order
        pos = np.empty(len(order), np.int64)
        pos[order] = np.arange(len(order))
        positions.append(pos)
        vals = b[order, 1:8].astype(np.float64).tolist()      # float32 -> python float (exact), nested lists built in C
        for v in vals:
            v[6] = int(v[6])
        row = BoxList(vals)
        row._row, row._n = r, len(order)
        all_boxes.append(row)
    batch = _DecodedBatch(boxes, counts, n, positions)
    for row in all_boxes:
        row._batch = batch
    return all_boxes


def get_region_boxes_v2(output, n_models, conf_thresh, num_classes, anchors, num_anchors, only_objectness=1,
                        validation=False):
    """Meta-detector decode (reference utils.py:195-290): output (bs*n_models, A*(5+C), H, W) -> per-row lists of
    [cx, cy, w, h, det_conf, cls_conf, cls_id]; class confidence = softmax across the n_models rows of an image."""
    if num_classes != 1:
        raise NotImplementedError("the meta detector uses classes=1")
    assert output.size(0) % n_models == 0
    return _decode(output, n_models, conf_thresh, num_classes, anchors, num_anchors, only_objectness, True)


def get_region_boxes(output, conf_thresh, num_classes, anchors, num_anchors, only_objectness=1, validation=False):
    """Plain YOLOv2 decode (reference utils.py:112-193), per-cell softmax over the class channels."""
    if validation and not only_objectness:
        raise NotImplementedError("validation-mode extra class scores are not on the MI355X path")
    return _decode(output, 1, conf_thresh, num_classes, anchors, num_anchors, only_objectness, False)
