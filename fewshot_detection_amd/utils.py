"""Box math and small helpers with the reference's names (utils.py) -- the whole public surface of that file, because
the reference's drivers and its dataset.py pull it in with `from utils import *` (train_meta.py:21, valid_ensemble.py:6,
dataset.py:11).

bbox_iou / bbox_ious / nms / read_data_cfg / convert2cpu / logging keep the reference's semantics
(utils.py:21-104, 460-475, 571-572).  get_region_boxes_v2 (utils.py:195-290) -- the inference-side
decode of the meta detector -- runs its softmax-over-classes + box decode on the device and only
ships the surviving boxes to the host.  The file / label / image-header helpers (utils.py:373-399, 488-569) are host
code with the reference's return values; do_detect (utils.py:413-458) runs the model and the device decode + NMS.
"""
import math
import os
import struct
import time

import numpy as np
import torch


def sigmoid(x):
    return 1.0 / (math.exp(-x) + 1.0)


def softmax(x):
    """Softmax of a whole tensor (all elements share one normaliser); reference utils.py:16-19."""
    e = torch.exp(x - torch.max(x))
    return e / e.sum()


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
    __slots__ = ("_batch", "_row", "_n", "_snapshot")


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
    # the device copy is only valid for the values it was decoded with: a caller that edited det_conf in place (the one
    # field drivers rescale or zero) gets the host loop on the edited values, like the reference.  Edits of the box
    # geometry between the two calls are not detected -- none of the reference's callers makes any.
    if [b[4] for b in boxes] != boxes._snapshot:
        return _nms_host(boxes, nms_thresh)
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
        row._snapshot = [v[4] for v in vals]              # det_conf as the device copy holds it (checked by nms)
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


# ---- label files, image lists, image headers (host side of dataset.py / valid_ensemble.py) -----------------------

def read_truths(lab_path):
    """Label file -> (n, 5) array of [cls, cx, cy, w, h]; empty array for a missing or empty file (utils.py:373-381)."""
    if not os.path.exists(lab_path) or not os.path.getsize(lab_path):
        return np.array([])
    truths = np.loadtxt(lab_path)
    return truths.reshape(truths.size // 5, 5)          # a single row comes back 1-D from loadtxt


def read_truths_args(lab_path, min_box_scale):
    """read_truths without the boxes narrower than min_box_scale (utils.py:383-390)."""
    truths = read_truths(lab_path)
    return np.array([[t[0], t[1], t[2], t[3], t[4]] for t in truths if not t[3] < min_box_scale])


def load_class_names(namesfile):
    """One class name per line, trailing whitespace stripped, blank lines kept (utils.py:392-399)."""
    with open(namesfile, "r") as fp:
        return [line.rstrip() for line in fp.readlines()]


def image2torch(img):
    """PIL RGB image -> (1, 3, H, W) float tensor in [0, 1] (utils.py:401-408)."""
    arr = np.frombuffer(img.tobytes(), dtype=np.uint8).reshape(img.height, img.width, 3)
    return torch.from_numpy(arr.transpose(2, 0, 1).copy()).view(1, 3, img.height, img.width).float().div(255.0)


def do_detect(model, img, conf_thresh, nms_thresh, use_cuda=1):
    """One image (PIL RGB or HxWx3 uint8 array) through a plain YOLOv2 `Darknet` -> boxes after NMS (utils.py:413-458).
    The forward pass, the decode and the suppression run on the device the model lives on."""
    model.eval()
    if hasattr(img, "tobytes") and hasattr(img, "height"):
        x = image2torch(img)
    elif isinstance(img, np.ndarray):
        x = torch.from_numpy(img.transpose(2, 0, 1)).float().div(255.0).unsqueeze(0)
    else:
        raise TypeError("do_detect: unknown image type %r" % type(img))
    if use_cuda:
        x = x.cuda()
    with torch.no_grad():
        output = model(x)
    boxes = get_region_boxes(output, conf_thresh, model.num_classes, model.anchors, model.num_anchors)[0]
    return nms(boxes, nms_thresh)


def scale_bboxes(bboxes, width, height):
    """Copies of the boxes with their first four fields scaled to pixels (utils.py:477-485)."""
    import copy
    dets = copy.deepcopy(bboxes)
    for d in dets:
        d[0], d[1], d[2], d[3] = d[0] * width, d[1] * height, d[2] * width, d[3] * height
    return dets


def is_dict(filename):
    """True for a "name path" list-of-lists file: its first line has exactly two fields (utils.py:488-494)."""
    with open(filename, "r") as f:
        return len(f.readline().strip().split()) == 2


def _file_lines(thefilepath):
    """Number of newline characters in a file (utils.py:504-513)."""
    count = 0
    with open(thefilepath, "rb") as fh:
        while True:
            chunk = fh.read(8192 * 1024)
            if not chunk:
                return count
            count += chunk.count(b"\n")


def all_file_lines(file_dict):
    """Distinct lines over all the list files a dict file names in its last column (utils.py:516-523)."""
    with open(file_dict, "r") as f:
        files = [line.rstrip().split()[-1] for line in f.readlines()]
    lines = set()
    for name in files:
        with open(name, "r") as f:
            lines.update(f.readlines())
    return len(lines)


def file_lines(thefilepath):
    return all_file_lines(thefilepath) if is_dict(thefilepath) else _file_lines(thefilepath)


def get_image_size(fname):
    """(width, height) from the header of a PNG / GIF / JPEG file, None for anything else or a malformed header
    (utils.py:536-569; the file types are told apart by their magic bytes instead of the deprecated `imghdr`)."""
    with open(fname, "rb") as fh:
        head = fh.read(24)
        if len(head) != 24:
            return None
        if head[:8] == b"\x89PNG\r\n\x1a\n":
            w, h = struct.unpack(">ii", head[16:24])
            return w, h
        if head[:6] in (b"GIF87a", b"GIF89a"):
            w, h = struct.unpack("<HH", head[6:10])
            return w, h
        if head[6:10] in (b"JFIF", b"Exif") or head[:2] == b"\xff\xd8":
            try:
                fh.seek(0)
                size, ftype = 2, 0
                while not 0xC0 <= ftype <= 0xCF:            # walk the segments up to the first SOFn
                    fh.seek(size, 1)
                    byte = fh.read(1)
                    while byte[0] == 0xFF:
                        byte = fh.read(1)
                    ftype = byte[0]
                    size = struct.unpack(">H", fh.read(2))[0] - 2
                fh.seek(1, 1)                                # precision byte
                h, w = struct.unpack(">HH", fh.read(4))
                return w, h
            except Exception:
                return None
        return None


def _class_colour(cls_id, n_classes):
    """The reference's colour wheel for a class id (utils.py:295-303, 341-349)."""
    wheel = [[1, 0, 1], [0, 0, 1], [0, 1, 1], [0, 1, 0], [1, 1, 0], [1, 0, 0]]
    ratio = float(cls_id * 123457 % n_classes) / n_classes * 5
    lo, hi = int(math.floor(ratio)), int(math.ceil(ratio))
    ratio -= lo
    return tuple(int(((1 - ratio) * wheel[lo][c] + ratio * wheel[hi][c]) * 255) for c in (2, 1, 0))


def plot_boxes(img, boxes, savename=None, class_names=None):
    """Draw boxes (relative centre format) on a PIL image (utils.py:331-370)."""
    from PIL import ImageDraw
    draw = ImageDraw.Draw(img)
    for box in boxes:
        x1, y1 = (box[0] - box[2] / 2.0) * img.width, (box[1] - box[3] / 2.0) * img.height
        x2, y2 = (box[0] + box[2] / 2.0) * img.width, (box[1] + box[3] / 2.0) * img.height
        rgb = (255, 0, 0)
        if len(box) >= 7 and class_names:
            print("%s: %f" % (class_names[box[6]], box[5]))
            rgb = _class_colour(box[6], len(class_names))
            draw.text((x1, y1), class_names[box[6]], fill=rgb)
        draw.rectangle([x1, y1, x2, y2], outline=rgb)
    if savename:
        print("save plot results to %s" % savename)
        img.save(savename)
    return img


def plot_boxes_cv2(img, boxes, savename=None, class_names=None, color=None):
    """Draw boxes on an OpenCV image (utils.py:292-329); needs cv2, which the reference imports lazily too."""
    import cv2
    height, width = img.shape[0], img.shape[1]
    for box in boxes:
        x1, y1 = int(round((box[0] - box[2] / 2.0) * width)), int(round((box[1] - box[3] / 2.0) * height))
        x2, y2 = int(round((box[0] + box[2] / 2.0) * width)), int(round((box[1] + box[3] / 2.0) * height))
        rgb = color if color else (255, 0, 0)
        if len(box) >= 7 and class_names:
            print("%s: %f" % (class_names[box[6]], box[5]))
            if color is None:
                rgb = _class_colour(box[6], len(class_names))
            img = cv2.putText(img, class_names[box[6]], (x1, y1), cv2.FONT_HERSHEY_SIMPLEX, 1.2, rgb, 1)
        img = cv2.rectangle(img, (x1, y1), (x2, y2), rgb, 1)
    if savename:
        print("save plot results to %s" % savename)
        cv2.imwrite(savename, img)
    return img
