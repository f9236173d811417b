"""Episode input pipeline around the hot path.

Label side (host, numpy; same arithmetic and order-dependent quirks as the reference, pinned by
tests/golden/episode.npz): the per-class target rows RegionLossV2 consumes, the support mask, the LR schedule.

    fill_truth_detection_meta  <- image.py:144-192       fill_truth_detection <- image.py:90-141
    support_mask               <- dataset.py:378-398     lr_factor / adjust_learning_rate <- train_meta.py:123-163

Image side (device): `draw_augmentation` consumes python's `random` exactly like image.data_augmentation +
random_distort_image (image.py:36-76), `index_tables` / `distort_luts` turn one draw into the small tables the gather
kernel needs, and `DeviceAugmenter` runs fsd_augment_batch: jitter crop + NEAREST resize + flip + HSV distortion +
ToTensor for a whole batch in one launch, output = the network input (NCHW, or channels-last RGB+mask pixels that the
first-layer kernels read without a layout pass).  Bit-exact with the reference on Pillow's 2018 defaults
(tests/golden/augment.npz); the host only decodes the image files.
"""
import os
import random as _random

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


# ---- image side ----------------------------------------------------------------------------------

def draw_augmentation(ow, oh, jitter=0.2, hue=0.1, saturation=1.5, exposure=1.5, rand=_random):
    """One image's random draws, in the order image.data_augmentation (image.py:52-76) and random_distort_image
    (image.py:36-47) make them: pleft, pright, ptop, pbot, flip, then hue, saturation, exposure (each scale = one
    uniform + one randint).  Returns the crop / flip / colour parameters and the (flip, dx, dy, sx, sy) the label
    warp needs (fill_truth_detection* take 1/sx, 1/sy, image.py:241-244)."""
    dw, dh = int(ow * jitter), int(oh * jitter)
    pleft = rand.randint(-dw, dw)
    pright = rand.randint(-dw, dw)
    ptop = rand.randint(-dh, dh)
    pbot = rand.randint(-dh, dh)
    flip = rand.randint(1, 10000) % 2
    swidth = ow - pleft - pright
    sheight = oh - ptop - pbot
    sx = float(swidth) / ow
    sy = float(sheight) / oh
    dx = (float(pleft) / ow) / sx
    dy = (float(ptop) / oh) / sy
    dhue = rand.uniform(-hue, hue)

    def rand_scale(s):
        scale = rand.uniform(1, s)
        return scale if rand.randint(1, 10000) % 2 else 1. / scale
    dsat = rand_scale(saturation)
    dexp = rand_scale(exposure)
    return dict(pleft=pleft, ptop=ptop, swidth=swidth, sheight=sheight, flip=flip, dx=dx, dy=dy, sx=sx, sy=sy,
                hue=dhue, sat=dsat, val=dexp)


def _nearest_table(box_w, out_w):
    """Pillow's NEAREST resize of a box_w-wide image to out_w columns: source column per output column.  The running
    sum `xo += scale` in double (Geometry.c ImagingScaleAffine) decides exact ties, so it is restated as a loop."""
    scale = float(box_w) / float(out_w)
    steps = np.full(out_w, scale, np.float64)
    steps[0] = scale * 0.5
    tab = np.cumsum(steps).astype(np.int64)      # cumsum adds left to right: the same doubles as the C loop `xo += scale`
    tab[tab >= box_w] = -1
    return tab


def index_tables(p, ow, oh, shape):
    """(xtab, ytab) int32 for fsd_augment_batch: source column / row of every output column / row, -1 where the
    jittered crop box leaves the image (Pillow fills with black).  p = draw_augmentation(...) or None (plain resize,
    data_augmentation(flag=False)).  shape = (width, height) like the reference's `shape`."""
    out_w, out_h = int(shape[0]), int(shape[1])
    if p is None:
        return _nearest_table(ow, out_w).astype(np.int32), _nearest_table(oh, out_h).astype(np.int32)
    cw, ch = p["swidth"] - 1, p["sheight"] - 1          # crop box (l, t, l + swidth - 1, t + sheight - 1), image.py:69
    if cw < 1 or ch < 1:
        raise ValueError("degenerate crop box %dx%d" % (cw, ch))
    xs, ys = _nearest_table(cw, out_w), _nearest_table(ch, out_h)
    xs = np.where(xs >= 0, xs + p["pleft"], -1)
    ys = np.where(ys >= 0, ys + p["ptop"], -1)
    xs = np.where((xs >= 0) & (xs < ow), xs, -1)
    ys = np.where((ys >= 0) & (ys < oh), ys, -1)
    if p["flip"]:
        xs = xs[::-1]
    return np.ascontiguousarray(xs, np.int32), np.ascontiguousarray(ys, np.int32)


def distort_luts(hue, sat, val):
    """(3, 256) uint8: the tables image.distort_image applies to the H, S, V bands (image.py:19-34)."""
    i = np.arange(256, dtype=np.float64)
    h = i + hue * 255
    h = np.where(h > 255, h - 255, h)
    h = np.where(h < 0, h + 255, h)
    # C (int) truncation, then clip to 0..255 -- the reference's Pillow converted point() tables that way
    return np.clip(np.trunc(np.stack([h, i * sat, i * val])), 0, 255).astype(np.uint8)


class DeviceAugmenter(object):
    """Batch of decoded uint8 RGB images (any sizes) -> the float network input on the device.

        aug = DeviceAugmenter(device)
        params = [draw_augmentation(im.shape[1], im.shape[0]) for im in images]      # or None per image: plain resize
        x = aug(images, params, (416, 416))                        # (B, 3, 416, 416) float32, contiguous NCHW
        x4 = aug(images, params, (416, 416), layout="nhwc4")      # (B, 4, 416, 416) channels_last: zero-copy input
        m4 = aug(supports, params, (416, 416), layout="nhwc4", mask_boxes=[(x1, y1, x2, y2), ...])   # RGB + mask
    """

    def __init__(self, device):
        import torch
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceAugmenter needs a HIP device (there is no CPU fallback)")

    def __call__(self, images, params, shape, layout="nchw", mask_boxes=None):
        import torch

        from ._lib import check, lib
        if layout not in ("nchw", "nhwc4"):
            raise ValueError("layout must be 'nchw' or 'nhwc4'")
        if mask_boxes is not None and layout != "nhwc4":
            raise ValueError("the support mask is channel 3 of the nhwc4 layout")
        B = len(images)
        if B < 1 or len(params) != B or (mask_boxes is not None and len(mask_boxes) != B):
            raise ValueError("images / params / mask_boxes must have the same, non-zero length")
        out_w, out_h = int(shape[0]), int(shape[1])
        offs, widths, chunks, xt, yt, luts = [], [], [], [], [], []
        any_lut = any(p is not None for p in params)
        pos = 0
        for im, p in zip(images, params):
            a = np.ascontiguousarray(np.asarray(im), np.uint8)
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("images must be (H, W, 3) uint8 RGB arrays, got %s" % (a.shape,))
            oh, ow = a.shape[:2]
            offs.append(pos)
            widths.append(ow)
            chunks.append(a.reshape(-1))
            pos += a.size
            xs, ys = index_tables(p, ow, oh, (out_w, out_h))
            xt.append(xs)
            yt.append(ys)
            if any_lut:
                luts.append(distort_luts(p["hue"], p["sat"], p["val"]) if p is not None
                            else np.tile(np.arange(256, dtype=np.uint8), (3, 1)))
        dev = self.device
        src = torch.from_numpy(np.concatenate(chunks)).to(dev, non_blocking=True)
        off_d = torch.tensor(offs, dtype=torch.int64).to(dev, non_blocking=True)
        w_d = torch.tensor(widths, dtype=torch.int32).to(dev, non_blocking=True)
        xt_d = torch.from_numpy(np.stack(xt)).to(dev, non_blocking=True)
        yt_d = torch.from_numpy(np.stack(yt)).to(dev, non_blocking=True)
        lut_d = torch.from_numpy(np.stack(luts)).to(dev, non_blocking=True) if any_lut else None
        on_d = None
        if any_lut and any(p is None for p in params):      # plain-resize images inside a distorting batch
            on_d = torch.tensor([0 if p is None else 1 for p in params], dtype=torch.int32).to(dev, non_blocking=True)
        mb_d = None
        if mask_boxes is not None:
            mb_d = torch.tensor([[int(v) for v in b] for b in mask_boxes], dtype=torch.int32).to(dev, non_blocking=True)
        if layout == "nchw":
            out = torch.empty((B, 3, out_h, out_w), dtype=torch.float32, device=dev)
        else:
            out = torch.empty((B, 4, out_h, out_w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        check(lib().fsd_augment_batch(src.data_ptr(), off_d.data_ptr(), w_d.data_ptr(), xt_d.data_ptr(), yt_d.data_ptr(),
                                      0 if lut_d is None else lut_d.data_ptr(), 0 if on_d is None else on_d.data_ptr(),
                                      0 if mb_d is None else mb_d.data_ptr(),
                                      out.data_ptr(), B, out_h, out_w, 0 if layout == "nchw" else 1,
                                      torch.cuda.current_stream().cuda_stream), "fsd_augment_batch")
        return out
