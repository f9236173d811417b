"""Oracle: box IoU in the two numeric flavours the reference uses.  TEST INFRASTRUCTURE.

  * `iou_scalar`  -- python-double arithmetic on centre-format boxes, reference
                     utils.py:21-52 (`bbox_iou`, x1y1x2y2=False branch and True branch)
  * `iou_vector`  -- float32 tensor arithmetic, reference utils.py:54-83 (`bbox_ious`);
                     operation order is kept so fp32 roundings match
"""
import numpy as np


def iou_scalar(b1, b2, corners=False):
    if corners:
        lo_x, hi_x = min(b1[0], b2[0]), max(b1[2], b2[2])
        lo_y, hi_y = min(b1[1], b2[1]), max(b1[3], b2[3])
        w1, h1 = b1[2] - b1[0], b1[3] - b1[1]
        w2, h2 = b2[2] - b2[0], b2[3] - b2[1]
    else:
        lo_x = min(b1[0] - b1[2] / 2.0, b2[0] - b2[2] / 2.0)
        hi_x = max(b1[0] + b1[2] / 2.0, b2[0] + b2[2] / 2.0)
        lo_y = min(b1[1] - b1[3] / 2.0, b2[1] - b2[3] / 2.0)
        hi_y = max(b1[1] + b1[3] / 2.0, b2[1] + b2[3] / 2.0)
        w1, h1, w2, h2 = b1[2], b1[3], b2[2], b2[3]
    ov_w = w1 + w2 - (hi_x - lo_x)
    ov_h = h1 + h2 - (hi_y - lo_y)
    if ov_w <= 0 or ov_h <= 0:
        return 0.0
    inter = ov_w * ov_h
    return inter / (w1 * h1 + w2 * h2 - inter)


def iou_vector(p, g):
    """p, g: float32 arrays (4, n) centre format (g may be (4,1)); returns float32 (n,)."""
    f = np.float32
    p = np.asarray(p, dtype=f)
    g = np.asarray(g, dtype=f)
    two = f(2.0)
    lo_x = np.minimum(p[0] - p[2] / two, g[0] - g[2] / two)
    hi_x = np.maximum(p[0] + p[2] / two, g[0] + g[2] / two)
    lo_y = np.minimum(p[1] - p[3] / two, g[1] - g[3] / two)
    hi_y = np.maximum(p[1] + p[3] / two, g[1] + g[3] / two)
    ov_w = p[2] + g[2] - (hi_x - lo_x)
    ov_h = p[3] + g[3] - (hi_y - lo_y)
    a1 = p[2] * p[3]
    a2 = g[2] * g[3]
    inter = ov_w * ov_h
    inter = np.where((ov_w <= 0) | (ov_h <= 0), f(0), inter).astype(f)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / (a1 + a2 - inter)).astype(f)
