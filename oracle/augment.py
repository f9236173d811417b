"""Oracle: the reference's image-side episode pipeline, restated on numpy.  TEST INFRASTRUCTURE.

Follows image.py:13-87 (`distort_image`, `rand_scale`, `random_distort_image`, `data_augmentation`) and the Pillow
primitives those call.  Pillow's behaviour is version dependent in two places, so the semantics are STATED here and
pinned to the reference's era (Pillow 5.x, 2018; requirements.txt pins no version):

  * `Image.resize(shape)` without a filter meant NEAREST until Pillow 7.0 (BICUBIC since).  Stated: nearest
    neighbour, source index = trunc(xo) with xo = box_x0 + 0.5*scale and `xo += scale` per output pixel in double
    arithmetic (Pillow Geometry.c ImagingScaleAffine; the ACCUMULATED sum, not (x+0.5)*scale, decides exact ties);
  * `Image.point(lambda)` builds a 256-entry table; Pillow <= 8 converted each float entry with C `(int)` (truncation)
    and clipped to 0..255, Pillow >= 9 rounds half-to-even first.  Stated: truncation, then clip.

Unchanged across versions and restated from their published algorithm: crop with out-of-image area filled with 0,
FLIP_LEFT_RIGHT, and the RGB <-> HSV conversions (Pillow Convert.c `rgb2hsv_row` / `hsv2rgb`, themselves following
colorsys.py with uint8 quantisation).  tests/test_augment_cpu.py checks every function here against the INSTALLED
Pillow exhaustively (all 2^24 colours both ways) and against fixtures minted from the reference's own
`data_augmentation` (tests/golden/augment.npz).
"""
import numpy as np


def nearest_index_table(box_x0, box_w, out_w, in_w):
    """Source column of every output column (or -1 = outside the source) for a NEAREST resize of the horizontal box
    [box_x0, box_x0 + box_w) to out_w pixels."""
    scale = float(box_w) / float(out_w)
    xo = float(box_x0) + scale * 0.5
    tab = np.empty(out_w, np.int32)
    for x in range(out_w):
        xin = -1 if xo < 0.0 else int(xo)
        tab[x] = xin if 0 <= xin < in_w else -1
        xo += scale
    return tab


def rgb_to_hsv_u8(rgb):
    """(..., 3) uint8 -> (..., 3) uint8, Pillow convert('HSV')."""
    rgb = np.asarray(rgb, np.uint8)
    r, g, b = (rgb[..., i].astype(np.int32) for i in range(3))
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    f32 = np.float32
    cr = (maxc - minc).astype(f32)
    safe = np.where(cr == 0, f32(1), cr)
    s = cr / np.where(maxc == 0, 1, maxc).astype(f32)
    rc = (maxc - r).astype(f32) / safe
    gc = (maxc - g).astype(f32) / safe
    bc = (maxc - b).astype(f32) / safe
    d64 = np.float64               # `2.0 + rc - bc` has a double literal: evaluated in double, stored into a float
    h = np.where(r == maxc, (bc - gc).astype(d64),
                 np.where(g == maxc, 2.0 + rc.astype(d64) - bc.astype(d64), 4.0 + gc.astype(d64) - rc.astype(d64))).astype(f32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(f32)       # stored back into a C float
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    grey = minc == maxc
    out = np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], axis=-1)
    return out.astype(np.uint8)


def hsv_to_rgb_u8(hsv):
    """(..., 3) uint8 -> (..., 3) uint8, Pillow convert('RGB') of an HSV image."""
    hsv = np.asarray(hsv, np.uint8)
    h, s, v = (hsv[..., i].astype(np.float64) for i in range(3))
    h6 = np.float32(h).astype(np.float64) * 6.0 / 255.0
    i = np.floor(h6)
    f = h6 - i
    fs = s / 255.0
    rnd = lambda a: np.floor(a + 0.5)            # C round() on non-negative values
    p = np.clip(rnd(v * (1.0 - fs)), 0, 255)
    q = np.clip(rnd(v * (1.0 - fs * f)), 0, 255)
    t = np.clip(rnd(v * (1.0 - fs * (1.0 - f))), 0, 255)
    k = i.astype(np.int64) % 6
    vv = v
    r = np.choose(k, [vv, q, p, p, t, vv])
    g = np.choose(k, [t, vv, vv, q, p, p])
    b = np.choose(k, [p, p, t, vv, vv, q])
    grey = hsv[..., 1] == 0
    out = np.stack([np.where(grey, vv, r), np.where(grey, vv, g), np.where(grey, vv, b)], axis=-1)
    return out.astype(np.uint8)


def _clip8(v):
    v = int(v)                   # C (int): truncation toward zero
    return 0 if v < 0 else 255 if v > 255 else v


def distort_luts(hue, sat, val):
    """The three 256-entry tables image.distort_image applies to the H, S, V bands (image.py:19-34)."""
    def change_hue(x):
        x += hue * 255
        if x > 255:
            x -= 255
        if x < 0:
            x += 255
        return x
    lh = np.array([_clip8(change_hue(i)) for i in range(256)], np.uint8)
    ls = np.array([_clip8(i * sat) for i in range(256)], np.uint8)
    lv = np.array([_clip8(i * val) for i in range(256)], np.uint8)
    return lh, ls, lv


def distort_image(rgb, hue, sat, val):
    """image.distort_image on a (H, W, 3) uint8 array."""
    hsv = rgb_to_hsv_u8(rgb)
    lh, ls, lv = distort_luts(hue, sat, val)
    hsv = np.stack([lh[hsv[..., 0]], ls[hsv[..., 1]], lv[hsv[..., 2]]], axis=-1)
    return hsv_to_rgb_u8(hsv)


def draw_params(ow, oh, jitter, hue, saturation, exposure, rand):
    """The random draws of image.data_augmentation + random_distort_image, in the reference's order
    (image.py:36-47, 52-76).  `rand`: an object with randint / uniform (python's `random` module)."""
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
        if rand.randint(1, 10000) % 2:
            return scale
        return 1. / scale
    dsat = rand_scale(saturation)
    dexp = rand_scale(exposure)
    return dict(pleft=pleft, ptop=ptop, swidth=swidth, sheight=sheight, flip=flip, dx=dx, dy=dy, sx=sx, sy=sy,
                hue=dhue, sat=dsat, val=dexp)


def augment(img, p, shape):
    """image.data_augmentation(flag=True) given the draws `p`: (oh, ow, 3) uint8 -> (shape[1], shape[0], 3) uint8."""
    oh, ow = img.shape[:2]
    out_w, out_h = shape
    # crop box (pleft, ptop, pleft + swidth - 1, ptop + sheight - 1): note the reference's -1 (image.py:69)
    cw, ch = p["swidth"] - 1, p["sheight"] - 1
    xs = nearest_index_table(0, cw, out_w, cw)          # index inside the CROPPED image ...
    ys = nearest_index_table(0, ch, out_h, ch)
    xs = np.where(xs >= 0, xs + p["pleft"], -1)         # ... which sits at (pleft, ptop) of the source
    ys = np.where(ys >= 0, ys + p["ptop"], -1)
    xs = np.where((xs >= 0) & (xs < ow), xs, -1)
    ys = np.where((ys >= 0) & (ys < oh), ys, -1)
    sized = np.zeros((out_h, out_w, 3), np.uint8)
    okx, oky = xs >= 0, ys >= 0
    sized[np.ix_(oky, okx)] = img[np.ix_(ys[oky], xs[okx])]
    if p["flip"]:
        sized = sized[:, ::-1]
    return distort_image(sized, p["hue"], p["sat"], p["val"])


def resize_only(img, shape):
    """image.data_augmentation(flag=False): a plain NEAREST resize (validation / ensemble inputs)."""
    oh, ow = img.shape[:2]
    xs = nearest_index_table(0, ow, shape[0], ow)
    ys = nearest_index_table(0, oh, shape[1], oh)
    return img[np.ix_(ys, xs)]


def to_tensor(img_u8):
    """torchvision ToTensor: (H, W, 3) uint8 -> (3, H, W) float32 in [0, 1] (train_meta.py:176-178, no normalisation)."""
    return (np.asarray(img_u8, np.float32) / np.float32(255.0)).transpose(2, 0, 1)
