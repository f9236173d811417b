"""Image side of the episode input pipeline, CPU half: the numpy oracle (oracle/augment.py) against
fixtures minted from the reference's own image.data_augmentation (tests/golden/augment.npz) and against the INSTALLED
Pillow primitive by primitive; the product's host tables (episode.draw_augmentation / index_tables / distort_luts)
against the oracle."""
import os
import random

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_reproduces_reference_data_augmentation_goldens():
    from oracle import augment as A
    d = np.load(os.path.join(GOLD, "augment.npz"))
    for k in range(int(d["n"])):
        arr, shape = d["in%d" % k], tuple(int(v) for v in d["shape%d" % k])
        random.seed(100 + k)
        p = A.draw_params(arr.shape[1], arr.shape[0], 0.2, 0.1, 1.5, 1.5, random)
        assert random.random() == float(d["next_random%d" % k])          # same number of draws consumed
        assert np.array_equal([p["flip"], p["dx"], p["dy"], p["sx"], p["sy"]], d["par%d" % k])
        assert np.array_equal(A.augment(arr, p, shape), d["out%d" % k]), k
        assert np.array_equal(A.resize_only(arr, shape), d["plain%d" % k]), k


def test_oracle_colour_conversions_equal_installed_pillow():
    """Every 4th value per channel (262k colours) + the full grey / primary axes, both directions."""
    Image = pytest.importorskip("PIL.Image")
    from oracle import augment as A
    v = np.arange(0, 256, 4, dtype=np.uint8)
    v[-1] = 255
    r, g, b = np.meshgrid(v, v, v, indexing="ij")
    grid = np.stack([r, g, b], -1).reshape(512, 512, 3)
    axes = np.zeros((4, 256, 3), np.uint8)
    axes[0, :, :] = np.arange(256)[:, None]
    for c in range(3):
        axes[1 + c, :, c] = np.arange(256)
    for arr in (grid, axes):
        assert np.array_equal(A.rgb_to_hsv_u8(arr), np.array(Image.fromarray(arr, "RGB").convert("HSV")))
        assert np.array_equal(A.hsv_to_rgb_u8(arr), np.array(Image.fromarray(arr, "HSV").convert("RGB")))


def test_oracle_nearest_resize_and_crop_equal_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    from oracle import augment as A
    for sw in (1, 2, 3, 7, 10, 33, 50, 64):
        src = Image.fromarray(np.arange(sw, dtype=np.uint8).reshape(1, sw))
        for dw in (1, 2, 5, 15, 32, 48, 97):
            got = np.array(src.resize((dw, 1), Image.NEAREST)).reshape(-1)
            assert np.array_equal(got, A.nearest_index_table(0, sw, dw, sw)), (sw, dw)     # incl. the accumulated ties
    arr = np.random.RandomState(0).randint(1, 256, (9, 11, 3)).astype(np.uint8)
    crop = np.array(Image.fromarray(arr).crop((-3, -2, 14, 12)))       # outside area is black
    ref = np.zeros((14, 17, 3), np.uint8)
    ref[2:11, 3:14] = arr
    assert np.array_equal(crop, ref)


def test_host_tables_equal_oracle_and_consume_rng_identically():
    from fewshot_detection_amd import episode as E
    from oracle import augment as A
    rng = np.random.RandomState(3)
    for k in range(40):
        ow, oh = int(rng.randint(8, 300)), int(rng.randint(8, 300))
        shape = (int(rng.choice([32, 160, 416])), int(rng.choice([32, 160, 416])))
        random.seed(k)
        p = E.draw_augmentation(ow, oh)
        nxt = random.random()
        random.seed(k)
        q = A.draw_params(ow, oh, 0.2, 0.1, 1.5, 1.5, random)
        assert p == q and nxt == random.random()
        xs, ys = E.index_tables(p, ow, oh, shape)
        cw, ch = p["swidth"] - 1, p["sheight"] - 1
        ox = A.nearest_index_table(0, cw, shape[0], cw)
        ox = np.where(ox >= 0, ox + p["pleft"], -1)
        ox = np.where((ox >= 0) & (ox < ow), ox, -1)
        assert np.array_equal(xs, ox[::-1] if p["flip"] else ox)
        oy = A.nearest_index_table(0, ch, shape[1], ch)
        oy = np.where(oy >= 0, oy + p["ptop"], -1)
        assert np.array_equal(ys, np.where((oy >= 0) & (oy < oh), oy, -1))
        assert all(np.array_equal(a, b) for a, b in zip(E.distort_luts(p["hue"], p["sat"], p["val"]),
                                                        A.distort_luts(p["hue"], p["sat"], p["val"])))
    xs, ys = E.index_tables(None, 50, 37, (32, 48))
    assert np.array_equal(xs, A.nearest_index_table(0, 50, 32, 50)) and np.array_equal(ys, A.nearest_index_table(0, 37, 48, 37))
