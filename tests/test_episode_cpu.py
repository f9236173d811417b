"""Host-side episode bookkeeping (fewshot_detection_amd/episode.py) against fixtures minted from the reference's own
functions (tests/golden/make_golden.py: gold_episode) and, when /root/reference is present, against the live reference."""
import os
import sys

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)


@pytest.mark.parametrize("k", range(5))
def test_target_rows_equal_reference_golden(k, tmp_path):
    from fewshot_detection_amd import episode
    d = np.load(os.path.join(GOLD, "episode.npz"))
    flip, dx, dy, sx, sy = d["par%d" % k]
    base_ids = d["base_ids"].tolist()
    rows = d["in%d" % k]
    meta = episode.fill_truth_detection_meta(rows, 416, 416, int(flip), dx, dy, sx, sy, base_ids=base_ids, n_cls=15,
                                             max_boxes=50)
    det = episode.fill_truth_detection(rows, 416, 416, int(flip), dx, dy, sx, sy, base_ids=base_ids, max_boxes=50)
    assert meta.shape == (15, 250) and det.shape == (250,)
    assert np.array_equal(meta, d["meta%d" % k])            # bit-exact float64
    assert np.array_equal(det, d["det%d" % k])
    # same through a darknet label file (the reference's calling convention)
    path = os.path.join(str(tmp_path), "000001.txt")
    if len(rows):
        np.savetxt(path, rows, fmt="%.17g")
    else:
        open(path, "w").close()
    assert np.array_equal(episode.fill_truth_detection_meta(path, 416, 416, int(flip), dx, dy, sx, sy, base_ids=base_ids,
                                                            n_cls=15, max_boxes=50), d["meta%d" % k])
    # structure the loss relies on: row n only holds class n, zero-terminated on cx, at most 50 boxes in total
    boxes = meta.reshape(15, 50, 5)
    used = boxes[:, :, 3] > 0
    assert used.sum() <= 50
    for n in range(15):
        assert np.all(boxes[n, used[n], 0] == n)
        assert not used[n, used[n].sum():].any()


def test_missing_label_file_and_keep_all():
    from fewshot_detection_amd import episode
    z = episode.fill_truth_detection_meta("/nonexistent/label.txt", 1, 1, 0, 0, 0, 1, 1, base_ids=[0, 1], n_cls=2, max_boxes=50)
    assert z.shape == (2, 250) and not z.any()
    rows = np.array([[7, 0.5, 0.5, 0.2, 0.2]])
    assert not episode.fill_truth_detection(rows, 1, 1, 0, 0, 0, 1, 1, base_ids=[0], max_boxes=50).any()
    kept = episode.fill_truth_detection(rows, 1, 1, 0, 0, 0, 1, 1, base_ids=[0], max_boxes=50, keep_all=True)
    assert np.allclose(kept[:5], [7, 0.5, 0.5, 0.2, 0.2])


def test_live_reference_agrees():
    import ref_shim
    if not ref_shim.available():
        pytest.skip("reference tree not present")
    from fewshot_detection_amd import episode
    ref_shim.load("cfg")
    im = ref_shim.load("image")
    base_ids = [1, 4, 6]
    im.cfg.base_ids, im.cfg.base_classes, im.cfg.yolo_joint, im.cfg.metaids = base_ids, ["a", "b", "c"], False, []
    rng = np.random.RandomState(3)
    import tempfile
    tmp = tempfile.mkdtemp()
    for k in range(6):
        n = int(rng.randint(1, 12))
        rows = np.column_stack([rng.randint(0, 8, n), rng.uniform(0, 1, (n, 2)), rng.uniform(0.0, 0.5, (n, 2))])
        path = os.path.join(tmp, "%d.txt" % k)
        np.savetxt(path, rows, fmt="%.17g")
        flip, (dx, dy), (sx, sy) = int(k % 2), rng.uniform(-0.2, 0.2, 2), rng.uniform(0.8, 1.25, 2)
        assert np.array_equal(im.fill_truth_detection_meta(path, 9, 9, flip, dx, dy, sx, sy),
                              episode.fill_truth_detection_meta(path, 9, 9, flip, dx, dy, sx, sy, base_ids=base_ids, n_cls=3,
                                                                max_boxes=50))


def test_support_mask_and_lr_schedule():
    from fewshot_detection_amd import episode
    (x1, y1, x2, y2), m = episode.support_mask((0.5, 0.5, 0.25, 0.5), 416, 416)
    assert (x1, y1, x2, y2) == (156, 104, 260, 312) and m.shape == (1, 416, 416)
    assert m.sum() == (x2 - x1) * (y2 - y1) and m[0, y1, x1] == 1 and m[0, y1 - 1, x1] == 0
    assert episode.support_mask((0.5, 0.5, 0.0, 0.3), 416, 416)[1] is None           # empty rectangle -> no mask
    assert episode.support_mask((0.99, 0.5, 0.5, 0.5), 100, 100)[0][2] == 100        # clipped to the image
    # train_meta.py:123-163 with cfg/darknet_dynamic.cfg: steps=-1,500,40000,60000 scales=.1,10,.1,.1, lr=0.001
    assert [episode.lr_factor(r, 20) for r in ("full", 1, 0, 5, 3)] == [15.0, 3.0, 1.5, 8.0, 20]
    steps, scales = [-1, 500, 40000, 60000], [0.1, 10, 0.1, 0.1]
    base = 0.001 / 3.0
    def lr(b):
        return episode.adjust_learning_rate(b, base, steps, scales, 64)
    assert np.isclose(lr(0)[0], base * 0.1) and np.isclose(lr(0)[1], base * 0.1 / 64)
    assert np.isclose(lr(499)[0], base * 0.1)
    assert np.isclose(lr(500)[0], base)
    assert np.isclose(lr(39999)[0], base)
    assert np.isclose(lr(40000)[0], base * 0.1)
    assert np.isclose(lr(70000)[0], base * 0.01)
