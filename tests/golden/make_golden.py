"""Mint golden vectors by running the REFERENCE ITSELF (see ref_shim.py) on seeded inputs.

Run once in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

Outputs (committed, small):
    boxes.npz            utils.bbox_iou / bbox_ious cases
    dconv.npz            dynamic_conv.dynamic_conv2d(is_first=True) forward
    mini.weights         darknet weight stream written by the reference's save_weights
    mini_forward.npz     darknet_meta.Darknet forward (train + eval), reweighting vectors, BN stats
    mini_yolo.weights / mini_yolo_forward.npz   darknet.Darknet (non-meta twin) forward
    region_v2_*.npz      RegionLossV2 forward + autograd gradient + build_targets tensors
    region_v1.npz        RegionLoss (v1) forward + gradient
    decode_v2.npz        utils.get_region_boxes_v2 + utils.nms
    decode_valid.npz     the call valid_ensemble.py:148 makes: get_region_boxes_v2(.., only_objectness=0, validation=1)
                         at conf 0.005 + utils.nms at 0.45 (dense rows: ~all cells survive the threshold)
    ensemble.npz         valid_ensemble.py:86-100,137-166 on the mini net: running mean of the reweighting vectors over
                         support batches -> detect_forward -> decode -> NMS

    augment.npz          image.data_augmentation (jitter crop, NEAREST resize, flip, HSV distortion) on synthetic images
    utils_host.npz       utils.read_truths(_args) / load_class_names / image2torch / scale_bboxes / is_dict / file_lines /
                         get_image_size / softmax on stored input files
    region_fns.npz       region_loss.build_targets called directly on decoded boxes; region_loss.neg_filter with seeded RNG
    drivers.npz          (drivers_golden.py) the loop body of train_meta.py:201-226 and valid_ensemble.valid() run from the
                         reference's own source on a synthetic tmp dataset

`python make_golden.py NAME...` regenerates only the named fixtures (decode_valid, ensemble, augment, utils_host, region_fns, drivers); no argument = all.

The GPU box has no /root/reference; tests there read only these files.
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ANCH = [1.3221, 1.73145, 3.19275, 4.00944, 5.05587, 8.09892, 9.47112, 4.84053, 11.2364, 10.0071]
ANCH_V1 = [1.08, 1.19, 3.42, 4.41, 6.63, 11.38, 9.42, 5.11, 16.62, 10.52]


def synth_targets(rng, bs, cs, max_per_img=5, same_cell=True):
    """(bs, cs, 250) float64 targets laid out like image.py:144-192 packs them."""
    tgt = np.zeros((bs, cs, 250), np.float64)
    fill = np.zeros((bs, cs), np.int64)
    for b in range(bs):
        for _ in range(rng.randint(1, max_per_img + 1)):
            n = rng.randint(0, cs)
            w, h = rng.uniform(0.05, 0.5, 2)
            cx = float(np.clip(rng.uniform(0.1, 0.9), w / 2, 0.999 - w / 2))
            cy = float(np.clip(rng.uniform(0.1, 0.9), h / 2, 0.999 - h / 2))
            t = fill[b, n]
            tgt[b, n, 5 * t:5 * t + 5] = [n, cx, cy, w, h]
            fill[b, n] += 1
    if same_cell and bs > 1 and cs > 1:
        # two classes of image 0 claim the same (anchor, cell): the class mask must drop it
        tgt[0, 0, :5] = [0, 0.52, 0.52, 0.30, 0.40]
        tgt[0, 1, :5] = [1, 0.53, 0.51, 0.31, 0.41]
        # two boxes of the same row in the same cell: the later one wins
        tgt[1, 1, :10] = [1, 0.30, 0.70, 0.20, 0.25, 1, 0.31, 0.71, 0.21, 0.24]
    return tgt


def gold_boxes(u):
    rng = np.random.RandomState(1)
    a = rng.uniform(0.1, 5, (64, 4))
    b = rng.uniform(0.1, 5, (64, 4))
    a[0] = [1, 1, 2, 2]; b[0] = [5, 5, 1, 1]          # disjoint
    a[1] = [1, 1, 2, 2]; b[1] = [1, 1, 2, 2]          # identical
    a[2] = [1, 1, 2, 2]; b[2] = [2, 1, 2, 2]          # half overlap
    a[3] = [1, 1, 2, 2]; b[3] = [3, 1, 2, 2]          # touching edge -> 0
    sc = np.array([u.bbox_iou(list(map(float, a[i])), list(map(float, b[i])), x1y1x2y2=False)
                   for i in range(64)])
    ta = torch.from_numpy(a.astype(np.float32)).t().contiguous()
    tb = torch.from_numpy(b.astype(np.float32)).t().contiguous()
    vec = u.bbox_ious(ta, tb, x1y1x2y2=False).numpy()
    np.savez_compressed(os.path.join(HERE, "boxes.npz"), a=a, b=b, scalar=sc, vector=vec)


def gold_dconv(dc):
    torch.manual_seed(2)
    x = torch.randn(3, 8, 5, 5)
    w = torch.randn(4, 8, 1, 1)
    layer = dc.dynamic_conv2d(True)(8, 8, 1, 1, 0)
    out = layer((x, w))
    np.savez_compressed(os.path.join(HERE, "dconv.npz"), x=x.numpy(), w=w.numpy(), out=out.detach().numpy())


def gold_mini_forward(dm):
    torch.manual_seed(3)
    net = dm.Darknet(os.path.join(HERE, "mini_dynamic.cfg"), os.path.join(HERE, "mini_reweight.cfg"))
    # non-trivial BN affine + running stats so every field of the weight stream matters
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.3, 0.3)
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    net.seen = 4242
    net.save_weights(os.path.join(HERE, "mini.weights"))
    x = torch.rand(2, 3, 64, 64)
    metax = torch.rand(3, 3, 64, 64)
    mask = torch.zeros(3, 1, 64, 64)
    mask[0, 0, 10:40, 5:30] = 1; mask[1, 0, 0:20, 30:64] = 1; mask[2, 0, 25:60, 20:50] = 1
    net.eval()
    with torch.no_grad():
        dyn_eval = net.meta_forward(metax, mask)[0]
        out_eval = net.detect_forward(x, [dyn_eval])
    net.train()
    dyn_train = net.meta_forward(metax, mask)[0]
    out_train = net.detect_forward(x, [dyn_train])
    g = torch.randn(out_train.shape, generator=torch.Generator().manual_seed(33))
    out_train.backward(g)
    sd = net.state_dict()
    grads = {k: p.grad.numpy() for k, p in net.named_parameters()
             if k in ("models.0.conv1.weight", "models.0.bn1.weight", "models.0.bn1.bias",
                      "models.21.conv14.weight", "models.23.conv16.weight", "models.23.conv16.bias",
                      "learnet_models.0.conv1.weight", "learnet_models.10.conv6.weight",
                      "learnet_models.10.bn6.weight")}
    np.savez_compressed(os.path.join(HERE, "mini_forward.npz"), x=x.numpy(), metax=metax.numpy(), mask=mask.numpy(),
             dyn_eval=dyn_eval.numpy(), out_eval=out_eval.numpy(),
             dyn_train=dyn_train.detach().numpy(), out_train=out_train.detach().numpy(),
             grad_out=g.numpy(),
             bn1_mean_after=sd["models.0.bn1.running_mean"].numpy(),
             bn1_var_after=sd["models.0.bn1.running_var"].numpy(),
             lbn6_mean_after=sd["learnet_models.10.bn6.running_mean"].numpy(),
             lbn6_var_after=sd["learnet_models.10.bn6.running_var"].numpy(),
             **{"grad:" + k: v for k, v in grads.items()})


def gold_mini_yolo(dk):
    torch.manual_seed(4)
    net = dk.Darknet(os.path.join(HERE, "mini_tiny_yolo.cfg"))
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.3, 0.3)
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    net.save_weights(os.path.join(HERE, "mini_yolo.weights"))
    x = torch.rand(2, 3, 64, 64)
    net.eval()
    with torch.no_grad():
        out_eval = net(x)
    net.train()
    out_train = net(x)
    np.savez_compressed(os.path.join(HERE, "mini_yolo_forward.npz"), x=x.numpy(), out_eval=out_eval.numpy(),
             out_train=out_train.detach().numpy())


def _run_loss(rl, module, out, tgt):
    captured = {}
    orig = rl.build_targets

    def spy(*a, **k):
        r = orig(*a, **k)
        captured["bt"] = r
        return r

    rl.build_targets = spy
    try:
        out = out.clone().requires_grad_(True)
        loss = module(out, tgt)
        loss.backward()
    finally:
        rl.build_targets = orig
    names = ["nGT", "nCorrect", "coord_mask", "conf_mask", "cls_mask", "tx", "ty", "tw", "th", "tconf", "tcls"]
    res = {"loss": np.float32(loss.item()), "grad": out.grad.numpy()}
    for n, v in zip(names, captured["bt"]):
        res[n] = v.numpy() if torch.is_tensor(v) else np.int64(v)
    return res


def gold_region_v2(rl, cfgmod):
    rng = np.random.RandomState(5)
    bs, cs, g = 3, 4, 13
    dense = synth_targets(rng, bs, cs)
    torch.manual_seed(5)
    out = torch.randn(bs * cs, 30, g, g) * 0.7
    # make a few predictions overlap their ground truth strongly so the silence rule fires
    out[:, 2::6] *= 0.3
    out[:, 3::6] *= 0.3
    mod = rl.RegionLossV2()
    mod.anchors, mod.num_anchors, mod.num_classes, mod.anchor_step = ANCH, 5, 1, 2
    cases = [("full_seen0", "full", 0), ("full_seen20000", "full", 20000),
             ("neg0_seen20000", 0, 20000), ("neg1_seen20000", 1, 20000)]
    sparse = synth_targets(np.random.RandomState(55), bs, cs, max_per_img=1, same_cell=False)
    for name, neg, seen in cases:
        cfgmod.cfg.neg_ratio = neg
        mod.seen = seen
        random.seed(77)
        tgt = sparse if name.startswith("neg1") else dense   # few positives -> the random drop path runs
        res = _run_loss(rl, mod, out, torch.from_numpy(tgt))
        np.savez_compressed(os.path.join(HERE, "region_v2_%s.npz" % name), output=out.numpy(), target=tgt,
                 neg_ratio=str(neg), seen=seen, py_seed=77, **res)
    cfgmod.cfg.neg_ratio = "full"


def gold_region_v1(rl, cfgmod):
    rng = np.random.RandomState(6)
    bs, nc, g = 2, 3, 13
    tgt = synth_targets(rng, bs, 1, same_cell=False)[:, 0]
    tgt[:, 0::5] = np.where(tgt[:, 1::5] != 0, rng.randint(0, nc, tgt[:, 0::5].shape), 0)
    torch.manual_seed(6)
    out = torch.randn(bs, 5 * (5 + nc), g, g) * 0.7
    mod = rl.RegionLoss(nc, ANCH_V1, 5)
    cfgmod.cfg.neg_ratio = "full"
    for name, meta, seen in [("seen0", False, 0), ("seen20000", False, 20000), ("metayolo", True, 20000)]:
        cfgmod.cfg.metayolo = meta
        mod.seen = seen
        res = _run_loss(rl, mod, out, torch.from_numpy(tgt))
        np.savez_compressed(os.path.join(HERE, "region_v1_%s.npz" % name), output=out.numpy(), target=tgt,
                 metayolo=meta, seen=seen, **res)
    cfgmod.cfg.metayolo = True


def gold_decode(u):
    torch.manual_seed(7)
    bs, cs, g = 2, 3, 13
    out = torch.randn(bs * cs, 30, g, g)
    boxes = u.get_region_boxes_v2(out, cs, 0.3, 1, ANCH, 5)
    flat = []
    for r, bl in enumerate(boxes):
        for bx in bl:
            flat.append([r] + [float(v) for v in bx])
    kept = []
    for r, bl in enumerate(boxes):
        for bx in u.nms(bl, 0.45):
            kept.append([r] + [float(v) for v in bx])
    np.savez_compressed(os.path.join(HERE, "decode_v2.npz"), output=out.numpy(), n_models=cs, conf_thresh=0.3,
             nms_thresh=0.45, boxes=np.array(flat, np.float64), kept=np.array(kept, np.float64))


def gold_decode_extra(u):
    """More shapes for the two decoders: meta (softmax across class rows) and plain YOLOv2 (per-cell softmax)."""
    out = {}
    for k, (bs, cs, g, th) in enumerate([(1, 1, 7, 0.5), (2, 5, 19, 0.5), (1, 20, 13, 0.5)]):
        torch.manual_seed(20 + k)
        o = torch.randn(bs * cs, 30, g, g) * 1.3
        o[:, 4::6] -= 3.0              # ~1 % of the cells survive: the reference NMS is O(n^2) python
        boxes = u.get_region_boxes_v2(o, cs, th, 1, ANCH, 5)
        flat = [[r] + [float(v) for v in bx] for r, bl in enumerate(boxes) for bx in bl]
        kept = [[r] + [float(v) for v in bx] for r, bl in enumerate(boxes) for bx in u.nms(bl, 0.45)]
        out.update({"m%d_output" % k: o.numpy(), "m%d_cfg" % k: np.array([bs, cs, g, th]),
                    "m%d_boxes" % k: np.array(flat, np.float64).reshape(-1, 8),
                    "m%d_kept" % k: np.array(kept, np.float64).reshape(-1, 8)})
    for k, (bs, nc, g, th) in enumerate([(2, 3, 13, 0.5), (1, 20, 7, 0.3)]):
        torch.manual_seed(40 + k)
        o = torch.randn(bs, 5 * (5 + nc), g, g) * 1.3
        o[:, 4::5 + nc] -= 3.0
        boxes = u.get_region_boxes(o, th, nc, ANCH_V1, 5)
        flat = [[r] + [float(v) for v in bx] for r, bl in enumerate(boxes) for bx in bl]
        out.update({"y%d_output" % k: o.numpy(), "y%d_cfg" % k: np.array([bs, nc, g, th]),
                    "y%d_boxes" % k: np.array(flat, np.float64).reshape(-1, 8)})
    np.savez_compressed(os.path.join(HERE, "decode_extra.npz"), **out)


def _flat(rows):
    return np.array([[r] + [float(v) for v in bx] for r, bl in enumerate(rows) for bx in bl], np.float64).reshape(-1, 8)


def gold_decode_valid(u):
    """valid_ensemble.py:148: get_region_boxes_v2(output, n_cls, 0.005, 1, anchors, 5, 0, 1) and :166 nms(boxes, 0.45)."""
    out = {}
    for k, (bs, cs, g, scale, shift) in enumerate([(1, 2, 13, 1.0, 0.0), (1, 2, 19, 1.3, -1.0), (1, 2, 7, 0.02, 0.0)]):
        torch.manual_seed(60 + k)
        o = torch.randn(bs * cs, 30, g, g) * scale
        o[:, 4::6] += shift
        if k == 2:
            # nearly equal objectness everywhere: many boxes share the float32 sort key 1 - det_conf
            o[:, 4::6] = torch.round(o[:, 4::6] * 2e4) / 2e4 + 3.0
        boxes = u.get_region_boxes_v2(o, cs, 0.005, 1, ANCH, 5, 0, 1)
        flat = _flat(boxes)
        kept = _flat([u.nms(bl, 0.45) for bl in boxes])
        out.update({"v%d_output" % k: o.numpy(), "v%d_cfg" % k: np.array([bs, cs, g]),
                    "v%d_boxes" % k: flat, "v%d_kept" % k: kept})
    np.savez_compressed(os.path.join(HERE, "decode_valid.npz"), conf_thresh=0.005, nms_thresh=0.45, **out)


def _load_stream(net, path):
    """Read a darknet weight stream into the REFERENCE model.  (The reference's own load_conv_bn copies a flat buffer
    into a 4-D parameter, cfg.py:455, which torch 0.3.1 allowed and current torch refuses; the field order below is
    the one its save_conv_bn / save_conv wrote the file in, cfg.py:457-481.)"""
    buf = np.fromfile(path, dtype=np.float32)[4:]
    net.header = torch.from_numpy(np.fromfile(path, count=4, dtype=np.int32))     # darknet_meta.py:355-359
    net.seen = net.header[3]
    pos = 0

    def pull(t):
        nonlocal pos
        t.data.copy_(torch.from_numpy(buf[pos:pos + t.numel()]).view_as(t))
        pos += t.numel()

    for blocks, models in ((net.blocks, net.models), (net.learnet_blocks, net.learnet_models)):
        for ind, blk in enumerate(blocks[1:]):
            if blk["type"] != "convolutional":
                continue
            m = models[ind]
            if net.is_dynamic(blk) and m[0].weight is None:
                continue
            if int(blk["batch_normalize"]):
                for t in (m[1].bias, m[1].weight, m[1].running_mean, m[1].running_var, m[0].weight):
                    pull(t)
            else:
                pull(m[0].bias)
                pull(m[0].weight)
    assert pos == buf.size, (pos, buf.size)


def gold_ensemble(dm, u):
    """valid_ensemble.py:86-100 (running mean of the reweighting vectors over all support batches, by class id) and
    :137-166 (detect_forward with the averaged vectors, decode, NMS) on the reduced-width twin of the meta detector."""
    torch.manual_seed(70)
    net = dm.Darknet(os.path.join(HERE, "mini_dynamic.cfg"), os.path.join(HERE, "mini_reweight.cfg"))
    _load_stream(net, os.path.join(HERE, "mini.weights"))
    net.eval()
    n_cls = 3
    clsids = [0, 2, 1, 0, 2, 2, 1, 0, 0]
    metax = torch.rand(len(clsids), 3, 64, 64)
    mask = torch.zeros(len(clsids), 1, 64, 64)
    for i in range(len(clsids)):
        mask[i, 0, 3 * i:30 + 3 * i, 2 * i:25 + 4 * i] = 1
    enews = [0.0] * n_cls
    cnt = [0.0] * n_cls
    with torch.no_grad():
        for lo, hi in ((0, 4), (4, 8), (8, 9)):                      # the loader's batches
            dw = net.meta_forward(metax[lo:hi], mask[lo:hi])[0]
            for ci, c in enumerate(clsids[lo:hi]):
                enews[c] = enews[c] * cnt[c] / (cnt[c] + 1) + dw[ci] / (cnt[c] + 1)
                cnt[c] += 1
        dynamic_weights = [torch.stack(enews)]
        x = torch.rand(2, 3, 160, 160)
        output = net.detect_forward(x, dynamic_weights)
    boxes = u.get_region_boxes_v2(output, n_cls, 0.005, net.num_classes, net.anchors, net.num_anchors, 0, 1)
    flat = _flat(boxes)                                              # before nms zeroes suppressed det_confs in place
    kept = [u.nms(bl, 0.45) for bl in boxes]
    np.savez_compressed(os.path.join(HERE, "ensemble.npz"), metax=metax.numpy(), mask=mask.numpy(), clsids=np.array(clsids),
                        batches=np.array([[0, 4], [4, 8], [8, 9]]), x=x.numpy(), vectors=dynamic_weights[0].numpy(),
                        output=output.numpy(), boxes=flat, kept=_flat(kept))


def gold_augment(im):
    """image.data_augmentation (crop with jitter, NEAREST resize, flip, HSV distortion; image.py:13-87) on synthetic
    images with python's seeded `random`, as dataset.py:240-247 calls it (jitter .2, hue .1, saturation / exposure 1.5)."""
    from PIL import Image
    rng = np.random.RandomState(11)
    out = {}
    cases = [((50, 37), (32, 32)), ((64, 48), (32, 32)), ((33, 70), (48, 48)), ((41, 41), (64, 64)), ((20, 90), (32, 32)),
             ((120, 80), (96, 96))]
    for k, ((ow, oh), shape) in enumerate(cases):
        arr = rng.randint(0, 256, (oh, ow, 3)).astype(np.uint8)
        arr[: oh // 3, : ow // 2] = rng.randint(0, 256, 3)               # flat patches: grey / saturated colours too
        arr[oh // 2:, ow // 2:] = np.array([200, 200, 200])
        random.seed(100 + k)
        img, flip, dx, dy, sx, sy = im.data_augmentation(Image.fromarray(arr), shape, 0.2, 0.1, 1.5, 1.5)
        nxt = random.random()                                            # pins how many draws were consumed
        plain, *_ = im.data_augmentation(Image.fromarray(arr), shape, 0.2, 0.1, 1.5, 1.5, flag=False)
        out.update({"in%d" % k: arr, "shape%d" % k: np.array(shape), "out%d" % k: np.array(img),
                    "plain%d" % k: np.array(plain), "par%d" % k: np.array([flip, dx, dy, sx, sy]),
                    "next_random%d" % k: nxt})
    out["n"] = len(cases)
    np.savez_compressed(os.path.join(HERE, "augment.npz"), **out)


def gold_episode(im):
    """image.fill_truth_detection_meta / fill_truth_detection on random label files (out-of-range boxes, degenerate
    boxes, classes outside the base set, more than 50 boxes)."""
    import tempfile
    rng = np.random.RandomState(7)
    cases = {}
    base_ids = [0, 2, 3, 5, 7, 8, 9, 11, 12, 13, 14, 15, 16, 18, 19]
    im.cfg.base_ids = base_ids
    im.cfg.base_classes = ["c%d" % i for i in base_ids]
    im.cfg.yolo_joint = False
    im.cfg.metaids = []
    tmp = tempfile.mkdtemp()
    for k, nbox in enumerate([0, 1, 7, 30, 80]):
        rows = np.zeros((nbox, 5))
        rows[:, 0] = rng.randint(0, 20, nbox)
        rows[:, 1:3] = rng.uniform(-0.1, 1.1, (nbox, 2))
        rows[:, 3:5] = rng.uniform(0.0005, 0.6, (nbox, 2))
        flip = int(rng.randint(0, 2))
        sx, sy = rng.uniform(0.7, 1.3, 2)
        dx, dy = rng.uniform(-0.2, 0.2, 2)
        path = os.path.join(tmp, "%06d.txt" % k)
        if nbox:
            np.savetxt(path, rows, fmt="%.17g")
        else:
            open(path, "w").close()
        cases["in%d" % k] = rows
        cases["par%d" % k] = np.array([flip, dx, dy, sx, sy])
        cases["meta%d" % k] = im.fill_truth_detection_meta(path, 416, 416, flip, dx, dy, sx, sy)
        cases["det%d" % k] = im.fill_truth_detection(path, 416, 416, flip, dx, dy, sx, sy)
    cases["base_ids"] = np.array(base_ids)
    np.savez_compressed(os.path.join(HERE, "episode.npz"), **cases)


def gold_public_fns(rl, cfgmod):
    """The module-level functions of region_loss.py called on their own: build_targets (:37-132) on decoded boxes, and
    neg_filter (:15-34) with python's seeded RNG."""
    rng = np.random.RandomState(21)
    out = {}
    for k, (rows, g, seen) in enumerate([(6, 13, 0), (6, 13, 20000), (4, 19, 20000), (3, 7, 20000)]):
        tgt = synth_targets(rng, rows, 1)[:, 0]                     # (rows, 250)
        cells = 5 * g * g
        pb = np.zeros((rows * cells, 4), np.float32)
        ii = np.tile(np.arange(g), g * 5 * rows)
        jj = np.tile(np.repeat(np.arange(g), g), 5 * rows)
        pb[:, 0] = ii + rng.uniform(0, 1, rows * cells)
        pb[:, 1] = jj + rng.uniform(0, 1, rows * cells)
        aw = np.tile(np.repeat(np.array(ANCH[0::2]), g * g), rows)
        ah = np.tile(np.repeat(np.array(ANCH[1::2]), g * g), rows)
        pb[:, 2] = aw * np.exp(rng.normal(0, 0.4, rows * cells))
        pb[:, 3] = ah * np.exp(rng.normal(0, 0.4, rows * cells))
        # plant near-perfect predictions on some ground truths so silence / nCorrect fire
        for b in range(rows):
            for t in range(3):
                if tgt[b, t * 5 + 1] == 0:
                    break
                gx, gy, gw, gh = tgt[b, t * 5 + 1] * g, tgt[b, t * 5 + 2] * g, tgt[b, t * 5 + 3] * g, tgt[b, t * 5 + 4] * g
                for a in range(5):
                    if rng.rand() < 0.5:
                        pb[b * cells + a * g * g + int(gy) * g + int(gx)] = [gx, gy, gw * rng.uniform(0.8, 1.2), gh]
        res = rl.build_targets(torch.from_numpy(pb), rl._PyFloatRows(torch.from_numpy(tgt)), ANCH, 5, 1, g, g, 1, 5, 0.6, seen)
        names = ["nGT", "nCorrect", "coord_mask", "conf_mask", "cls_mask", "tx", "ty", "tw", "th", "tconf", "tcls"]
        out.update({"bt%d_pred" % k: pb, "bt%d_target" % k: tgt, "bt%d_cfg" % k: np.array([rows, g, seen])})
        for n, v in zip(names, res):
            out["bt%d_%s" % (k, n)] = v.numpy() if torch.is_tensor(v) else np.int64(v)
    out["bt_n"] = 4
    # neg_filter: (rows, 4) dummy predictions, targets with a few positive rows
    k = 0
    for neg in ["full", 0, 1, 2, 5]:
        for npos in [2, 9]:
            rows = 30
            tgt = np.zeros((rows, 250))
            pos = np.random.RandomState(100 + k).choice(rows, npos, replace=False)
            tgt[pos, 1] = 0.5
            tgt[pos, 2] = 0.5
            tgt[pos, 3] = 0.2
            tgt[pos, 4] = 0.2
            pred = torch.arange(rows * 4, dtype=torch.float32).view(rows, 4)
            cfgmod.cfg.neg_ratio = neg
            random.seed(500 + k)
            p2, t2, inds = rl.neg_filter(pred, torch.from_numpy(tgt), withids=True)
            nxt = random.random()
            out.update({"nf%d_target" % k: tgt, "nf%d_neg" % k: str(neg), "nf%d_inds" % k: np.asarray(inds).reshape(-1),
                        "nf%d_pred" % k: p2.numpy(), "nf%d_next_random" % k: nxt})
            k += 1
    out["nf_n"] = k
    cfgmod.cfg.neg_ratio = "full"
    np.savez_compressed(os.path.join(HERE, "region_fns.npz"), **out)


def gold_utils_host(u):
    """The host helpers of utils.py that dataset.py / valid_ensemble.py import: read_truths(_args) (:373-390),
    load_class_names (:392-399), image2torch (:401-408), scale_bboxes (:477-485), is_dict / file_lines (:488-523),
    get_image_size (:536-569), softmax (:16-19).  The input FILES are stored as byte arrays so the test can recreate
    them where the reference is absent."""
    import io
    import tempfile
    from PIL import Image
    rng = np.random.RandomState(31)
    tmp = tempfile.mkdtemp(prefix="fsdgold_")
    out = {}
    files = {}

    def put(name, data):
        files[name] = np.frombuffer(data, dtype=np.uint8)
        with open(os.path.join(tmp, name), "wb") as fh:
            fh.write(data)
        return os.path.join(tmp, name)

    # label files: empty, one row (loadtxt gives 1-D), many rows with narrow boxes
    labs = {"empty.txt": np.zeros((0, 5)), "one.txt": np.array([[3, .5, .4, .2, .1]]),
            "many.txt": np.column_stack([rng.randint(0, 20, 12), rng.uniform(0, 1, (12, 2)), rng.uniform(0.0005, 0.3, (12, 2))])}
    for name, rows in labs.items():
        buf = io.StringIO()
        if len(rows):
            np.savetxt(buf, rows, fmt="%.17g")
        path = put(name, buf.getvalue().encode())
        out["truths_" + name] = np.asarray(u.read_truths(path), np.float64)
        out["truths_args_" + name] = np.asarray(u.read_truths_args(path, 0.05), np.float64)
    out["truths_missing"] = np.asarray(u.read_truths(os.path.join(tmp, "nope.txt")), np.float64)
    # names file with trailing blanks and an empty line
    path = put("x.names", b"aeroplane\nbicycle  \n\npotted plant\t\nlast")
    out["names"] = np.array(u.load_class_names(path))
    # list files: plain list and "dict" (two fields per line) pointing at lists that share a line
    l1 = put("list1.txt", b"/a/images/1.jpg\n/a/images/2.jpg\n/a/images/3.jpg\n")
    l2 = put("list2.txt", b"/a/images/3.jpg\n/a/images/4.jpg\n")
    dct = put("dict.txt", ("bird %s\ncat %s\n" % ("@TMP@/list1.txt", "@TMP@/list2.txt")).encode())
    with open(dct, "w") as fh:                      # the stored bytes keep the placeholder, the live file has real paths
        fh.write("bird %s\ncat %s\n" % (l1, l2))
    out["is_dict"] = np.array([u.is_dict(l1), u.is_dict(dct)])
    out["file_lines"] = np.array([u.file_lines(l1), u.file_lines(l2), u.file_lines(dct), u._file_lines(dct)])
    # images: PNG, GIF, JPEG (baseline and progressive), a truncated file and a text file
    arr = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    sizes = []
    for name, fmt, kw in [("a.png", "PNG", {}), ("b.gif", "GIF", {}), ("c.jpg", "JPEG", {}),
                          ("d.jpg", "JPEG", {"progressive": True, "quality": 60})]:
        bio = io.BytesIO()
        Image.fromarray(arr[:, : 53 - 7 * len(sizes)]).save(bio, fmt, **kw)
        sizes.append(u.get_image_size(put(name, bio.getvalue())))
    out["image_sizes"] = np.array(sizes)
    out["image_size_short"] = np.array([u.get_image_size(put("short.png", b"\x89PNG\r\n")) is None,
                                        u.get_image_size(put("text.jpg", b"this is not an image, just thirty bytes")) is None])
    img = Image.fromarray(arr)
    out["image2torch_in"] = arr
    out["image2torch"] = u.image2torch(img).numpy()
    boxes = [[.5, .25, .1, .2, .9, .8, 3], [.1, .9, .3, .4, .5, .6, 1]]
    out["scale_bboxes"] = np.array(u.scale_bboxes(boxes, 640, 480), np.float64)
    out["scale_bboxes_in"] = np.array(boxes, np.float64)
    x = torch.from_numpy(rng.normal(0, 3, (4, 5)).astype(np.float32))
    out["softmax_in"], out["softmax"] = x.numpy(), u.softmax(x).numpy()
    for name, data in files.items():
        out["file_" + name] = data
    np.savez_compressed(os.path.join(HERE, "utils_host.npz"), **out)


def main():
    assert ref_shim.available(), "needs /root/reference"
    only = set(sys.argv[1:])
    if only:
        u = ref_shim.load("utils")
        if "decode_valid" in only:
            gold_decode_valid(u)
        if "ensemble" in only:
            gold_ensemble(ref_shim.load("darknet_meta"), u)
        if "augment" in only:
            gold_augment(ref_shim.load("image"))
        if "utils_host" in only:
            gold_utils_host(u)
        if "region_fns" in only:
            gold_public_fns(ref_shim.load("region_loss"), ref_shim.load("cfg"))
        if "drivers" in only:
            import drivers_golden
            drivers_golden.mint()
        print("golden vectors written to", HERE, sorted(only))
        return
    u = ref_shim.load("utils")
    cfgmod = ref_shim.load("cfg")
    rl = ref_shim.load("region_loss")
    dc = ref_shim.load("dynamic_conv")
    dm = ref_shim.load("darknet_meta")
    dk = ref_shim.load("darknet")
    gold_boxes(u)
    gold_dconv(dc)
    gold_mini_forward(dm)
    gold_mini_yolo(dk)
    gold_region_v2(rl, cfgmod)
    gold_region_v1(rl, cfgmod)
    gold_decode(u)
    gold_decode_extra(u)
    gold_decode_valid(u)
    gold_ensemble(dm, u)
    gold_episode(ref_shim.load("image"))
    gold_augment(ref_shim.load("image"))
    gold_utils_host(u)
    gold_public_fns(rl, cfgmod)
    import drivers_golden
    drivers_golden.mint()
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
