"""Mint tests/golden/drivers.npz by running the REFERENCE'S OWN DRIVER SOURCE on a synthetic on-disk dataset:

  * train_meta.py as a whole script (argument parsing, cfg plumbing, dataset.listDataset / MetaDataset with the real
    image.py augmentation, SGD set-up, LR schedule, the loop body :201-226) for a few base-training steps, and
  * valid_ensemble.valid() (:13-178: support-set ensembling over dataset.MetaDataset(ensemble=True), detect_forward,
    validation-mode decode, NMS, comp4_det_test_<class>.txt result files).

TEST INFRASTRUCTURE ONLY.  The sources are exec'd where they lie with the textual py2 -> py3 / torch-0.3 -> 2.x
substitutions listed below (and ref_shim's for the modules they import); nothing of them is stored.  What IS stored: the
batches the reference's loaders produced (uint8 images, masks, float64 targets), the loss of every step, the parameters
after training, the head output of the first step, and the result files valid() wrote -- so that tests/test_gpu_drivers.py
can replay the same steps through the py3 spelling of the two loops (tests/drivers_py3.py) against the MI355X modules
on a box that has no /root/reference.

The reference has no GPU here: `.cuda()` is substituted away and everything runs on PyTorch-CPU.
"""
import io
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

VOC = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog", "horse",
       "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]

_TRAIN_SUBS = [
    ("from models.tiny_yolo import TinyYoloNet", ""),
    ("seed          = int(time.time())", "seed          = 1234"),
    ("metaloader.next()", "next(metaloader)"),
    ("model.seen/batch_size", "model.seen//batch_size"),
    ("model.seen/nsamples", "model.seen//nsamples"),
    ("max_batches*batch_size/nsamples+1", "max_batches*batch_size//nsamples+1"),
    # the reference's load_conv_bn copies a flat buffer into a 4-D parameter (cfg.py:455): torch 0.3.1 allowed it
    ("model.load_weights(weightfile)", "_load_stream(model, weightfile)"),
    ("'pin_memory': True", "'pin_memory': False"),
    ("pin_memory=True", "pin_memory=False"),
]
_VALID_SUBS = [
    ("m.load_weights(weightfile)", "_load_stream(m, weightfile)"),
    ("kwargs = {'num_workers': 4, 'pin_memory': True}", "kwargs = {'num_workers': 0}"),
    ("Variable(metax, volatile=True), Variable(mask, volatile=True)", "Variable(metax), Variable(mask)"),
    ("data = Variable(data, volatile = True)", "data = Variable(data)"),
    ("for j in range((len(box)-5)/2):", "for j in range((len(box)-5)//2):"),
]


def smooth_image(rng, w, h):
    """Low-frequency colour field + a few flat rectangles (compresses well, still exercises the HSV distortion)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for c in range(3):
        fx, fy, ph = rng.uniform(0.02, 0.09), rng.uniform(0.02, 0.09), rng.uniform(0, 6.28)
        img[:, :, c] = 127 + 110 * np.sin(fx * xx + fy * yy + ph)
    for _ in range(3):
        x0, y0 = rng.randint(0, w - 10), rng.randint(0, h - 10)
        img[y0:y0 + rng.randint(8, 30), x0:x0 + rng.randint(8, 30)] = rng.randint(0, 256, 3)
    return np.clip(img, 0, 255).astype(np.uint8)


def build_dataset(root, n_images=10, seed=3):
    """VOC-shaped tree: images/, labels/ (all classes), labels_1c/<class>/ (one class), list files and the metadict."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "VOC", "images"))
    os.makedirs(os.path.join(root, "VOC", "labels"))
    for c in VOC:
        os.makedirs(os.path.join(root, "VOC", "labels_1c", c))
    paths, per_class = [], {c: [] for c in VOC}
    for i in range(n_images):
        w, h = int(rng.randint(90, 150)), int(rng.randint(70, 130))
        name = "%06d" % (i + 1)
        ipath = os.path.join(root, "VOC", "images", name + ".png")
        Image.fromarray(smooth_image(rng, w, h)).save(ipath)
        paths.append(ipath)
        # every class appears in images i with (i + class) % 5 in {0, 1} (so every class has >= 2 images), 1-2 boxes each
        rows = []
        for ci, c in enumerate(VOC):
            own = []
            if (i + ci) % 5 in (0, 1):
                for _ in range(1 + (i + ci) % 2):
                    bw, bh = rng.uniform(0.15, 0.6), rng.uniform(0.15, 0.6)
                    cx, cy = rng.uniform(bw / 2, 1 - bw / 2), rng.uniform(bh / 2, 1 - bh / 2)
                    own.append([ci, cx, cy, bw, bh])
                per_class[c].append(ipath)
            with open(os.path.join(root, "VOC", "labels_1c", c, name + ".txt"), "w") as fh:
                for r in own:
                    fh.write("0 %.6f %.6f %.6f %.6f\n" % tuple(r[1:]))
            rows += own
        with open(os.path.join(root, "VOC", "labels", name + ".txt"), "w") as fh:
            for r in rows:
                fh.write("%d %.6f %.6f %.6f %.6f\n" % tuple(r))
    with open(os.path.join(root, "train.txt"), "w") as fh:
        fh.write("".join(p + "\n" for p in paths[:8]))
    with open(os.path.join(root, "valid.txt"), "w") as fh:
        fh.write("".join(p + "\n" for p in paths[5:10]))
    with open(os.path.join(root, "metadict_2shot.txt"), "w") as fh:
        for c in VOC:
            lst = os.path.join(root, "meta_%s.txt" % c)
            with open(lst, "w") as g:
                g.write("".join(p + "\n" for p in per_class[c][:2]))
            fh.write("%s %s\n" % (c, lst))
    with open(os.path.join(root, "novels.txt"), "w") as fh:
        fh.write("bird,bus,cow,motorbike,sofa\n")
    common = ("metayolo=1\nmetain_type=2\ndata=voc\nrand = 0\nnovel = %s\nnovelid = 0\nscale = 0\nnum_workers = 0\n"
              "meta = %s\ntrain = %s\nvalid = %s\ngpus=0\n" % (os.path.join(root, "novels.txt"),
                                                              os.path.join(root, "metadict_2shot.txt"),
                                                              os.path.join(root, "train.txt"), os.path.join(root, "valid.txt")))
    with open(os.path.join(root, "base.data"), "w") as fh:
        fh.write(common + "neg = 1\nbackup = backup/metayolo\n")
    with open(os.path.join(root, "tune.data"), "w") as fh:
        fh.write(common + "neg = 0\ntuning = 1\nmax_epoch = 2\nrepeat = 1\ndynamic = 0\nbackup = backup/metatune\n")
    # a net cfg that trains for two epochs of two batches; mini.weights carries seen = 4242 -> 1060 processed batches
    src = open(os.path.join(HERE, "mini_dynamic.cfg")).read()
    assert "batch=4" in src
    with open(os.path.join(root, "net.cfg"), "w") as fh:
        fh.write(src.replace("momentum=0.9", "momentum=0.9\nmax_batches=1062\nsteps=-1,1061,1063\nscales=.1,10,.1"))
    shutil.copy(os.path.join(HERE, "mini_reweight.cfg"), os.path.join(root, "learnet.cfg"))
    os.makedirs(os.path.join(root, "backup", "w"))
    shutil.copy(os.path.join(HERE, "mini.weights"), os.path.join(root, "backup", "w", "mini.weights"))
    return paths


def _exec_driver(name, subs, argv, extra):
    """exec REF/<name>.py with ref_shim's common substitutions + `subs`, the shimmed reference modules served under their
    bare names, and `extra` pre-seeded in the namespace."""
    ref_shim._torchvision_stub()
    src = open(os.path.join(ref_shim.REF, name + ".py")).read()
    for a, b in ref_shim._SUBS_COMMON + subs:
        assert a in src or (a, b) in ref_shim._SUBS_COMMON, (name, a)
        src = src.replace(a, b)
    mods = {m: ref_shim.load(m) for m in ("utils", "cfg", "region_loss", "dynamic_conv", "pooling", "darknet_meta", "image",
                                          "dataset")}
    saved = {m: sys.modules.get(m) for m in mods}
    saved_argv = sys.argv
    sys.modules.update(mods)
    sys.argv = argv
    ns = {"__name__": "ref_" + name, "__file__": os.path.join(ref_shim.REF, name + ".py")}
    ns.update(extra)
    try:
        exec(compile(src, ns["__file__"], "exec"), ns)
    finally:
        sys.argv = saved_argv
        for m, v in saved.items():
            if v is None:
                sys.modules.pop(m, None)
            else:
                sys.modules[m] = v
    return ns, mods


def _u8(t):
    a = (t.detach().cpu().numpy() * 255.0).round()
    assert np.abs(a / 255.0 - t.detach().cpu().numpy()).max() < 1e-6
    return a.astype(np.uint8)


def mint():
    from make_golden import _load_stream
    assert ref_shim.available()
    root = tempfile.mkdtemp(prefix="fsdgold_drv_")
    build_dataset(root)
    cwd = os.getcwd()
    out = {}
    os.chdir(root)
    try:
        # ---------------- train_meta.py, whole script, base training -------------------------------------------------
        dm = ref_shim.load("darknet_meta")
        rl = ref_shim.load("region_loss")
        rec = {"steps": []}
        orig_fwd, orig_loss = dm.Darknet.forward, rl.RegionLossV2.forward

        def spy_fwd(self, x, metax, mask, ids=None):
            o = orig_fwd(self, x, metax, mask, ids)
            if self.training:
                rec["steps"].append({"data": _u8(x), "metax": _u8(metax), "mask": _u8(mask), "output": o.detach().numpy().copy()})
            return o

        def spy_loss(self, output, target):
            step = rec["steps"][-1]
            random.seed(9000 + len(rec["steps"]))            # neg_filter's draws: same stream for the replay
            loss = orig_loss(self, output, target)
            step.update(target=target.detach().numpy().copy(), loss=float(loss.item()), seen=int(self.seen))
            return loss

        dm.Darknet.forward, rl.RegionLossV2.forward = spy_fwd, spy_loss
        random.seed(11)
        np.random.seed(11)
        try:
            ns, mods = _exec_driver("train_meta", _TRAIN_SUBS,
                                    ["train_meta.py", "base.data", "net.cfg", "learnet.cfg", "backup/w/mini.weights"],
                                    {"_load_stream": _load_stream})
        finally:
            dm.Darknet.forward, rl.RegionLossV2.forward = orig_fwd, orig_loss
        model = ns["model"]
        steps = rec["steps"]
        assert len(steps) == 4, len(steps)
        out["train_n"] = len(steps)
        for i, s in enumerate(steps):
            for k, v in s.items():
                out["train%d_%s" % (i, k)] = np.asarray(v)
        opt = ns["optimizer"].param_groups[0]
        out["train_hparams"] = np.array([opt["lr"], opt["momentum"], opt["weight_decay"], ns["batch_size"], ns["learning_rate"]])
        out["train_lrs"] = np.array([ns["adjust_learning_rate"](ns["optimizer"], b) for b in range(1059, 1066)])
        out["train_processed_batches"] = ns["processed_batches"]
        for k, v in model.state_dict().items():
            out["train_final/" + k] = v.numpy().copy()
        out["train_cfg"] = np.array([str(mods["cfg"].cfg.neg_ratio), str(mods["cfg"].cfg.backup)])
        # ---------------- valid_ensemble.valid() on the tuning data cfg ----------------------------------------------
        cfgm = ref_shim.load("cfg")
        ds = ref_shim.load("dataset")
        u = ref_shim.load("utils")
        darknet = cfgm.parse_cfg("net.cfg")
        learnet = cfgm.parse_cfg("learnet.cfg")
        data_options = u.read_data_cfg("tune.data")
        data_options["gpus"] = "0"
        cfgm.cfg.config_data(data_options)
        cfgm.cfg.config_meta(learnet[0])
        cfgm.cfg.config_net(darknet[0])
        vrec = {"meta": [], "data": []}
        orig_meta, orig_det = dm.Darknet.meta_forward, dm.Darknet.detect_forward

        def spy_meta(self, metax, mask):
            vrec["meta"].append((_u8(metax), _u8(mask)))
            return orig_meta(self, metax, mask)

        def spy_det(self, x, dws):
            o = orig_det(self, x, dws)
            vrec["data"].append((_u8(x), o.detach().numpy().copy()))
            vrec["vectors"] = dws[0].detach().numpy().copy()
            return o

        class _IdLoader(torch.utils.data.DataLoader):        # records the class ids the meta loader hands out
            def __iter__(self):
                for batch in super(_IdLoader, self).__iter__():
                    if len(batch) == 3:
                        vrec.setdefault("clsids", []).append(np.asarray(batch[2]))
                    yield batch

        dm.Darknet.meta_forward, dm.Darknet.detect_forward = spy_meta, spy_det
        real_loader = torch.utils.data.DataLoader
        torch.utils.data.DataLoader = _IdLoader
        try:
            vns, _ = _exec_driver("valid_ensemble", _VALID_SUBS, ["valid_ensemble.py"], {"_load_stream": _load_stream})
            with torch.no_grad():
                vns["valid"]("tune.data", darknet, learnet, "backup/w/mini.weights", "comp4_det_test_", False)
        finally:
            dm.Darknet.meta_forward, dm.Darknet.detect_forward = orig_meta, orig_det
            torch.utils.data.DataLoader = real_loader
        res_dir = os.path.join(root, "results", "w", "enemini")
        files = sorted(os.listdir(res_dir))
        assert files == sorted("comp4_det_test_%s.txt" % c for c in VOC), files
        for c in VOC:
            rows = [ln.split() for ln in open(os.path.join(res_dir, "comp4_det_test_%s.txt" % c))]
            out["valid_ids/" + c] = np.array([r[0] for r in rows])
            out["valid_rows/" + c] = np.array([[float(v) for v in r[1:]] for r in rows], np.float64).reshape(-1, 5)
        out["valid_n_meta"] = len(vrec["meta"])
        for i, (mx, mk) in enumerate(vrec["meta"]):
            out["valid_metax%d" % i], out["valid_mask%d" % i] = mx, mk
            out["valid_clsids%d" % i] = vrec["clsids"][i]
        out["valid_n_data"] = len(vrec["data"])
        for i, (x, o) in enumerate(vrec["data"]):
            out["valid_data%d" % i], out["valid_output%d" % i] = x, o
        out["valid_vectors"] = vrec["vectors"]
        lines = [ln.rstrip() for ln in open(os.path.join(root, "valid.txt"))]
        out["valid_lines"] = np.array([os.path.relpath(p, root) for p in lines])
        out["valid_sizes"] = np.array([u.get_image_size(p) for p in lines])
        out["valid_image_files"] = np.array([os.path.basename(p) for p in lines])
        for p in lines:                                       # the PNG headers the replay's get_image_size reads
            out["valid_png/" + os.path.basename(p)] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
        out["net_cfg"] = np.frombuffer(open(os.path.join(root, "net.cfg"), "rb").read(), dtype=np.uint8)
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "drivers.npz"), **out)
    shutil.rmtree(root, ignore_errors=True)
    print("drivers.npz:", os.path.getsize(os.path.join(HERE, "drivers.npz")) // 1024, "KiB;", len(steps), "train steps,",
          len(vrec["data"]), "valid batches")


if __name__ == "__main__":
    mint()
