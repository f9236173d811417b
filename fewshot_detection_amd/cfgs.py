"""Generators for the standard network definitions of the path (darknet .cfg text).

The GPU box has no copy of the reference tree, so bench.py / smoke() / the tests build the
architectures from these compact layer tables; tests/test_cfgs.py checks (in the build container)
that parse_cfg of the generated text equals parse_cfg of the reference's cfg/*.cfg files block by
block (cfg/darknet_dynamic.cfg, cfg/reweighting_net.cfg, cfg/tiny-yolo-voc.cfg).
"""
import os

_NET_COMMON = [("momentum", "0.9"), ("decay", "0.0005"), ("angle", "0"), ("saturation", "1.5"),
               ("exposure", "1.5"), ("hue", ".1"), ("learning_rate", "0.001")]

ANCHORS_META = "1.3221, 1.73145, 3.19275, 4.00944, 5.05587, 8.09892, 9.47112, 4.84053, 11.2364, 10.0071"
ANCHORS_TINY_VOC = "1.08,1.19,  3.42,4.41,  6.63,11.38,  9.42,5.11,  16.62,10.52"


def _block(name, items):
    return "[%s]\n%s\n" % (name, "\n".join("%s=%s" % kv for kv in items))


def _conv(filters, size, bn=True, act="leaky", dynamic=False):
    items = []
    if dynamic:
        items.append(("dynamic", "1"))
    if bn or dynamic:
        items.append(("batch_normalize", "1" if bn else "0"))
    items += [("filters", str(filters)), ("size", str(size)), ("stride", "1"), ("pad", "1"),
              ("activation", act)]
    return _block("convolutional", items)


def _pool(stride=2):
    return _block("maxpool", [("size", "2"), ("stride", str(stride))])


def _region(anchors, classes, jitter):
    return _block("region", [("anchors", anchors), ("bias_match", "1"), ("classes", str(classes)),
                             ("coords", "4"), ("num", "5"), ("softmax", "1"), ("jitter", jitter),
                             ("rescore", "1"), ("object_scale", "5"), ("noobject_scale", "1"),
                             ("class_scale", "1"), ("coord_scale", "1"), ("absolute", "1"),
                             ("thresh", ".6"), ("random", "1")])


def darknet_dynamic(height=416, width=416):
    """Darknet-19 backbone + passthrough + dynamic (reweighting) conv + 1x1 head, 5 anchors, classes=1."""
    net = _block("net", [("batch", "64"), ("subdivisions", "8"), ("height", str(height)), ("width", str(width)),
                         ("channels", "3")] + _NET_COMMON +
                 [("burn_in", "1000"), ("max_batches", "80200"), ("policy", "steps"),
                  ("steps", "-1,500,40000,60000"), ("scales", "0.1,10,.1,.1")])
    body = [_conv(32, 3), _pool(), _conv(64, 3), _pool(),
            _conv(128, 3), _conv(64, 1), _conv(128, 3), _pool(),
            _conv(256, 3), _conv(128, 1), _conv(256, 3), _pool(),
            _conv(512, 3), _conv(256, 1), _conv(512, 3), _conv(256, 1), _conv(512, 3), _pool(),
            _conv(1024, 3), _conv(512, 1), _conv(1024, 3), _conv(512, 1), _conv(1024, 3),
            _conv(1024, 3), _conv(1024, 3),
            _block("route", [("layers", "-9")]), _conv(64, 1), _block("reorg", [("stride", "2")]),
            _block("route", [("layers", "-1,-4")]),
            _conv(1024, 3), _conv(1024, 1, bn=False, act="linear", dynamic=True),
            _conv(30, 1, bn=False, act="linear"), _region(ANCHORS_META, 1, ".3")]
    return net + "".join(body)


def reweighting_net(height=416, width=416, channels=4):
    """Support branch: 6 x (3x3 conv, BN, leaky, maxpool), one more conv, global max -> 1024 weights."""
    net = _block("learnet", [("feat_layer", "0"), ("channels", str(channels)), ("height", str(height)),
                             ("width", str(width))])
    body = []
    for f in (32, 64, 128, 256, 512, 1024):
        body += [_conv(f, 3), _pool()]
    body += [_conv(1024, 3), "[globalmax]\n"]
    return net + "".join(body)


def tiny_yolo_voc(height=416, width=416, classes=20):
    net = _block("net", [("batch", "64"), ("subdivisions", "8"), ("width", str(width)), ("height", str(height)),
                         ("channels", "3")] + _NET_COMMON +
                 [("max_batches", "40200"), ("policy", "steps"), ("steps", "-1,100,20000,30000"),
                  ("scales", ".1,10,.1,.1")])
    body = []
    for f in (16, 32, 64, 128, 256):
        body += [_conv(f, 3), _pool()]
    body += [_conv(512, 3), _pool(1), _conv(1024, 3), _conv(1024, 3),
             _conv(5 * (5 + classes), 1, bn=False, act="linear"), _region(ANCHORS_TINY_VOC, classes, ".2")]
    return net + "".join(body)


def write_standard_cfgs(directory):
    """Write the three cfg files; returns their paths (dynamic, reweighting, tiny_yolo)."""
    os.makedirs(directory, exist_ok=True)
    out = []
    for name, text in (("darknet_dynamic.cfg", darknet_dynamic()), ("reweighting_net.cfg", reweighting_net()),
                       ("tiny-yolo-voc.cfg", tiny_yolo_voc())):
        p = os.path.join(directory, name)
        with open(p, "w") as fh:
            fh.write(text)
        out.append(p)
    return out
