"""Image side of the episode input pipeline on the MI355X: fsd_augment_batch == the reference's
image.data_augmentation + ToTensor, bit for bit (tests/golden/augment.npz, minted from the reference on Pillow with the
2018 defaults), on both output layouts; random batches against the oracle; and the channels-last output feeds the
network without a layout pass."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _u8(t_nchw):
    """float NCHW in [0,1] -> (H, W, 3) uint8, requiring every value to be EXACTLY k/255 in float32."""
    a = t_nchw.cpu().numpy().transpose(1, 2, 0)
    k = np.rint(a * 255.0).astype(np.int64)
    assert np.array_equal((k.astype(np.float32) / np.float32(255.0)), a)
    return k.astype(np.uint8)


def test_device_pipeline_equals_reference_goldens(dev):
    from fewshot_detection_amd import episode as E
    d = np.load(os.path.join(GOLD, "augment.npz"))
    aug = E.DeviceAugmenter(dev)
    by_shape = {}
    for k in range(int(d["n"])):
        by_shape.setdefault(tuple(int(v) for v in d["shape%d" % k]), []).append(k)
    for shape, ks in by_shape.items():                       # images of different sizes share one launch
        imgs, params = [], []
        for k in ks:
            arr = d["in%d" % k]
            random.seed(100 + k)
            p = E.draw_augmentation(arr.shape[1], arr.shape[0])
            assert np.array_equal([p["flip"], p["dx"], p["dy"], p["sx"], p["sy"]], d["par%d" % k])
            imgs.append(arr)
            params.append(p)
        out = aug(imgs, params, shape)
        out4 = aug(imgs, params, shape, layout="nhwc4")
        plain = aug(imgs, [None] * len(ks), shape)
        assert out.shape == (len(ks), 3, shape[1], shape[0]) and out.is_contiguous()
        assert out4.shape == (len(ks), 4, shape[1], shape[0]) and out4.is_contiguous(memory_format=torch.channels_last)
        for i, k in enumerate(ks):
            assert np.array_equal(_u8(out[i]), d["out%d" % k]), k
            assert np.array_equal(_u8(plain[i]), d["plain%d" % k]), k
        assert torch.equal(out4[:, :3].contiguous(), out) and float(out4[:, 3].abs().max()) == 0.0


@pytest.mark.parametrize("seed", range(3))
def test_device_pipeline_random_batches_vs_oracle(dev, seed):
    from fewshot_detection_amd import episode as E
    from oracle import augment as A
    rng = np.random.RandomState(seed)
    random.seed(seed)
    shape = [(416, 416), (224, 224), (96, 160)][seed]
    imgs, params, want, boxes = [], [], [], []
    for i in range(12):
        ow, oh = int(rng.randint(20, 700)), int(rng.randint(20, 500))
        arr = rng.randint(0, 256, (oh, ow, 3)).astype(np.uint8)
        if i % 3 == 0:
            arr[:] = rng.randint(0, 256, 3)                                 # a flat colour (grey when all equal)
        p = E.draw_augmentation(ow, oh, jitter=[0.2, 0.45, 0.2][seed]) if i != 5 else None
        imgs.append(arr)
        params.append(p)
        want.append(A.augment(arr, p, shape) if p is not None else A.resize_only(arr, shape))
        x1, y1 = int(rng.randint(0, shape[0] // 2)), int(rng.randint(0, shape[1] // 2))
        boxes.append((x1, y1, x1 + int(rng.randint(1, shape[0] // 2)), y1 + int(rng.randint(1, shape[1] // 2))))
    aug = E.DeviceAugmenter(dev)
    out = aug(imgs, params, shape)
    for i in range(len(imgs)):
        assert np.array_equal(_u8(out[i]), want[i]), i
    out4 = aug(imgs, params, shape, layout="nhwc4", mask_boxes=boxes)
    assert torch.equal(out4[:, :3].contiguous(), out)
    m = out4[:, 3].cpu().numpy()
    for i, (x1, y1, x2, y2) in enumerate(boxes):
        ref = np.zeros((shape[1], shape[0]), np.float32)
        ref[y1:y2, x1:x2] = 1
        assert np.array_equal(m[i], ref)


def test_channels_last_input_feeds_the_network_in_place(dev):
    """(B, 4, S, S) channels_last = 16-byte NHWC4 pixels: the detector / reweighting net consume it without a layout
    pass and give the same result as the (x, metax, mask) NCHW call."""
    from fewshot_detection_amd.darknet_meta import Darknet
    d = np.load(os.path.join(GOLD, "mini_forward.npz"))
    net = Darknet(os.path.join(GOLD, "mini_dynamic.cfg"), os.path.join(GOLD, "mini_reweight.cfg"))
    net.load_weights(os.path.join(GOLD, "mini.weights"))
    net = net.to(dev).eval()
    x, metax, mask = (torch.from_numpy(d[k]).to(dev) for k in ("x", "metax", "mask"))
    with torch.no_grad():
        ref = net(x, metax, mask)
        x4 = torch.cat([x, torch.zeros_like(x[:, :1])], 1).contiguous(memory_format=torch.channels_last)
        m4 = torch.cat([metax, mask], 1).contiguous(memory_format=torch.channels_last)
        got = net(x4, m4, None)
    assert torch.equal(got, ref)
    net.train()
    out = net(x4, m4, None)                                        # and the backward pass runs from it as well
    out.sum().backward()
    assert net.models[0][0].weight.grad is not None and torch.isfinite(net.models[0][0].weight.grad).all()
