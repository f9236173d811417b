"""Side streams (fewshot_detection_amd/streams.py): the reweighting net beside the detector, the weight gradients beside
the data-gradient chain (the target upload is an asynchronous copy out of pinned staging, streams.upload).  Kernels are unchanged and deterministic, so a step with
the side streams must be BIT-identical to the same step on one stream -- any difference is a missing dependency."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def cfg_paths(tmp_path_factory):
    from fewshot_detection_amd import cfgs
    return cfgs.write_standard_cfgs(str(tmp_path_factory.mktemp("cfgs")))


def _episode(seed, B, N, S, Sm):
    import bench
    return bench.synth_episode(seed, B, N, S, Sm)


def _run_steps(dev, cfg_paths, enabled, dtype, steps, B=8, N=5, S=224, Sm=128, neg="full", lr=1e-9, early=None, info=None):
    """`steps` train steps from a fixed seed -> (outputs of every step, losses, final flat parameters, flat gradient)."""
    from fewshot_detection_amd import streams
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.darknet_meta import Darknet
    from fewshot_detection_amd.dp import EpisodeTrainer
    old = streams.ENABLED
    streams.ENABLED = enabled
    cfg.neg_ratio = neg
    try:
        torch.manual_seed(3)
        random.seed(3)
        net = Darknet(cfg_paths[0], cfg_paths[1]).to(dev).train()
        net.set_compute_dtype(dtype)
        region = net.models[len(net.models) - 1]
        region.verbose = False
        opt = EpisodeTrainer(net, lr=lr, momentum=0.9, weight_decay=5e-4)        # random init: a real step size diverges
        if early is not None:
            opt.EARLY_STEP = early
        flat0 = opt.flat.detach().clone()
        outs, losses = [], []
        for i in range(steps):
            x, metax, mask, target = _episode(100 + i, B, N, S, Sm)
            out = net(x.to(dev), metax.to(dev), mask.to(dev))
            region.seen += B
            loss = region(out, target)
            outs.append(out.detach().clone())
            losses.append(loss.detach().clone())
            opt.backward_and_step(loss)
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(v)) for v in losses) and bool(torch.isfinite(opt.flat).all())
        if info is not None:
            info["early_steps_last"] = opt.early_steps_last
            info["buckets"] = len(opt.buckets)
            info["moved"] = float((opt.flat - flat0).norm() / flat0.norm())
        return outs, losses, opt.flat.detach().clone(), opt.grad.detach().clone()
    finally:
        streams.ENABLED = old
        cfg.neg_ratio = "full"


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_train_steps_bit_identical_with_side_streams(dev, cfg_paths, dtype):
    ref = _run_steps(dev, cfg_paths, False, dtype, 3)
    for attempt in range(3):                      # a missing dependency is a race: give it a few chances to show
        got = _run_steps(dev, cfg_paths, True, dtype, 3)
        for a, b in zip(ref[0], got[0]):
            assert torch.equal(a, b), "head output differs (attempt %d)" % attempt
        for a, b in zip(ref[1], got[1]):
            assert torch.equal(a, b), "loss differs (attempt %d)" % attempt
        assert torch.equal(ref[3], got[3]), "flat gradient of the last step differs (attempt %d)" % attempt
        assert torch.equal(ref[2], got[2]), "parameters after 3 steps differ (attempt %d)" % attempt


def test_side_streams_with_negative_row_filter(dev, cfg_paths):
    """neg_ratio = 1 consumes python's RNG on the host and uploads a keep map: same draw, same bits."""
    ref = _run_steps(dev, cfg_paths, False, "f32", 2, neg=1)
    got = _run_steps(dev, cfg_paths, True, "f32", 2, neg=1)
    assert all(torch.equal(a, b) for a, b in zip(ref[1], got[1]))
    assert torch.equal(ref[2], got[2])


def test_autograd_without_trainer_and_public_meta_forward(dev, cfg_paths):
    """Plain autograd (.grad tensors, no gradient sink) and the public meta_forward / detect_forward pair: the vectors a
    caller receives from meta_forward are complete on the CURRENT stream (no deferred wait leaks out of the API)."""
    from fewshot_detection_amd import streams
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.darknet_meta import Darknet
    cfg.neg_ratio = "full"
    res = []
    for enabled in (False, True):
        old = streams.ENABLED
        streams.ENABLED = enabled
        try:
            torch.manual_seed(5)
            net = Darknet(cfg_paths[0], cfg_paths[1]).to(dev).train()
            region = net.models[len(net.models) - 1]
            region.verbose = False
            x, metax, mask, target = _episode(7, 4, 3, 160, 96)
            vecs = net.meta_forward(metax.to(dev), mask.to(dev))
            v_host = vecs[0].detach().cpu()                      # read right away on the current stream
            out = net.detect_forward(x.to(dev), vecs)
            loss = region(out, target)
            loss.backward()
            torch.cuda.synchronize()
            grads = [p.grad.detach().clone() for p in net.parameters()]
            res.append((v_host, out.detach().clone(), grads))
        finally:
            streams.ENABLED = old
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a, b)


def test_autotune_keeps_streams_that_pay_and_drops_streams_that_do_not():
    """streams.autotune: a step that is faster with the side streams keeps them; one that is slower gets new streams up to
    `tries` times and then falls back to one stream."""
    import time
    from fewshot_detection_amd import streams
    before = streams.ENABLED
    try:
        streams.ENABLED = True
        rep = streams.autotune(lambda: time.sleep(0.004 if streams.ENABLED else 0.008), tries=2, reps=2)
        assert rep["enabled_after"] is True and streams.ENABLED is True and len(rep["tries"]) == 1
        streams.ENABLED = True
        made = []
        orig = streams.reset
        streams.reset = lambda: (made.append(1), orig())[1]
        try:
            rep = streams.autotune(lambda: time.sleep(0.008 if streams.ENABLED else 0.004), tries=2, reps=2)
        finally:
            streams.reset = orig
        assert rep["enabled_after"] is False and streams.ENABLED is False and len(rep["tries"]) == 2 and len(made) == 2
        # a set that overlaps only a little is traded for a new draw; the better of the sets seen is the one that stays
        dev = torch.device("cuda:0")
        streams.ENABLED = True
        streams.reset()
        seen = []

        def step():
            if streams.ENABLED:
                st = streams.side(dev, "probe")
                if not any(st is t for t in seen):
                    seen.append(st)
                time.sleep(0.0078 if st is seen[0] else 0.0090)        # first set: 2.5 % under one stream; later sets: slower
            else:
                time.sleep(0.008)
        rep = streams.autotune(step, tries=3, reps=2)
        assert rep["enabled_after"] is True and len(rep["tries"]) == 3 and len(seen) == 3
        assert streams.side(dev, "probe") is seen[0]
        assert abs(rep["kept_ms"] - rep["tries"][0]["streams_ms"]) < 1e-9
    finally:
        streams.reset()
        streams.ENABLED = before


def test_upload_through_pinned_staging_is_exact_and_reuses_its_slots():
    """streams.upload: host array -> device tensor by a kernel that reads a pinned staging slot (no memcpy, no side stream).
    More uploads than slots, sizes that grow, odd byte counts, float64 targets of the loss's shape; every one bit-exact, also
    when queued behind pending GPU work."""
    import numpy as np
    from fewshot_detection_amd import streams
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    busy = torch.randn(4096, 4096, device=dev)
    outs, refs = [], []
    for i in range(20):
        (busy @ busy).sum()                                  # pending work on the stream the upload is queued on
        if i % 3 == 0:
            a = rng.standard_normal((960, 250))              # the (B*N, 250) float64 target
        elif i % 3 == 1:
            a = rng.integers(-5, 5, size=(960 + i,), dtype=np.int32)
        else:
            a = rng.integers(0, 255, size=(1001 + 2 * i,), dtype=np.uint8)    # odd byte count
        outs.append(streams.upload(a, dev))
        refs.append(a.copy())
        a[...] = 0                                           # the caller may overwrite its array at once
    e = streams.upload(np.zeros((0, 250)), dev)
    torch.cuda.synchronize()
    assert e.shape == (0, 250)
    for o, r in zip(outs, refs):
        assert o.shape == r.shape and np.array_equal(o.cpu().numpy(), r)
    assert len(streams._PINNED[0]) <= streams._PIN_SLOTS


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_early_optimizer_step_is_bit_identical(dev, cfg_paths, dtype):
    """EpisodeTrainer.EARLY_STEP: the optimizer kernel of a gradient bucket and the in-place re-packing of its conv operands
    are queued on the side stream while the backward pass still runs.  With a step size that really moves the weights, four
    steps (each forward reads the operands the previous step re-packed) must equal the after-the-backward form bit for bit:
    a re-pack ordered before its optimizer kernel, or under a reader of the old copy, shows up in the next head output."""
    ia, ib = {}, {}
    ref = _run_steps(dev, cfg_paths, True, dtype, 4, lr=2e-7, early=False, info=ia)
    assert ia["early_steps_last"] == 0
    for attempt in range(2):
        got = _run_steps(dev, cfg_paths, True, dtype, 4, lr=2e-7, early=True, info=ib)
        assert ib["early_steps_last"] >= ib["buckets"] - 2, ib          # every bucket but the last one or two
        for k, (a, b) in enumerate(zip(ref[0], got[0])):
            assert torch.equal(a, b), "head output of step %d differs (attempt %d)" % (k, attempt)
        assert torch.equal(ref[3], got[3]) and torch.equal(ref[2], got[2])
    assert ia["moved"] > 1e-5, ia           # ... and the weights did move
