"""Host-side enqueue time of one train step per mode (fp32 / bf16, side streams on / off) against its GPU time: is a mode
launch-bound?  The first two steps after a device synchronise are timed on the host (nothing blocks them: the trainer keeps two
steps in flight), then ten steps end to end.   python tools/probes/enqueue_modes.py"""
import random
import sys
import tempfile
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
import bench  # noqa: E402
from fewshot_detection_amd import cfgs, streams  # noqa: E402
from fewshot_detection_amd.cfg import cfg  # noqa: E402
from fewshot_detection_amd.darknet_meta import Darknet  # noqa: E402
from fewshot_detection_amd.dp import EpisodeTrainer  # noqa: E402

cfg.neg_ratio = 1
dev = torch.device("cuda:0")
dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(tempfile.mkdtemp())
for dtype in ("f32", "bf16"):
    for on in (True, False):
        streams.ENABLED = on
        torch.manual_seed(0)
        random.seed(0)
        net = Darknet(dyn_cfg, rw_cfg).to(dev).train().set_compute_dtype(dtype)
        region = net.models[len(net.models) - 1]
        region.verbose = False
        x, metax, mask, target = bench.synth_episode(1000, 64, 20, 416, 224)
        x, metax, mask = x.to(dev), metax.to(dev), mask.to(dev)
        opt = EpisodeTrainer(net, lr=1e-9, momentum=0.9, weight_decay=0.0)

        def step():
            region.seen += 64
            opt.backward_and_step(region(net(x, metax, mask), target))
        for _ in range(6):
            step()
        enq = []
        for _ in range(4):
            torch.cuda.synchronize()
            a = time.perf_counter()
            step()
            b = time.perf_counter()
            step()
            c = time.perf_counter()
            enq += [(b - a) * 1e3, (c - b) * 1e3]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        print("%s streams=%d: host enqueue per step %s ms (median %.1f); step %.2f ms" %
              (dtype, on, [round(v, 1) for v in enq], sorted(enq)[len(enq) // 2], ms), flush=True)
        del net, opt
        torch.cuda.empty_cache()
