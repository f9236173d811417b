"""Per-step wall time (synchronised) of the C2 train step: shows warm-up transients.  python tools/step_times.py [steps]"""
import random
import sys
import tempfile
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402
from fewshot_detection_amd import cfgs  # noqa: E402
from fewshot_detection_amd.cfg import cfg  # noqa: E402
from fewshot_detection_amd.darknet_meta import Darknet  # noqa: E402
from fewshot_detection_amd.dp import EpisodeTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg.neg_ratio = 1
dev = torch.device("cuda:0")
dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(tempfile.mkdtemp())
torch.manual_seed(0)
random.seed(0)
net = Darknet(dyn_cfg, rw_cfg).to(dev).train()
region = net.models[len(net.models) - 1]
region.verbose = False
x, metax, mask, target = bench.synth_episode(1000, 64, 15, 416, 416)
x, metax, mask = x.to(dev), metax.to(dev), mask.to(dev)
opt = EpisodeTrainer(net, lr=1e-9, momentum=0.9, weight_decay=0.0)
ts = []
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    region.seen += 64
    opt.backward_and_step(region(net(x, metax, mask), target))
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("ms per step:", " ".join("%.1f" % t for t in ts))
