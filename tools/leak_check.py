"""Memory stability of the train step: device memory (allocated / reserved) after 20 and after 220 steps."""
import random
import sys
import tempfile

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402
from fewshot_detection_amd import cfgs  # noqa: E402
from fewshot_detection_amd.cfg import cfg  # noqa: E402
from fewshot_detection_amd.darknet_meta import Darknet  # noqa: E402
from fewshot_detection_amd.dp import EpisodeTrainer  # noqa: E402

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


def run(n):
    for _ in range(n):
        region.seen += 64
        opt.backward_and_step(region(net(x, metax, mask), target))
    torch.cuda.synchronize()
    return torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, torch.cuda.max_memory_allocated() / 2**30


a = run(20)
b = run(200)
print("after  20 steps: allocated %.2f GiB reserved %.2f GiB peak %.2f GiB" % a)
print("after 220 steps: allocated %.2f GiB reserved %.2f GiB peak %.2f GiB" % b)
assert b[0] <= a[0] + 0.01 and b[1] <= a[1] + 0.5, "device memory grows with the step count"
print("stable")
