"""Probe: does a process that starts slow (a fresh box, the first GPU process of a call) stay slow?  Runs the headline train step in
blocks of 20 pipelined steps for ~15 s and prints ms/step per block, with the one-stream step time every fifth block.
python tools/probes/slow_start.py [blocks]"""
import os
import sys
import time
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import bench  # noqa: E402
from fewshot_detection_amd import cfgs, streams  # noqa: E402
from fewshot_detection_amd.cfg import cfg  # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cfg.neg_ratio = 1
dev = torch.device("cuda:0")
dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(tempfile.mkdtemp())
t_start = time.perf_counter()
leg = bench.Leg(dyn_cfg, rw_cfg, "f32", dev, None, 64, "train")
x, metax, mask, target = bench.synth_episode(1000, 64, 20, 416, 224)
step = leg.stepper(x.to(dev).contiguous(), metax.to(dev), mask.to(dev), target)
for _ in range(3):
    step()
torch.cuda.synchronize()


def block(n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = []
for b in range(blocks):
    ms = block()
    tag = ""
    if b % 6 == 5:
        streams.ENABLED = False
        one = block(6)
        streams.ENABLED = True
        tag = " (one stream %.2f)" % one
    out.append("%.2f%s" % (ms, tag))
print("t=%.0fs after start; ms/step per block of 20: %s" % (time.perf_counter() - t_start, "  ".join(out)))
