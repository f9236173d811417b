"""Probe: per-step wall times of the headline train step (streams on), host enqueue time per step, and the GPU time of each
step from events -- to tell host-bound steps from GPU-side interference.  python tools/probes/step_jitter.py [steps]"""
import os
import sys
import time
import tempfile
import random

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from fewshot_detection_amd import cfgs, streams  # noqa: E402
from fewshot_detection_amd.cfg import cfg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg.neg_ratio = 1
dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(tmp)
leg = bench.Leg(dyn_cfg, rw_cfg, "f32", dev, None, 64, "train")
x, metax, mask, target = bench.synth_episode(1000, 64, 20, 416, 224)
step = leg.stepper(x.to(dev).contiguous(), metax.to(dev), mask.to(dev), target)
for _ in range(5):
    step()
torch.cuda.synchronize()
host, wall, gpu = [], [], []
for i in range(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    step()
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3); gpu.append(e0.elapsed_time(e1))
print("streams", streams.ENABLED)
print("host enqueue ms:", " ".join("%.1f" % v for v in host))
print("wall ms        :", " ".join("%.1f" % v for v in wall))
print("gpu (main) ms  :", " ".join("%.1f" % v for v in gpu))

# pipelined blocks: the host runs ahead of the GPU as in bench.py (no sync inside a block)
def stats():
    s = torch.cuda.memory_stats()
    return s.get("num_device_alloc", 0), s.get("num_device_free", 0), s.get("num_alloc_retries", 0), s.get("reserved_bytes.all.current", 0) >> 20
print("pipelined blocks of 20 steps: ms/step, (device allocs, frees, retries, reserved MiB) after each")
for b in range(int(os.environ.get("BLOCKS", "8"))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("  block %d: %.2f ms/step (host %.2f)  %s" % (b, (t2 - t0) / 20 * 1e3, (t1 - t0) / 20 * 1e3, stats()))
