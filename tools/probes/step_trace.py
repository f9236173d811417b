"""Probe: 400 pipelined train steps (as bench.py queues them); per step the host enqueue time and the GPU time between the
step-end events -- to see whether a slow stretch is the host (enqueue time up) or the GPU (enqueue flat, GPU time up)."""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cfg.neg_ratio = 1
dev = torch.device("cuda:0")
dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(tempfile.mkdtemp())
leg = bench.Leg(dyn_cfg, rw_cfg, "f32", dev, None, 64, "train")
x, metax, mask, target = bench.synth_episode(1000, 64, 20, 416, 224)
step = leg.stepper(x.to(dev).contiguous(), metax.to(dev), mask.to(dev), target)
for _ in range(5):
    step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
ev[0].record()
t00 = time.perf_counter()
for i in range(n):
    t0 = time.perf_counter()
    step()
    ev[i + 1].record()
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
wall = (time.perf_counter() - t00) / n * 1e3
gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("streams", streams.ENABLED, "mean wall %.2f ms/step" % wall)
slow = [i for i, g in enumerate(gpu) if g > 1.15 * sorted(gpu)[n // 2]]
print("median gpu %.2f, host median %.2f; slow steps (>1.15x median): %d" % (sorted(gpu)[n // 2], sorted(host)[n // 2], len(slow)))
for i in slow[:40]:
    print("  step %3d gpu %.1f host %.1f" % (i, gpu[i], host[i]))
blocks = [sum(gpu[i:i + 20]) / 20 for i in range(0, n, 20)]
print("20-step block means (gpu):", " ".join("%.1f" % b for b in blocks))
hb = [sum(host[i:i + 20]) / 20 for i in range(0, n, 20)]
print("20-step block means (host):", " ".join("%.1f" % b for b in hb))
