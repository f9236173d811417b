"""Probe: pure-write, pure-read and copy bandwidth of this box (what the HBM-bound kernels are measured against).
fill = fsd_fill (16-B stores), read = torch.sum (library reduction), copy = torch copy_ (read + write)."""
import sys
import time
import torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
from fewshot_detection_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
n = 1420 * 1000 * 1000 // 4          # 1.42 GB, the first layer's output at B = 64
a = torch.empty(n, device=dev)
b = torch.empty(n, device=dev)


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
w = t(lambda: ops.fill(a, 1.0))
w2 = t(lambda: a.fill_(2.0))
r = t(lambda: torch.sum(a))
c = t(lambda: b.copy_(a))
gb = n * 4 / 1e9
print("pure write (fsd_fill) %.3f ms %.2f TB/s | torch fill_ %.3f ms %.2f TB/s | pure read (sum) %.3f ms %.2f TB/s | copy %.3f ms %.2f TB/s (r+w)"
      % (w * 1e3, gb / w / 1e3, w2 * 1e3, gb / w2 / 1e3, r * 1e3, gb / r / 1e3, c * 1e3, 2 * gb / c / 1e3))
