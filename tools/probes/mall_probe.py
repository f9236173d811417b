"""Probe: does a working set that fits the 256 MB memory-side cache move faster than one that does not?  Per size: repeated fill
(16-byte stores), repeated read (sum), write-then-read pairs and copies of a buffer of that size."""
import sys
import time
import torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
from fewshot_detection_amd import ops  # noqa: E402
dev = torch.device("cuda:0")


def t(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device=dev)
    b = torch.empty(n, device=dev)
    reps = max(10, 4000 // mb)
    gb = n * 4 / 1e9
    w = t(lambda: ops.fill(a, 1.0), reps)
    r = t(lambda: torch.sum(a), reps)
    wr = t(lambda: (ops.fill(a, 1.0), torch.sum(a)), reps)
    c = t(lambda: b.copy_(a), reps)
    print("%5d MB: write %.2f TB/s | read %.2f TB/s | write+read %.2f TB/s | copy (r+w) %.2f TB/s"
          % (mb, gb / w / 1e3, gb / r / 1e3, 2 * gb / wr / 1e3, 2 * gb / c / 1e3), flush=True)
    del a, b
