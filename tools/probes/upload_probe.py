"""Host-side cost of streams.upload() phases while the GPU is busy (why did a step take 35 ms with it?)."""
import time
import numpy as np
import torch
from fewshot_detection_amd import ops, streams

dev = torch.device("cuda:0")
a = np.random.default_rng(0).standard_normal((960, 250))
t = torch.from_numpy(a)
pin = torch.empty(a.nbytes, dtype=torch.uint8).pin_memory()
busy = torch.randn(8192, 8192, device=dev)
torch.cuda.synchronize()


def phase(name, fn, n=10, load=True):
    ts = []
    for _ in range(n):
        if load:
            for _ in range(3):
                busy @ busy                   # ~ms of pending GPU work
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    print("%-40s host ms: median %.3f max %.3f" % (name, sorted(ts)[len(ts) // 2], max(ts)))


out = torch.empty(a.shape, dtype=torch.float64, device=dev)
phase("cpu copy into pinned", lambda: pin[:a.nbytes].copy_(t.reshape(-1).view(torch.uint8)))
phase("torch.empty device", lambda: torch.empty(a.shape, dtype=torch.float64, device=dev))
phase("upload_words kernel launch", lambda: ops.upload_words(pin, out, a.nbytes // 4))
ev = torch.cuda.Event()
phase("event record", lambda: ev.record(torch.cuda.current_stream(dev)))
phase("event record + later synchronize (idle)", lambda: (ev.record(), torch.cuda.synchronize(), ev.synchronize()), load=False)
phase("streams.upload whole", lambda: streams.upload(a, dev))
phase("pageable .to(dev) on busy stream", lambda: t.to(dev))
phase("pinned copy_ non_blocking on busy stream", lambda: out.copy_(pin[:a.nbytes].view(torch.float64).view(a.shape), non_blocking=True))
