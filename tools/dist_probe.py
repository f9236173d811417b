"""Probe: can two ranks share one GPU for a functional check of the N>1 bench path? (gloo on device tensors / RCCL)"""
import os
import sys

import torch
import torch.distributed as dist

backend = sys.argv[1] if len(sys.argv) > 1 else "gloo"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(backend, rank=rank, world_size=world)
t = torch.full((1 << 20,), float(rank + 1), device="cuda:0")
try:
    w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    print(rank, backend, "all_reduce on a device tensor ->", float(t[0]), flush=True)
except Exception as e:  # noqa: BLE001
    print(rank, backend, "FAILED:", repr(e)[:300], flush=True)
dist.barrier()
dist.destroy_process_group()
