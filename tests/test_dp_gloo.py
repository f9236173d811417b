"""Multi-process data-parallel logic on CPU (gloo, world_size 2): the flat-buffer bucketed SUM
all-reduce + step of dp.EpisodeTrainer reproduces a single-process full-batch SGD step.
(The HIP step kernel is swapped for a torch one here; GPU parity of the kernel is in test_gpu_backward.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(3, 5, 3, 1, 1, bias=False)
        self.b = nn.Conv2d(5, 4, 1, bias=True)

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


def _worker(rank, world, port, out_dir):
    from fewshot_detection_amd.dp import EpisodeTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = _Tiny()
    data = torch.randn(8, 3, 6, 6, generator=torch.Generator().manual_seed(1))
    shard = data[rank * 4:(rank + 1) * 4]                       # each rank owns B/R samples
    lr, mom, wd = 0.001, 0.9, 0.01
    holder = {}

    def torch_step(lo, hi):
        t = holder["t"]
        g = t.grad[lo:hi] + wd * t.flat[lo:hi]
        buf = g if t.steps == 0 else mom * t.mom[lo:hi] + g
        t.mom[lo:hi] = buf
        t.flat[lo:hi] -= lr * buf

    tr = EpisodeTrainer(net, lr, mom, wd, process_group=dist, n_buckets=3, step_fn=torch_step)
    holder["t"] = tr
    for _ in range(3):
        loss = (net(shard) ** 2).sum()                           # a SUM loss, like the region loss
        tr.backward_and_step(loss)
    torch.save(tr.flat.clone(), os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_sum_allreduce_equals_full_batch_sgd(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    assert torch.equal(r0, r1)                                   # replicas stay identical
    torch.manual_seed(0)
    net = _Tiny()
    data = torch.randn(8, 3, 6, 6, generator=torch.Generator().manual_seed(1))
    opt = torch.optim.SGD(net.parameters(), lr=0.001, momentum=0.9, weight_decay=0.01)
    for _ in range(3):
        opt.zero_grad()
        (net(data) ** 2).sum().backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(r0, ref, rtol=1e-5, atol=1e-6)


def test_bucket_bounds_cover_buffer():
    from fewshot_detection_amd.dp import bucket_bounds
    for total in (1, 1023, 1024, 5000, 66287742):
        b = bucket_bounds(total, 4)
        assert b[0][0] == 0 and b[-1][1] == total
        assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
