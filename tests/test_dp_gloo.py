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
    assert tr.launch_order_last == list(range(len(tr.buckets)))   # every bucket was reduced, in ascending = readiness order
    # the flat buffer is laid out in gradient-readiness order (last layer first): compare in registration order
    assert tr.params[0] is net.b.bias and tr.params[-1] is net.a.weight
    torch.save(torch.cat([p.detach().reshape(-1) for p in net.parameters()]), os.path.join(out_dir, "rank%d.pt" % rank))
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
    from fewshot_detection_amd.dp import bucket_bounds, tapered_bounds
    for total in (1, 1023, 1024, 5000, 66287742):
        for b in (bucket_bounds(total, 4), tapered_bounds(total, 6)):
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
            assert all(lo % 1024 == 0 for lo, _ in b)
    b = tapered_bounds(66287742, 6)
    assert len(b) == 6 and (b[-1][1] - b[-1][0]) < 0.08 * 66287742   # the bucket nothing can hide is the small one
    b = tapered_bounds(66287742, 8)                                  # optional finer tail: five equal buckets, then 5.7 % / 1 % / 0.3 %
    sizes = [hi - lo for lo, hi in b]
    assert len(b) == 8 and b[0][0] == 0 and b[-1][1] == 66287742 and all(b[i][1] == b[i + 1][0] for i in range(7))
    assert sizes[-1] < 0.004 * 66287742 and sizes[-2] < 0.011 * 66287742 and sizes[-3] < 0.06 * 66287742
    assert max(sizes[:5]) - min(sizes[:5]) <= 1024 * 5
    for total in (1, 5000, 40000, 66287742):
        bb = tapered_bounds(total, 8)
        assert bb[0][0] == 0 and bb[-1][1] == total and all(bb[i][1] == bb[i + 1][0] for i in range(len(bb) - 1))


def test_flat_buffer_follows_gradient_readiness_order(tmp_path):
    """The meta detector's flat parameter buffer: reweighting net first (its backward sweep is queued as soon as
    d(vectors) exists), then the detector from the head down to layer 0 -- SURVEY 5: L29 / L24 / L23 lead, L0 is last."""
    from fewshot_detection_amd import cfgs
    from fewshot_detection_amd.darknet_meta import Darknet
    from fewshot_detection_amd.dp import readiness_order
    dyn_cfg, rw_cfg, _ = cfgs.write_standard_cfgs(str(tmp_path))
    net = Darknet(dyn_cfg, rw_cfg)
    order = readiness_order(net)
    assert len(order) == len(list(net.parameters())) and len({id(p) for p in order}) == len(order)
    n_learnet = len(list(net.learnet_models.parameters()))
    assert all(any(p is q for q in net.learnet_models.parameters()) for p in order[:n_learnet])
    det = order[n_learnet:]
    assert det[-1] is net.models[0][0].weight                      # layer 0's 3x3x3x32 filter is the last gradient
    big = [p.numel() for p in det if p.numel() > 9_000_000]
    assert big == [1280 * 1024 * 9, 1024 * 1024 * 9, 1024 * 1024 * 9]   # L29, L24, L23 in that order


class _TinyBN(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(3, 5, 3, 1, 1, bias=False)
        self.bn = nn.BatchNorm2d(5)

    def forward(self, x):
        return self.bn(self.a(x))


def _worker_sync(rank, world, port, out_dir):
    from fewshot_detection_amd.dp import EpisodeTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                                 # every rank initialises DIFFERENTLY
    net = _TinyBN()
    net.bn.running_mean.fill_(float(rank + 1))
    net.bn.running_var.fill_(float(rank + 2))
    tr = EpisodeTrainer(net, 0.01, 0.9, 0.0, process_group=dist, n_buckets=2, step_fn=lambda lo, hi: None)
    torch.save(dict(flat=torch.cat([p.detach().reshape(-1) for p in net.parameters()]), mean=net.bn.running_mean.clone(), var=net.bn.running_var.clone(),
                    world=tr.world_size), os.path.join(out_dir, "sync%d.pt" % rank))
    # buckets must be reduced in ascending order on every rank: a descending launch is refused loudly
    tr._launch_order = [1]
    try:
        tr._launch_ready((), final=True)
        ok = False
    except RuntimeError:
        ok = True
    torch.save(ok, os.path.join(out_dir, "order%d.pt" % rank))
    dist.destroy_process_group()


def test_replicas_start_from_rank0_state_and_bucket_order_is_enforced(tmp_path):
    """EpisodeTrainer broadcasts parameters, momentum and BN running statistics from rank 0 at
    construction (the reference's DataParallel re-broadcasts module 0 every step, train_meta.py:137-141)."""
    port = _free_port()
    mp.spawn(_worker_sync, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), "sync0.pt"))
    b = torch.load(os.path.join(str(tmp_path), "sync1.pt"))
    assert a["world"] == b["world"] == 2
    assert torch.equal(a["flat"], b["flat"]) and torch.equal(a["mean"], b["mean"]) and torch.equal(a["var"], b["var"])
    torch.manual_seed(100)
    ref = _TinyBN()
    assert torch.equal(a["flat"], torch.cat([p.detach().reshape(-1) for p in ref.parameters()]))
    assert float(a["mean"][0]) == 1.0 and float(b["var"][0]) == 2.0      # rank 0's statistics everywhere
    assert torch.load(os.path.join(str(tmp_path), "order0.pt")) and torch.load(os.path.join(str(tmp_path), "order1.pt"))


class _TinyWithLoss(_Tiny):
    """A replica that owns a region-loss module, like darknet_meta.Darknet (models[-1])."""

    def __init__(self):
        super().__init__()
        from fewshot_detection_amd.region_loss import RegionLossV2
        self.loss = RegionLossV2(1, [1.0, 1.0], 1)


def _neg_worker(rank, world, port, out_dir):
    import random
    import numpy as np
    from fewshot_detection_amd import region_loss
    from fewshot_detection_amd.cfg import cfg
    from fewshot_detection_amd.dp import EpisodeTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, other = _TinyWithLoss(), _TinyWithLoss()
    tr = EpisodeTrainer(net, 0.001, 0.9, 0.0, process_group=dist, n_buckets=2, step_fn=lambda lo, hi: None)
    # the reducer belongs to THIS trainer's loss module and to nothing else (no process-global hook)
    assert tr.neg_counts is not None and net.loss.neg_counts is tr.neg_counts and other.loss.neg_counts is None
    assert not hasattr(region_loss, "GLOBAL_NEG_COUNTS")
    # rank 0: 2 positive rows of 10; rank 1: 6 of 10 -> the batch: 8 positive, 12 negative
    rows = np.zeros((10, 250))
    rows[:2 if rank == 0 else 6, 1] = 0.5
    assert tr.neg_counts(int((rows.sum(1) != 0).sum()), 10) == (8, 20)
    keep = cfg.neg_ratio
    try:
        cfg.neg_ratio = 1                      # ratio = 8 / 12 everywhere (per rank it would be 2/8 and 6/4 -> "keep all")
        random.seed(100 + rank)
        inds = region_loss.neg_filter_indices(rows, tr.neg_counts)
        random.seed(100 + rank)
        n_pos = 2 if rank == 0 else 6
        want = [i for i in range(10) if i < n_pos or not (random.random() > 8.0 / 12.0)]
        assert inds == want, (rank, inds, want)
        cfg.neg_ratio = 2                      # 2 * 8 / 12 >= 1: every rank keeps every row
        assert region_loss.neg_filter_indices(rows, tr.neg_counts) == list(range(10))
        cfg.neg_ratio = "full"                 # no collective, no change
        assert region_loss.neg_filter_indices(rows, tr.neg_counts) == list(range(10))
        # a free-standing call (rank-0-only validation, another model's loss, utils callers) is NOT a collective: it uses
        # the shard's own ratio and cannot block -- only rank 0 makes this call
        cfg.neg_ratio = 1
        if rank == 0:
            random.seed(7)
            local = region_loss.neg_filter_indices(rows)
            random.seed(7)
            assert local == [i for i in range(10) if i < 2 or not (random.random() > 2.0 / 8.0)]
        # a second live trainer on the same model would pair its collectives with the first one's: refused
        try:
            EpisodeTrainer(net, 0.001, 0.9, 0.0, process_group=dist, n_buckets=2, step_fn=lambda lo, hi: None)
            raise AssertionError("a second live trainer was accepted")
        except RuntimeError as e:
            assert "live EpisodeTrainer" in str(e)
        # close(): the model's loss is back on the local ratio, the reducer is dead, a new trainer may take the model
        reducer = tr.neg_counts
        tr.close()
        tr.close()                             # idempotent
        assert net.loss.neg_counts is None and tr.neg_counts is None
        try:
            reducer(1, 2)
            raise AssertionError("the reducer outlived its trainer")
        except RuntimeError:
            pass
        tr2 = EpisodeTrainer(net, 0.001, 0.9, 0.0, process_group=dist, n_buckets=2, step_fn=lambda lo, hi: None)
        assert net.loss.neg_counts is tr2.neg_counts and tr2.neg_counts(1, 10) == (2, 20)
        tr2.close()
    finally:
        cfg.neg_ratio = keep
    open(os.path.join(out_dir, "neg%d.ok" % rank), "w").write("ok")
    dist.destroy_process_group()
    tr2.close()                                # after the default group is gone: still harmless


def test_neg_filter_ratio_is_the_gathered_batchs_under_data_parallelism(tmp_path):
    """The reference drops negative (image, class) rows with probability 1 - neg_ratio * n_pos / n_neg of the batch that
    nn.DataParallel GATHERS (region_loss.py:15-34 on train_meta.py:137-141's outputs).  With one process per GPU the trainer
    sums the two counts over the ranks (host-side gloo group): every rank uses the batch's ratio, not its shard's.  The
    reducer is owned by the trainer (VERDICT r5 / ADVICE r5): scoped to its model's loss module, removed by close(), a second
    live trainer refused, free-standing neg_filter calls local."""
    port = _free_port()
    mp.spawn(_neg_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "neg%d.ok" % r)) for r in (0, 1))


def _single_rank_worker(rank, port, out_dir):
    from fewshot_detection_amd.dp import EpisodeTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    data = torch.randn(4, 3, 6, 6, generator=torch.Generator().manual_seed(1))
    out = {}
    for name, group, force in (("plain", None, False), ("group_idle", dist, False), ("group_forced", dist, True)):
        torch.manual_seed(0)
        net = _Tiny()
        holder = {}

        def torch_step(lo, hi):
            t = holder["t"]
            buf = t.grad[lo:hi] if t.steps == 0 else 0.9 * t.mom[lo:hi] + t.grad[lo:hi]
            t.mom[lo:hi] = buf
            t.flat[lo:hi] -= 0.001 * buf

        tr = EpisodeTrainer(net, 0.001, 0.9, 0.0, process_group=group, n_buckets=3, step_fn=torch_step,
                            single_rank_collectives=force)
        holder["t"] = tr
        for _ in range(3):
            tr.backward_and_step((net(data) ** 2).sum())
        out[name] = (tr.flat.detach().clone(), tr.collective, list(tr.launch_order_last), len(tr.buckets))
        tr.close()
    torch.save(out, os.path.join(out_dir, "single.pt"))
    dist.destroy_process_group()


def test_single_rank_collectives_run_every_collective_and_change_nothing(tmp_path):
    """EpisodeTrainer(single_rank_collectives=True): with a process group of ONE rank the trainer still launches its bucketed
    all-reduces (in ascending order) -- the switch behind the one-rank RCCL runs of the GPU box -- and the parameters equal the
    group-less trainer's bit for bit; without the switch a one-rank group issues nothing."""
    mp.spawn(_single_rank_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    out = torch.load(os.path.join(str(tmp_path), "single.pt"))
    assert out["plain"][1] is False and out["group_idle"][1] is False and out["group_forced"][1] is True
    assert out["plain"][2] == [] and out["group_idle"][2] == [] and out["group_forced"][2] == list(range(out["group_forced"][3]))
    assert out["group_forced"][3] >= 1
    assert torch.equal(out["plain"][0], out["group_idle"][0]) and torch.equal(out["plain"][0], out["group_forced"][0])
