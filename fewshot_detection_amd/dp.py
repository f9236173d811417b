"""Data-parallel episodic training: one process per GPU, RCCL gradient SUM all-reduce over xGMI.

The reference trains with single-process nn.DataParallel (train_meta.py:137-141): per step it
re-broadcasts 265 MB of parameters, gathers the head outputs to GPU 0 and reduces the gradients
there.  Here every rank keeps its own replica, runs its own episode shard (B/R queries + its own N
supports, like the per-GPU MetaDataset draw, dataset.py:348) and the only exchange is one bucketed
SUM all-reduce of the flat gradient buffer (no averaging: the loss is a sum and lr is already divided
by the global batch, train_meta.py:144).  BatchNorm statistics stay per-rank, as under DataParallel.

Parameters and momentum live in ONE flat fp32 buffer each, so the optimizer step is a single fused
HIP kernel per bucket (fsd_sgd_step) instead of ~200 small launches, and each collective moves tens
of MB (xGMI is point-to-point: few large transfers beat many small ones).
"""
import torch

from . import ops
from .engine import bump_weight_epoch


def flatten_parameters(module):
    """Re-home every parameter of `module` as a view into one contiguous fp32 buffer."""
    params = [p for p in module.parameters()]
    total = sum(p.numel() for p in params)
    flat = torch.empty(total, dtype=torch.float32, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view_as(p.data)
        off += n
    return flat, params


def bucket_bounds(total, n_buckets):
    step = (total + n_buckets - 1) // n_buckets
    step = (step + 1023) // 1024 * 1024
    return [(s, min(total, s + step)) for s in range(0, total, step)]


class EpisodeTrainer(object):
    """SGD(momentum, weight decay) + gradient all-reduce for one Darknet replica."""

    def __init__(self, net, lr, momentum=0.9, weight_decay=0.0, process_group=None, n_buckets=4, step_fn=None,
                 grad_dtype=torch.float32):
        """grad_dtype: wire format of the gradient all-reduce.  torch.bfloat16 (BASELINE configs[2] / [4]) halves the
        xGMI payload (133 MB instead of 265 MB per step): each bucket is rounded to bf16 right before its
        collective and widened back before the fp32 optimizer step; master weights, momentum and the local gradient
        stay fp32."""
        self.net = net
        self.grad_dtype = grad_dtype
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.dist = process_group            # the torch.distributed module (or None for one GPU)
        self.flat, self.params = flatten_parameters(net)
        self.grad = torch.zeros_like(self.flat)
        self.mom = torch.zeros_like(self.flat)
        self.buckets = bucket_bounds(self.flat.numel(), n_buckets)
        self.sink, off = {}, 0               # parameter -> its slice of the flat gradient buffer
        self._bucket_params = [[] for _ in self.buckets]         # ids of the parameters overlapping each bucket
        for p in self.params:
            self.sink[id(p)] = self.grad[off:off + p.numel()]
            for i, (lo, hi) in enumerate(self.buckets):
                if off < hi and off + p.numel() > lo:
                    self._bucket_params[i].append(id(p))
            off += p.numel()
        self._works = [None] * len(self.buckets)
        self._launch_order = []              # bucket indices in the order their all-reduce was started this step
        self.steps = 0
        self._step_fn = step_fn or self._hip_step
        self.world_size = 1 if self.dist is None else int(self.dist.get_world_size())
        self.grad_lp = None
        if self.grad_dtype != torch.float32 and self.world_size > 1:
            self.grad_lp = torch.empty_like(self.grad, dtype=self.grad_dtype)
        # per-bucket time the optimizer loop spent blocked in work.wait() (exposed all-reduce), accumulated over steps
        self.allreduce_wait_ms = [0.0] * len(self.buckets)
        self.time_allreduce = False
        self.sync_replicas()

    def sync_replicas(self):
        """Every replica starts from rank 0's parameters, momentum and BatchNorm running statistics (the reference's
        nn.DataParallel re-broadcasts module 0's state every step, train_meta.py:137-141; DDP does this once at
        construction).  Without it, ranks that seeded or loaded differently would apply the SUM of their gradients to
        different weights and silently diverge."""
        if self.world_size <= 1:
            return
        self.dist.broadcast(self.flat, 0)
        self.dist.broadcast(self.mom, 0)
        for b in self.net.buffers():
            if b.is_floating_point():
                self.dist.broadcast(b, 0)
            else:                            # num_batches_tracked (int64): gloo/RCCL both take integer tensors
                self.dist.broadcast(b, 0)
        bump_weight_epoch()

    def _hip_step(self, lo, hi):
        ops.sgd_step(self.flat[lo:hi], self.grad[lo:hi], self.mom[lo:hi], self.lr, self.momentum,
                     self.weight_decay, self.steps == 0)

    def gather_grads(self, sunk=()):
        """Copy p.grad of every parameter into the flat gradient buffer (missing grads count as zero); parameters
        in `sunk` already had their gradient written there by the backward kernels."""
        off = 0
        for p in self.params:
            n = p.numel()
            if id(p) in sunk:
                pass
            elif p.grad is None:
                self.grad[off:off + n].zero_()
            else:
                self.grad[off:off + n].copy_(p.grad.reshape(-1))
                p.grad = None
            off += n

    def _launch_ready(self, sunk, final=False):
        """Start the all-reduce of every bucket whose gradients are complete (all of its parameters were written by
        the backward kernels, or -- `final` -- everything has been gathered).  Called once per network as its backward
        finishes (ops.GRAD_HOOK), so the reduction of the detector's buckets runs under the reweighting net's backward;
        every rank launches the same buckets in the same order."""
        if self.dist is None or self.dist.get_world_size() <= 1:
            return
        for i, (lo, hi) in enumerate(self.buckets):
            if self._works[i] is not None:
                continue
            if not (final or all(pid in sunk for pid in self._bucket_params[i])):
                break          # strictly ascending bucket order on every rank, whatever order the networks finish in
            if self._launch_order and self._launch_order[-1] >= i:
                raise RuntimeError("gradient buckets must be reduced in ascending order on every rank (bucket %d after "
                                   "%d): ranks would pair different buckets in one collective" % (i, self._launch_order[-1]))
            self._launch_order.append(i)
            buf = self.grad[lo:hi]
            if self.grad_lp is not None:
                buf = self.grad_lp[lo:hi]
                buf.copy_(self.grad[lo:hi])
            self._works[i] = self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, async_op=True)

    def reduce_and_step(self):
        """Bucketed SUM all-reduce overlapped with the per-bucket optimizer kernel."""
        self._launch_ready((), final=True)
        if self.world_size > 1 and self._launch_order != list(range(len(self.buckets))):
            raise RuntimeError("all-reduce launch order %r is not 0..%d ascending" % (self._launch_order, len(self.buckets) - 1))
        for i, (lo, hi) in enumerate(self.buckets):
            if self._works[i] is not None:
                if self.time_allreduce:
                    import time
                    t0 = time.perf_counter()
                    self._works[i].wait()
                    self.allreduce_wait_ms[i] += (time.perf_counter() - t0) * 1e3
                else:
                    self._works[i].wait()
                if self.grad_lp is not None:
                    self.grad[lo:hi].copy_(self.grad_lp[lo:hi])
            self._step_fn(lo, hi)
        self._works = [None] * len(self.buckets)
        self._launch_order = []
        self.steps += 1
        bump_weight_epoch()

    def backward_and_step(self, loss):
        # While this backward runs, the HIP gradient kernels write dW / dgamma / dbeta straight into the flat
        # buffer (ops.GRAD_SINK); whatever still arrives through autograd is gathered afterwards.
        ops.GRAD_SINK, ops.GRAD_SUNK = self.sink, set()
        ops.GRAD_HOOK = lambda: self._launch_ready(ops.GRAD_SUNK)
        try:
            loss.backward()
            sunk = ops.GRAD_SUNK
        finally:
            ops.GRAD_SINK, ops.GRAD_SUNK, ops.GRAD_HOOK = None, set(), None
        # a bucket that is already being reduced had all of its parameters sunk: nothing of it is left to gather
        self.gather_grads(sunk)
        self.reduce_and_step()
