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

Overlap ("bucket the grads in reverse layer order, overlap with the remaining dgrad / wgrad").  The flat
buffer is laid out in the order in which the backward pass FINISHES the gradients: the reweighting net first (its
backward starts as soon as the head's backward has produced d(vectors) and is a few ms long), then the detector from
its last layer down to layer 0 -- the 47 / 38 / 38 MB tensors of L29 / L24 / L23 come first, the small early layers last.
Buckets are contiguous ranges of that order, the last one small.  The backward sweep calls the trainer after EVERY layer
(ops.GRAD_HOOK); for a bucket whose last gradient kernel has just been queued the "meta" side stream (idle by then) waits for
the streams those kernels run on, and its all-reduce is started there -- while the sweep is still queueing (and the GPU still
running) the layers below.  Buckets start in ascending index on every rank (enforced), which is their readiness order.
"""
import os
import sys
import time

import torch

from . import ops, streams
from .engine import bump_weight_epoch


def readiness_order(module):
    """Parameters in the order their gradients are completed by one backward pass: for the meta detector the reweighting
    net (its sweep is queued as soon as d(vectors) exists, see backward.run_early), then the detector from the head down;
    for any other module the reverse of the registration order."""
    learnet, det = getattr(module, "learnet_models", None), getattr(module, "models", None)
    if learnet is not None and det is not None:
        order = list(reversed(list(learnet.parameters()))) + list(reversed(list(det.parameters())))
        seen = {id(p) for p in order}
        order += [p for p in reversed(list(module.parameters())) if id(p) not in seen]
        return order
    return list(reversed(list(module.parameters())))


def flatten_parameters(module, order=None):
    """Re-home every parameter of `module` as a view into one contiguous fp32 buffer, laid out in `order`."""
    params = list(order) if order is not None else [p for p in module.parameters()]
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


def tapered_bounds(total, n_buckets, tail=0.07, fine=(0.010, 0.003)):
    """n_buckets contiguous ranges, 1024-aligned, the last ones small: `tail` of the buffer behind the equal-sized head
    buckets, of which -- from 8 buckets on -- the last two take `fine[0]` and `fine[1]` of the buffer.  The end of the buffer
    holds the gradients the backward pass finishes last (the early layers of the detector: L0..L16 are 7 % of the parameters,
    L0..L10 1.2 %, L0..L6 0.3 %), so the all-reduce nothing is left to hide behind -- and the optimizer step + re-pack that
    cannot run under the backward pass (EARLY_STEP) -- are those of a handful of small layers."""
    if n_buckets <= 1 or total < 4096 * n_buckets:
        return bucket_bounds(total, max(1, n_buckets))
    cut = int(total * (1.0 - tail)) // 1024 * 1024
    if n_buckets < 8:
        return bucket_bounds(cut, n_buckets - 1) + [(cut, total)]
    c1 = int(total * (1.0 - fine[0] - fine[1])) // 1024 * 1024
    c2 = int(total * (1.0 - fine[1])) // 1024 * 1024
    if not (cut < c1 < c2 < total):
        return bucket_bounds(cut, n_buckets - 1) + [(cut, total)]
    return bucket_bounds(cut, n_buckets - 3) + [(cut, c1), (c1, c2), (c2, total)]


class EpisodeTrainer(object):
    """SGD(momentum, weight decay) + gradient all-reduce for one Darknet replica."""

    def __init__(self, net, lr, momentum=0.9, weight_decay=0.0, process_group=None, n_buckets=6, step_fn=None,
                 grad_dtype=torch.float32, single_rank_collectives=False):
        """single_rank_collectives: with a process group of ONE rank, issue every collective anyway (broadcasts, the
        bucketed all-reduce launched from a side stream, the host-side neg_filter reduction).  The sums over one rank are the
        identity, so the result must equal the group-less trainer's bit for bit -- this is how the RCCL transport, its
        stream semantics and the bf16 wire format are exercised on a one-GPU box (tests/test_gpu_dp.py).
        grad_dtype: wire format of the gradient all-reduce.  torch.bfloat16 (BASELINE configs[2] / [4]) halves the
        xGMI payload (133 MB instead of 265 MB per step): each bucket is rounded to bf16 right before its
        collective and widened back before the fp32 optimizer step; master weights, momentum and the local gradient
        stay fp32."""
        self.net = net
        self.grad_dtype = grad_dtype
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.dist = process_group            # the torch.distributed module (or None for one GPU)
        self.flat, self.params = flatten_parameters(net, readiness_order(net))
        self.grad = torch.zeros_like(self.flat)
        self.mom = torch.zeros_like(self.flat)
        self.buckets = tapered_bounds(self.flat.numel(), n_buckets)
        self.sink, off = {}, 0               # parameter -> its slice of the flat gradient buffer
        self._bucket_params = [[] for _ in self.buckets]         # ids of the parameters overlapping each bucket
        for p in self.params:
            self.sink[id(p)] = self.grad[off:off + p.numel()]
            for i, (lo, hi) in enumerate(self.buckets):
                if off < hi and off + p.numel() > lo:
                    self._bucket_params[i].append(id(p))
            off += p.numel()
        self._works = [None] * len(self.buckets)
        self._stepped = [False] * len(self.buckets)              # buckets whose optimizer step was queued during the backward
        self.early_steps_last = 0                                # ... how many of them in the last step
        self._conv_by_bucket = None
        self._launch_order = []              # bucket indices in the order their all-reduce was started this step
        self.launch_order_last = []          # ... of the last completed step
        self._t_backward0 = 0.0
        self._launch_host_ms = [None] * len(self.buckets)        # host time since the start of backward() at launch
        self._ready_events = [None] * len(self.buckets)          # GPU: "this bucket's gradients are complete"
        self._bw_end_event, self._bw_host_ms = None, 0.0
        self._streams_seen = []              # streams the current backward pass has queued gradient kernels on
        self._overlap = None                 # last step's measurements (time_allreduce only)
        self.steps = 0
        self._step_fn = step_fn or self._hip_step
        self.world_size = 1 if self.dist is None else int(self.dist.get_world_size())
        # True when this trainer exchanges gradients at all (several ranks, or one rank asked to run its collectives)
        self.collective = self.dist is not None and (self.world_size > 1 or bool(single_rank_collectives))
        self.grad_lp = None
        if self.grad_dtype != torch.float32 and self.collective:
            self.grad_lp = torch.empty_like(self.grad, dtype=self.grad_dtype)
        # per-bucket time the optimizer loop spent blocked in work.wait() (exposed all-reduce), accumulated over steps
        self.allreduce_wait_ms = [0.0] * len(self.buckets)
        self.time_allreduce = False
        self._neg_group = None
        self.neg_counts = None               # (n_pos, n_rows) -> sums over the ranks; None: the local counts are the batch's
        self._loss_modules = []
        if self.collective:
            self._install_global_neg_counts()
        self.sync_replicas()

    def _install_global_neg_counts(self):
        """neg_filter's keep ratio over the WHOLE batch, as the reference computes it on the gathered outputs
        (region_loss.py:15-34 on train_meta.py:137-141): two integers per step, summed over the ranks on a gloo group of their
        own -- host tensors, so the call neither touches nor waits for a GPU stream (an .item() behind an RCCL all-reduce would
        drain the two steps the trainer keeps in flight).  The reducer is handed to the loss modules of THIS trainer's model
        (`module.neg_counts`) and to nothing else; they call it only in training mode with gradients enabled.
        CONTRACT: every rank calls its model's loss exactly once per training step, in step order -- the collective matches
        up by construction.  close() (or the trainer's destruction) removes the reducer again."""
        from .region_loss import _RegionBase
        mods = [m for m in self.net.modules() if isinstance(m, _RegionBase)]
        for m in mods:
            if getattr(m, "neg_counts", None) is not None:
                raise RuntimeError("this model's loss module already belongs to a live EpisodeTrainer (close() it first): two "
                                   "trainers would pair their neg_filter collectives with each other")
        group, err = None, None
        try:
            group = self.dist.new_group(backend="gloo")
        except Exception as e:           # no gloo in this build / rendezvous without TCP
            err = e
        # the fallback is a COLLECTIVE decision: a rank that could not build the group must not leave the others waiting in
        # its all-reduce, and ranks must not mix the global with the per-rank ratio
        ok = torch.tensor([0 if group is None else 1], dtype=torch.int32, device=self.flat.device)
        self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            print("EpisodeTrainer: global neg_filter ratio unavailable on some rank (%s); every rank keeps its per-rank ratio"
                  % (err,), file=sys.stderr)
            if group is not None:
                self.dist.destroy_process_group(group)
            return
        self._neg_group = group
        buf = torch.zeros(2, dtype=torch.int64)

        def reduce_counts(n_pos, n_rows):
            if self._neg_group is None:
                raise RuntimeError("EpisodeTrainer.close() was called: this reducer is gone")
            buf[0], buf[1] = int(n_pos), int(n_rows)
            self.dist.all_reduce(buf, group=self._neg_group)
            return int(buf[0]), int(buf[1])
        self.neg_counts = reduce_counts
        self._loss_modules = mods
        for m in mods:
            m.neg_counts = reduce_counts

    def close(self):
        """Give the model back: remove the whole-batch neg_filter reducer from its loss modules (they use the local ratio
        again) and drop the host-side group.  Idempotent; call it on every rank before destroy_process_group()."""
        for m in self._loss_modules:
            if getattr(m, "neg_counts", None) is self.neg_counts:
                m.neg_counts = None
        self._loss_modules = []
        self.neg_counts = None
        group, self._neg_group = self._neg_group, None
        if group is not None:
            try:
                if self.dist.is_initialized():
                    self.dist.destroy_process_group(group)
            except Exception:            # the default group is already gone: nothing left to free
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync_replicas(self):
        """Every replica starts from rank 0's parameters, momentum and BatchNorm running statistics (the reference's
        nn.DataParallel re-broadcasts module 0's state every step, train_meta.py:137-141; DDP does this once at
        construction).  Without it, ranks that seeded or loaded differently would apply the SUM of their gradients to
        different weights and silently diverge."""
        if not self.collective:
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
        multi = self.__dict__.get("_multi_now")      # looked up once per step by reduce_and_step
        if multi is not None:
            tab = multi[self.buckets.index((lo, hi))]
            if tab is not None:              # (a bucket whose tensors all end in a later bucket has nothing to do)
                ops.sgd_step_multi(self.flat, self.grad, self.mom, tab[0], tab[1], tab[2], tab[3], self.lr, self.momentum,
                                   self.weight_decay, self.steps == 0)
            return
        ops.sgd_step(self.flat[lo:hi], self.grad[lo:hi], self.mom[lo:hi], self.lr, self.momentum,
                     self.weight_decay, self.steps == 0)

    # ---- bf16 storage mode: optimizer step + re-packing of the bf16 conv operands in ONE pass ------------------------------
    # The bf16 kernels read packed bf16 copies of every conv weight (forward and data-gradient operand).  They used to be
    # rebuilt by 20 pack launches at the start of the next forward (0.28 ms on the critical path of a 12.3 ms step, VERDICT r5
    # weak #6); fsd_sgd_step_multi updates a [64 cout][64 cin][taps] block of the master weight and writes both bf16 copies of
    # the NEW values while it has them in LDS -- one launch per gradient bucket, no second read of the weights.
    FUSE_PACK = True

    def _multi_tables(self):
        """Per bucket: (device table, entries, workgroup blocks, elements) of fsd_sgd_step_multi, or None when the fused form
        does not apply (fp32 mode, a step function of the caller's, packed copies not built yet).  A tensor belongs to the
        bucket that holds its LAST element (its gradient is complete once that bucket's all-reduce is)."""
        if not self.FUSE_PACK or self._step_fn != self._hip_step or not self.flat.is_cuda:
            return None
        nets = [n for n in (getattr(self.net, "_det", None), getattr(self.net, "_meta", None)) if n is not None]
        if not nets or any(n.compute_dtype != "bf16" for n in nets):
            return None
        pairs = {}
        for p in self.params:
            if p.dim() == 4:
                for n in nets:
                    pr = n.cache.bf16_pair(p)
                    if pr is not None:
                        pairs[id(p)] = (n.cache, pr)
        sig = tuple((k, v[1][0].data_ptr(), v[1][1].data_ptr()) for k, v in pairs.items())
        built = self.__dict__.get("_multi_built")
        if built is not None and built[0] == sig:
            return built[1]
        if not pairs:
            return None
        rows = [[] for _ in self.buckets]            # per bucket: [offset, count, cout, cin, taps, first_block, p0, p1]
        off = 0
        for p in self.params:
            n_el = p.numel()
            b = next(i for i, (lo, hi) in enumerate(self.buckets) if lo <= off + n_el - 1 < hi)
            if id(p) in pairs:
                cout, cin, k, _ = p.shape
                pr = pairs[id(p)][1]
                rows[b].append([off, n_el, cout, cin, k * k, 0, pr[0].data_ptr(), pr[1].data_ptr()])
            elif rows[b] and rows[b][-1][4] == 0 and rows[b][-1][0] + rows[b][-1][1] == off:
                rows[b][-1][1] += n_el                 # adjacent plain tensors: one range
            else:
                rows[b].append([off, n_el, 0, 0, 0, 0, 0, 0])
            off += n_el
        tables = []
        for r in rows:
            if not r:
                tables.append(None)
                continue
            blocks = 0
            for e in r:
                e[5] = blocks
                blocks += ops.sgd_multi_blocks(e[1], e[2], e[3], e[4])
            tables.append((torch.tensor(r, dtype=torch.int64).to(self.flat.device), len(r), blocks, sum(e[1] for e in r)))
        self._multi_built = (sig, tables, [v[0] for v in pairs.values()], [p for p in self.params if id(p) in pairs])
        return tables

    def _mark_packed_fresh(self):
        built = self.__dict__.get("_multi_built")
        if built is not None and self.__dict__.get("_multi_now") is built[1]:
            for cache, p in zip(built[2], built[3]):
                cache.mark_fresh(p)
        self._multi_now = None

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

    # Optimizer step of a bucket as soon as its gradients are final -- DURING the backward pass, on the side stream the
    # collectives are launched from: the SGD kernel and the re-packing of the bucket's conv operands (weight transforms of the
    # Winograd layers, K-major copies: ~0.9 ms of HBM-bound kernels per fp32 step, 0.4 ms in bf16 mode) run under the
    # MFMA-bound layers the sweep still has to do, instead of between the steps.  Same kernels on the same values: results
    # are bit-identical.  EARLY_STEP = False: every bucket is stepped after the backward pass, on the main stream.
    EARLY_STEP = os.environ.get("FSD_EARLY_STEP", "1") != "0"

    def _early_ok(self):
        if not (self.EARLY_STEP and streams.ENABLED and self.grad.is_cuda and self._step_fn == self._hip_step):
            return False
        # (gloo's wait() blocks the HOST until the reduction is done: only a stream-ordered backend can be stepped early)
        return (not self.collective) or str(self.dist.get_backend()) == "nccl"

    def _launch_ready(self, sunk, final=False, wait_streams=()):
        """Start the all-reduce of every bucket whose gradients are complete (all of its parameters were written by
        the backward kernels, or -- `final` -- everything has been gathered).  Called by the backward sweep after every
        layer (ops.GRAD_HOOK) with the streams its gradient kernels were queued on: the collective is started from a
        side stream that waits for exactly those, so it runs under the rest of the sweep -- and, where the backend allows,
        the bucket's optimizer step right behind it (EARLY_STEP).  Every rank launches the same buckets in the same
        (ascending = readiness) order."""
        early = (not final) and self._early_ok()
        if not self.collective and not early:
            return
        if self.grad.is_cuda:
            for s_ in (torch.cuda.current_stream(),) + tuple(wait_streams):
                if s_ is not None and not any(s_ == t for t in self._streams_seen):
                    self._streams_seen.append(s_)
        for i, (lo, hi) in enumerate(self.buckets):
            if self._works[i] is not None or self._stepped[i]:
                continue
            if not (final or all(pid in sunk for pid in self._bucket_params[i])):
                break          # strictly ascending bucket order on every rank, whatever order the networks finish in
            if self.collective:
                if self._launch_order and self._launch_order[-1] >= i:
                    raise RuntimeError("gradient buckets must be reduced in ascending order on every rank (bucket %d after "
                                       "%d): ranks would pair different buckets in one collective" % (i, self._launch_order[-1]))
                self._launch_order.append(i)
                self._launch_host_ms[i] = (time.perf_counter() - self._t_backward0) * 1e3
            if self.grad.is_cuda:
                # launched from the "meta" stream: it is idle by now (the reweighting net's sweep is the first thing a backward
                # pass queues, and its gradients are the first bucket), and a stream of the collectives' own is one stream
                # more than the four hardware queues carry without sharing (streams.py)
                comm = streams.side(self.grad.device, self.collective_stream)
                # A bucket can hold gradients queued on several streams at different times (the tail of the reweighting
                # net's share, queued on the "meta" stream by the early sweep, shares a bucket with the detector's head):
                # wait for EVERY stream this backward pass has reported so far, not only the reporting call's.  Streams are
                # FIFO, so this waits for nothing that is not already due.
                for s_ in self._streams_seen:
                    if s_ != comm:
                        comm.wait_stream(s_)
                with torch.cuda.stream(comm):
                    if self.collective:
                        if self.time_allreduce:
                            self._ready_events[i] = comm.record_event(torch.cuda.Event(enable_timing=True))
                        self._works[i] = self._start(lo, hi)
                    if early:
                        self._finish_bucket(i)
            else:
                self._works[i] = self._start(lo, hi)

    def _finish_bucket(self, i):
        """Bucket i on the CURRENT stream: wait for its collective, widen a bf16 wire buffer, optimizer kernel, re-pack."""
        lo, hi = self.buckets[i]
        if self._works[i] is not None:
            if self.time_allreduce:
                t0 = time.perf_counter()
                self._works[i].wait()
                self.allreduce_wait_ms[i] += (time.perf_counter() - t0) * 1e3
            else:
                self._works[i].wait()
            if self.grad_lp is not None:
                self.grad[lo:hi].copy_(self.grad_lp[lo:hi])
        self._step_fn(lo, hi)
        self._repack(i)
        self._stepped[i] = True

    def _repack(self, i):
        """fp32 modes: rewrite the packed operand copies of the conv weights that bucket i completes (a tensor belongs to the
        bucket that holds its LAST element), valid from the weight epoch that the end of this step starts."""
        if self.__dict__.get("_multi_now") is not None:
            return                           # bf16 mode: the fused optimizer kernel wrote the bf16 copies itself
        from .engine import _WEIGHT_EPOCH
        nets = [n for n in (getattr(self.net, "_det", None), getattr(self.net, "_meta", None)) if n is not None]
        if not nets:
            return
        if self._conv_by_bucket is None:
            by, off = [[] for _ in self.buckets], 0
            for p in self.params:
                if p.dim() == 4:
                    last = off + p.numel() - 1
                    by[next(k for k, (lo, hi) in enumerate(self.buckets) if lo <= last < hi)].append(p)
                off += p.numel()
            self._conv_by_bucket = by
        for p in self._conv_by_bucket[i]:
            for n in nets:
                n.cache.refresh(p, _WEIGHT_EPOCH[0] + 1)

    def _start(self, lo, hi):
        buf = self.grad[lo:hi]
        if self.grad_lp is not None:
            buf = self.grad_lp[lo:hi]
            buf.copy_(self.grad[lo:hi])
        return self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, async_op=True)

    def overlap_report(self):
        """Where the collectives of the LAST step started relative to its backward pass (needs time_allreduce and > 1
        rank; synchronises): per bucket the host time since loss.backward() began at which its all-reduce was queued,
        and how long before the END of the backward pass on the GPU its gradients were complete (positive = the
        collective could start that many ms before the backward finished)."""
        o = self._overlap
        if not o:
            return None
        torch.cuda.synchronize() if self.grad.is_cuda else None
        ready = [None if (e is None or o["bw_end"] is None) else e.elapsed_time(o["bw_end"]) for e in o["ready"]]
        return {"launch_order": o["order"], "launch_host_ms_after_backward_start": o["host_ms"],
                "backward_enqueue_host_ms": o["bw_host_ms"], "gpu_ms_ready_before_backward_end": ready,
                "buckets_launched_before_backward_enqueue_ended": sum(1 for v in o["host_ms"] if v is not None and v < o["bw_host_ms"])}

    def reduce_and_step(self):
        """Bucketed SUM all-reduce overlapped with the per-bucket optimizer kernel."""
        self._launch_ready((), final=True)
        if self.collective and self._launch_order != list(range(len(self.buckets))):
            raise RuntimeError("all-reduce launch order %r is not 0..%d ascending" % (self._launch_order, len(self.buckets) - 1))
        if self.time_allreduce and self.collective:
            self._overlap = {"order": list(self._launch_order), "host_ms": list(self._launch_host_ms),
                             "ready": list(self._ready_events), "bw_end": self._bw_end_event,
                             "bw_host_ms": self._bw_host_ms}
        if self.__dict__.get("_multi_now") is None:
            self._multi_now = self._multi_tables()
        self.early_steps_last = sum(1 for v in self._stepped if v)
        if self.early_steps_last and self.grad.is_cuda:
            # what follows on this stream -- the remaining buckets (a tensor can straddle two of them), then the next forward --
            # reads the weights and operand copies the side stream has (re)written
            torch.cuda.current_stream().wait_stream(streams.side(self.grad.device, self.collective_stream))
        for i in range(len(self.buckets)):
            if not self._stepped[i]:
                self._finish_bucket(i)
        self._stepped = [False] * len(self.buckets)
        self._works = [None] * len(self.buckets)
        self.launch_order_last, self._launch_order = self._launch_order, []
        self._launch_host_ms = [None] * len(self.buckets)
        self._ready_events = [None] * len(self.buckets)
        self._streams_seen = []
        self.steps += 1
        bump_weight_epoch()
        self._mark_packed_fresh()            # (fused step: the bf16 operand copies already hold the new weights)

    def backward_and_step(self, loss):
        # While this backward runs, the HIP gradient kernels write dW / dgamma / dbeta straight into the flat
        # buffer (ops.GRAD_SINK); whatever still arrives through autograd is gathered afterwards.
        ops.GRAD_SINK, ops.GRAD_SUNK = self.sink, set()
        ops.GRAD_HOOK = lambda wait_streams=(): self._launch_ready(ops.GRAD_SUNK, wait_streams=wait_streams)
        self._multi_now = self._multi_tables()       # (bf16 mode: the fused optimizer + re-pack tables, needed by an early step)
        self._t_backward0 = time.perf_counter()
        self._bw_end_event, self._bw_host_ms = None, 0.0
        try:
            loss.backward()
            sunk = ops.GRAD_SUNK
        except BaseException:
            # a backward pass that died half-way: with EARLY_STEP the buckets finished so far are already stepped (their
            # gradients were final); forget the step's bookkeeping so that the next call starts clean
            self._stepped = [False] * len(self.buckets)
            self._works = [None] * len(self.buckets)
            self._launch_order, self._streams_seen, self._multi_now = [], [], None
            raise
        finally:
            ops.GRAD_SINK, ops.GRAD_SUNK, ops.GRAD_HOOK = None, set(), None
        self._bw_host_ms = (time.perf_counter() - self._t_backward0) * 1e3
        if self.time_allreduce and self.grad.is_cuda and self.collective:
            self._bw_end_event = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
        # a bucket that is already being reduced had all of its parameters sunk: nothing of it is left to gather
        self.gather_grads(sunk)
        self.reduce_and_step()
        self._throttle()

    # How many steps the host may queue ahead of the GPU.  Unbounded, a host that enqueues a step in ~8 ms against ~25 ms of
    # GPU time is 20 steps ahead after 25 steps; tensors that crossed a stream (record_stream) cannot be re-used by the caching
    # allocator until the GPU has passed them, so every queued step takes fresh memory -- measured 153 GB reserved after 160
    # pipelined steps of a step whose live peak is 14 GB, and one run in four of bench.py hitting the 288 GB ceiling: the
    # allocator then synchronises and frees its cache mid-run (36-43 ms per step instead of 26).  Two steps in flight keep the
    # GPU fed (the next step's ~8 ms of enqueue hide behind the current one) and the footprint at ~2 steps' worth.
    max_steps_in_flight = 2

    # the side stream the collectives are launched from (see _launch_ready)
    collective_stream = "meta"

    def _throttle(self):
        if not self.grad.is_cuda or self.max_steps_in_flight is None:
            return
        q = self.__dict__.setdefault("_step_events", [])
        q.append(torch.cuda.current_stream().record_event())
        while len(q) > self.max_steps_in_flight:
            q.pop(0).synchronize()
