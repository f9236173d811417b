"""RegionLoss / RegionLossV2 with the reference's interface (region_loss.py:134-366) on top of
the fused HIP kernel fsd_region_loss_fwd_bwd: decode, IoU silence test, anchor assignment,
objectness / box terms, the softmax over the N episode classes and the gradient are computed on
the device in one pass; nothing is copied back to the host unless the stats line is printed.

Globals read at call time, as in the reference: cfg.neg_ratio, cfg.metayolo, cfg.max_boxes.
"""
from numbers import Number
from random import random

import numpy as np
import torch
import torch.nn as nn

from . import ops, streams
from ._lib import check, lib
from .cfg import cfg


# Data parallelism: the reference computes the keep ratio on the GATHERED batch (nn.DataParallel hands RegionLossV2 the outputs
# of every replica, train_meta.py:137-141 -> region_loss.py:269); one process per GPU sees only its shard.  A trainer over
# several ranks therefore gives ITS model's loss module a reducer (`module.neg_counts`: (n_pos, n_rows) of this rank -> their
# sums over all ranks; dp.EpisodeTrainer, a two-integer all-reduce on a host-side gloo group, off the GPU streams), so that
# every rank drops negatives with the reference's probability 1 - neg_ratio * N_pos / N_neg of the whole batch.  The reducer is
# a COLLECTIVE: it belongs to the one loss module the trainer owns, runs only for a training-mode call with gradients enabled
# (every rank makes exactly one such call per step), and is removed by EpisodeTrainer.close().  Validation under
# torch.no_grad() / .eval(), another model's loss and the free functions below never touch it.


def neg_filter_indices(target_rows, counts=None):
    """Host half of reference neg_filter (region_loss.py:15-34): which (image, class) rows keep
    their box/objectness loss.  `target_rows`: (R, L) array on the host.  python's global
    `random()` is consumed once per negative row, in row order, exactly like the reference.
    `counts`: optional reducer (n_pos, n_rows) -> the sums over the whole data-parallel batch."""
    n_rows = target_rows.shape[0]
    if cfg.neg_ratio == "full":
        return list(range(n_rows))
    if not isinstance(cfg.neg_ratio, Number):
        raise NotImplementedError("neg_ratio not recognized")
    pos = (np.asarray(target_rows, dtype=np.float64).sum(axis=1) != 0).tolist()
    n_pos = sum(pos)
    g_pos, g_rows = (n_pos, n_rows) if counts is None else counts(n_pos, n_rows)
    if g_pos == g_rows:
        return list(range(n_rows))
    ratio = cfg.neg_ratio * g_pos * 1.0 / (g_rows - g_pos)
    if ratio >= 1:
        return list(range(n_rows))
    return [i for i, p in enumerate(pos) if p or not (random() > ratio)]


def neg_filter(pred_boxes, target, withids=False):
    """The reference's function (region_loss.py:15-34) on tensors: drop negative (image, class) rows with probability
    1 - neg_ratio * n_pos / n_neg.  `pred_boxes` / `target`: tensors with one leading row per (image, class)."""
    assert pred_boxes.size(0) == target.size(0)
    t = target.detach()
    inds = neg_filter_indices(t.reshape(t.size(0), -1).cpu().numpy())
    if len(inds) != target.size(0):
        idx = torch.as_tensor(inds, dtype=torch.long)
        pred_boxes, target = pred_boxes[idx.to(pred_boxes.device)], target[idx.to(target.device)]
    return (pred_boxes, target, inds) if withids else (pred_boxes, target)


def build_targets(pred_boxes, target, anchors, num_anchors, num_classes, nH, nW, noobject_scale, object_scale,
                  sil_thresh, seen):
    """The reference's function (region_loss.py:37-132) on the device: decoded boxes (rows*A*H*W, 4) in grid cells +
    float64 targets (rows, 250) -> nGT, nCorrect, coord_mask, conf_mask, cls_mask, tx, ty, tw, th, tconf, tcls, each
    (rows, A, H, W) float32 on the boxes' device.  RegionLoss[V2].forward does not call it (the fused loss kernel builds
    the same targets from the raw head output); it exists for callers that used the function directly."""
    ops.require_device(pred_boxes)
    rows = target.size(0)
    if pred_boxes.shape != (rows * num_anchors * nH * nW, 4):
        raise ValueError("pred_boxes must be (rows*A*H*W, 4)")
    if len(anchors) // num_anchors != 2:
        raise NotImplementedError("anchor_step != 2 is not used by any shipped cfg")
    tr = np.ascontiguousarray(target.detach().cpu().reshape(rows, -1).numpy(), dtype=np.float64)
    _validate_targets(tr)
    dev = pred_boxes.device
    pb = pred_boxes.detach().contiguous().float()
    tgt = torch.from_numpy(tr).to(dev)
    out = torch.empty((9, rows, num_anchors, nH, nW), dtype=torch.float32, device=dev)
    stats = torch.empty(16, dtype=torch.float64, device=dev)
    anc = np.ctypeslib.as_ctypes(np.asarray(anchors, dtype=np.float64)[:2 * num_anchors].copy())
    check(lib().fsd_region_build_targets(pb.data_ptr(), tgt.data_ptr(), out.data_ptr(), stats.data_ptr(), rows,
                                         num_anchors, nH, nW, tr.shape[1], anc, float(noobject_scale),
                                         float(object_scale), float(sil_thresh), int(seen), int(cfg.max_boxes),
                                         torch.cuda.current_stream().cuda_stream), "fsd_region_build_targets")
    s = stats.tolist()
    if s[9] > 0:
        raise ValueError("build_targets: %d ground-truth entries had no matching anchor or left the grid" % int(s[9]))
    return (int(s[6]), int(s[7])) + tuple(out[k] for k in range(9))


def select_classes(pred, tgt, ids):
    """Rows of `pred` whose label is one of `ids` (not the first), restricted to those columns, with the labels
    renumbered by position in `ids` (region_loss.py:369-378; only referenced from commented-out code there)."""
    t = tgt.detach().cpu().numpy()
    new_tgt = np.max(np.stack([(t == d) * i for i, d in enumerate(ids)]), axis=0)
    idx = np.argwhere(new_tgt > 0).reshape(-1)
    sel = torch.as_tensor(idx, dtype=torch.long, device=pred.device)
    return pred[sel][:, list(ids)], new_tgt[idx]


def _validate_targets(rows, n_labels=None):
    """The reference raises (math.log domain error / bad index) on these; say why instead.
    n_labels: number of valid class ids (rows per image for the meta loss, num_classes for v1); a label outside
    [0, n_labels) makes the reference's CrossEntropyLoss raise, so it is refused here on the host rather than being
    counted by a device-side statistic nobody reads when verbose=False."""
    cx = rows[:, 1::5]
    w, h = rows[:, 3::5], rows[:, 4::5]
    n = min(cx.shape[1], w.shape[1], h.shape[1])
    live = np.cumprod(cx[:, :n] != 0, axis=1).astype(bool)
    if np.any(live & ((w[:, :n] <= 0) | (h[:, :n] <= 0))):
        raise ValueError("region loss target has a box with non-positive width/height")
    cy = rows[:, 2::5][:, :n]
    if np.any(live & ((cx[:, :n] < 0) | (cx[:, :n] >= 1) | (cy < 0) | (cy >= 1))):
        raise ValueError("region loss target has a box centre outside [0, 1)")
    if n_labels is not None:
        cls = rows[:, 0::5][:, :n]
        if np.any(live & ((cls < 0) | (np.floor(cls) >= n_labels))):
            raise ValueError("region loss target carries a class id outside [0, %d)" % n_labels)


class _RegionLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target_dev, keep_dev, mod, rows_per_image, softmax_over_rows, zero_tcls, dbg):
        ops.require_device(output)
        out = output.contiguous()
        rows, chans, H, W = out.shape
        A, Cn = mod.num_anchors, mod.num_classes
        if chans != A * (5 + Cn):
            raise ValueError("output has %d channels, expected %d" % (chans, A * (5 + Cn)))
        L = lib()
        ws_bytes = L.fsd_region_loss_workspace_bytes(rows, rows_per_image, A, H, W)
        ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=out.device)
        grad = torch.empty_like(out)
        loss = torch.empty((), dtype=torch.float32, device=out.device)
        anchors = (np.ctypeslib.as_ctypes(np.asarray(mod.anchors, dtype=np.float64)[:2 * A].copy())
                   if int(mod.anchor_step) == 2 else None)
        if anchors is None:
            raise NotImplementedError("anchor_step != 2 is not used by any shipped cfg")
        check(L.fsd_region_loss_fwd_bwd(
            out.data_ptr(), target_dev.data_ptr(), keep_dev.data_ptr(), grad.data_ptr(), loss.data_ptr(),
            ws.data_ptr(), ws_bytes, rows, rows_per_image, A, Cn, H, W, target_dev.shape[1], anchors,
            float(mod.coord_scale), float(mod.noobject_scale), float(mod.object_scale), float(mod.class_scale),
            float(mod.thresh), int(mod.seen), int(cfg.max_boxes), int(softmax_over_rows), int(zero_tcls),
            0 if dbg is None else dbg.data_ptr(), torch.cuda.current_stream().cuda_stream),
            "fsd_region_loss_fwd_bwd")
        ctx.save_for_backward(grad)
        mod._stats = ws[:16]
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None, None, None


class _RegionBase(nn.Module):
    verbose = True       # print the reference's per-batch stats line (costs one device->host sync)

    def __init__(self, num_classes=0, anchors=(), num_anchors=1):
        super(_RegionBase, self).__init__()
        self.num_classes = num_classes
        self.anchors = list(anchors)
        self.num_anchors = num_anchors
        self.anchor_step = len(self.anchors) // num_anchors if num_anchors else 0
        self.coord_scale = 1
        self.noobject_scale = 1
        self.object_scale = 5
        self.class_scale = 1
        self.thresh = 0.6
        self.seen = 0
        self._stats = None
        self.debug_targets = False      # tests: also emit build_targets' nine tensors
        self.last_targets = None
        self.last_keep = None
        self.neg_counts = None          # data parallelism: whole-batch (n_pos, n_rows) reducer, owned by dp.EpisodeTrainer

    def __getstate__(self):
        # a trainer's collective reducer is process state, not model state: a pickled / deep-copied model comes back local
        state = dict(self.__dict__)
        state["neg_counts"] = None
        return state

    def stats(self):
        """(dict) loss parts + nGT / nCorrect / nProposals of the last call (synchronises)."""
        s = self._stats.tolist()
        if s[9] > 0:
            raise ValueError("region loss: %d ground-truth entries had no matching anchor, left the grid or "
                             "carried a class id outside the episode" % int(s[9]))
        return dict(loss_x=s[0], loss_y=s[1], loss_w=s[2], loss_h=s[3], loss_conf=s[4], loss_cls=s[5],
                    nGT=int(s[6]), nCorrect=int(s[7]), nProposals=int(s[8]))

    def _run(self, output, target_rows_host, rows_per_image, softmax_over_rows, zero_tcls):
        rows = output.shape[0]
        tr = np.ascontiguousarray(target_rows_host, dtype=np.float64)
        if tr.shape[0] != rows:
            raise ValueError("target has %d rows, output has %d" % (tr.shape[0], rows))
        _validate_targets(tr, None if zero_tcls else (rows_per_image if softmax_over_rows else self.num_classes))
        # the whole-batch ratio is a collective: only the training step's call takes part in it (see the note at the top)
        counts = getattr(self, "neg_counts", None) if (self.training and torch.is_grad_enabled()) else None
        keep = neg_filter_indices(tr, counts)
        keep_map = np.full(rows, -1, np.int32)
        keep_map[keep] = np.arange(len(keep), dtype=np.int32)
        dev = output.device
        # Asynchronous copies out of pinned staging slots on the current stream (streams.upload): a pageable copy would block
        # the host until the forward pass has run (it then queues loss + backward against an idle GPU).
        target_dev = streams.upload(tr, dev)
        keep_dev = streams.upload(keep_map, dev)
        dbg = None
        if self.debug_targets:
            dbg = torch.zeros((9, rows, self.num_anchors, output.shape[2], output.shape[3]),
                              dtype=torch.float32, device=dev)
        loss = _RegionLossFn.apply(output, target_dev, keep_dev, self, rows_per_image, softmax_over_rows,
                                   zero_tcls, dbg)
        self.last_keep = keep
        if dbg is not None:
            self.last_targets = dbg[:, :len(keep)]
        if self.verbose:
            s = self.stats()
            print("%d: nGT %d, recall %d, proposals %d, loss: x %f, y %f, w %f, h %f, conf %f, cls %f, total %f" % (
                self.seen, s["nGT"], s["nCorrect"], s["nProposals"], s["loss_x"], s["loss_y"], s["loss_w"],
                s["loss_h"], s["loss_conf"], s["loss_cls"], float(loss.detach())))
        return loss


def _host_rows(target):
    t = target.detach()
    if t.is_cuda:
        t = t.cpu()
    return t.reshape(-1, t.shape[-1]).numpy()


class RegionLoss(_RegionBase):
    """YOLOv2 region loss with the per-cell softmax over `num_classes` (reference region_loss.py:134-232).
    output (B, A*(5+C), H, W), target (B, 250) or (B, N, 250) -> scalar (sum over the batch)."""

    def forward(self, output, target):
        return self._run(output, _host_rows(target), 1, False, bool(cfg.metayolo))


class RegionLossV2(_RegionBase):
    """Region loss + softmax classification across the N meta-inputs (reference region_loss.py:234-366).
    output (B*N, A*6, H, W) with rows ordered b*N+n, target (B, N, 250) -> scalar."""

    def __init__(self, num_classes=0, anchors=(), num_anchors=1):
        super(RegionLossV2, self).__init__(num_classes, anchors, num_anchors)
        print("class_scale", self.class_scale)

    def forward(self, output, target):
        if target.dim() != 3:
            raise ValueError("RegionLossV2 expects a (batch, n_classes, 250) target")
        bs, cs = target.shape[0], target.shape[1]
        if output.shape[0] != bs * cs:
            raise ValueError("output rows %d != batch %d x classes %d" % (output.shape[0], bs, cs))
        return self._run(output, _host_rows(target), cs, True, False)
