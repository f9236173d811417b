"""Executes a parsed darknet .cfg network on the HIP kernels, in NHWC, with layer fusion.

The reference walks `self.blocks` and calls one PyTorch module per block (darknet_meta.py:130-195).
Here the same walk drives a small set of fused device ops:

  [convolutional](+BN)(+leaky)(+[maxpool])  ->  implicit-GEMM conv (BN partial sums in its epilogue)
                                                 -> statistics finalise -> affine+leaky+pool pass
  [route] a            ->  alias;   [route] a,b  ->  producers write channel slices of one buffer
  [reorg]              ->  space-to-depth straight into its slice of the concat buffer
  dynamic [convolutional] + 1x1 head  ->  ONE GEMM with the reweighting vectors folded into the
                                          head weights (the (B*N,1024,H,W) tensor never exists)
  [globalmax] / [globalavg]  ->  per-channel max / mean over the support feature map

A forward pass records a tape (views + per-channel statistics) that `backward` replays in reverse.
"""
import os

import torch

from . import ops, streams
from .ops import View


def is_dynamic(block):
    return "dynamic" in block and int(block["dynamic"]) == 1


def _slope(activation):
    if activation == "leaky":
        return 0.1
    if activation == "relu":
        return 0.0
    if activation == "linear":
        return 1.0
    raise NotImplementedError("activation %r" % activation)


_WEIGHT_EPOCH = [0]


def bump_weight_epoch():
    """Call after updating parameters through raw pointers (fused SGD kernel): torch's version
    counter does not see such writes, so packed weight copies must be invalidated explicitly."""
    _WEIGHT_EPOCH[0] += 1


_STATS_EPOCH = [0]


def bump_stats_epoch():
    """A training-mode BatchNorm finalize rewrote running_mean / running_var through raw pointers (no version bump):
    eval-mode folds of those statistics (and hipGraphs captured over them) must be rebuilt."""
    _STATS_EPOCH[0] += 1


class _WeightCache(object):
    """Packed (K-major) copies of conv weights, refreshed when the parameter changes.
    Entries: key -> [tag, packed, weight]."""

    def __init__(self):
        self._store = {}

    @staticmethod
    def _tag(w):
        return (w.data_ptr(), w._version, _WEIGHT_EPOCH[0])

    def _build(self, w, mode, dtype, old=None):
        # `old`: the stale packed copy of the same parameter, rewritten in place (a step then re-packs ~2.9 GB of operands
        # without a single allocator call; every reader of the old copy is stream-ordered before the optimizer step)
        if dtype in ("wino2", "wino4"):
            return ops.pack_weight_wino(w.detach(), mode, int(dtype[4]), out=old)
        return ops.pack_weight(w.detach(), mode, dtype, out=old)

    def refresh(self, w, epoch):
        """Rewrite every fp32-family packed copy of `w` (plain, Winograd; forward and data-gradient operand) from its CURRENT
        value, in place, on the current stream, and tag it valid for weight epoch `epoch` -- the trainer does this right after
        the optimizer kernel of w's gradient bucket, while the backward pass is still running (dp.EpisodeTrainer).  The bf16
        copies are written by the fused optimizer kernel itself (mark_fresh).  -> number of copies rewritten."""
        n = 0
        for key, ent in self._store.items():
            if key[0] != id(w) or key[2] == "bf16" or ent[2] is not w:
                continue
            ent[1] = self._build(w, key[1], key[2], ent[1])
            ent[0] = (w.data_ptr(), w._version, epoch)
            n += 1
        return n

    def bf16_pair(self, w):
        """The kept (forward, data-gradient) bf16 operand buffers of `w`, or None if this network has not packed it yet."""
        a, b = self._store.get((id(w), 0, "bf16")), self._store.get((id(w), 1, "bf16"))
        if a is None or b is None or a[2] is not w or b[2] is not w:
            return None
        return a[1], b[1]

    def mark_fresh(self, w, dtype="bf16"):
        """The packed copies of `w` were just rewritten from its current value by someone else (the fused optimizer + pack
        kernel, dp.EpisodeTrainer): they are valid for the weight epoch that starts now."""
        tag = self._tag(w)
        for mode in (0, 1):
            ent = self._store.get((id(w), mode, dtype))
            if ent is not None:
                ent[0] = tag

    def get(self, w, mode=0, dtype="f32"):
        key = (id(w), mode, dtype)
        tag = self._tag(w)
        hit = self._store.get(key)
        if dtype == "bf16" and (hit is None or hit[0] != tag):
            # both bf16 operands (forward, data gradient) in ONE pass over the weight, into buffers that are kept
            other = self._store.get((id(w), 1 - mode, dtype))
            bufs = None
            if hit is not None and other is not None:
                bufs = (hit[1], other[1]) if mode == 0 else (other[1], hit[1])
            pair = ops.pack_weight_bf16_pair(w.detach(), bufs)
            self._store[(id(w), 0, dtype)] = [tag, pair[0], w]
            self._store[(id(w), 1, dtype)] = [tag, pair[1], w]
            return pair[mode]
        if hit is None or hit[0] != tag:
            # in-place only for the SAME parameter object at the same address (hit[2] is w): a folded eval weight that was
            # rebuilt, or a parameter that moved, gets a fresh buffer
            old = hit[1] if (hit is not None and hit[2] is w and hit[0][0] == tag[0] and not torch.cuda.is_current_stream_capturing()) else None
            hit = [tag, self._build(w, mode, dtype, old), w]
            self._store[key] = hit
        return hit[1]



FOLD_EVAL_BN = True     # inference: BatchNorm folded into the conv operands, leaky in the conv epilogue (Network._conv_eval)
# fp32 training: a conv + BatchNorm + leaky layer whose only reader is the next convolution hands over its raw output and the
# reader forms the activation on load (Network._defers_to_consumer).  DEFER_ACTIVATION = False: every activation is materialised.
DEFER_ACTIVATION = True


class Network(object):
    """One cfg network (detector or reweighting net) bound to its nn.ModuleList of parameters."""

    def __init__(self, blocks, models):
        self.blocks = blocks
        self.models = models
        self.layers = blocks[1:]
        self.cache = _WeightCache()
        self._folded = {}             # id(conv.weight) -> (tag, folded weight, folded bias): eval-mode BatchNorm folds
        # "f32": fp32 activations and results (GEMMs: fp32-accurate split arithmetic or the native fp32 MFMA,
        # ops.f32_gemm_mode).  "bf16" (BASELINE configs[2] / [4]): activations and their gradients are
        # STORED in HBM as bfloat16, convolutions run bf16 x bf16 -> fp32 on the bf16 matrix cores, BatchNorm statistics come
        # from the fp32 accumulators; loss, parameter gradients, master weights and the optimizer stay fp32.
        self.compute_dtype = "f32"
        self.fallback_convs = 0       # bf16 mode: convolutions whose channel counts the bf16 kernel does not take
        # static analysis: who is read by a [route], and which producers write into a concat buffer
        self.route_src = {}
        self.concat_of = {}
        widths = []
        width = int(blocks[0].get("channels", 3))
        for ind, blk in enumerate(self.layers):
            kind = blk["type"]
            if kind == "convolutional":
                width = int(blk["filters"])
            elif kind == "reorg":
                width *= int(blk["stride"]) ** 2
            elif kind == "route":
                src = [int(v) if int(v) > 0 else int(v) + ind for v in blk["layers"].split(",")]
                self.route_src[ind] = src
                width = sum(widths[s] for s in src)
                if len(src) == 2:
                    off = 0
                    for s in src:
                        if s in self.concat_of:
                            raise NotImplementedError("a layer feeding two concatenating routes")
                        self.concat_of[s] = (ind, off, width)
                        off += widths[s]
                elif len(src) != 1:
                    raise NotImplementedError("route with %d inputs" % len(src))
            widths.append(width)
        self.widths = widths
        self.tapped = set(s for src in self.route_src.values() for s in src)

    # ---- helpers ---------------------------------------------------------------------------
    @property
    def act_dtype(self):
        return torch.bfloat16 if self.compute_dtype == "bf16" else torch.float32

    def _dest(self, ind, B, H, W, C, dev, bufs):
        """Output view for layer `ind`: a slice of its route's concat buffer, or a fresh tensor."""
        if ind in self.concat_of:
            route, off, total = self.concat_of[ind]
            if route not in bufs:
                bufs[route] = ops.new_view(B, H, W, total, dev, dtype=self.act_dtype)
            big = bufs[route]
            if (big.B, big.H, big.W) != (B, H, W):
                raise NotImplementedError("route over feature maps of different size (maybe_repeat)")
            return View(big.t, B, H, W, C, off)
        return ops.new_view(B, H, W, C, dev, dtype=self.act_dtype)

    def _defers_to_consumer(self, ind, y, cout, training):
        """fp32 training pass: may layer `ind` (conv + BatchNorm + leaky, no pool) hand its RAW output on, the activation being
        formed by the consumer on load?  Only when the consumer is exactly one convolution that takes a deferred view: the
        next layer, an F(4x4) Winograd conv or a direct fp32 conv over whole 32-channel chunks, and nothing else (no route)
        reads this layer."""
        if not DEFER_ACTIVATION or self.compute_dtype != "f32" or not (training or self._record) or y.bf16:
            return False
        if ind in self.tapped or ind in self.concat_of or ind + 1 >= len(self.layers):
            return False
        nxt = self.layers[ind + 1]
        if nxt["type"] != "convolutional" or is_dynamic(nxt) or int(nxt.get("stride", 1)) != 1:
            return False
        k2, cout2 = int(nxt["size"]), int(nxt["filters"])
        if k2 not in (1, 3) or (k2 == 3 and not int(nxt.get("pad", 0))) or cout % 32:
            return False
        wino = ops.wino_tile(cout, cout2, k2, y.H, y.W)
        return wino == 4 or (wino == 0 and k2 == 1)       # (the direct 3x3 weight gradient reads a materialised x)

    def _conv_any(self, xv, conv, cout, k, bias, out, bn_partial):
        """One non-Winograd, non-first-layer convolution in the network's storage mode -> (y view, partial sums)."""
        if self.compute_dtype != "bf16":
            wp = self.cache.get(conv.weight, 0, "f32")
            return ops.conv2d(xv, wp, cout, k, bias=bias, out=out, bn_partial=bn_partial, cin_true=conv.weight.shape[1])
        if xv.bf16 and xv.C % 32 == 0 and cout % 2 == 0 and xv.c0 % 8 == 0:
            return ops.conv2d(xv, self.cache.get(conv.weight, 0, "bf16"), cout, k, bias=bias, out=out, bn_partial=bn_partial)
        # channel counts outside the bf16 kernel (no shipped cfg has any past the first layer): the same arithmetic --
        # bf16-rounded operands, fp32 accumulation, bf16 result -- on the fp32 kernel
        self.fallback_convs += 1
        xf = ops.cast_view(xv, torch.float32)
        wr = conv.weight.detach().to(torch.bfloat16).float()
        yf, partial = ops.conv2d(xf, ops.pack_weight(wr, 0, "f32"), cout, k, bias=bias, bn_partial=bn_partial,
                                 cin_true=conv.weight.shape[1])
        y = ops.cast_view(yf, torch.bfloat16, out=out)
        return y, partial

    def _fold_bn(self, conv, bn):
        """Eval-mode BatchNorm as part of the convolution: w' = w * gamma / sqrt(var + eps) per output channel,
        bias' = beta - mean * gamma / sqrt(var + eps) (+ the conv's own bias scaled).  Rebuilt when any of the five
        tensors changes (versions / the weight epoch of raw-pointer updates)."""
        tag = tuple(t._version for t in (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)) + \
            (conv.weight.data_ptr(), _WEIGHT_EPOCH[0], _STATS_EPOCH[0])
        hit = self._folded.get(id(conv.weight))
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
                wf = (conv.weight.detach() * scale.view(-1, 1, 1, 1)).contiguous()
                bf = bn.bias.detach() - bn.running_mean.detach() * scale
                if conv.bias is not None:
                    bf = bf + conv.bias.detach() * scale
                bf = bf.contiguous()
            if hit is not None:          # keep the tensor objects (the packed-operand cache is keyed by them)
                hit[1].copy_(wf)
                hit[2].copy_(bf)
                wf, bf = hit[1], hit[2]
            hit = (tag, wf, bf)
            self._folded[id(conv.weight)] = hit
        return hit[1], hit[2]

    def _conv_eval(self, ind, xv, conv, bn, k, cout, slope, pool, wino, first, bufs, tape):
        """Inference form of a conv + BatchNorm + leaky (+ max pool) block: one launch without a pool (the activation is
        in the conv epilogue), two with one -- instead of conv, statistics -> scale / shift, affine + leaky + pool."""
        dev = xv.t.device
        wf, bf = self._fold_bn(conv, bn)
        tapped = pool and ind in self.tapped
        full = self._dest(ind, xv.B, xv.H, xv.W, cout, dev, bufs) if (not pool or tapped) else None
        if first:
            # first-layer kernel: linear output, the (always present) pool pass applies the leaky
            y, _ = ops.conv3x3_c4(xv, wf, cout, bias=bf, out_dtype=self.act_dtype)
            act = ops.bn_act_pool(y, None, None, slope, 0, out=full) if full is not None else None
            z = ops.bn_act_pool(y, None, None, slope, pool, out=self._dest(ind + 1, xv.B, xv.H // 2 if pool == 1 else xv.H,
                                                                            xv.W // 2 if pool == 1 else xv.W, cout, dev, bufs)) \
                if pool else act
        else:
            if wino:
                act, _ = ops.conv3x3_wino(xv, self.cache.get(wf, 0, "wino%d" % wino), cout, bias=bf, out=full, tile=wino,
                                          slope=slope)
            elif self.compute_dtype == "bf16":
                act, _ = ops.conv2d(xv, self.cache.get(wf, 0, "bf16"), cout, k, bias=bf, out=full, slope=slope)
            else:
                act, _ = ops.conv2d(xv, self.cache.get(wf, 0, "f32"), cout, k, bias=bf, out=full, slope=slope,
                                    cin_true=conv.weight.shape[1])
            if pool:                     # leaky is monotonic: max pool of the activated tensor = activated max pool
                OH, OW = (xv.H // 2, xv.W // 2) if pool == 1 else (xv.H, xv.W)
                z = ops.bn_act_pool(act, None, None, 1.0, pool, out=self._dest(ind + 1, xv.B, OH, OW, cout, dev, bufs))
            else:
                z = act
        tape.append(dict(kind="conv_eval", ind=ind))
        return z, (act if tapped else None)

    def _conv(self, ind, blk, xv, training, pool, bufs, tape):
        seq = self.models[ind]
        conv = seq[0]
        bn = seq[1] if int(blk["batch_normalize"]) else None
        k = int(blk["size"])
        if int(blk["stride"]) != 1:
            raise NotImplementedError("strided convolution")
        if k not in (1, 3) or (k == 3 and not int(blk["pad"])):
            raise NotImplementedError("conv size=%d pad=%s" % (k, blk["pad"]))
        cout = int(blk["filters"])
        slope = _slope(blk["activation"])
        # Winograd layers always run the fp32 Winograd pipeline: in bf16 mode it is both faster and more accurate than
        # the bf16 direct kernel (4x fewer multiplications beat the bf16 MFMA rate of a staging-bound kernel)
        bf16 = self.compute_dtype == "bf16"
        wino = 0 if bf16 else ops.wino_tile(xv.C, cout, k, xv.H, xv.W)
        first = not wino and not xv.bf16 and ops.c4_bnfused_eligible(xv, cout, k)   # NHWC4 input: direct-operand first-layer kernel
        if (FOLD_EVAL_BN and bn is not None and not training and not self._record
                and (not bf16 or (xv.bf16 and xv.C % 32 == 0 and cout % 2 == 0 and xv.c0 % 8 == 0) or first)):
            if wino:
                wino = ops.wino_tile_inference(xv.C, cout, k, xv.H, xv.W, xv.B)
            return self._conv_eval(ind, xv, conv, bn, k, cout, slope, pool, wino, first, bufs, tape)
        wp = self.cache.get(conv.weight, 0, "wino%d" % wino) if wino else None
        dev = xv.t.device
        input_cast = False
        if bf16 and not first and not xv.bf16:      # a float network input that is not first-layer shaped
            xv = ops.cast_view(xv, torch.bfloat16)
            input_cast = True
        cin_true = conv.weight.shape[1]
        # Winograd layers keep their transformed input for the weight gradient when a backward pass will follow
        keep = [] if (wino and self._record) else None
        rec = dict(kind="conv", ind=ind, x=xv, conv=conv, bn=bn, k=k, cout=cout, slope=slope, pool=pool, wino_v=keep,
                   wino_tile=wino, input_cast=input_cast)
        if bn is None and slope == 1.0 and pool == 0:
            z = self._dest(ind, xv.B, xv.H, xv.W, cout, dev, bufs)
            if wino:
                ops.conv3x3_wino(xv, wp, cout, bias=conv.bias, out=z, keep_v=keep, tile=wino)
            elif first:
                ops.conv3x3_c4(xv, conv.weight, cout, bias=conv.bias, out=z)
            else:
                self._conv_any(xv, conv, cout, k, conv.bias, z, False)
            rec.update(x=xv, y=z, z=z, z_full=None)
            tape.append(rec)
            return z, None
        if wino:
            y, partial = ops.conv3x3_wino(xv, wp, cout, bias=None if bn is not None else conv.bias,
                                          bn_partial=bn is not None and training, keep_v=keep, tile=wino)
        elif first:
            y, partial = ops.conv3x3_c4(xv, conv.weight, cout, bias=None if bn is not None else conv.bias,
                                        bn_partial=bn is not None and training, out_dtype=self.act_dtype)
        else:
            y, partial = self._conv_any(xv, conv, cout, k, None if bn is not None else conv.bias, None,
                                        bn is not None and training)
        scale = shift = mean = invstd = None
        if bn is not None:
            scale, shift, mean, invstd = ops.bn_finalize(partial, xv.pixels, bn, training)
        if bn is not None and pool == 0 and self._defers_to_consumer(ind, y, cout, training):
            # the single consumer forms leaky(y * scale + shift) on load: no BatchNorm + leaky pass, no activation tensor
            z = View(y.t, y.B, y.H, y.W, y.C, y.c0, lazy=(scale, shift, slope))
            rec.update(x=xv, y=y, z=z, z_full=None, scale=scale, shift=shift, mean=mean, invstd=invstd, training=training)
            tape.append(rec)
            return z, None
        z_full = None
        if pool and ind in self.tapped:      # a [route] reads the activation before its maxpool
            z_full = ops.bn_act_pool(y, scale, shift, slope, 0,
                                     out=self._dest(ind, xv.B, xv.H, xv.W, cout, dev, bufs))
        OH, OW = (xv.H // 2, xv.W // 2) if pool == 1 else (xv.H, xv.W)
        z = ops.bn_act_pool(y, scale, shift, slope, pool,
                            out=self._dest(ind + 1 if pool else ind, xv.B, OH, OW, cout, dev, bufs))
        rec.update(x=xv, y=y, z=z, z_full=z_full, scale=scale, shift=shift, mean=mean, invstd=invstd,
                   training=training)
        tape.append(rec)
        return z, z_full

    # ---- forward ---------------------------------------------------------------------------
    def forward(self, inputs, dyn=None, training=False, tape=None, record=False):
        """inputs: list of NCHW tensors concatenated along channels (e.g. [metax, mask]).
        Returns an NCHW tensor (detector: (B*N, A*(5+C), G, G)) or (N, C, 1, 1) after [globalmax]."""
        self._record = bool(record)          # a backward pass will replay the tape: keep what it can reuse
        if tape is None:
            tape = []
        ops.require_device(*inputs)
        for t in inputs:
            if t.dim() != 4:
                raise ValueError("network inputs must be (B, C, H, W) tensors, got shape %s" % (tuple(t.shape),))
            if t.dtype != torch.float32:
                raise ValueError("network inputs must be float32 (got %s): the kernels read raw fp32 storage" % t.dtype)
            if (t.shape[0],) + tuple(t.shape[2:]) != (inputs[0].shape[0],) + tuple(inputs[0].shape[2:]):
                raise ValueError("inputs concatenated along channels must share batch and spatial size: %s vs %s"
                                 % (tuple(t.shape), tuple(inputs[0].shape)))
        if dyn is not None:
            for v in dyn:
                ops.require_device(v)
                if v.dtype != torch.float32:
                    raise ValueError("reweighting vectors must be float32 (got %s)" % v.dtype)
        B, _, H, W = inputs[0].shape
        ctot = sum(t.shape[1] for t in inputs)
        t0 = inputs[0]
        if (len(inputs) == 1 and t0.shape[1] == 4 and t0.stride(1) == 1 and (H * W > 1)
                and t0.is_contiguous(memory_format=torch.channels_last) and t0.data_ptr() % 16 == 0):
            # 16-byte NHWC4 pixels already (episode.DeviceAugmenter layout="nhwc4"): the first layer reads them in place
            x = View(t0.detach().permute(0, 2, 3, 1).reshape(B * H * W, 4), B, H, W, 4)
        elif len(inputs) == 1:
            x = ops.nchw_to_nhwc(inputs[0])
        else:
            x = ops.new_view(B, H, W, (ctot + 3) // 4 * 4, inputs[0].device)
            if x.C != ctot:
                ops.fill(x.t, 0.0)
            off = 0
            for t in inputs:
                ops.write_channels(t, x, off)
                off += t.shape[1]
        tape.append(dict(kind="input", x=x, n_in=ctot))
        outs = {}
        bufs = {}
        n_dyn = 0
        result = None
        skip = -1
        n_layers = len(self.layers)
        for ind, blk in enumerate(self.layers):
            if ind <= skip:
                continue
            kind = blk["type"]
            if kind == "convolutional" and is_dynamic(blk):
                nxt = self.layers[ind + 1] if ind + 1 < n_layers else None
                fusable = (nxt is not None and nxt["type"] == "convolutional" and not is_dynamic(nxt)
                           and int(nxt["size"]) == 1 and not int(nxt["batch_normalize"])
                           and nxt["activation"] == "linear" and blk["activation"] == "linear"
                           and not int(blk["batch_normalize"])
                           and all(b["type"] in ("region", "cost") for b in self.layers[ind + 2:])
                           and ind not in self.tapped and (ind + 1) not in self.tapped)
                if dyn is None or n_dyn >= len(dyn):
                    raise ValueError("dynamic convolution without reweighting vectors")
                if not fusable:
                    raise NotImplementedError("a dynamic conv that is not directly followed by the 1x1 "
                                              "detection head is outside the fused path")
                head = self.models[ind + 1][0]
                vec = dyn[n_dyn]
                n_dyn += 1
                if vec.shape[1] != x.C or tuple(vec.shape[2:]) != (1, 1):
                    raise ValueError("reweighting vectors %s do not match %d feature channels"
                                     % (tuple(vec.shape), x.C))
                n_cls, o_ch = vec.shape[0], head.weight.shape[0]
                streams.await_tensor(vec)        # vectors still in flight on the reweighting net's stream (Darknet.forward)
                w_op, b_eff, w_eff = ops.fold_reweight_head(head.weight.detach(), None if head.bias is None
                                                            else head.bias.detach(), vec.detach(), self.compute_dtype)
                if self.compute_dtype == "bf16" and not (x.bf16 and x.C % 32 == 0):
                    raise NotImplementedError("bf16 mode: the fused head needs a bf16 feature map with channels % 32 == 0")
                y, _ = ops.conv2d(x, w_op, n_cls * o_ch, 1, bias=b_eff, nchw_out=True)
                result = y.view(x.B * n_cls, o_ch, x.H, x.W)
                tape.append(dict(kind="head", x=x, head=head, dyn=vec, w_eff=w_eff, n_cls=n_cls, o_ch=o_ch))
                skip = ind + 1
                x = None
                continue
            if kind == "convolutional":
                pool = 0
                nxt = self.layers[ind + 1] if ind + 1 < n_layers else None
                if nxt is not None and nxt["type"] == "maxpool" and int(nxt["size"]) == 2:
                    pool = 1 if int(nxt["stride"]) == 2 else (2 if int(nxt["stride"]) == 1 else 0)
                z, z_full = self._conv(ind, blk, x, training, pool, bufs, tape)
                if pool:
                    outs[ind] = z_full
                    outs[ind + 1] = z
                    skip = ind + 1
                else:
                    outs[ind] = z
                x = z
            elif kind == "maxpool":
                size, stride = int(blk["size"]), int(blk["stride"])
                if size != 2 or stride not in (1, 2):
                    raise NotImplementedError("maxpool %dx%d/%d" % (size, size, stride))
                pool = 1 if stride == 2 else 2
                OH, OW = (x.H // 2, x.W // 2) if pool == 1 else (x.H, x.W)
                z = ops.bn_act_pool(x, None, None, 1.0, pool,
                                    out=self._dest(ind, x.B, OH, OW, x.C, x.t.device, bufs))
                tape.append(dict(kind="pool", x=x, z=z, pool=pool))
                outs[ind] = x = z
            elif kind == "reorg":
                s = int(blk["stride"])
                z = ops.reorg(x, s, out=self._dest(ind, x.B, x.H // s, x.W // s, x.C * s * s, x.t.device, bufs))
                tape.append(dict(kind="reorg", x=x, z=z, stride=s))
                outs[ind] = x = z
            elif kind == "route":
                src = self.route_src[ind]
                if len(src) == 1:
                    x = outs[src[0]]
                    if x is None:
                        raise RuntimeError("route reads layer %d whose output was fused away" % src[0])
                else:
                    big = bufs[ind]
                    x = View(big.t, big.B, big.H, big.W, self.widths[ind], 0)
                tape.append(dict(kind="route", src=src, z=x, src_views=[outs[s_] for s_ in src]))
                outs[ind] = x
            elif kind == "globalmax":
                vals, arg = ops.global_maxpool(x, want_argmax=True)
                tape.append(dict(kind="globalmax", x=x, arg=arg))
                result = vals.view(x.B, x.C, 1, 1)
                x = None
            elif kind in ("globalavg", "avgpool"):
                # reference create_network maps both to GlobalAvgPool2d (darknet_meta.py:281-286 -> pooling.py:29-45)
                vals = ops.global_avgpool(x)
                tape.append(dict(kind="globalavg", x=x))
                result = vals.view(x.B, x.C, 1, 1)
                x = None
            elif kind in ("region", "cost"):
                continue
            else:
                raise NotImplementedError("block type %r is outside the MI355X hot path" % kind)
        if result is None:
            if x is None:
                raise RuntimeError("network produced no output")
            result = ops.nhwc_to_nchw(x)
            tape.append(dict(kind="output", x=x))
        return result, tape
