"""`Darknet(darknet_cfg, learnet_cfg)`: the meta detector with the reference's interface
(darknet_meta.py:86-482) -- attributes blocks / learnet_blocks / models / learnet_models / loss /
width / height / anchors / num_anchors / anchor_step / num_classes / header / seen; methods forward,
meta_forward, detect_forward, print_network, create_network, load_weights, save_weights, is_dynamic.

The nn.Module tree only OWNS the parameters (same names and shapes as the reference, so
state_dicts and darknet .weights files interchange); the arithmetic runs in engine.Network on
the HIP kernels.  Gradients flow through two autograd nodes (reweighting net, detector) whose
backward replays the engine tape.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops, streams
from .cfg import (cfg, load_conv, load_conv_bn, load_fc, parse_cfg, print_cfg, save_conv, save_conv_bn,
                  save_fc)
from .dynamic_conv import dynamic_conv2d
from .engine import Network, is_dynamic
from .pooling import GlobalAvgPool2d, GlobalMaxPool2d, Split
from .region_loss import RegionLossV2


class ConvBlock(nn.Sequential):
    """conv(+bn)(+activation) container.  Standalone call: NCHW in, NCHW out, forward only."""

    def forward(self, x):
        conv = self[0]
        if getattr(conv, "is_first", None) is not None:          # dynamic conv: x is (features, vectors)
            return conv(x)
        bn = self[1] if len(self) > 1 and isinstance(self[1], nn.BatchNorm2d) else None
        act = self[len(self) - 1]
        slope = 0.1 if isinstance(act, nn.LeakyReLU) else 0.0 if isinstance(act, nn.ReLU) else 1.0
        k = conv.kernel_size[0]
        # the same limits engine.Network._conv enforces: stride 1, 1x1 or "same"-padded 3x3, no groups / dilation
        if (tuple(conv.stride) != (1, 1) or conv.kernel_size[0] != conv.kernel_size[1] or k not in (1, 3)
                or tuple(conv.padding) != ((k - 1) // 2,) * 2 or tuple(conv.dilation) != (1, 1) or conv.groups != 1):
            raise NotImplementedError("standalone ConvBlock: only stride-1 1x1 / same-padded 3x3 convolutions run on the "
                                      "HIP path (got kernel %s stride %s padding %s)"
                                      % (conv.kernel_size, conv.stride, conv.padding))
        if x.shape[1] != conv.in_channels:
            raise ValueError("ConvBlock expects %d input channels, got %d" % (conv.in_channels, x.shape[1]))
        with torch.no_grad():
            xv = ops.nchw_to_nhwc(x)
            wp = ops.pack_weight(conv.weight)
            training = self.training and bn is not None
            y, part = ops.conv2d(xv, wp, conv.out_channels, k, bias=conv.bias, bn_partial=training)
            if bn is not None:
                scale, shift, _, _ = ops.bn_finalize(part, xv.pixels, bn, training)
                y = ops.bn_act_pool(y, scale, shift, slope, 0)
            elif slope != 1.0:
                y = ops.bn_act_pool(y, None, None, slope, 0)
            return ops.nhwc_to_nchw(y)


class _Pool2x2(nn.Module):
    def __init__(self, mode):
        super(_Pool2x2, self).__init__()
        self.mode = mode

    def forward(self, x):
        with torch.no_grad():       # the kernels work on channel quads: zero-padded channels pool to zero, slice them off
            out = ops.nhwc_to_nchw(ops.bn_act_pool(ops.nchw_to_nhwc(x), None, None, 1.0, self.mode))
            return out if out.shape[1] == x.shape[1] else out[:, :x.shape[1]].contiguous()


class MaxPoolStride1(_Pool2x2):
    """2x2 max, stride 1, replicate padding right/bottom (darknet_meta.py:47-53)."""

    def __init__(self):
        super(MaxPoolStride1, self).__init__(2)


class MaxPool2x2(_Pool2x2):
    """nn.MaxPool2d(2, 2) equivalent (floor mode)."""

    def __init__(self):
        super(MaxPool2x2, self).__init__(1)


class Reorg(nn.Module):
    """Space to depth: out[b,(di*s+dj)*C+c,i,j] = x[b,c,s*i+di,s*j+dj]  (darknet_meta.py:55-74)."""

    def __init__(self, stride=2):
        super(Reorg, self).__init__()
        self.stride = stride

    def forward(self, x):
        assert x.dim() == 4 and x.shape[2] % self.stride == 0 and x.shape[3] % self.stride == 0
        with torch.no_grad():
            out = ops.nhwc_to_nchw(ops.reorg(ops.nchw_to_nhwc(x), self.stride))
            c, s2 = x.shape[1], self.stride * self.stride
            if out.shape[1] != c * s2:      # C was padded to a channel quad: drop the padding inside every (di, dj) group
                b, cp, h, w = out.shape[0], out.shape[1] // s2, out.shape[2], out.shape[3]
                out = out.view(b, s2, cp, h, w)[:, :, :c].reshape(b, s2 * c, h, w)
            return out


class EmptyModule(nn.Module):
    def forward(self, x):
        return x


class Reshape(nn.Module):
    """View (B, ...) as (B, *shape) (darknet_meta.py:38-44; no shipped cfg instantiates it)."""

    def __init__(self, *args):
        super(Reshape, self).__init__()
        self.shape = args

    def forward(self, x):
        return x.view(x.size(0), *self.shape)


def maybe_repeat(x1, x2):
    """Batch-align two tensors by repeating the smaller one per class (darknet_meta.py:16-35)."""
    n1, n2 = x1.size(0), x2.size(0)
    if n1 == n2:
        return x1, x2
    if n1 < n2:
        assert n2 % n1 == 0
        return x1.repeat_interleave(n2 // n1, dim=0), x2
    assert n1 % n2 == 0
    return x1, x2.repeat_interleave(n1 // n2, dim=0)


def _flat_params(models):
    return [p for p in models.parameters()]


def build_modules(blocks, region_cls):
    """cfg blocks -> nn.ModuleList owning the parameters (reference create_network, darknet_meta.py:208-353)."""
    models = nn.ModuleList()
    prev_filters = 3
    out_filters = []
    conv_id = 0
    dynamic_count = 0
    for block in blocks:
        kind = block["type"]
        if kind in ("net", "learnet"):
            prev_filters = int(block["channels"])
            continue
        if kind == "convolutional":
            conv_id += 1
            bn_on = int(block["batch_normalize"])
            filters = int(block["filters"])
            k = int(block["size"])
            pad = (k - 1) // 2 if int(block["pad"]) else 0
            want_bias = bool(int(block["bias"])) if "bias" in block else True
            groups = int(block["groups"]) if "groups" in block else 1
            if groups != 1:
                raise NotImplementedError("grouped convolution")
            if is_dynamic(block):
                partial = int(block["partial"]) if "partial" in block else None
                conv = dynamic_conv2d(dynamic_count == 0, partial=partial)(
                    prev_filters, filters, k, int(block["stride"]), pad, groups=groups, bias=False)
                dynamic_count += 1
            else:
                conv = nn.Conv2d(prev_filters, filters, k, int(block["stride"]), pad,
                                 bias=False if bn_on else want_bias)
            seq = ConvBlock()
            seq.add_module("conv{0}".format(conv_id), conv)
            if bn_on:
                seq.add_module("bn{0}".format(conv_id), nn.BatchNorm2d(filters))
            if block["activation"] == "leaky":
                seq.add_module("leaky{0}".format(conv_id), nn.LeakyReLU(0.1, inplace=True))
            elif block["activation"] == "relu":
                seq.add_module("relu{0}".format(conv_id), nn.ReLU(inplace=True))
            prev_filters = filters
            models.append(seq)
        elif kind == "maxpool":
            if int(block["size"]) != 2:
                raise NotImplementedError("maxpool size %s" % block["size"])
            models.append(MaxPool2x2() if int(block["stride"]) > 1 else MaxPoolStride1())
        elif kind == "reorg":
            stride = int(block["stride"])
            prev_filters = stride * stride * prev_filters
            models.append(Reorg(stride))
        elif kind == "route":
            ind = len(models)
            layers = [int(i) if int(i) > 0 else int(i) + ind for i in block["layers"].split(",")]
            prev_filters = sum(out_filters[i] for i in layers)
            models.append(EmptyModule())
        elif kind == "shortcut":
            prev_filters = out_filters[len(models) - 1]
            models.append(EmptyModule())
        elif kind == "region":
            loss = region_cls()
            loss.anchors = [float(i) for i in block["anchors"].split(",")]
            loss.num_classes = int(block["classes"])
            loss.num_anchors = int(block["num"])
            loss.anchor_step = len(loss.anchors) // loss.num_anchors
            loss.object_scale = float(block["object_scale"])
            loss.noobject_scale = float(block["noobject_scale"])
            loss.class_scale = float(block["class_scale"])
            loss.coord_scale = float(block["coord_scale"])
            models.append(loss)
        elif kind == "globalmax":
            models.append(GlobalMaxPool2d())
        elif kind in ("globalavg", "avgpool"):
            models.append(GlobalAvgPool2d())
        elif kind == "split":
            splits = [int(sz) for sz in block["splits"].split(",")]
            prev_filters = splits[-1]
            models.append(Split(splits))
        else:
            raise NotImplementedError("block type %r is outside the MI355X hot path" % kind)
        out_filters.append(prev_filters)
    ops.install_bn_counter_hooks(models)       # BatchNorm batch counters live on the host between looks (ops.bn_finalize)
    return models



_GRAD_MODE = [True]      # torch.is_grad_enabled() of the caller of _NetFn.apply (set by _apply_net)


def _apply_net(*args):
    _GRAD_MODE[0] = torch.is_grad_enabled()
    return _NetFn.apply(*args)


class _NetFn(torch.autograd.Function):
    """One engine.Network as a single autograd node: inputs (activations, optional reweighting
    vectors, parameters) -> output; backward replays the tape on the HIP kernels.

    side: None, or the name of the side stream this network runs on (streams.py; the reweighting net runs on
    "meta" beside the detector).  defer: the caller guarantees that the output's first reader calls
    streams.await_tensor (the fused head does); otherwise the current stream waits for the side stream right away."""

    @staticmethod
    def forward(ctx, net, training, n_inputs, has_dyn, side, defer, *tensors):
        inputs = list(tensors[:n_inputs])
        dyn = [tensors[n_inputs]] if has_dyn else None
        # a backward pass can only follow if the caller runs with autograd on (inside forward() grad mode is always off,
        # and under torch.no_grad() needs_input_grad still reports the parameters)
        record = any(ctx.needs_input_grad) and _GRAD_MODE[0]
        ctx.side = None
        if not has_dyn and inputs[0].is_cuda:
            streams.clear_early(inputs[0].device)      # a producer of vectors starts: entries of earlier forwards are stale
        if side is not None and streams.ENABLED and streams.META and inputs[0].is_cuda:
            main = torch.cuda.current_stream()
            s = streams.side(inputs[0].device, side)
            s.wait_stream(main)               # inputs, and the weights the optimizer just updated on the main stream
            with torch.cuda.stream(s):
                out, tape = net.forward(inputs, dyn=dyn, training=training, record=record)
                done = s.record_event()
            streams.keep_alive(s, *inputs)
            streams.keep_alive(main, out)
            streams.mark_side_output(out)
            if defer:
                streams.publish(out, done)
            else:
                main.wait_event(done)
            ctx.side = side
        else:
            out, tape = net.forward(inputs, dyn=dyn, training=training, record=record)
        ctx.net, ctx.tape, ctx.n_inputs, ctx.has_dyn = net, tape, n_inputs, has_dyn
        ctx.params = tensors[n_inputs + (1 if has_dyn else 0):]
        ctx.early_result = None
        if defer and record:                  # forward(): the vectors have one consumer, the detector of this very call
            streams.register_early(out, ctx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import backward as bw
        grad_out = grad_out.contiguous()
        early = ctx.early_result
        if ctx.tape is None and early is None:
            raise RuntimeError("backward through this network a second time: the tape of its forward pass was released by the "
                               "first backward (like autograd's saved tensors without retain_graph); run the forward again")
        if early is not None and early[0].data_ptr() == grad_out.data_ptr() and tuple(early[0].shape) == tuple(grad_out.shape):
            # the detector's sweep already ran this network's backward on this very gradient (backward.run_early)
            grads = early[1]
            ctx.early_result = None
            if ctx.side is not None:
                main = torch.cuda.current_stream()
                main.wait_stream(streams.side(grad_out.device, ctx.side))
                streams.keep_alive(main, *[g for g in grads["params"] if g is not None])
        elif early is not None:
            # the sweep ran early on another tensor than autograd now presents (a hook or a second consumer changed the
            # gradient): its results may already be inside a running all-reduce -- refuse rather than overwrite them
            raise RuntimeError("the reweighting vectors of Darknet.forward() received a gradient that differs from the one "
                               "their early backward sweep consumed; call meta_forward() / detect_forward() separately")
        elif ctx.side is not None:
            main = torch.cuda.current_stream()
            s = streams.side(grad_out.device, ctx.side)
            if not streams.await_tensor(grad_out, s):     # published by the head's backward: start as soon as it is there
                s.wait_stream(main)
            with torch.cuda.stream(s):
                grads = bw.run(ctx.net, ctx.tape, grad_out, ctx.params, wgrad_stream=False)
            streams.keep_alive(s, grad_out)
            main.wait_stream(s)               # gradients (flat buffer or tensors) are complete for whatever follows
            streams.keep_alive(main, *[g for g in grads["params"] if g is not None])
        else:
            grads = bw.run(ctx.net, ctx.tape, grad_out, ctx.params)
        if ops.GRAD_HOOK is not None:
            ops.GRAD_HOOK()
        # Release the tape NOW, as autograd releases saved tensors after a backward without retain_graph: it holds every kept
        # activation of the step (y, x, V: ~8 GB at the headline episode).  A caller that keeps `loss` (or the output) alive --
        # `loss = step()` in a loop, a logging list -- would otherwise keep the whole previous step's activations alive through
        # the next forward: twice the memory, and a caching allocator that needs fresh segments whenever the overlap changes
        # (tools/alloc_trace.py: this was the source of the hipMallocs inside bench.py's timed region).
        ctx.tape = None
        head = [None] * ctx.n_inputs + ([grads["dyn"]] if ctx.has_dyn else [])
        return (None, None, None, None, None, None) + tuple(head) + tuple(grads["params"])


class Darknet(nn.Module):
    def __init__(self, darknet_file, learnet_file):
        super(Darknet, self).__init__()
        self.blocks = darknet_file if isinstance(darknet_file, list) else parse_cfg(darknet_file)
        self.learnet_blocks = learnet_file if isinstance(learnet_file, list) else parse_cfg(learnet_file)
        self.models = self.create_network(self.blocks)
        self.learnet_models = self.create_network(self.learnet_blocks)
        self.loss = self.models[len(self.models) - 1]
        self.width = int(self.blocks[0]["width"])
        self.height = int(self.blocks[0]["height"])
        if self.blocks[len(self.blocks) - 1]["type"] == "region":
            self.anchors = self.loss.anchors
            self.num_anchors = self.loss.num_anchors
            self.anchor_step = self.loss.anchor_step
            self.num_classes = self.loss.num_classes
        self.header = torch.IntTensor([0, 0, 0, 0])
        self.seen = 0
        self._det = Network(self.blocks, self.models)
        self._meta = Network(self.learnet_blocks, self.learnet_models)
        # Inference at the reference's validation batch size (valid_ensemble.py: 2 images per batch) is bound by the ~100
        # kernel launches of a forward pass, not by the GPU (1.4 ms of host enqueue per batch against 0.3 ms of kernels):
        # with inference_graphs the eval-mode detect_forward of a (shape, vectors, weights) combination is captured once
        # into a hipGraph and replayed.  Opt-in (or FSD_INFER_GRAPHS=1): the returned tensor is a fresh copy, but capture
        # needs a warm-up pass and pins the activation memory of every captured shape.
        self.inference_graphs = os.environ.get("FSD_INFER_GRAPHS", "0") == "1"
        self._graphs = {}

    def set_compute_dtype(self, dtype):
        """"f32" (default: fp32 storage and results; GEMM arithmetic per ops.f32_gemm_mode) or "bf16" (conv operands in bf16, fp32
        accumulate: BASELINE C3/C5)."""
        if dtype not in ("f32", "bf16"):
            raise ValueError("compute dtype must be 'f32' or 'bf16'")
        self._det.compute_dtype = self._meta.compute_dtype = dtype
        return self

    # ---- forward ---------------------------------------------------------------------------
    def meta_forward(self, metax, mask, _defer=False):
        """Support images (+ masks) -> list of reweighting vectors [(N, C, 1, 1)].
        _defer (internal, used by forward()): the vectors are still being computed on the "meta" side stream when this
        returns and their only reader -- detect_forward's fused head -- waits for them (streams.py)."""
        if int(self.learnet_blocks[0]["feat_layer"]) != 0:
            raise NotImplementedError("feat_layer != 0 is not used by any shipped cfg")
        inputs = [metax, mask] if cfg.metain_type in (2, 3) else [metax]
        if mask is None:          # RGB + mask already interleaved per pixel (episode.DeviceAugmenter layout="nhwc4")
            inputs = [metax]
        params = _flat_params(self.learnet_models)
        out = _apply_net(self._meta, self.training, len(inputs), False, "meta", bool(_defer), *(inputs + params))
        return [out]

    def _detect_graphed(self, x, vec):
        """Eval-mode detect_forward through a captured hipGraph (one per input shape / vector count / weight state).
        The graph reads its images and its reweighting vectors from static buffers that every call refreshes, so callers
        may pass new tensors each time (valid_ensemble.py re-uses one set of averaged vectors, others recompute them)."""
        from .engine import _STATS_EPOCH, _WEIGHT_EPOCH
        # the vectors may still be in flight on the "meta" side stream (forward() defers the wait to the first reader,
        # which on the eager path is the fused head): every read below is on the current stream, so wait here
        streams.await_tensor(vec)
        # (the BatchNorm batch counters are host bookkeeping the kernels never read: their flush must not invalidate a graph)
        versions = tuple(p._version for p in self.models.parameters()) + \
            tuple(b._version for n, b in self.models.named_buffers() if not n.endswith("num_batches_tracked"))
        # the arithmetic of the fp32 GEMMs is a process-wide launch-time switch: a graph replays the one it was captured in
        key = (tuple(x.shape), x.device.index, tuple(vec.shape), _WEIGHT_EPOCH[0], _STATS_EPOCH[0], self._det.compute_dtype,
               ops.f32_gemm_mode(), hash(versions))
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= 8:                 # each entry pins its activations: keep a handful of shapes
                self._graphs.pop(next(iter(self._graphs)))
            static_x, static_vec = x.detach().clone(), vec.detach().clone()
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):                     # warm-up off the capture: packs the weights, primes the allocator
                    self._det.forward([static_x], dyn=[static_vec], training=False)
            cur.wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph), torch.no_grad():
                static_out, _ = self._det.forward([static_x], dyn=[static_vec], training=False)
            ent = (graph, static_x, static_vec, static_out)
            self._graphs[key] = ent
        graph, static_x, static_vec, static_out = ent
        static_x.copy_(x)
        static_vec.copy_(vec.detach())
        graph.replay()
        return static_out.clone()

    def detect_forward(self, x, dynamic_weights):
        """Query images + reweighting vectors -> (B*N, A*(5+C), G, G), rows ordered b*N+n."""
        self.loss = None       # the reference clears it here too (darknet_meta.py:134)
        if (self.inference_graphs and not self.training and not torch.is_grad_enabled() and x.is_cuda
                and x.dtype == torch.float32 and x.dim() == 4 and not dynamic_weights[0].requires_grad):
            return self._detect_graphed(x, dynamic_weights[0])
        params = _flat_params(self.models)
        return _apply_net(self._det, self.training, 1, True, None, False, x, dynamic_weights[0], *params)

    def forward(self, x, metax, mask, ids=None):
        # the reweighting net runs on its own stream beside the detector backbone; the two meet at the fused head
        return self.detect_forward(x, self.meta_forward(metax, mask, _defer=True))

    def state_dict(self, *args, **kwargs):
        ops.flush_bn_counters(self)        # (the per-BatchNorm hooks of ops.install_bn_counter_hooks cover submodule calls)
        return super(Darknet, self).state_dict(*args, **kwargs)

    def print_network(self):
        print_cfg(self.blocks)
        print("---------------------------------------------------------------------")
        print_cfg(self.learnet_blocks)

    # ---- construction ------------------------------------------------------------------------
    def is_dynamic(self, block):
        return is_dynamic(block)

    def create_network(self, blocks):
        return build_modules(blocks, RegionLossV2)

    # ---- darknet .weights files (byte-compatible with the reference) ------------------------
    def _streams(self):
        return [(self.blocks, self.models), (self.learnet_blocks, self.learnet_models)]

    def load_weights(self, weightfile):
        with open(weightfile, "rb") as fp:
            header = np.fromfile(fp, count=4, dtype=np.int32)
            buf = np.fromfile(fp, dtype=np.float32)
        self.header = torch.from_numpy(header)
        self.seen = self.header[3]
        start = 0
        for blocks, models in self._streams():
            for ind, block in enumerate(blocks[1:]):
                if start >= buf.size:        # partial files (e.g. darknet19_448.conv.23) stop here
                    break
                if block["type"] == "convolutional":
                    model = models[ind]
                    if self.is_dynamic(block) and model[0].weight is None:
                        continue
                    if int(block["batch_normalize"]):
                        start = load_conv_bn(buf, start, model[0], model[1])
                    else:
                        start = load_conv(buf, start, model[0])
                elif block["type"] == "connected":
                    model = models[ind]
                    start = load_fc(buf, start, model if block["activation"] == "linear" else model[0])

    def save_weights(self, outfile, cutoff=0):
        if cutoff <= 0:
            cutoff = len(self.blocks) - 1 + len(self.learnet_blocks)
        with open(outfile, "wb") as fp:
            self.header[3] = int(self.seen)
            self.header.numpy().tofile(fp)
            done = 0
            for blocks, models in self._streams():
                # the reference counts the [learnet] header block as one step of `cutoff`
                if blocks is self.learnet_blocks:
                    done += 1
                for ind, block in enumerate(blocks[1:]):
                    done += 1
                    if done > cutoff:
                        return
                    if block["type"] == "convolutional":
                        model = models[ind]
                        if self.is_dynamic(block) and model[0].weight is None:
                            continue
                        if int(block["batch_normalize"]):
                            save_conv_bn(fp, model[0], model[1])
                        else:
                            save_conv(fp, model[0])
                    elif block["type"] == "connected":
                        model = models[ind]
                        save_fc(fp, model if block["activation"] == "linear" else model[0])
