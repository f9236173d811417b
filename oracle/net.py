"""Oracle: the .cfg-driven meta detector on PyTorch-CPU.  TEST INFRASTRUCTURE.

Restates reference darknet_meta.py (Darknet: create_network :208-353, meta_forward :107-128,
detect_forward :130-195, load/save_weights :355-479), the helper layers (Reorg :55-74,
MaxPoolStride1 :47-53), dynamic_conv.py:125-164 (channel re-weighting written as the plain
broadcast product it is equal to), pooling.py:8-60 and the weight-stream helpers of
cfg.py:411-481.  `OracleYolo` is the non-meta twin (reference darknet.py:61-341) used by
BASELINE config C1 (tiny-yolo-voc).

Module/parameter names match the reference (`models.<i>.conv<k>.weight`, `bn<k>` ...) so a
state_dict moves freely between the reference, this oracle and the product.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .cfgparse import parse_cfg


def reorg(x, stride=2):
    """out[b, (di*s+dj)*C + c, i, j] = x[b, c, s*i+di, s*j+dj]   (darknet_meta.py:55-74)."""
    b, c, h, w = x.shape
    s = stride
    x = x.view(b, c, h // s, s, w // s, s)          # b c i di j dj
    x = x.permute(0, 3, 5, 1, 2, 4).contiguous()     # b di dj c i j
    return x.view(b, s * s * c, h // s, w // s)


def reweight(x, dyn):
    """First dynamic conv: out[b*N+n, c, h, w] = x[b, c, h, w] * dyn[n, c]  (dynamic_conv.py:125-164)."""
    n_cls, n_ch = dyn.shape[0], x.shape[1]
    assert dyn.shape[1] == n_ch and tuple(dyn.shape[2:]) == (1, 1)
    out = x.unsqueeze(1) * dyn.view(1, n_cls, n_ch, 1, 1)
    return out.reshape(-1, n_ch, x.shape[2], x.shape[3])


class _Passthrough(nn.Module):
    def forward(self, x):
        return x


class _Reorg(nn.Module):
    def __init__(self, stride):
        super().__init__()
        self.stride = stride

    def forward(self, x):
        return reorg(x, self.stride)


class _PoolStride1(nn.Module):
    def forward(self, x):
        return F.max_pool2d(F.pad(x, (0, 1, 0, 1), mode="replicate"), 2, stride=1)


class _GlobalMax(nn.Module):
    def forward(self, x):
        return F.max_pool2d(x, x.size(-1), 1)


class _GlobalAvg(nn.Module):
    def forward(self, x):
        return F.adaptive_avg_pool2d(x, 1)


class _Reweight(nn.Module):
    """Parameter-free dynamic 1x1 conv (weight/bias registered as None like the reference)."""

    def __init__(self):
        super().__init__()
        self.register_parameter("weight", None)
        self.register_parameter("bias", None)

    def forward(self, pair):
        return reweight(pair[0], pair[1])


class _RegionSpec(nn.Module):
    """Holds the [region] hyper-parameters (the reference stores a loss module here)."""

    def __init__(self, block):
        super().__init__()
        self.anchors = [float(v) for v in block["anchors"].split(",")]
        self.num_classes = int(block["classes"])
        self.num_anchors = int(block["num"])
        self.anchor_step = len(self.anchors) // self.num_anchors
        self.object_scale = float(block["object_scale"])
        self.noobject_scale = float(block["noobject_scale"])
        self.class_scale = float(block["class_scale"])
        self.coord_scale = float(block["coord_scale"])
        self.thresh = 0.6        # constructor default; the cfg key is never copied (region_loss.py:248)
        self.seen = 0


def is_dynamic(block):
    return "dynamic" in block and int(block["dynamic"]) == 1


def build_modules(blocks):
    mods = nn.ModuleList()
    width = 3
    widths = []
    n_conv = 0
    for blk in blocks:
        kind = blk["type"]
        if kind in ("net", "learnet"):
            width = int(blk["channels"])
            continue
        if kind == "convolutional":
            n_conv += 1
            k = int(blk["size"])
            pad = (k - 1) // 2 if int(blk["pad"]) else 0
            out_ch = int(blk["filters"])
            use_bn = int(blk["batch_normalize"])
            seq = nn.Sequential()
            if is_dynamic(blk):
                assert "partial" not in blk, "partial dynamic conv is not used by any shipped cfg"
                seq.add_module("conv%d" % n_conv, _Reweight())
            else:
                has_bias = (bool(int(blk["bias"])) if "bias" in blk else True) and not use_bn
                seq.add_module("conv%d" % n_conv,
                               nn.Conv2d(width, out_ch, k, int(blk["stride"]), pad, bias=has_bias))
            if use_bn:
                seq.add_module("bn%d" % n_conv, nn.BatchNorm2d(out_ch))
            if blk["activation"] == "leaky":
                seq.add_module("leaky%d" % n_conv, nn.LeakyReLU(0.1, inplace=True))
            elif blk["activation"] == "relu":
                seq.add_module("relu%d" % n_conv, nn.ReLU(inplace=True))
            width = out_ch
            mods.append(seq)
        elif kind == "maxpool":
            s = int(blk["stride"])
            mods.append(nn.MaxPool2d(int(blk["size"]), s) if s > 1 else _PoolStride1())
        elif kind == "reorg":
            s = int(blk["stride"])
            width = s * s * width
            mods.append(_Reorg(s))
        elif kind == "route":
            here = len(mods)
            src = [int(v) if int(v) > 0 else int(v) + here for v in blk["layers"].split(",")]
            width = sum(widths[i] for i in src)
            mods.append(_Passthrough())
        elif kind == "globalmax":
            mods.append(_GlobalMax())
        elif kind in ("globalavg", "avgpool"):
            mods.append(_GlobalAvg())
        elif kind == "region":
            mods.append(_RegionSpec(blk))
        else:
            raise NotImplementedError("block type %r is outside the oracle's scope" % kind)
        widths.append(width)
    return mods


def _walk(blocks, mods, x, dyn=None):
    """Shared layer walk (darknet_meta.py:130-195 / darknet.py:80-129)."""
    outs = {}
    n_dyn = 0
    for idx, blk in enumerate(blocks[1:]):
        kind = blk["type"]
        if kind == "route":
            src = [int(v) if int(v) > 0 else int(v) + idx for v in blk["layers"].split(",")]
            x = outs[src[0]] if len(src) == 1 else torch.cat([outs[s] for s in src], 1)
        elif kind in ("region", "cost"):
            continue
        elif kind == "convolutional" and is_dynamic(blk):
            x = mods[idx]((x, dyn[n_dyn]))
            n_dyn += 1
        else:
            x = mods[idx](x)
        outs[idx] = x
    return x


# ---- bf16 storage mode (BASELINE configs[2] / [4]) -------------------------------------------------
# What the product's bf16 mode is DEFINED to compute, restated with float tensors that are rounded to bfloat16 wherever
# the product stores one: every conv output y, every activation z, the packed conv weights and the folded head weight;
# products accumulate in fp32 and the BatchNorm statistics come from the UNROUNDED conv output.  First layers (<= 4 input
# channels: the fp32 image / image+mask) keep fp32 operands.  Rounding is a straight-through estimator for autograd.

def _q(t):
    return t + (t.to(torch.bfloat16).float() - t).detach()


def _conv_block_bf16(seq, x, training):
    conv = seq[0]
    bn = seq[1] if len(seq) > 1 and isinstance(seq[1], nn.BatchNorm2d) else None
    act = seq[len(seq) - 1]
    first = conv.in_channels <= 4
    w = conv.weight if first else _q(conv.weight)
    y32 = F.conv2d(x, w, conv.bias, conv.stride, conv.padding)
    y16 = _q(y32)
    if bn is not None:
        if training:
            mean = y32.mean(dim=(0, 2, 3))
            var = y32.var(dim=(0, 2, 3), unbiased=False)
            with torch.no_grad():
                n = y32.numel() / y32.shape[1]
                mom = 0.1 if bn.momentum is None else bn.momentum
                bn.running_mean.mul_(1 - mom).add_(mom * mean)
                bn.running_var.mul_(1 - mom).add_(mom * var * (n / max(n - 1.0, 1.0)))
        else:
            mean, var = bn.running_mean, bn.running_var
        inv = torch.rsqrt(var + bn.eps)
        t = (y16 - mean.view(1, -1, 1, 1)) * (inv * bn.weight).view(1, -1, 1, 1) + bn.bias.view(1, -1, 1, 1)
    else:
        t = y16
    if isinstance(act, nn.LeakyReLU):
        t = F.leaky_relu(t, 0.1)
    elif isinstance(act, nn.ReLU):
        t = F.relu(t)
    if bn is None and act is conv:      # linear conv without BatchNorm: the stored value is the rounded conv output itself
        return y16
    return _q(t)


def _walk_bf16(blocks, mods, x, dyn=None, training=True):
    """_walk in bf16 storage mode.  The reweighting + 1x1 head pair is evaluated the way the product fuses it: the folded
    weight W[o,c] * w[n,c] is rounded to bf16 once, the (B*N, C, H, W) tensor never exists, the head output is float."""
    outs = {}
    n_dyn = 0
    layers = blocks[1:]
    skip = -1
    for idx, blk in enumerate(layers):
        if idx <= skip:
            continue
        kind = blk["type"]
        if kind == "route":
            src = [int(v) if int(v) > 0 else int(v) + idx for v in blk["layers"].split(",")]
            x = outs[src[0]] if len(src) == 1 else torch.cat([outs[s] for s in src], 1)
        elif kind in ("region", "cost"):
            continue
        elif kind == "convolutional" and is_dynamic(blk):
            head = mods[idx + 1][0]
            vec = dyn[n_dyn]
            n_dyn += 1
            n_cls, o_ch, c = vec.shape[0], head.weight.shape[0], head.weight.shape[1]
            w_eff = _q((head.weight.view(1, o_ch, c) * vec.view(n_cls, 1, c)).reshape(n_cls * o_ch, c, 1, 1))
            bias = None if head.bias is None else head.bias.repeat(n_cls)
            y = F.conv2d(x, w_eff, bias)
            x = y.view(x.shape[0] * n_cls, o_ch, x.shape[2], x.shape[3])
            skip = idx + 1
            outs[idx + 1] = x
            continue
        elif kind == "convolutional":
            x = _conv_block_bf16(mods[idx], x, training)
        else:
            x = mods[idx](x)          # max pools / reorg / global max commute with the rounding
        outs[idx] = x
    return x


# ---- darknet weight stream (cfg.py:411-481, darknet_meta.py:355-479) -------------------------

def _take(buf, pos, t):
    n = t.numel()
    t.data.copy_(torch.from_numpy(buf[pos:pos + n]).view_as(t))
    return pos + n


def load_stream(buf, pos, blocks, mods):
    for idx, blk in enumerate(blocks[1:]):
        if pos >= buf.size:
            break
        if blk["type"] != "convolutional" or is_dynamic(blk):
            continue
        seq = mods[idx]
        conv = seq[0]
        if int(blk["batch_normalize"]):
            bn = seq[1]
            for t in (bn.bias, bn.weight, bn.running_mean, bn.running_var, conv.weight):
                pos = _take(buf, pos, t)
        else:
            if conv.bias is not None:
                pos = _take(buf, pos, conv.bias)
            pos = _take(buf, pos, conv.weight)
    return pos


def dump_stream(fh, blocks, mods):
    for idx, blk in enumerate(blocks[1:]):
        if blk["type"] != "convolutional" or is_dynamic(blk):
            continue
        seq = mods[idx]
        conv = seq[0]
        if int(blk["batch_normalize"]):
            bn = seq[1]
            parts = (bn.bias, bn.weight, bn.running_mean, bn.running_var, conv.weight)
        else:
            parts = (conv.bias, conv.weight) if conv.bias is not None else (conv.weight,)
        for t in parts:
            t.detach().cpu().numpy().astype(np.float32).tofile(fh)


class OracleDarknet(nn.Module):
    """Meta detector = detector `models` + re-weighting net `learnet_models`."""

    def __init__(self, darknet_cfg, learnet_cfg, metain_type=2):
        super().__init__()
        self.blocks = darknet_cfg if isinstance(darknet_cfg, list) else parse_cfg(darknet_cfg)
        self.learnet_blocks = learnet_cfg if isinstance(learnet_cfg, list) else parse_cfg(learnet_cfg)
        self.models = build_modules(self.blocks)
        self.learnet_models = build_modules(self.learnet_blocks)
        self.region = self.models[len(self.models) - 1]
        self.metain_type = metain_type
        self.header = torch.IntTensor([0, 0, 0, 0])
        self.seen = 0

    def meta_forward(self, metax, mask):
        assert int(self.learnet_blocks[0]["feat_layer"]) == 0
        if self.metain_type in (2, 3):
            metax = torch.cat([metax, mask], dim=1)
        for m in self.learnet_models:
            metax = m(metax)
        return [metax]

    def detect_forward(self, x, dynamic_weights):
        return _walk(self.blocks, self.models, x, dynamic_weights)

    def forward(self, x, metax, mask):
        return self.detect_forward(x, self.meta_forward(metax, mask))

    def forward_bf16(self, x, metax, mask):
        """The product's bf16 storage mode (see _walk_bf16): -> (head output float, reweighting vectors)."""
        if self.metain_type in (2, 3):
            metax = torch.cat([metax, mask], dim=1)
        dyn = _walk_bf16(self.learnet_blocks, self.learnet_models, metax, training=self.training)
        return _walk_bf16(self.blocks, self.models, x, [dyn], training=self.training), dyn

    def load_weights(self, path):
        with open(path, "rb") as fh:
            self.header = torch.from_numpy(np.fromfile(fh, count=4, dtype=np.int32))
            buf = np.fromfile(fh, dtype=np.float32)
        self.seen = int(self.header[3])
        pos = load_stream(buf, 0, self.blocks, self.models)
        load_stream(buf, pos, self.learnet_blocks, self.learnet_models)

    def save_weights(self, path):
        with open(path, "wb") as fh:
            self.header[3] = self.seen
            self.header.numpy().tofile(fh)
            dump_stream(fh, self.blocks, self.models)
            dump_stream(fh, self.learnet_blocks, self.learnet_models)


class OracleYolo(nn.Module):
    """Plain YOLOv2 twin (reference darknet.py) -- BASELINE config C1."""

    def __init__(self, cfg):
        super().__init__()
        self.blocks = cfg if isinstance(cfg, list) else parse_cfg(cfg)
        self.models = build_modules(self.blocks)
        self.region = self.models[len(self.models) - 1]
        self.header = torch.IntTensor([0, 0, 0, 0])
        self.seen = 0

    def forward(self, x):
        return _walk(self.blocks, self.models, x)

    def load_weights(self, path):
        with open(path, "rb") as fh:
            self.header = torch.from_numpy(np.fromfile(fh, count=4, dtype=np.int32))
            buf = np.fromfile(fh, dtype=np.float32)
        self.seen = int(self.header[3])
        load_stream(buf, 0, self.blocks, self.models)

    def save_weights(self, path):
        with open(path, "wb") as fh:
            self.header[3] = self.seen
            self.header.numpy().tofile(fh)
            dump_stream(fh, self.blocks, self.models)
