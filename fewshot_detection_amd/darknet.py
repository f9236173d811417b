"""Plain (non-meta) YOLOv2 `Darknet(cfgfile)` with the reference's interface (darknet.py:61-341):
same kernels as the meta detector, no reweighting branch, RegionLoss v1.  BASELINE config C1."""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .cfg import load_conv, load_conv_bn, parse_cfg, print_cfg, save_conv, save_conv_bn
from .darknet_meta import (EmptyModule, MaxPoolStride1, Reorg, _apply_net, _flat_params,  # noqa: F401  (the reference's
                           build_modules)                                                 # darknet.py defines them too)
from .engine import Network
from .pooling import GlobalAvgPool2d  # noqa: F401
from .region_loss import RegionLoss


class Darknet(nn.Module):
    def __init__(self, cfgfile):
        super(Darknet, self).__init__()
        self.blocks = cfgfile if isinstance(cfgfile, list) else parse_cfg(cfgfile)
        self.models = self.create_network(self.blocks)
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
        self._net = Network(self.blocks, self.models)

    def create_network(self, blocks):
        return build_modules(blocks, RegionLoss)

    def forward(self, x):
        self.loss = None
        return _apply_net(self._net, self.training, 1, False, None, False, x, *_flat_params(self.models))

    def state_dict(self, *args, **kwargs):
        ops.flush_bn_counters(self)        # BatchNorm batch counters are kept on the host between looks
        return super(Darknet, self).state_dict(*args, **kwargs)

    def print_network(self):
        print_cfg(self.blocks)

    def load_weights(self, weightfile):
        with open(weightfile, "rb") as fp:
            header = np.fromfile(fp, count=4, dtype=np.int32)
            buf = np.fromfile(fp, dtype=np.float32)
        self.header = torch.from_numpy(header)
        self.seen = self.header[3]
        start = 0
        for ind, block in enumerate(self.blocks[1:]):
            if start >= buf.size:
                break
            if block["type"] == "convolutional":
                model = self.models[ind]
                if int(block["batch_normalize"]):
                    start = load_conv_bn(buf, start, model[0], model[1])
                else:
                    start = load_conv(buf, start, model[0])

    def save_weights(self, outfile, cutoff=0):
        if cutoff <= 0:
            cutoff = len(self.blocks) - 1
        with open(outfile, "wb") as fp:
            self.header[3] = int(self.seen)
            self.header.numpy().tofile(fp)
            for ind, block in enumerate(self.blocks[1:cutoff + 1]):
                if block["type"] == "convolutional":
                    model = self.models[ind]
                    if int(block["batch_normalize"]):
                        save_conv_bn(fp, model[0], model[1])
                    else:
                        save_conv(fp, model[0])
