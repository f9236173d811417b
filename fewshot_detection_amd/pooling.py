"""Pooling helpers with the reference's names (pooling.py:8-60), on the HIP kernels."""
import numpy as np
import torch.nn as nn

from . import ops


class GlobalMaxPool2d(nn.Module):
    """max over the whole (square) feature map: (B,C,H,W) -> (B,C,1,1)   (pooling.py:8-27)."""

    def __init__(self, stride=None, padding=0, dilation=1, return_indices=False, ceil_mode=False):
        super(GlobalMaxPool2d, self).__init__()
        self.stride = stride or 1
        self.padding = padding
        self.dilation = dilation
        self.return_indices = return_indices
        self.ceil_mode = ceil_mode

    def extra_repr(self):
        return "global max pooling, stride={stride}, padding={padding}, dilation={dilation}, " \
               "ceil_mode={ceil_mode}".format(**self.__dict__)

    def forward(self, input):
        vals, _ = ops.global_maxpool(ops.nchw_to_nhwc(input, pad_to=1))
        return vals.view(input.shape[0], input.shape[1], 1, 1)


class GlobalAvgPool2d(nn.Module):
    """mean over the whole feature map: (B,C,H,W) -> (B,C,1,1)   (pooling.py:29-45, F.adaptive_avg_pool2d(x, 1)); the
    [globalavg] / [avgpool] cfg blocks.  Inside a cfg network the engine calls the kernel on its NHWC view (forward and
    backward); this module form serves direct callers."""

    def __init__(self, stride=None, padding=0, dilation=1, return_indices=False, ceil_mode=False):
        super(GlobalAvgPool2d, self).__init__()
        self.stride = stride or 1
        self.padding = padding
        self.dilation = dilation
        self.return_indices = return_indices
        self.ceil_mode = ceil_mode

    def extra_repr(self):
        return "global avg pooling"

    def forward(self, input):
        vals = ops.global_avgpool(ops.nchw_to_nhwc(input, pad_to=1))
        return vals.view(input.shape[0], input.shape[1], 1, 1)


class Split(nn.Module):
    def __init__(self, splits):
        super(Split, self).__init__()
        self.splits = splits

    def extra_repr(self):
        return "split layer, splits={splits}".format(**self.__dict__)

    def forward(self, input):
        edges = np.cumsum([0] + self.splits)
        return [input[:, edges[i]:edges[i + 1], :, :].contiguous() for i in range(len(edges) - 1)]
