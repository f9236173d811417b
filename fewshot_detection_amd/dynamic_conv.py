"""Channel-wise feature reweighting with the reference's factory name (dynamic_conv.py:110-168).

`dynamic_conv2d(is_first, partial)` returns a module class whose forward takes `(x, w)` with
x (B,C,H,W), w (N,C,1,1) and returns (B*N,C,H,W): out[b*N+n,c] = x[b,c] * w[n,c].  Inside
`Darknet.detect_forward` this module is never run on its own: it is folded into the 1x1 detection
head (engine.py); the standalone forward below exists for API parity and materialises the product
with a HIP kernel.
"""
import torch.nn as nn
from torch.nn.modules.utils import _pair

from . import ops


def dynamic_conv2d(is_first, partial=None):
    if partial is not None:
        raise NotImplementedError("partial dynamic convolution is not used by any shipped cfg")
    if not is_first:
        raise NotImplementedError("only the first dynamic convolution of a network is supported")

    class DynamicConv2d(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                     groups=1, bias=False):
            super(DynamicConv2d, self).__init__()
            self.in_channels = in_channels
            self.out_channels = out_channels
            self.kernel_size = _pair(kernel_size)
            self.stride = _pair(stride)
            self.padding = _pair(padding)
            self.dilation = _pair(dilation)
            self.groups = groups
            self.register_parameter("weight", None)     # parameter-free, like the reference
            self.register_parameter("bias", None)

        def forward(self, inputs):
            x, w = inputs
            assert tuple(w.shape[-2:]) == self.kernel_size == (1, 1)
            assert w.shape[1] == x.shape[1], "reweighting vector width != feature channels"
            return ops.dynamic_conv(x, w)

    DynamicConv2d.is_first = is_first
    DynamicConv2d.partial = partial
    return DynamicConv2d
