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


class _ConvNd(nn.Module):
    """Parameter-free convolution base with torch's conv attributes and repr (dynamic_conv.py:10-76): the kernel of a
    dynamic convolution arrives with the input, so `weight` / `bias` are registered as None."""

    partial = None

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, transposed, output_padding,
                 groups, bias):
        super(_ConvNd, self).__init__()
        if in_channels % groups != 0:
            raise ValueError("in_channels must be divisible by groups")
        if out_channels % groups != 0:
            raise ValueError("out_channels must be divisible by groups")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = kernel_size, stride, padding, dilation
        self.transposed, self.output_padding, self.groups = transposed, output_padding, groups
        self.register_parameter("weight", None)
        self.register_parameter("bias", None)

    def reset_parameters(self):
        pass

    def extra_repr(self):
        s = "{in_channels}, {out_channels}, kernel_size={kernel_size}, stride={stride}"
        if self.padding != (0,) * len(self.padding):
            s += ", padding={padding}"
        if self.dilation != (1,) * len(self.dilation):
            s += ", dilation={dilation}"
        if self.groups != 1:
            s += ", groups={groups}"
        return s.format(**self.__dict__) + ", bias=False"


class DynamicConv2d(_ConvNd):
    """The one variant the shipped cfgs build (is_first=True, partial=None).  Module-level -- the reference defines the class
    inside its factory (dynamic_conv.py:112), which makes every model that holds one unpicklable; this one pickles."""
    is_first = True
    partial = None

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=False):
        super(DynamicConv2d, self).__init__(in_channels, out_channels, _pair(kernel_size), _pair(stride),
                                            _pair(padding), _pair(dilation), False, _pair(0), groups, bias)

    def forward(self, inputs):
        x, w = inputs
        assert tuple(w.shape[-2:]) == self.kernel_size == (1, 1)
        assert w.shape[1] == x.shape[1], "reweighting vector width != feature channels"
        return ops.dynamic_conv(x, w)


def dynamic_conv2d(is_first, partial=None):
    if partial is not None:
        raise NotImplementedError("partial dynamic convolution is not used by any shipped cfg")
    if not is_first:
        raise NotImplementedError("only the first dynamic convolution of a network is supported")
    return DynamicConv2d
