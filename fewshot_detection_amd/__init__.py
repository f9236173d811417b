"""fewshot_detection_amd -- MI355X-native hot path of bingykang/Fewshot_Detection.

Host-side mirror of the reference's Python interface (cfg.parse_cfg, darknet_meta.Darknet,
region_loss.RegionLoss / RegionLossV2, dynamic_conv.dynamic_conv2d, pooling.*) over the C-ABI
of include/fsdet.h (hand-written HIP kernels for gfx950 in csrc/).  PyTorch is used for device
memory, streams, autograd plumbing and torch.distributed only.

There is no CPU fallback: every op raises if libfsdet_hip.so is missing or a tensor is not on
a HIP device.
"""
__version__ = "0.1.0"
