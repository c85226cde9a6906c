"""Shared blocks (reference: net/unit/base.py).  The 2-D blocks run stock PyTorch-ROCm (they are not on
the hand-kernel path); ConvBNReLU3D is a parameter container whose arithmetic is done by the fused
MFMA conv kernel; homo_warping is the HIP warp."""
import torch
import torch.nn as nn

from mdfnet_hip import hostmirror, layers, ops


class ConvBNReLU(nn.Module):
    """Conv2d(bias=False) + BatchNorm2d + ReLU   (base.py:7-25); children named conv / bn / relu."""

    def __init__(self, inchs, outchs, kernel_size=3, stride=1, padding=1, groups=1, bias=False):
        super().__init__()
        self.conv = nn.Conv2d(inchs, outchs, kernel_size, stride, (kernel_size - 1) // 2, groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(outchs)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.relu(self.bn(self.conv(x)))


class ConvBNReLU3D(nn.Module):
    """Conv3d(bias=False) + BatchNorm3d + ReLU (base.py:50-68) as ONE fused kernel in eval mode."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, groups=1, bias=False):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride, padding, groups=groups, bias=bias)
        self.bn = nn.BatchNorm3d(out_channels)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        if layers.use_hip(self, x):
            from .regular import run_layer
            return ops.from_ndhwc(run_layer(self, ops.to_ndhwc(x)))
        return self.relu(self.bn(self.conv(x)))   # training path: stock ops


class Res(nn.Module):
    """x + 0.1 * conv(relu(conv(x)))   (base.py:71-82)."""

    def __init__(self, chs):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(chs, chs, 3, 1, 1, bias=False), nn.ReLU(inplace=True),
                                  nn.Conv2d(chs, chs, 3, 1, 1, bias=False))

    def forward(self, x):
        return x + self.conv(x) * 0.1


def homo_warping(src_fea, src_proj, ref_proj, depth_hypos):
    """base.py:85-126.  src_fea [B,C,h,w]; projections [B,4,4]; hypos [B,D,1,1] | [B,D,h,w] -> [B,C,D,h,w]
    (bit-identical to the reference's CPU result).  With autograd / on the CPU: the stock-op training path."""
    if not layers.use_hip(None, src_fea, depth_hypos):
        return layers.stock().homo_warping(src_fea, src_proj, ref_proj, depth_hypos)
    proj = ops.relative_projections(hostmirror.get(ref_proj), [hostmirror.get(src_proj)])[0]
    return ops.homo_warp(src_fea, proj.to(src_fea.device, non_blocking=True), depth_hypos)
