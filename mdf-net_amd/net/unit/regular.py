"""3-D conv cost regularisers (reference: net/unit/regular.py).  The modules keep the reference's
parameter tree (so checkpoints load strictly) but run a small layer program on the fused MFMA conv
kernel: Conv3d/ConvTranspose3d + folded BN + ReLU + residual add in one launch per layer, activations
in NDHWC; the `prob` conv + softmax over D is one bandwidth-bound kernel."""
from typing import Tuple

import torch
import torch.nn as nn

import torch.nn.functional as F

from mdfnet_hip import layers, ops
from mdfnet_hip.layers import cache_of as _cache
from .base import ConvBNReLU3D


def run_layer(block, x, res=None, tape=None):
    """block: ConvBNReLU3D, or (conv|convT, bn) pair.  x/res/return: [B,D,H,W,C] NDHWC.  With a `tape` (training mode,
    mdfnet_hip/train_ops.py:Tape) the layer runs conv -> BatchNorm3d(batch statistics) -> ReLU (+ res) on the training
    kernels and is recorded for the backward pass; without, it is one fused eval launch (BN folded)."""
    conv, bn = (block.conv, block.bn) if isinstance(block, ConvBNReLU3D) else block
    tr = isinstance(conv, nn.ConvTranspose3d)
    if tuple(conv.kernel_size) != (3, 3, 3):
        raise NotImplementedError("conv3d kernel is built for 3x3x3 only")
    if tape is not None:
        return tape.layer(conv, bn, x, res)
    if conv.training or bn.training:
        raise RuntimeError("regulariser layer in training mode without a tape (use the module's forward)")
    tensors = [conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
    wpack, alpha, beta = _cache(conv).get(tensors, lambda: (ops.pack_conv3d_weight(conv.weight, tr),)
                                          + ops.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps))
    stride = conv.stride[0]
    return ops.conv3d_ndhwc(x, wpack, conv.in_channels, conv.out_channels, stride, tr, alpha, beta, True, res)


def _up(cin, cout, out_pad=1, stride=2):
    return [nn.ConvTranspose3d(cin, cout, kernel_size=3, padding=1, output_padding=out_pad, stride=stride, bias=False),
            nn.BatchNorm3d(cout), nn.ReLU(inplace=True)]


class RegularNet_3Scales(nn.Module):
    """Stage-0 regulariser (regular.py:9-69): 16/32/64-channel 3-level U-Net -> softmax over D."""

    def __init__(self, in_chs: int = 8, inner_chs: int = 16) -> None:
        super().__init__()
        a, b, c = inner_chs, inner_chs * 2, inner_chs * 4
        self.conv01 = nn.Sequential(ConvBNReLU3D(in_chs, a), ConvBNReLU3D(a, a))
        self.conv12 = nn.Sequential(ConvBNReLU3D(a, b, stride=2), ConvBNReLU3D(b, b), ConvBNReLU3D(b, b))
        self.conv232 = nn.Sequential(ConvBNReLU3D(b, c, stride=2), ConvBNReLU3D(c, c), ConvBNReLU3D(c, c), *_up(c, b))
        self.conv10 = nn.Sequential(*_up(b, a))
        self.prob = nn.Conv3d(a, 1, 3, stride=1, padding=1, bias=False)

    def features(self, x, tape=None):
        """cost [B,D,H,W,C] -> last feature volume before `prob` (NDHWC)."""
        assert x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0, f"cost volume H,W must be divisible by 4: {tuple(x.shape)}"

        def L(block, t, res=None):
            return run_layer(block, t, res, tape)
        x = L(self.conv01[1], L(self.conv01[0], x))
        x1 = L(self.conv12[0], x)
        x1 = L(self.conv12[2], L(self.conv12[1], x1))
        y = L(self.conv232[2], L(self.conv232[1], L(self.conv232[0], x1)))
        x1 = L((self.conv232[3], self.conv232[4]), y, res=x1)     # x1 + relu(bn(convT(y)))
        return L((self.conv10[0], self.conv10[1]), x1, res=x)     # x + relu(bn(convT(x1)))

    def _stock_forward(self, x):
        """Training path (autograd, batch-stat BN): the same layer program with the stock modules."""
        x = self.conv01(x)
        x1 = self.conv12(x)
        x1 = x1 + self.conv232(x1)
        x = x + self.conv10(x1)
        return F.softmax(self.prob(x).squeeze(1), dim=1)

    fused_regress = True   # forward(cost, hypos) also returns the soft-argmin depth

    def forward(self, x: torch.Tensor, depth_hypos=None):
        """cost [B,C,D,H,W] -> prob [B,D,H,W]; with depth_hypos also returns the soft-argmin depth."""
        if layers.hip_train(self, x):
            # training on the GPU: every layer forward + backward on the hand-written kernels (mdfnet_hip/train_ops.py)
            from mdfnet_hip import train_ops
            return train_ops.regulariser_train(self, x, depth_hypos)
        if not layers.use_hip(self, x):
            prob = self._stock_forward(x)
            return prob if depth_hypos is None else (prob, torch.sum(prob * depth_hypos, 1))
        with torch.no_grad():
            wpack = layers.cache_of(self.prob).get((self.prob.weight,), lambda: ops.pack_prob_weight(self.prob.weight))
            return ops.prob_head(self.features(ops.to_ndhwc(x)), self.prob.weight, depth_hypos, wpack=wpack)


class RegularNet_4Scales(nn.Module):
    """Stage-1/2 regulariser (regular.py:72-133): 8/16/32/64-channel 4-level U-Net -> softmax over D."""

    def __init__(self, in_chs: int, base_chs: int = 8, sample_stride: Tuple = (2, 2, 2), sample_padding: Tuple = (1, 1, 1)):
        super().__init__()
        if tuple(sample_stride) != (2, 2, 2) or tuple(sample_padding) != (1, 1, 1):
            raise NotImplementedError("conv kernels are built for sample_stride=(2,2,2), sample_padding=(1,1,1)")
        a, b, c, d = base_chs, base_chs * 2, base_chs * 4, base_chs * 8
        self.conv01 = ConvBNReLU3D(in_chs, a)
        self.conv12 = nn.Sequential(ConvBNReLU3D(a, b, 3, sample_stride, 1), ConvBNReLU3D(b, b, 3, 1, 1))
        self.conv23 = nn.Sequential(ConvBNReLU3D(b, c, 3, sample_stride, 1), ConvBNReLU3D(c, c, 3, 1, 1))
        self.conv343 = nn.Sequential(ConvBNReLU3D(c, d, 3, sample_stride, 1), ConvBNReLU3D(d, d, 3, 1, 1),
                                     *_up(d, c, sample_padding, sample_stride))
        self.trconv32 = nn.Sequential(*_up(c, b, sample_padding, sample_stride))
        self.trconv21 = nn.Sequential(*_up(b, a, sample_padding, sample_stride))
        self.prob = nn.Conv3d(a, 1, 3, stride=1, padding=1, bias=False)

    def features(self, x, tape=None):
        assert x.shape[2] % 8 == 0 and x.shape[3] % 8 == 0, f"cost volume H,W must be divisible by 8: {tuple(x.shape)}"

        def L(block, t, res=None):
            return run_layer(block, t, res, tape)
        x1 = L(self.conv01, x)
        x2 = L(self.conv12[1], L(self.conv12[0], x1))
        x3 = L(self.conv23[1], L(self.conv23[0], x2))
        y = L(self.conv343[1], L(self.conv343[0], x3))
        x3 = L((self.conv343[2], self.conv343[3]), y, res=x3)
        x2 = L((self.trconv32[0], self.trconv32[1]), x3, res=x2)
        return L((self.trconv21[0], self.trconv21[1]), x2, res=x1)

    def _stock_forward(self, x):
        x1 = self.conv01(x)
        x2 = self.conv12(x1)
        x3 = self.conv23(x2)
        x3 = x3 + self.conv343(x3)
        x2 = x2 + self.trconv32(x3)
        x1 = x1 + self.trconv21(x2)
        return F.softmax(self.prob(x1).squeeze(1), dim=1)

    fused_regress = True   # forward(cost, hypos) also returns the soft-argmin depth

    def forward(self, x: torch.Tensor, depth_hypos=None):
        if layers.hip_train(self, x):
            # training on the GPU: every layer forward + backward on the hand-written kernels (mdfnet_hip/train_ops.py)
            from mdfnet_hip import train_ops
            return train_ops.regulariser_train(self, x, depth_hypos)
        if not layers.use_hip(self, x):
            prob = self._stock_forward(x)
            return prob if depth_hypos is None else (prob, torch.sum(prob * depth_hypos, 1))
        with torch.no_grad():
            wpack = layers.cache_of(self.prob).get((self.prob.weight,), lambda: ops.pack_prob_weight(self.prob.weight))
            return ops.prob_head(self.features(ops.to_ndhwc(x)), self.prob.weight, depth_hypos, wpack=wpack)
