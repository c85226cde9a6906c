"""Depth refinement / x2 upsampling (reference: net/unit/refine.py:8-46).  Eval on a GPU: fused implicit-GEMM conv
kernels (NHWC) with the Res-block update x + 0.1*conv(...) and the skip add in the conv epilogues; training / CPU: stock
PyTorch modules."""
import os

import torch
import torch.nn as nn

from mdfnet_hip import controlplane, layers, ops

from .base import Res

_TAIL = os.environ.get("MDF_REFINE_TAIL", "1") != "0"      # dev A/B: the last three stages as three launches
_HEAD = os.environ.get("MDF_REFINE_HEAD", "1") != "0"           # dev A/B: range mapping and conv0 as two launches
_RES_PAIR = os.environ.get("MDF_REFINE_RES_PAIR", "1") != "0"   # dev A/B: a Res block as two launches


class RefineNet2(nn.Module):
    def __init__(self, base_chs: int = 8, nres: int = 3):
        super().__init__()
        self.conv0 = nn.Conv2d(1, base_chs, 3, 1, 1, bias=False)
        self.ress = nn.ModuleList(Res(base_chs) for _ in range(nres))
        self.conv1 = nn.Conv2d(base_chs, base_chs, 3, 1, 1, bias=False)
        self.conv2 = nn.Sequential(nn.Conv2d(base_chs, base_chs * 4, 3, 1, 1, bias=False), nn.PixelShuffle(2),
                                   nn.Conv2d(base_chs, 1, 3, 1, 1, bias=False))

    def forward(self, depth: torch.Tensor, depth_range: torch.Tensor) -> torch.Tensor:
        """depth [B,h,w] (detached), range [B,2] -> [B,2h,2w]: work in [0,1] then map back to the range."""
        b = depth.shape[0]
        plan = controlplane.for_range(depth_range) if depth.is_cuda else None
        if plan is not None:                                   # (lo, span) came up with the forward's control plane
            lo, span = plan.lo.view(b, 1, 1, 1), plan.span.view(b, 1, 1, 1)
        else:
            lo = depth_range[:, 0].float().view(b, 1, 1, 1)
            span = depth_range[:, 1].float().view(b, 1, 1, 1) - lo
        if layers.hip_eval(self, depth):
            with torch.no_grad():
                # (depth - lo) / span and lo + y * span: one launch each, torch's roundings (mdf_range_affine_fwd)
                if _HEAD and self.conv0.out_channels == 8:
                    x0 = ops.refine_head(depth.detach(), lo.reshape(b), span.reshape(b), self.conv0.weight)   # one launch, bit-identical
                else:
                    x = ops.range_affine(depth.detach(), lo.reshape(b), span.reshape(b), 0).unsqueeze(-1)   # [B,h,w,1]
                    x0 = layers.conv2d_layer(self.conv0, None, x)
                y = x0
                for blk in self.ress:                                                               # x + 0.1*conv(relu(conv(x)))
                    if _RES_PAIR and self.conv1.in_channels == 8:
                        y = layers.res_block(blk.conv[0], blk.conv[2], y, 0.1)                      # one launch (res_pair.hip), bit-identical
                    else:
                        t = layers.conv2d_layer(blk.conv[0], None, y, relu=True)
                        y = layers.conv2d_layer(blk.conv[2], None, t, res=y, res_scale=0.1)
                y = layers.conv2d_layer(self.conv1, None, y, res=x0)                               # x0 + conv1(y)
                if _TAIL:                                                                          # conv2 + the range mapping in one launch
                    return layers.refine_tail(self.conv2[0], self.conv2[2], y, lo.reshape(b), span.reshape(b))
                y = layers.conv2d_layer(self.conv2[0], None, y, pixel_shuffle2=True)               # conv + PixelShuffle(2): [B,2h,2w,8]
                y = layers.conv2d_layer(self.conv2[2], None, y)                                    # [B,2h,2w,1]
                return ops.range_affine(y.squeeze(-1), lo.reshape(b), span.reshape(b), 1)
        if layers.hip_train(self, depth):
            from mdfnet_hip import train_ops
            return train_ops.refine_train(self, depth, lo, span)
        x0 = self.conv0((depth.detach().unsqueeze(1) - lo) / span)
        y = x0
        for blk in self.ress:
            y = blk(y)
        y = self.conv2(x0 + self.conv1(y))
        return (lo + y * span).squeeze(1)
