"""Feature pyramid (reference: net/unit/backbone.py:9-66).  Eval on a GPU: every Conv2d+BN+ReLU is one fused
implicit-GEMM MFMA kernel (conv_lds.hip, NHWC), the FPN top-down step (bilinear x2 + lateral 1x1 conv + add) is fused
into the lateral conv's epilogue, and the three outputs are NHWC in memory, ready for the aggregation kernel.
Training / CPU: the stock PyTorch modules (same parameters)."""
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from mdfnet_hip import layers, ops
from .base import ConvBNReLU


import os as _os
_HEADS = bool(int(_os.environ.get("MDF_FPN_HEADS_FUSED", "1")))    # dev A/B: 0 = every composed head as a launch of its own
_PAIR = bool(int(_os.environ.get("MDF_CONV_PAIR", "1")))      # dev A/B: 0 = the two full-resolution layers as separate launches


def _stage(cin, cout, first_k, first_s):
    return nn.Sequential(ConvBNReLU(cin, cout, first_k, first_s, (first_k - 1) // 2), ConvBNReLU(cout, cout, 3, 1, 1),
                         ConvBNReLU(cout, cout, 3, 1, 1))


class FPN_4Scales(nn.Module):
    batch_views = True   # CoreNet may push all views through in one batched call (eval BN is per-sample)

    def __init__(self, out_chs: Tuple = (8, 16, 32, 64)) -> None:
        super().__init__()
        c0, c1, c2, c3 = out_chs
        self.conv01 = nn.Sequential(ConvBNReLU(3, c0, 3, 1, 1), ConvBNReLU(c0, c0, 3, 1, 1))   # 1/1
        self.conv12 = _stage(c0, c1, 5, 2)                                                     # 1/2
        self.conv23 = _stage(c1, c2, 5, 2)                                                     # 1/4
        self.conv34 = _stage(c2, c3, 5, 2)                                                     # 1/8
        self.lat2 = nn.Conv2d(c1, c3, 1, bias=True)
        self.lat3 = nn.Conv2d(c2, c3, 1, bias=True)
        self.out2 = nn.Conv2d(c3, c1, 1, bias=False)
        self.out3 = nn.Conv2d(c3, c2, 1, bias=False)
        self.out4 = nn.Conv2d(c3, c3, 1, bias=False)

    def _composed_heads(self):
        """The FPN head is linear: out(up(x) + lat(t) + b) = up(out(x)) + (out.lat)(t) + out.b  (1x1 convs commute with
        bilinear upsampling).  Composing the 1x1 weights once (fp64, rounded to fp32) means the 64-channel 1/4- and
        1/2-resolution tensors of backbone.py:60-63 (606 MB written and re-read per 5-view step) are never formed."""
        mods = (self.out2, self.out3, self.out4, self.lat2, self.lat3)
        tensors = [t for m in mods for t in (m.weight, m.bias) if t is not None]

        def build():
            def mat(m):
                return m.weight.detach().double().reshape(m.out_channels, m.in_channels)
            o2, o3, l2, l3 = mat(self.out2), mat(self.out3), mat(self.lat2), mat(self.lat3)
            b2, b3 = self.lat2.bias.detach().double(), self.lat3.bias.detach().double()

            def pack(w, bias=None):
                w4 = w.float().reshape(w.shape[0], w.shape[1], 1, 1).contiguous()
                return ops.pack_conv2d_weight(w4), (None if bias is None else bias.float().contiguous()), w.shape[1], w.shape[0]
            return {"y4": pack(mat(self.out4)), "a4": pack(o3), "y3": pack(o3 @ l3, o3 @ b3),
                    "c4": pack(o2), "c3": pack(o2 @ l3, o2 @ b3), "y2": pack(o2 @ l2, o2 @ b2)}
        return layers.cache_of(self.out2).get(tensors, build)

    def _hip_forward(self, x):
        def seq(blocks, t):
            for blk in blocks:
                t = layers.conv2d_layer(blk.conv, blk.bn, t, relu=True)
            return t

        def head(h, t, res_up=None):
            wp, bias, cin, cout = h
            return ops.conv2d_nhwc(t, wp, cin, cout, 1, 1, None, bias, False, None, 1.0, res_up)
        with torch.no_grad():
            # the first conv reads the planar NCHW images as they arrive (no 113 MB layout copy at cfg2)
            first, second = self.conv01[0], self.conv01[1]
            xin = x.float().contiguous()
            if _PAIR and first.conv.in_channels == 3 and first.conv.out_channels == 8 and second.conv.out_channels == 8 and len(self.conv01) == 2:
                t1 = layers.conv2d_pair(first, second, xin)        # both full-resolution layers in one launch (conv_pair.hip)
            else:
                t1 = layers.conv2d_layer(first.conv, first.bn, xin, relu=True, planar_in=True)
                t1 = seq(self.conv01[1:], t1)
            t2 = seq(self.conv12, t1)
            t3 = seq(self.conv23, t2)
            t4 = seq(self.conv34, t3)
            hd = self._composed_heads()
            if _HEADS and (hd["y4"][2], hd["y4"][3], hd["a4"][3], hd["c4"][3], hd["y3"][2], hd["y3"][3], hd["c3"][2], hd["c3"][3]) == (64, 64, 32, 16, 32, 32, 32, 16) \
                    and tuple(t3.shape[1:3]) == (2 * t4.shape[1], 2 * t4.shape[2]):       # the full tuple the kernel is built for: any other pyramid takes the per-head branch
                # the three heads of t4 and the two of t3 as one launch each: the input is streamed once (conv1x1_heads_kernel)
                y4, a4, c4 = ops.conv1x1_heads(t4, [hd["y4"], hd["a4"], hd["c4"]], [None, None, None])
                y3, c3 = ops.conv1x1_heads(t3, [hd["y3"], hd["c3"]], [a4, c4])    # out3(up(t4) + lat3(t3)); out2(up(t4) + lat3(t3)) @1/4
            else:
                y4 = head(hd["y4"], t4)
                y3 = head(hd["y3"], t3, res_up=head(hd["a4"], t4))            # out3(up(t4) + lat3(t3))
                c3 = head(hd["c3"], t3, res_up=head(hd["c4"], t4))            # out2(up(t4) + lat3(t3))        @1/4
            y2 = head(hd["y2"], t2, res_up=c3)                            # out2(up(up3) + lat2(t2))        @1/2
        return ops.from_nhwc(y4), ops.from_nhwc(y3), ops.from_nhwc(y2)

    def _heads(self, t2, t3, t4):
        if layers.hip_train(self, t2, t3, t4):
            from mdfnet_hip import train_ops
            return train_ops.fpn_heads_train(self, t2, t3, t4)
        up3 = F.interpolate(t4, scale_factor=2.0, mode="bilinear", align_corners=False) + self.lat3(t3)
        up2 = F.interpolate(up3, scale_factor=2.0, mode="bilinear", align_corners=False) + self.lat2(t2)
        return self.out4(t4), self.out3(up3), self.out2(up2)

    def forward_views(self, imgs):
        """Training on the GPU, all views at once: imgs [B,V,3,H,W] -> list over views of (f8, f4, f2).  The conv trunk
        (11 Conv2d + BatchNorm2d + ReLU layers, forward and backward) runs on the hand-written training kernels with one
        BatchNorm group per view -- the statistics of V separate calls, net/core.py:42 -- and the bias-only 1x1 heads with their
        bilinear top-down adds, which have no batch statistics, run batched on the same kernels (train_ops.FPNHeadsTrainFn)."""
        from mdfnet_hip import train_ops
        b, v = imgs.shape[:2]
        x = imgs.transpose(0, 1).reshape(v * b, *imgs.shape[2:])           # view-major: group g = view g
        ys = self._heads(*train_ops.trunk_train(self, x, groups=v))
        out = []
        for g in range(v):
            trip = tuple(y[g * b:(g + 1) * b] for y in ys)
            for t, y in zip(trip, ys):
                t._mdf_parent = (y, g, v)        # lets the aggregation slot take (and return the gradient of) the whole tensor at once
            out.append(trip)
        return out

    def forward(self, x: torch.Tensor):
        """[B,3,H,W] -> (1/8: 64ch, 1/4: 32ch, 1/2: 16ch)   (backbone.py:50-66)."""
        if layers.hip_eval(self, x):
            return self._hip_forward(x)
        if layers.hip_train(self, x):
            from mdfnet_hip import train_ops
            return self._heads(*train_ops.trunk_train(self, x, groups=1))
        t2 = self.conv12(self.conv01(x))
        t3 = self.conv23(t2)
        t4 = self.conv34(t3)
        up3 = F.interpolate(t4, scale_factor=2.0, mode="bilinear", align_corners=False) + self.lat3(t3)
        up2 = F.interpolate(up3, scale_factor=2.0, mode="bilinear", align_corners=False) + self.lat2(t2)
        return self.out4(t4), self.out3(up3), self.out2(up2)
