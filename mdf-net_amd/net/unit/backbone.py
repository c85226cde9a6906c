"""Feature pyramid (reference: net/unit/backbone.py:9-66).  Surface only: stock PyTorch-ROCm 2-D convs
(MIOpen), run in channels_last so the three outputs are already NHWC for the aggregation kernel."""
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .base import ConvBNReLU


def _stage(cin, cout, first_k, first_s):
    return nn.Sequential(ConvBNReLU(cin, cout, first_k, first_s, (first_k - 1) // 2), ConvBNReLU(cout, cout, 3, 1, 1),
                         ConvBNReLU(cout, cout, 3, 1, 1))


class FPN_4Scales(nn.Module):
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

    def forward(self, x: torch.Tensor):
        """[B,3,H,W] -> (1/8: 64ch, 1/4: 32ch, 1/2: 16ch)   (backbone.py:50-66)."""
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        t2 = self.conv12(self.conv01(x))
        t3 = self.conv23(t2)
        t4 = self.conv34(t3)
        up3 = F.interpolate(t4, scale_factor=2.0, mode="bilinear", align_corners=False) + self.lat3(t3)
        up2 = F.interpolate(up3, scale_factor=2.0, mode="bilinear", align_corners=False) + self.lat2(t2)
        return self.out4(t4), self.out3(up3), self.out2(up2)
