"""Depth refinement / x2 upsampling (reference: net/unit/refine.py:8-46).  Surface only: stock 2-D convs."""
import torch
import torch.nn as nn

from .base import Res


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
        lo = depth_range[:, 0].float().view(b, 1, 1, 1)
        span = depth_range[:, 1].float().view(b, 1, 1, 1) - lo
        x0 = self.conv0((depth.detach().unsqueeze(1) - lo) / span)
        y = x0
        for blk in self.ress:
            y = blk(y)
        y = self.conv2(x0 + self.conv1(y))
        return (lo + y * span).squeeze(1)
