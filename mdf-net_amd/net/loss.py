"""Training loss (reference: net/loss.py:10-27): smooth-L1 (mean) over the 4 output scales on gt > depth_min."""
from typing import Dict

import torch
import torch.nn.functional as F


class Loss(torch.nn.Module):
    def forward(self, outputs: Dict, depth_gt: Dict, depth_range: torch.Tensor) -> torch.Tensor:
        total = 0.0
        floor = depth_range[:, 0].view(-1, 1, 1)
        for est, gt in zip(outputs["depth"], depth_gt.values()):
            valid = gt > floor
            for e in (est if isinstance(est, (list, tuple)) else [est]):
                if e.is_cuda:
                    # the same masked mean without boolean indexing: `e[valid]` has a data-dependent size, i.e. a device->host
                    # synchronisation in the middle of every training step (2.7 ms of stalled issue at cfg3)
                    per = F.smooth_l1_loss(e, gt, reduction="none")
                    total = total + (per * valid).sum() / valid.sum()
                else:
                    total = total + F.smooth_l1_loss(e[valid], gt[valid], reduction="mean")
        return total
