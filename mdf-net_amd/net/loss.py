"""Training loss (reference: net/loss.py:10-27): smooth-L1 (mean) over the 4 output scales on gt > depth_min."""
from typing import Dict

import torch
import torch.nn.functional as F


class Loss(torch.nn.Module):
    def forward(self, outputs: Dict, depth_gt: Dict, depth_range: torch.Tensor) -> torch.Tensor:
        pairs = [(e, gt) for est, gt in zip(outputs["depth"], depth_gt.values())
                 for e in (est if isinstance(est, (list, tuple)) else [est])]
        if pairs and all(e.is_cuda and gt.is_cuda and e.dtype == gt.dtype == torch.float32 and e.shape == gt.shape for e, gt in pairs) \
                and depth_range.is_cuda:
            # on the GPU: one reduction launch per scale + one finalize (and one backward launch per scale) instead of ~55
            # tiny elementwise / reduction launches (mdfnet_hip/train_ops.py:LossTrainFn, csrc/loss.hip)
            from mdfnet_hip import layers, train_ops
            if not layers._TRAIN_STOCK:      # (rehearsal.enable(on_gpu=True): the autograd baseline)
                return train_ops.loss_train(depth_range[:, 0], pairs)
        total = 0.0
        floor = depth_range[:, 0].view(-1, 1, 1)
        for e, gt in pairs:
            valid = gt > floor
            if e.is_cuda:
                # the same masked mean without boolean indexing: `e[valid]` has a data-dependent size, i.e. a device->host
                # synchronisation in the middle of every training step
                per = F.smooth_l1_loss(e, gt, reduction="none")
                # where(), not `per * valid`: a non-finite estimate at an INVALID pixel must not poison the sum (NaN * 0 = NaN);
                # the reference's e[valid] never sees it
                total = total + torch.where(valid, per, per.new_zeros(())).sum() / valid.sum()
            else:
                total = total + F.smooth_l1_loss(e[valid], gt[valid], reduction="mean")
        return total
