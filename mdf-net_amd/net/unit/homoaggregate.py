"""Cost-volume aggregation slot (reference: net/unit/homoaggregate.py)."""
import torch
import torch.nn as nn

from mdfnet_hip import controlplane, hostmirror, layers, ops
from .base import ConvBNReLU3D


def _projections(ref_proj, src_projs, device):
    ready = controlplane.projections(ref_proj)                 # uploaded with the rest of the forward's control plane
    if ready is not None and ready.shape[0] == len(src_projs):
        return ready
    host = ops.relative_projections(hostmirror.get(ref_proj), [hostmirror.get(s) for s in src_projs])
    return host.to(device, non_blocking=True)


class VectorAggregate(nn.Module):
    """Group-wise softmax similarity with a learned per-voxel view weight (homoaggregate.py:8-46),
    fused with the plane-sweep warp into one kernel.  state_dict keys: depth_weight.0.{conv,bn}.*,
    depth_weight.1.{weight,bias}.  Returns cost [B,G,D,h,w] whose memory is NDHWC (what the
    regulariser's conv kernels read)."""

    def __init__(self, ngroups: int = 8):
        super().__init__()
        self.ngroups = ngroups
        self.depth_weight = nn.Sequential(ConvBNReLU3D(ngroups, 1, 1, 1, 0), nn.Conv3d(1, 1, 1, 1, 0), nn.Sigmoid())

    def _params(self):
        sd = {k: v for k, v in self.depth_weight.state_dict(keep_vars=True).items()}
        return layers.cache_of(self).get(list(sd.values()), lambda: ops.fold_view_weight(sd, self.ngroups, prefix=""))

    def forward(self, features, ref_proj, src_projs, depth_hypos):
        if layers.hip_train(self, *features, depth_hypos):
            # training on the GPU: two-pass batch-statistics BatchNorm3d(1), backward as an atomic scatter (train_ops.py)
            from mdfnet_hip import train_ops
            proj = _projections(ref_proj, src_projs, features[0].device)
            return train_ops.aggregate_train(self, list(features), proj, depth_hypos)
        if not layers.use_hip(self, *features, depth_hypos):
            # training path (autograd, batch-stat BN in depth_weight): the rehearsal backend only (mdf-net_amd/rehearsal/stockops.py)
            return layers.stock().vector_aggregate(self.depth_weight, self.ngroups, features, ref_proj, src_projs, depth_hypos)
        with torch.no_grad():
            proj = _projections(ref_proj, src_projs, features[0].device)
            return ops.warp_aggregate_vec(list(features), proj, depth_hypos, self._params(), self.ngroups)


def homo_aggregate_by_variance(features, ref_proj, src_projs, depth_hypos):
    """homoaggregate.py:49-69: variance over {ref, softmax_C(warped src)} -> [B,C,D,h,w]."""
    if not layers.use_hip(None, *features, depth_hypos):
        return layers.stock().variance_aggregate(features, ref_proj, src_projs, depth_hypos)
    with torch.no_grad():
        proj = _projections(ref_proj, src_projs, features[0].device)
        return ops.warp_aggregate_var(list(features), proj, depth_hypos)
