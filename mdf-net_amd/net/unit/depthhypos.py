"""Depth-hypothesis slot (reference: net/unit/depthhypos.py:10-215)."""
import torch
import torch.nn as nn

from mdfnet_hip import controlplane, hostmirror, layers, ops

_MODES = {"gauss1": 1, "laplace": 2}


class HyposByFit(nn.Module):
    """HyposByFit(ndepths, curve_calss, prob_thresh); no parameters or buffers (contributes nothing to
    the state_dict, like the reference whose prob_thresh is a plain tensor attribute)."""

    def __init__(self, ndepths: int = 16, curve_calss: str = "gauss1", prob_thresh: float = 0.95) -> None:
        super().__init__()
        self.ndepths, self.curve_calss = ndepths, curve_calss
        self.prob_thresh = torch.tensor(prob_thresh)

    def uniform_host(self, depth_range):
        """depthhypos.py:31-38 on the host (B*D floats; GPU `tensor / int` rounds differently) -> [B,D,1,1] CPU."""
        dr = hostmirror.get(depth_range)
        b = dr.shape[0]
        lo = dr[:, 0].float().reshape(b, 1)
        step = (dr[:, 1].float().reshape(b, 1) - lo) / (self.ndepths - 1)
        return (lo + torch.arange(0, self.ndepths).reshape(1, -1) * step).reshape(b, self.ndepths, 1, 1).contiguous()

    def _uniform(self, depth_range):
        plan = controlplane.for_range(depth_range)
        if plan is not None and plan.hyp0 is not None and plan.hyp0.shape[1] == self.ndepths:
            return plan.hyp0                                  # uploaded with the rest of the forward's control plane
        host = self.uniform_host(depth_range)
        if depth_range.is_cuda:
            return hostmirror.put(host.to(depth_range.device, non_blocking=True), host)
        return host

    def forward(self, depth, depth_range, prob_volume, depth_hypos, upsample=False):
        if depth is None:
            return self._uniform(depth_range)
        on_gpu = depth.is_cuda and prob_volume.is_cuda      # no gradient flows here (depthhypos.py:40): same kernels in training
        if not on_gpu and not layers.use_hip(self, depth.detach(), prob_volume.detach()):
            if self.curve_calss not in _MODES:
                raise NotImplementedError(f"HyposByFit curve '{self.curve_calss}' is not built (gauss1, laplace are)")
            return layers.stock().hypos_by_fit(self.curve_calss, self.prob_thresh, self.ndepths, depth.detach(), depth_range,
                                         prob_volume.detach(), depth_hypos.detach(), upsample)
        mode = _MODES.get(self.curve_calss)
        if mode is None:
            raise NotImplementedError(f"HyposByFit curve '{self.curve_calss}' is not built (gauss1, laplace are)")
        with torch.no_grad():
            row = None
            plan = controlplane.for_range(depth_range)
            if mode == 1:
                if depth_hypos.shape[-1] != 1:
                    raise NotImplementedError("gauss1 fit needs hypotheses shared by all pixels ([B,D,1,1])")
                if plan is not None and plan.fit_row is not None and depth_hypos is plan.hyp0:
                    row = plan.fit_row
                else:
                    row = ops.gauss1_fit_row(hostmirror.get(depth_hypos)).to(depth.device, non_blocking=True)
            s = ops.hypos_fit(mode, prob_volume, depth, depth_hypos, row)
            rng = plan.rng if plan is not None else hostmirror.get(depth_range).float().contiguous().to(depth.device, non_blocking=True)
            log_thr = ops.recorded("log_thresh", mode)
            if log_thr is None:
                log_thr = float(torch.log(self.prob_thresh))
            return ops.hypos_from_fit(mode, s, depth, rng, log_thr, self.ndepths, bool(upsample))
