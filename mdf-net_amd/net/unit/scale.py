"""Camera scaling slot (`scale`), reference: net/unit/scale.py:4-20."""
import torch

from mdfnet_hip import controlplane, hostmirror, ops


def host_cameras(intrinsics, extrinsics, stage):
    """HOST part of scale_cam: -> [V,B,4,4] float32 CPU (view v = P_v of the stage)."""
    k = hostmirror.get(intrinsics).float().clone()
    e = hostmirror.get(extrinsics).float()
    k[:, :, :2, :] = k[:, :, :2, :] / float(2 ** (3 - stage))
    p = e.clone()
    cam = ops.recorded("cams")                             # parity tests only: the build host's own product
    p[:, :, :3, :4] = torch.matmul(k, e[:, :, :3, :4]) if cam is None else cam
    return p.permute(1, 0, 2, 3).contiguous()


def scale_cam(intrinsics, extrinsics, stage):
    """(K [B,V,3,3], E [B,V,4,4], stage) -> (ref_proj [B,4,4], tuple of V-1 src_proj [B,4,4]).

    level = 3 - stage; K[:2] / 2**level; P[:3,:4] = K @ E[:3,:4].  The 4x4s are control-plane data:
    they are computed on the HOST (fp32, same torch ops as the reference's CPU path) and handed back on
    the inputs' device with a host mirror attached, so the aggregation slot can form
    src_proj @ inverse(ref_proj) with the reference's LAPACK arithmetic and without a device sync.
    Inputs are not modified (scale.py:14)."""
    dev = intrinsics.device
    ready = controlplane.cams(intrinsics, extrinsics, stage)      # uploaded with the rest of the forward's control plane
    if ready is not None:
        return ready
    host = host_cameras(intrinsics, extrinsics, stage)   # [V,B,4,4]
    if dev.type == "cpu":
        views = [v.contiguous() for v in host.unbind(0)]
        return views[0], tuple(views[1:])
    on_dev = host.to(dev, non_blocking=True)             # ONE host->device copy, the views share its storage
    out = [hostmirror.put(on_dev[v], host[v]) for v in range(host.shape[0])]
    return out[0], tuple(out[1:])
