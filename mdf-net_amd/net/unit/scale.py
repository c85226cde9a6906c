"""Camera scaling slot (`scale`), reference: net/unit/scale.py:4-20."""
import torch

from mdfnet_hip import hostmirror, ops


def scale_cam(intrinsics, extrinsics, stage):
    """(K [B,V,3,3], E [B,V,4,4], stage) -> (ref_proj [B,4,4], tuple of V-1 src_proj [B,4,4]).

    level = 3 - stage; K[:2] / 2**level; P[:3,:4] = K @ E[:3,:4].  The 4x4s are control-plane data:
    they are computed on the HOST (fp32, same torch ops as the reference's CPU path) and handed back on
    the inputs' device with a host mirror attached, so the aggregation slot can form
    src_proj @ inverse(ref_proj) with the reference's LAPACK arithmetic and without a device sync.
    Inputs are not modified (scale.py:14)."""
    dev = intrinsics.device
    k = hostmirror.get(intrinsics).float().clone()
    e = hostmirror.get(extrinsics).float()
    k[:, :, :2, :] = k[:, :, :2, :] / float(2 ** (3 - stage))
    p = e.clone()
    cam = ops.recorded("cams")                             # parity tests only: the build host's own product
    p[:, :, :3, :4] = torch.matmul(k, e[:, :, :3, :4]) if cam is None else cam
    if dev.type == "cpu":
        views = [v.contiguous() for v in p.unbind(1)]
        return views[0], tuple(views[1:])
    host = p.permute(1, 0, 2, 3).contiguous()            # [V,B,4,4]: ONE host->device copy, the views share its storage
    on_dev = host.to(dev, non_blocking=True)
    out = [hostmirror.put(on_dev[v], host[v]) for v in range(host.shape[0])]
    return out[0], tuple(out[1:])
