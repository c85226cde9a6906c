"""Regress slot: soft-argmin depth and photometric confidence (reference: net/unit/regress.py)."""
from mdfnet_hip import layers, ops


def depth_regression(prob_volume, depth_hypos):
    """regress.py:5-7: sum_d prob * hypos.  prob [B,D,h,w]; hypos [B,D,1,1] | [B,D,h,w] -> [B,h,w]."""
    if not layers.use_hip(None, prob_volume, depth_hypos):
        return layers.stock().depth_regression(prob_volume, depth_hypos)   # rehearsal backend only
    return ops.depth_regress(prob_volume, depth_hypos)


depth_regression.mdf_builtin = True   # lets CoreNet fuse it into the regulariser's last kernel


def confidence_regress(prob_volume, last_confidence=None, n=4, pad=(0, 0, 0, 0, 1, 2)):
    """regress.py:9-25: sum of the 4 probabilities around trunc(E[d]).  Only the configuration the model
    uses (n=4, pad=(1,2) on D, no last_confidence) is built."""
    if last_confidence is not None or n != 4 or tuple(pad) != (0, 0, 0, 0, 1, 2):
        raise NotImplementedError("confidence_regress: only n=4, pad=(0,0,0,0,1,2), last_confidence=None is built")
    return ops.confidence(prob_volume.detach())


confidence_regress.mdf_builtin = True   # lets CoreNet fold the nearest x2 upsampling of core.py:76 into the same launch
