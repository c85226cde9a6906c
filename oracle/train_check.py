"""TEST INFRASTRUCTURE: autograd runs of the oracle's training-mode operators in float32 and float64 -- the yardstick of the
per-operator tests in tests/test_train_gpu.py (the product package's own CPU route, mdfnet_hip/stockops.py, is NOT the checker).

Every runner takes a module's state_dict (the product's modules are only parameter containers here), re-creates the floating-point
entries as autograd leaves of the requested dtype and calls the functional restatement in oracle/mvs_oracle.py under
`precision(dtype)`.  nn.BatchNorm's running-statistics side effect is reproduced from the recorded batch statistics
(`expected_buffers`)."""
import contextlib

import torch

from . import mvs_oracle as O


def leaf_params(sd, dtype):
    """state_dict -> dict of tensors of `dtype`; parameters (not running statistics / counters) require grad."""
    out = {}
    for k, v in sd.items():
        if v.is_floating_point():
            t = v.detach().cpu().to(dtype).clone()
            if "running_" not in k:
                t.requires_grad_(True)
            out[k] = t
        else:
            out[k] = v.detach().cpu().clone()
    return out


@contextlib.contextmanager
def bn_trace():
    prev, O.BN_TRACE = O.BN_TRACE, []
    try:
        yield O.BN_TRACE
    finally:
        O.BN_TRACE = prev


def expected_buffers(sd, trace, momentum=0.1):
    """The BatchNorm buffers after the traced training-mode calls, applied in call order (a module called once per source view
    or per image updates its running statistics that many times): running = (1 - m) * running + m * batch (variance unbiased)."""
    out = {k: v.detach().cpu().clone() for k, v in sd.items() if "running_" in k or "num_batches_tracked" in k}
    for pre, mean, var in trace:
        out[pre + "running_mean"] = (1 - momentum) * out[pre + "running_mean"].to(mean.dtype) + momentum * mean
        out[pre + "running_var"] = (1 - momentum) * out[pre + "running_var"].to(var.dtype) + momentum * var
        out[pre + "num_batches_tracked"] = out[pre + "num_batches_tracked"] + 1
    return out


@contextlib.contextmanager
def relu_hook(fn):
    """While active, every ReLU of the regularisers (oracle.mvs_oracle._relu3) is `fn(pre_activation)`: record the pre-activations, or
    replace the decision `pre > 0` by a given mask (the run then follows another implementation's ReLU decisions)."""
    prev, O.RELU3_HOOK = O.RELU3_HOOK, fn
    try:
        yield
    finally:
        O.RELU3_HOOK = prev


def grads(params):
    return {k: v.grad for k, v in params.items() if v.is_floating_point() and v.requires_grad}


def regulariser(sd, cost, hypos, ddepth, dtype=torch.float32, relu=None):
    """Regular[s] + soft-argmin (regular.py:47-69 / :114-133, regress.py:5-7), training mode, backward from d depth.
    relu: optional replacement of the ReLUs (relu_hook).  -> dict(prob, depth, dcost, grads, buffers)."""
    p = leaf_params(sd, dtype)
    c = cost.detach().to(dtype).clone().requires_grad_(True)
    with O.precision(dtype), bn_trace() as tr, relu_hook(relu):
        prob = O.regular(c, p, training=True)
        depth = O.depth_regression(prob, hypos.to(dtype))
    depth.backward(ddepth.to(dtype))
    return {"prob": prob.detach(), "depth": depth.detach(), "dcost": c.grad, "grads": grads(p), "buffers": expected_buffers(sd, tr)}


def aggregate(sd, ngroups, feats, ref_proj, src_projs, hypos, dcost, dtype=torch.float32):
    """VectorAggregate.forward (homoaggregate.py:25-46) in training mode (batch-statistics BatchNorm3d(1) per source view),
    backward from d cost.  -> dict(cost, dfeats, grads, buffers)."""
    p = leaf_params(sd, dtype)
    f = [t.detach().to(dtype).clone().requires_grad_(True) for t in feats]
    with O.precision(dtype), bn_trace() as tr:
        cost = O.vector_aggregate(f, ref_proj.to(dtype), tuple(s.to(dtype) for s in src_projs), hypos.to(dtype), ngroups, p, training=True)
    cost.backward(dcost.to(dtype))
    return {"cost": cost.detach(), "dfeats": [t.grad for t in f], "grads": grads(p), "buffers": expected_buffers(sd, tr)}


def pyramid(sd, imgs, gouts, dtype=torch.float32):
    """FPN_4Scales (backbone.py:50-66) called once per view in training mode (core.py:42), backward from the given output gradients
    [view][level].  -> dict(outs [view][level], grads, buffers)."""
    p = leaf_params(sd, dtype)
    with O.precision(dtype), bn_trace() as tr:
        outs = [O.fpn_4scales(imgs[:, i].to(dtype), p, training=True) for i in range(imgs.shape[1])]
    sum((t * g.to(dtype)).sum() for o, go in zip(outs, gouts) for t, g in zip(o, go)).backward()
    return {"outs": [[t.detach() for t in o] for o in outs], "grads": grads(p), "buffers": expected_buffers(sd, tr)}


def refine(sd, depth, depth_range, gout, dtype=torch.float32):
    """RefineNet2 (refine.py:25-46), backward from d output.  -> dict(out, grads)."""
    p = leaf_params(sd, dtype)
    with O.precision(dtype):
        out = O.refine_net2(depth.to(dtype), depth_range, p)
    (out * gout.to(dtype)).sum().backward()
    return {"out": out.detach(), "grads": grads(p)}
