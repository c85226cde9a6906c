"""Glue between torch parameter containers (the reference's module tree, kept for state_dict compatibility) and
the fused conv kernels: per-layer cache of packed weights + folded BatchNorm, rebuilt when a tensor changes."""
import torch

from . import ops


class _Folded:
    """One cached, derived device value (packed weights, folded BN, ...).  The value is built by kernels enqueued on the
    stream current at build time; items in flight on OTHER streams (pipeline.InFlight) may consume it right away, so the
    build records an event and every consumer stream waits for it once -- until the event has completed, after which the
    check is a single `is None`."""

    def __init__(self):
        self.key, self.val, self.ev, self.seen = None, None, None, ()

    def get(self, tensors, build):
        key = tuple((t.data_ptr(), t._version, t.device.index) for t in tensors if t is not None)
        dev = next((t.device for t in tensors if t is not None and t.is_cuda), None)
        if key != self.key:
            with torch.no_grad():
                self.val = build()
            self.ev, self.seen = None, ()
            if dev is not None:
                cur = torch.cuda.current_stream(dev)
                self.ev = torch.cuda.Event()
                self.ev.record(cur)
                self.seen = (cur.cuda_stream,)
            self.key = key
        elif self.ev is not None:
            if self.ev.query():
                self.ev = None                      # build finished: nothing to order against any more
            else:
                cur = torch.cuda.current_stream(dev)
                if cur.cuda_stream not in self.seen:
                    cur.wait_event(self.ev)
                    self.seen = self.seen + (cur.cuda_stream,)
        return self.val


def cache_of(mod):
    c = mod.__dict__.get("_mdf_cache")
    if c is None:
        c = mod.__dict__["_mdf_cache"] = _Folded()
    return c


def cache_of_key(mod, name):
    """Several derived values per module (forward pack, input-gradient pack, ...)."""
    d = mod.__dict__.get("_mdf_caches")
    if d is None:
        d = mod.__dict__["_mdf_caches"] = {}
    c = d.get(name)
    if c is None:
        c = d[name] = _Folded()
    return c


_NO_CPU = ("the slots run on hand-written MI355X kernels only (got a CPU tensor in {} mode); there is no CPU route in the product.  "
           "Rehearsals of the drivers on machines without a GPU select the stock-op backend explicitly: `import rehearsal; rehearsal.enable()` "
           "(mdf-net_amd/rehearsal/)")


def hip_train(mod, *tensors):
    """True when the slot runs its TRAINING mode on the hand-written kernels: module in training mode (or an enclosing CoreNet in
    training mode) with its tensors on a GPU.  CPU tensors in training mode RAISE unless the rehearsal backend has been selected
    (mdf-net_amd/rehearsal: gloo rehearsals, the autograd baseline of bench.py) -- it is not part of the product's dispatch."""
    ts = [t for t in tensors if isinstance(t, torch.Tensor)]
    training = mod.training if mod is not None else bool(getattr(_mode, "training", False))
    if _TRAIN_STOCK and _REHEARSAL is not None:        # rehearsal.enable(on_gpu=True): PyTorch-ROCm autograd instead of the HIP training kernels
        return False
    if training and len(ts) > 0 and not all(t.is_cuda for t in ts) and _REHEARSAL is None:
        raise RuntimeError(_NO_CPU.format("training"))
    return training and len(ts) > 0 and all(t.is_cuda for t in ts)


def stock():
    """The rehearsal backend's slot functions (mdf-net_amd/rehearsal/stockops.py); raises unless it was selected."""
    if _REHEARSAL is None:
        raise RuntimeError(_NO_CPU.format("training"))
    return _REHEARSAL


def hip_eval(mod, x):
    """True when the fused HIP path applies: eval mode on a GPU tensor."""
    return (not mod.training) and x.is_cuda


import contextlib
import os
import threading

_TRAIN_STOCK = False     # set by rehearsal.enable(on_gpu=True)
_REHEARSAL = None        # the stock-op backend module once rehearsal.enable() was called

_mode = threading.local()


class StageCuts:
    """Where a recorded training step (graphstep.GraphedTrainStep) splits its backward pass: CoreNet's training forward replaces
    every stage's feature tensor and every stage's depth by a detached copy that requires grad and notes the pairs here.  The
    gradient is the same, but it can then be taken in pieces -- loss -> d depth per stage | one stage's regulariser + aggregation
    (three chains that share nothing: the hypotheses are built under no_grad, depthhypos.py:40,188) | feature pyramid + trunk --
    and each piece recorded as a hipGraph of its own, the three stage chains replayed side by side on three streams."""

    def __init__(self):
        self.feat, self.depth, self.streams = [], [], None      # [stage] -> [(tensor, cut)], (depth, cut); the stages' streams
        self.refine = None                                      # (refined depth, cut)


@contextlib.contextmanager
def stage_cuts():
    prev = getattr(_mode, "cuts", None)
    _mode.cuts = StageCuts()
    try:
        yield _mode.cuts
    finally:
        _mode.cuts = prev


def active_cuts():
    return getattr(_mode, "cuts", None)


@contextlib.contextmanager
def model_mode(training):
    """CoreNet.forward announces its mode so that the plain-function slots (depth_regression, homo_warping, ...) follow the
    model's mode even under torch.no_grad()."""
    prev = getattr(_mode, "training", None)
    _mode.training = bool(training)
    try:
        yield
    finally:
        _mode.training = prev


def use_hip(mod, *tensors):
    """Slot dispatch for INFERENCE.  A module in eval mode with no autograd graph wanted runs the hand-written eval kernels and
    REFUSES CPU tensors -- there is no CPU fallback.  In training mode (module.training, an enclosing CoreNet in training mode, or
    inputs that require grad) the callers have asked `hip_train` first (GPU tensors -> the hand-written training kernels of
    train_ops.py); what is left is the rehearsal backend if it was selected explicitly (returns False), an error otherwise.
    `mod` is None for plain functions."""
    ts = [t for t in tensors if isinstance(t, torch.Tensor)]
    training = mod.training if mod is not None else bool(getattr(_mode, "training", False))
    if training or (torch.is_grad_enabled() and any(t.requires_grad for t in ts)):
        if _REHEARSAL is None:
            raise RuntimeError(_NO_CPU.format("training") if not all(t.is_cuda for t in ts) else
                               "this slot has no hand-written training kernel for the given call (autograd wanted outside CoreNet's training "
                               "step); the stock-op route is the rehearsal backend: `import rehearsal; rehearsal.enable(on_gpu=True)`")
        return False
    if not all(t.is_cuda for t in ts):
        raise RuntimeError(_NO_CPU.format("eval"))
    return True


def _folded(conv, bn):
    """(packed weights, alpha, beta) of a Conv2d [+ BatchNorm2d eval], cached per layer."""
    tensors = [conv.weight, conv.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])

    def build():
        wp = ops.pack_conv2d_weight(conv.weight)
        if bn is not None:
            alpha, beta = ops.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
            if conv.bias is not None:
                beta = (beta + conv.bias.float() * alpha).contiguous()
            return wp, alpha, beta
        return wp, None, (None if conv.bias is None else conv.bias.detach().float().contiguous())
    return cache_of(conv).get(tensors, build)


def conv2d_pair(block0, block1, x):
    """The feature pyramid's two full-resolution ConvBNReLU layers (3 -> 8 -> 8, k3) on the planar images as one launch."""
    w1, a1, b1 = _folded(block0.conv, block0.bn)
    w2, a2, b2 = _folded(block1.conv, block1.bn)
    return ops.conv2d_pair_planar(x, w1, a1, b1, w2, a2, b2)


def res_block(conv_a, conv_b, x, scale=0.1):
    """Res (net/unit/base.py:39-47): x + scale * conv_b(relu(conv_a(x))), both Conv2d(8,8,k3,p1) without bias, as one launch."""
    assert conv_a.bias is None and conv_b.bias is None and conv_a.in_channels == conv_b.out_channels == 8
    wa, _, _ = cache_of(conv_a).get([conv_a.weight, conv_a.bias], lambda: (ops.pack_conv2d_weight(conv_a.weight), None, None))
    wb, _, _ = cache_of(conv_b).get([conv_b.weight, conv_b.bias], lambda: (ops.pack_conv2d_weight(conv_b.weight), None, None))
    return ops.conv2d_res_pair(x, wa, wb, scale)


def refine_tail(conv_a, conv_b, x, lo, span):
    """RefineNet2.conv2 (Conv2d(8,32) -> PixelShuffle(2) -> Conv2d(8,1), no bias) + the mapping lo + y*span as one launch."""
    assert conv_a.bias is None and conv_b.bias is None
    wp, _, _ = cache_of(conv_a).get([conv_a.weight, conv_a.bias], lambda: (ops.pack_conv2d_weight(ops.shuffle2_rows(conv_a.weight)), None, None))
    return ops.refine_tail(x, wp, conv_b.weight, lo, span)


def conv2d_layer(conv, bn, x, relu=False, res=None, res_scale=1.0, res_up=None, planar_in=False, pixel_shuffle2=False):
    """Conv2d [+ BatchNorm2d eval] [+ ReLU] [+ residual / upsample-add] [+ PixelShuffle(2)] as one kernel.  x, res: [B,H,W,C] NHWC."""
    assert not (pixel_shuffle2 and (bn is not None or conv.bias is not None))
    k = conv.kernel_size[0]
    tensors = [conv.weight, conv.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])

    def build():
        wp = ops.pack_conv2d_weight(ops.shuffle2_rows(conv.weight) if pixel_shuffle2 else conv.weight)
        if bn is not None:
            alpha, beta = ops.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
            if conv.bias is not None:
                beta = (beta + conv.bias.float() * alpha).contiguous()
            return wp, alpha, beta
        return wp, None, (None if conv.bias is None else conv.bias.detach().float().contiguous())

    wp, alpha, beta = cache_of(conv).get(tensors, build)
    return ops.conv2d_nhwc(x, wp, conv.in_channels, conv.out_channels, k, conv.stride[0], alpha, beta, relu, res, res_scale, res_up,
                           planar_in, pixel_shuffle2=pixel_shuffle2)
