"""Training path on the hand-written kernels: torch-facing wrappers over the `*_bwd` / train-mode entries of the C ABI
(include/mdfnet_hip.h, "Training path") and the two autograd nodes that put them behind the reference's slots:

  AggregateTrainFn    Homoaggre[s] in training mode (net/unit/homoaggregate.py:25-46 with the batch-statistics
                      BatchNorm3d(1) of :16-20; gradient to the features only, base.py:97)
  RegulariserTrainFn  Regular[s] + Depth_regress in training mode (net/unit/regular.py:47-69,114-133, regress.py:5-7):
                      every Conv3d/ConvTranspose3d + BatchNorm3d(batch statistics) + ReLU (+ skip) layer, the `prob`
                      conv, softmax over D and the soft-argmin, forward and backward

PyTorch is plumbing (memory, streams, the autograd graph between the slots); no op here falls back to ATen compute.
"""
import ctypes
import os as _os

import torch

from . import lib, check
from . import ops
from .ops import _abi, _f32c, _need_gpu, _stream, _src_array, _hypos_arg

def _momentum(bn):
    """nn.BatchNorm's momentum=None means a CUMULATIVE average (factor 1/num_batches_tracked), which the finalize kernels do
    not implement; the reference never uses it (base.py:50-68: default 0.1)."""
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm(momentum=None) (cumulative moving average) is not built into the training kernels")
    return bn.momentum


# --------------------------------------------------------------------------- BatchNorm(batch stats) + ReLU
# Tensors are channels-last [groups][n][c]: every group is one call of the module with its own batch statistics (the
# regulariser layers: 1 group; the feature pyramid, called once per view: one group per view).
class ZeroPool:
    """Zeroed fp64 scratch for the reduction kernels of one forward or backward sweep: one fill per sweep instead of one
    allocation + fill launch per layer (the step is host-bound at ~900 launches, every launch saved is ~10 us)."""

    def __init__(self, device, n=16384):
        self.device, self.n = device, n
        self.buf, self.off = torch.zeros(n, device=device, dtype=torch.float64), 0
        self.held_by_recording = 0          # graphstep: number of live hipGraph recordings that hold self.buf's address

    def take(self, k):
        if self.off + k > self.n:
            if self.held_by_recording:
                # replacing the buffer would free memory every replay still fills and accumulates into
                raise RuntimeError("the step's reduction pool overflowed while a recorded training step holds its buffer: call "
                                   "slots outside a training step with their own ZeroPool, or enlarge train_ops.step_pool")
            self.buf, self.off = torch.zeros(max(self.n, k), device=self.device, dtype=torch.float64), 0
        v = self.buf[self.off:self.off + k]
        self.off += k
        return v

    def reset(self):
        """Start of a training step: every slice handed out so far has been consumed (same stream), one fill re-arms them all."""
        self.buf.zero_()
        self.off = 0


_STEP_POOLS = {}


def step_pool(device):
    """The reduction scratch of the whole step (forward and backward sweeps of every slot): zeroed ONCE per step by prepack();
    without that reset (a slot called on its own) it simply keeps handing out fresh zeroed slices."""
    p = _STEP_POOLS.get(device)
    if p is None:
        p = _STEP_POOLS[device] = ZeroPool(device, 1 << 19)     # 4 MiB: the reductions of one step (the conv epilogues' sums in STAT_SLICES copies)
    return p


def bn_stats(y, n, c, groups=1, pool=None):
    sums = torch.zeros(groups * 2 * c, device=y.device, dtype=torch.float64) if pool is None else pool.take(groups * 2 * c)
    _abi("mdf_bn_stats_fwd", (y.data_ptr(), n, c, groups, sums.data_ptr(), _stream(y)), tag=f"stats C{c} N{n}x{groups}",
         work={"bytes": 4.0 * n * c * groups, "bound": "hbm"})
    return sums


def bn_finalize(sums, bn, n, c, groups=1):
    """-> aux [groups][4C] = (a, b, mean, invstd); updates the module's running statistics like `groups` successive calls of
    nn.BatchNorm.train()."""
    aux = torch.empty(groups * 4 * c, device=sums.device, dtype=torch.float32)
    track = bn.track_running_stats and bn.running_mean is not None
    mom = _momentum(bn)
    _abi("mdf_bn_finalize_fwd", (sums.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), ctypes.c_float(bn.eps), ctypes.c_float(mom),
                                 n, c, groups, aux.data_ptr(), bn.running_mean.data_ptr() if track else None,
                                 bn.running_var.data_ptr() if track else None,
                                 bn.num_batches_tracked.data_ptr() if track else None, sums.numel() // (groups * 2 * c), _stream(aux)))
    return aux


def bn_finalize_apply(sums, bn, y, res, n, c, groups=1):
    """finalize + apply in one launch -> (z, aux)."""
    aux = torch.empty(groups * 4 * c, device=y.device, dtype=torch.float32)
    z = torch.empty_like(y)
    track = bn.track_running_stats and bn.running_mean is not None
    mom = _momentum(bn)
    _abi("mdf_bn_finalize_apply_fwd", (y.data_ptr(), sums.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), ctypes.c_float(bn.eps),
                                       ctypes.c_float(mom), None if res is None else res.data_ptr(), z.data_ptr(), aux.data_ptr(),
                                       bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
                                       bn.num_batches_tracked.data_ptr() if track else None, n, c, groups, sums.numel() // (groups * 2 * c),
                                       _stream(z)),
         tag=f"finalize+apply C{c} N{n}x{groups}", work={"bytes": 4.0 * n * c * groups * (3 if res is not None else 2), "bound": "hbm"})
    return z, aux


def bn_relu_apply(y, aux, res, n, c, groups=1):
    z = torch.empty_like(y)
    _abi("mdf_bn_relu_apply_fwd", (y.data_ptr(), aux.data_ptr(), None if res is None else res.data_ptr(), z.data_ptr(), n, c, groups,
                                   _stream(z)),
         tag=f"apply C{c} N{n}x{groups}", work={"bytes": 4.0 * n * c * groups * (3 if res is not None else 2), "bound": "hbm"})
    return z


def bn_relu_backward(dz, y, aux, gamma, n, c, groups=1, pool=None, red=None):
    """-> (dy, dgamma, dbeta)  (parameter gradients summed over the groups).  `red` [groups][2c] = (sum dr, sum dr*xhat) when the
    conv that produced dz has already accumulated them in its epilogue (ops.conv*_train, stat_mode 2); else reduced here."""
    if red is None:
        red = torch.zeros(groups * 2 * c, device=y.device, dtype=torch.float64) if pool is None else pool.take(groups * 2 * c)
        _abi("mdf_bn_relu_bwd_reduce", (dz.data_ptr(), y.data_ptr(), aux.data_ptr(), n, c, groups, red.data_ptr(), _stream(y)),
             tag=f"bwd-reduce C{c} N{n}x{groups}", work={"bytes": 8.0 * n * c * groups, "bound": "hbm"})
    dy = torch.empty_like(y)
    dgamma = torch.empty(c, device=y.device, dtype=torch.float32)
    dbeta = torch.empty(c, device=y.device, dtype=torch.float32)
    _abi("mdf_bn_relu_bwd", (dz.data_ptr(), y.data_ptr(), aux.data_ptr(), red.data_ptr(), gamma.data_ptr(), n, c, groups, dy.data_ptr(),
                             dgamma.data_ptr(), dbeta.data_ptr(), red.numel() // (groups * 2 * c), _stream(y)),
         tag=f"bwd C{c} N{n}x{groups}", work={"bytes": 12.0 * n * c * groups, "bound": "hbm"})
    return dy, dgamma, dbeta


# --------------------------------------------------------------------------- conv weight / input gradients
# The partial tiles of a weight gradient are summed into its tensor LATER: nothing reads a weight gradient before the optimizer, so
# the sums of all the layers of a backward pass are ONE launch (mdf_wgrad_sum_batch) issued when the autograd engine has finished
# the pass -- still inside loss.backward(), so every .grad is complete when it returns.  Two conditions, else the sum is issued at
# once: (a) we are inside a backward pass, (b) the parameter's .grad is None, so that AccumulateGrad merely takes the tensor over
# (it would ADD an unfinished gradient into an existing .grad).  The pending list holds the gradient's ADDRESS, not the tensor: a
# second reference would make AccumulateGrad clone it instead of taking it (59 copies per step, and of unfinished data).
DEFER_WGRAD_SUMS = bool(int(_os.environ.get("MDF_WGRAD_DEFER", "1")))      # dev A/B
BATCH_WGRAD = DEFER_WGRAD_SUMS and bool(int(_os.environ.get("MDF_WGRAD_BATCH", "1")))   # dev A/B: the LAUNCHES deferred too and batched (mdf_wgrad_batch_flush)
_PENDING_SUMS = {}          # device index -> [streams the partial tiles were launched on, [(work, dw, nslab, n), ...]]


def sum_wgrad_jobs(jobs):
    """jobs: [(slab tensor, address of the gradient tensor, slabs, elements)] -> the gradients, ONE launch on the current stream."""
    k = len(jobs)
    if k == 0:
        return
    slabs = (ctypes.c_void_p * k)(*[j[0].data_ptr() for j in jobs])
    outs = (ctypes.c_void_p * k)(*[j[1] for j in jobs])
    nsl = (ctypes.c_int * k)(*[j[2] for j in jobs])
    ns = (ctypes.c_int * k)(*[j[3] for j in jobs])
    _abi("mdf_wgrad_sum_batch", (slabs, outs, nsl, ns, k, _stream(jobs[0][0])), tag=f"{k} weight gradients",
         work={"bytes": 4.0 * sum(j[2] * j[3] for j in jobs), "bound": "hbm"})


_HOLD_FLUSH = [False]


class hold_wgrad_flush:
    """Inside: the deferred weight gradients of the backward passes that run are NOT launched when a pass ends; they stay queued
    (operands alive) until `flush_held`.  The recorded training step uses it to take the three stages' weight gradients out of the
    stage chains and launch them beside the pyramid / trunk backward chain (graphstep.py)."""

    def __enter__(self):
        self.prev, _HOLD_FLUSH[0] = _HOLD_FLUSH[0], True

    def __exit__(self, *exc):
        _HOLD_FLUSH[0] = self.prev


def flush_held(device):
    """Launch what `hold_wgrad_flush` kept back, on the current stream.  The caller has ordered that stream behind the backward
    passes whose gradients are queued (no stream joins are issued here: inside a recording the chains' streams are not capturing)."""
    _flush_wgrad_sums(device.index if device.index is not None else torch.cuda.current_device(), force=True, join=False, held=True)


def has_held(device):
    ent = _PENDING_SUMS.get(("held", device.index if device.index is not None else torch.cuda.current_device()))
    return bool(ent and (ent[1] or ent[2]))


def _flush_wgrad_sums(dev_index, force=False, join=True, held=False):
    """held: the queue `hold_wgrad_flush` filled (its own key: a backward pass that runs OUTSIDE the hold while that queue waits --
    the pyramid / trunk piece of the recorded step -- queues and flushes its weight gradients as usual)."""
    if _HOLD_FLUSH[0] and not force:
        return
    ent = _PENDING_SUMS.pop(("held", dev_index) if held else dev_index, None)
    if not ent or not (ent[1] or ent[2]):
        return
    streams, jobs, deferred = ent
    cur = torch.cuda.current_stream(torch.device("cuda", dev_index))
    for st in streams:          # partial tiles may have been launched on other streams: join them
        if st != cur and join:
            cur.wait_stream(st)
    if deferred:
        # the weight gradients whose LAUNCH was deferred too (BATCH_WGRAD): recorded by the library, then launched grouped by
        # kernel instantiation -- the ~45 small layers of a cfg3 step as a handful of launches (mdf_wgrad_batch_flush)
        lib().mdf_wgrad_batch_begin()
        flops = nbytes = 0.0
        st_now = ctypes.c_void_p(cur.cuda_stream)
        try:
            for fn in deferred:
                job, fl, by = fn(st_now)
                jobs.append(job)
                flops += fl
                nbytes += by
        finally:
            _abi("mdf_wgrad_batch_flush", (st_now,), tag=f"{len(deferred)} weight gradients",
                 work={"flops": flops, "bytes": nbytes, "bound": "mfma"})
    sum_wgrad_jobs(jobs)


def drop_stale_wgrad_sums(dev_index=None):
    """A backward pass that raised (out of memory, a user interrupt) never ran the engine's final callback: its pending jobs hold
    addresses of tensors that are gone.  They are dropped -- never flushed -- at the start of the next step (prepack) so that the
    next pass queues a fresh callback and nothing is summed into freed memory."""
    if dev_index is None:
        _PENDING_SUMS.clear()
    else:
        _PENDING_SUMS.pop(dev_index, None)
        _PENDING_SUMS.pop(("held", dev_index), None)


def _can_defer(param):
    """True when the sum of a weight gradient's partial tiles may wait for the end of the backward pass (see _sum_later)."""
    return (DEFER_WGRAD_SUMS and param is not None and param.requires_grad and param.grad is None
            and not getattr(param, "_backward_hooks", None) and not getattr(param, "_post_accumulate_grad_hooks", None)
            and getattr(torch._C, "_current_graph_task_id", lambda: -1)() != -1)


# (r04, measured and dropped: the deferred weight gradients on a SIDE stream that forks from the backward chain per layer and joins
#  before the one sum launch -- eager step 8.5 -> 9.4-9.6 ms, recorded step 8.44 -> 8.70 ms with 6.4 ms of host time per replay.  r05
#  found the cause: a hipGraph with parallel branches is replayed node by node from the host.  The recorded step now keeps every
#  graph a chain and launches the stages' weight gradients as a graph of their own beside the trunk's backward chain: graphstep.py.)


def _sum_later(work, dw, nslab, n, param):
    dev = dw.device.index
    # deferred only when AccumulateGrad is certain to TAKE the returned tensor over: a trainable dense parameter without a
    # gradient yet and without tensor hooks (a frozen weight's dw is dropped and its block re-used within the same backward pass; a
    # hooked or already-populated .grad gets a clone or a sum of the unfinished tensor) -- ADVICE r03
    defer = _can_defer(param)
    ent = _pending_entry(dev, defer)
    cur = torch.cuda.current_stream(dw.device)
    if cur not in ent[0]:
        ent[0].append(cur)
    ent[1].append((work, dw.data_ptr(), nslab, n))
    if not defer:
        _flush_wgrad_sums(dev, force=True)


def _pending_entry(dev, defer):
    key = ("held", dev) if (_HOLD_FLUSH[0] and defer) else dev
    ent = _PENDING_SUMS.get(key)
    if ent is None:
        ent = _PENDING_SUMS[key] = [[], [], []]       # streams, sum jobs, deferred launches
        if defer and key == dev:
            torch.autograd.Variable._execution_engine.queue_callback(lambda d=dev: _flush_wgrad_sums(d))
    return ent


class _DeferredWgrad:
    """A weight-gradient launch put off to the end of the backward pass.  It keeps its OPERANDS and the workspace alive until then, and
    only the ADDRESS of the gradient tensor: a second reference to dw would make AccumulateGrad clone the (still empty) tensor
    instead of taking it over -- the rule of the deferred sums above."""

    def __init__(self, entry, args_of, small, big, dw, work, flops):
        self.entry, self.args_of, self.small, self.big, self.work, self.flops = entry, args_of, small, big, work, flops
        self.dw_ptr, self.dw_numel = dw.data_ptr(), dw.numel()

    def __call__(self, st):
        """st: the stream of the FLUSH (the backward pass's caller stream, which has joined every stream a stage's chain ran on) --
        not the stream that was current when the launch was put off."""
        nslab = ctypes.c_int(0)
        check(getattr(lib(), self.entry)(*self.args_of(nslab, st)), self.entry)
        return (self.work, self.dw_ptr, nslab.value, self.dw_numel), self.flops, 4.0 * (self.small.numel() + self.big.numel())


def conv3d_wgrad(small, big, stride, out_shape, param=None):
    """dw[a][b][27] = sum_o small[o][a] * big[stride*o + tap - 1][b]; small [B,Ds,Hs,Ws,A], big [B,s*Ds,..,Bc] NDHWC."""
    _need_gpu(small, big)
    b, ds, hs, ws, a = small.shape
    bc = big.shape[-1]
    assert tuple(big.shape[:4]) == (b, ds * stride, hs * stride, ws * stride), (small.shape, big.shape, stride)
    assert small.is_contiguous() and big.is_contiguous()
    n = lib().mdf_conv3d_wgrad_workspace(b, ds, hs, ws, a, bc)
    work = torch.empty(n, device=small.device, dtype=torch.float32)
    dw = torch.empty(out_shape, device=small.device, dtype=torch.float32)
    assert dw.numel() == a * bc * 27
    if BATCH_WGRAD and _can_defer(param):
        ent = _pending_entry(dw.device.index, True)
        cur = torch.cuda.current_stream(dw.device)
        if cur not in ent[0]:
            ent[0].append(cur)
        dwp = dw.data_ptr()         # (the closure must not capture dw itself)
        ent[2].append(_DeferredWgrad("mdf_conv3d_wgrad_partial", lambda ns, st: (small.data_ptr(), big.data_ptr(), dwp, work.data_ptr(), b, ds, hs, ws,
                                                                                 a, bc, stride, ctypes.byref(ns), st),
                                     small, big, dw, work, 2.0 * 27 * a * bc * b * ds * hs * ws))
        return dw
    nslab = ctypes.c_int(0)
    _abi("mdf_conv3d_wgrad_partial", (small.data_ptr(), big.data_ptr(), dw.data_ptr(), work.data_ptr(), b, ds, hs, ws, a, bc, stride,
                                      ctypes.byref(nslab), _stream(dw)), tag=f"wgrad {a}x{bc} s{stride} {ds}x{hs}x{ws}",
         work={"flops": 2.0 * 27 * a * bc * b * ds * hs * ws, "bytes": 4.0 * (small.numel() + big.numel()), "bound": "mfma"})
    _sum_later(work, dw, nslab.value, dw.numel(), param)
    return dw


def dgrad_pack(conv, transposed):
    """Packed weights of the layer's INPUT-gradient conv (cached per layer, rebuilt when the weight changes):
    stride-1 conv -> stride-1 conv with flipped taps and swapped channels; stride-2 conv -> the transposed conv with the
    same weight tensor read as [Cin'=Cout, Cout'=Cin]; transposed conv -> stride-2 conv likewise."""
    from .layers import cache_of_key
    w = conv.weight

    def build():
        wd = w.detach()
        if transposed:                       # ConvTranspose3d [Cin,Cout,k]: dgrad = Conv3d(stride 2) with weight [out=Cin,in=Cout]
            return ops.pack_conv3d_weight(wd, transposed=False)
        if conv.stride[0] == 2:              # Conv3d s2 [Cout,Cin,k]: dgrad = ConvTranspose3d with weight [in=Cout,out=Cin]
            return ops.pack_conv3d_weight(wd, transposed=True)
        return ops.pack_conv3d_weight(wd.flip(2, 3, 4).transpose(0, 1).contiguous(), transposed=False)
    return cache_of_key(conv, "dgrad").get((w,), build)


def conv3d_dgrad(conv, transposed, dy, add_to=None, stat=None):
    """dx = [add_to +] (input gradient of the layer) as one launch of the forward conv kernel family.  stat = (y_p, aux_p, red):
    dx is the complete dz of the layer that produced this layer's input (raw conv output y_p, BatchNorm constants aux_p) --
    its BatchNorm-backward sums are accumulated into `red` by the launch's epilogue."""
    wp = dgrad_pack(conv, transposed)
    if transposed:        # backward of ConvTranspose3d(Cin->Cout): stride-2 conv Cout -> Cin
        stride, tr = 2, False
    elif conv.stride[0] == 2:
        stride, tr = 2, True
    else:
        stride, tr = 1, False
    if stat is not None:
        return ops.conv3d_train(dy, wp, conv.out_channels, conv.in_channels, stride, tr, add_to, 2, stat[2], stat[0], stat[1])
    return ops.conv3d_ndhwc(dy, wp, conv.out_channels, conv.in_channels, stride, tr, None, None, False, add_to)


# --------------------------------------------------------------------------- prob head backward
def prob_head_backward(prob, hypos, ddepth, dprob, x_feat, weight, stat=None):
    """-> (dx_feat [B,D,h,w,C], dweight [1,C,3,3,3]).  stat = (y_p, aux_p, red): dx is the complete dz of the layer that produced x_feat
    (raw output y_p, BatchNorm constants aux_p); its backward sums are accumulated into red [STAT_SLICES][2C] on the way."""
    b, d, h, w = prob.shape
    c = x_feat.shape[-1]
    hyp, pp = (None, 0) if hypos is None else _hypos_arg(hypos, h, w)
    dlogit = torch.empty_like(prob)
    _abi("mdf_prob_softmax_regress_bwd", (prob.data_ptr(), None if hyp is None else hyp.data_ptr(), pp,
                                          None if ddepth is None else _f32c(ddepth).data_ptr(),
                                          None if dprob is None else _f32c(dprob).data_ptr(), dlogit.data_ptr(), b, d, h, w,
                                          _stream(prob)), tag=f"softmax-bwd {d}x{h}x{w}",
         work={"bytes": 4.0 * prob.numel() * (3 if dprob is not None else 2), "bound": "hbm"})
    dx = torch.empty_like(x_feat)
    if stat is not None:
        _abi("mdf_prob_conv_dgrad_stat", (dlogit.data_ptr(), _f32c(weight.detach()).data_ptr(), dx.data_ptr(), b, d, h, w, c, stat[0].data_ptr(),
                                          stat[1].data_ptr(), stat[2].data_ptr(), stat[2].numel() // (2 * c), _stream(dx)),
             tag=f"1->{c} dgrad {d}x{h}x{w} +sums2", work={"bytes": 4.0 * (dlogit.numel() + 2 * dx.numel()), "bound": "hbm"})
    else:
        _abi("mdf_prob_conv_dgrad", (dlogit.data_ptr(), _f32c(weight.detach()).data_ptr(), dx.data_ptr(), b, d, h, w, c, _stream(dx)),
             tag=f"1->{c} dgrad {d}x{h}x{w}", work={"bytes": 4.0 * (dlogit.numel() + dx.numel()), "bound": "hbm"})
    dw = conv3d_wgrad(dlogit.view(b, d, h, w, 1), x_feat, 1, tuple(weight.shape), weight if isinstance(weight, torch.nn.Parameter) else None)
    return dx, dw


# --------------------------------------------------------------------------- regulariser: layer tape
FUSE_BN_SUMS = bool(int(_os.environ.get("MDF_FUSE_BN_SUMS", "1")))      # dev A/B (tests, scripts): False = statistics / backward sums as separate passes (mdf_bn_stats_fwd, mdf_bn_relu_bwd_reduce)


class Tape:
    """Forward record of the regulariser's layer program (net/unit/regular.py: `features`), replayed backwards.
    The per-channel sums of every BatchNorm ride in the epilogue of the conv that produces the tensor they are taken over:
    forward, the layer's own conv (sum y, sum y^2); backward, the input-gradient conv of the NEXT layer, whose output (plus the
    skip gradient in its `res` operand) is this layer's complete dz (sum dr, sum dr*xhat)."""

    def __init__(self):
        self.layers, self.pool = [], None
        self.producer, self.uses = {}, {}      # id(z) -> layer index; id(tensor) -> consumers (conv input or skip) on the tape

    def layer(self, conv, bn, x, res):
        if self.pool is None:
            self.pool = step_pool(x.device)
        tr = isinstance(conv, torch.nn.ConvTranspose3d)
        stride = conv.stride[0]
        wp = ops_pack_fwd(conv, tr)
        c = conv.out_channels
        if FUSE_BN_SUMS:
            sums = self.pool.take(ops.STAT_SLICES * 2 * c)
            y = ops.conv3d_train(x, wp, conv.in_channels, c, stride, tr, None, 1, sums)                             # raw conv + statistics
        else:
            y = ops.conv3d_ndhwc(x, wp, conv.in_channels, c, stride, tr, None, None, False, None)                   # raw conv
            sums = bn_stats(y, y.numel() // c, c, pool=self.pool)
        n = y.numel() // c
        z, aux = bn_finalize_apply(sums, bn, y, res, n, c)
        self.producer[id(z)] = len(self.layers)
        for t in (x, res):
            if t is not None:
                self.uses[id(t)] = self.uses.get(id(t), 0) + 1
        self.layers.append((conv, bn, tr, stride, x, y, aux, res, z))
        return z

    def backward(self, grads, red_of=None):
        """grads: {id(tensor): gradient} holding the gradient of the last layer's output; returns parameter grads
        {param: grad} and leaves the input gradients in `grads`.  red_of: {layer index: BatchNorm-backward sums already taken by the
        kernel that produced that layer's dz} (the prob head's input-gradient launch for the last layer)."""
        pg = {}
        pool = step_pool(self.layers[0][4].device)
        pending, red_of = dict(self.uses), dict(red_of or {})
        for li in reversed(range(len(self.layers))):
            conv, bn, tr, stride, x, y, aux, res, z = self.layers[li]
            dz = grads.pop(id(z))
            if res is not None:
                grads[id(res)] = dz if id(res) not in grads else grads[id(res)] + dz
                pending[id(res)] -= 1
            c = conv.out_channels
            n = y.numel() // c
            dy, pg[bn.weight], pg[bn.bias] = bn_relu_backward(dz, y, aux, bn.weight, n, c, pool=pool, red=red_of.pop(li, None))
            if tr:      # ConvTranspose3d: small = x (input), big = dy (twice the size)
                pg[conv.weight] = conv3d_wgrad(x, dy, 2, tuple(conv.weight.shape), conv.weight)
            else:
                pg[conv.weight] = conv3d_wgrad(dy, x, stride, tuple(conv.weight.shape), conv.weight)
            pending[id(x)] -= 1
            prod = self.producer.get(id(x))
            stat = None
            if FUSE_BN_SUMS and prod is not None and pending[id(x)] == 0:
                # every other consumer of x (skip connections: later layers) has been walked: this launch's output is the COMPLETE
                # dz of layer `prod`
                pl = self.layers[prod]
                stat = (pl[5], pl[6], pool.take(ops.STAT_SLICES * 2 * pl[0].out_channels))
                red_of[prod] = stat[2]
            grads[id(x)] = conv3d_dgrad(conv, tr, dy, add_to=grads.get(id(x)), stat=stat)
        return pg


def ops_pack_fwd(conv, tr):
    from .layers import cache_of_key
    return cache_of_key(conv, "fwd").get((conv.weight,), lambda: ops.pack_conv3d_weight(conv.weight, tr))


class RegulariserTrainFn(torch.autograd.Function):
    """(cost [B,C,D,H,W], hypos) -> (prob [B,D,H,W], depth [B,H,W]) in training mode, on the HIP kernels."""

    @staticmethod
    def forward(ctx, module, hypos, cost, *params):
        tape = Tape()
        x0 = ops.to_ndhwc(cost.detach())
        feat = module.features(x0, tape=tape)
        from .layers import cache_of_key
        wpack = cache_of_key(module.prob, "probpack").get((module.prob.weight,), lambda: ops.pack_prob_weight(module.prob.weight))
        prob, depth = ops.prob_head(feat, module.prob.weight, hypos.detach(), wpack=wpack)
        ctx.module, ctx.tape, ctx.x0, ctx.feat, ctx.params = module, tape, x0, feat, params
        ctx.hypos = hypos.detach()
        ctx.save_for_backward(prob)
        ctx.set_materialize_grads(False)
        return prob, depth

    @staticmethod
    def backward(ctx, dprob, ddepth):
        (prob,) = ctx.saved_tensors
        module, tape = ctx.module, ctx.tape
        if dprob is None and ddepth is None:
            return (None,) * (3 + len(ctx.params))
        stat, red_of = None, None
        last = len(tape.layers) - 1
        if FUSE_BN_SUMS and last >= 0 and tape.layers[last][8] is ctx.feat and tape.uses.get(id(ctx.feat), 0) == 0 and ctx.feat.shape[-1] in (8, 16):
            # the prob head is the only consumer of the last layer's output: its input-gradient launch takes that layer's backward sums
            conv_l, _, _, _, _, y_l, aux_l, _, _ = tape.layers[last]
            red = step_pool(ctx.feat.device).take(ops.STAT_SLICES * 2 * conv_l.out_channels)
            stat, red_of = (y_l, aux_l, red), {last: red}
        dfeat, dwp = prob_head_backward(prob, ctx.hypos, ddepth, dprob, ctx.feat, module.prob.weight, stat=stat)
        grads = {id(ctx.feat): dfeat}
        pg = tape.backward(grads, red_of=red_of)
        pg[module.prob.weight] = dwp
        dcost = ops.from_ndhwc(grads.pop(id(ctx.x0)))
        ctx.tape = None
        return (None, None, dcost) + tuple(pg.get(p) for p in ctx.params)


def regulariser_train(module, cost, hypos):
    params = tuple(module.parameters())
    if hypos is None:       # Regular[s](cost) alone: prob only
        zeros = torch.zeros((cost.shape[0], cost.shape[2], 1, 1), device=cost.device, dtype=torch.float32)
        return RegulariserTrainFn.apply(module, zeros, cost, *params)[0]
    return RegulariserTrainFn.apply(module, hypos, cost, *params)


# --------------------------------------------------------------------------- VectorAggregate in training mode
_PASS_STATS, _PASS_FWD, _PASS_BWD_REDUCE, _PASS_BWD = 0, 1, 2, 3


def _agg_call(pass_, ref, srcs, proj, hyp, pp, par, red_in, dcost, cost, wsum, red_out, dref, dsrcs, dcw, b, c, g, d, h, w, aux=None):
    def ptr(t):
        return None if t is None else t.data_ptr()
    algo = 4.0 * b * ((len(srcs) + 1) * c * h * w + g * d * h * w)
    _abi("mdf_warp_aggregate_vec_train", (pass_, ref.data_ptr(), _src_array(srcs), proj.data_ptr(), hyp.data_ptr(), pp, par.data_ptr(),
                                          ptr(red_in), ptr(dcost), ptr(cost), ptr(wsum), ptr(red_out), ptr(dref),
                                          None if dsrcs is None else _src_array(dsrcs), ptr(dcw), ptr(aux), b, c, g, d, h, w, len(srcs),
                                          _stream(ref)), tag=f"train pass{pass_} C{c}G{g}D{d} {w}x{h} V{len(srcs) + 1}",
         work={"bytes": algo * (2 if pass_ == _PASS_BWD else 1), "bound": "hbm"})


class AggregateTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, proj, hypos, cw, gamma, beta, w2, b2, nviews, *features):
        """features: one [B,C,h,w] tensor per view -- or (nviews > 0) ONE view-major tensor [nviews*B,C,h,w] holding them all
        (FPN_4Scales.forward_views), whose gradient is then returned as one tensor too: no slice / zero-fill / add launches
        of the autograd engine around the slot."""
        bn = module.depth_weight[0].bn
        ctx.batched = nviews > 0
        if ctx.batched:
            allv = ops.nhwc(features[0].detach()).permute(0, 2, 3, 1).contiguous()              # [V*B,h,w,C] memory
            bb = allv.shape[0] // nviews
            feas = [allv[v * bb:(v + 1) * bb] for v in range(nviews)]
        else:
            feas = [ops.nhwc(f.detach()).permute(0, 2, 3, 1).contiguous() for f in features]  # [B,h,w,C] memory
        b, h, w, c = feas[0].shape
        g = module.ngroups
        d = hypos.shape[1]
        hyp, pp = _hypos_arg(hypos.detach(), h, w)
        nsrc = len(feas) - 1
        dev = feas[0].device
        n = b * d * h * w
        st = _stream(feas[0])
        # the scalars between the passes stay on the device (prepare / finalize kernels): no host arithmetic, no tiny ATen launches
        par = torch.empty(g + 4 + 4 * nsrc, device=dev, dtype=torch.float32)
        red = torch.empty(2 * nsrc, device=dev, dtype=torch.float64)
        cwd, gmd, btd, w2d, b2d = (_f32c(t.detach()) for t in (cw, gamma, beta, w2, b2))
        _abi("mdf_aggregate_train_prepare", (cwd.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), gmd.data_ptr(), n, g, nsrc, par.data_ptr(),
                                             red.data_ptr(), 2 * nsrc, st))
        _agg_call(_PASS_STATS, feas[0], feas[1:], proj, hyp, pp, par, None, None, None, None, red, None, None, None, b, c, g, d, h, w)
        track = bn.track_running_stats and bn.running_mean is not None
        mom = _momentum(bn)
        _abi("mdf_aggregate_train_finalize", (red.data_ptr(), gmd.data_ptr(), btd.data_ptr(), ctypes.c_float(bn.eps), ctypes.c_float(mom), n, g,
                                              nsrc, par.data_ptr(), bn.running_mean.data_ptr() if track else None,
                                              bn.running_var.data_ptr() if track else None,
                                              bn.num_batches_tracked.data_ptr() if track else None, st))
        cost = torch.empty((b, d, h, w, g), device=dev, dtype=torch.float32)
        wsum = torch.empty((b, d, h, w), device=dev, dtype=torch.float32)
        _agg_call(_PASS_FWD, feas[0], feas[1:], proj, hyp, pp, par, None, None, cost, wsum, None, None, None, None, b, c, g, d, h, w)
        ctx.feas, ctx.proj, ctx.hyp, ctx.pp, ctx.par, ctx.dims = feas, proj, hyp, pp, par, (b, c, g, d, h, w)
        ctx.cost, ctx.wsum = cost, wsum
        ctx.wshapes = (cw.shape, gamma.shape, beta.shape, w2.shape, b2.shape)
        return cost.permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, dcost):
        b, c, g, d, h, w = ctx.dims
        feas, nsrc = ctx.feas, len(ctx.feas) - 1
        dev = feas[0].device
        dc = ops.to_ndhwc(dcost)
        nhalf = b * h * w * g
        # one zero fill for everything the two passes accumulate into: fp64 reductions | even-channel gradients per view | d conv weight
        nred = 2 * nsrc + 2
        nref = b * h * w * c
        zero = torch.zeros(8 * nred + 4 * (nsrc * nhalf + nref + g), device=dev, dtype=torch.uint8)
        red = zero[:8 * nred].view(torch.float64)
        acc = zero[8 * nred:].view(torch.float32)
        dhalf = [acc[v * nhalf:(v + 1) * nhalf] for v in range(nsrc)]
        dref_acc = acc[nsrc * nhalf:nsrc * nhalf + nref]            # the depth slices of a pixel add their d ref here
        dcw = acc[nsrc * nhalf + nref:]
        aux = torch.empty((nsrc, b * d * h * w, 4), device=dev, dtype=torch.float32)     # per sample (w_v, dz_v, t_v, -): pass 2 -> pass 3
        _agg_call(_PASS_BWD_REDUCE, feas[0], feas[1:], ctx.proj, ctx.hyp, ctx.pp, ctx.par, None, dc, ctx.cost, ctx.wsum, red, None, None,
                  None, b, c, g, d, h, w, aux=aux)
        dall = torch.empty((nsrc + 1, b, h, w, c), device=dev, dtype=torch.float32)                # reference view first
        dref, dfull = dall[0], dall[1:]
        _agg_call(_PASS_BWD, feas[0], feas[1:], ctx.proj, ctx.hyp, ctx.pp, ctx.par, red, dc, ctx.cost, ctx.wsum, None, dref_acc, dhalf,
                  dcw, b, c, g, d, h, w, aux=aux)
        dpar = torch.empty(4, device=dev, dtype=torch.float32)
        _abi("mdf_aggregate_train_bwd_finalize", (acc.data_ptr(), red.data_ptr(), nsrc, nsrc * nhalf, dfull.data_ptr(), dpar.data_ptr(),
                                                  dref_acc.data_ptr(), dref.data_ptr(), nref, _stream(dfull)),
             tag=f"pairs {nsrc}x{h}x{w}x{c}", work={"bytes": 12.0 * nsrc * nhalf + 8.0 * nref, "bound": "hbm"})
        s_cw, s_gamma, s_beta, s_w2, s_b2 = ctx.wshapes
        if ctx.batched:
            dfeas = [dall.view((nsrc + 1) * b, h, w, c).permute(0, 3, 1, 2)]
        else:
            dfeas = [dall[v].permute(0, 3, 1, 2) for v in range(nsrc + 1)]
        ctx.cost = ctx.wsum = None
        return (None, None, None, dcw.reshape(s_cw), dpar[0].reshape(s_gamma), dpar[1].reshape(s_beta), dpar[2].reshape(s_w2),
                dpar[3].reshape(s_b2), None) + tuple(dfeas)


def aggregate_train(module, features, proj, hypos):
    head = module.depth_weight
    args = (module, proj, hypos, head[0].conv.weight, head[0].bn.weight, head[0].bn.bias, head[1].weight, head[1].bias)
    parents = [getattr(f, "_mdf_parent", None) for f in features]
    if all(p is not None and p[0] is parents[0][0] and p[1] == i and p[2] == len(features) for i, p in enumerate(parents)):
        return AggregateTrainFn.apply(*args, len(features), parents[0][0])      # the views are the slices of one view-major tensor
    return AggregateTrainFn.apply(*args, 0, *features)


# --------------------------------------------------------------------------- feature-pyramid trunk (2-D) in training mode
def conv2d_wgrad(small, big, ksize, stride, out_shape, param=None, hold=None):
    """dw[a][b][kh][kw] = sum_o small[o][a] * big[stride*o + (kh,kw) - pad][b]; small [B,Hs,Ws,A], big [B,s*Hs,s*Ws,Bc] NHWC.
    hold: a list -- the partial tiles are left un-summed and their job is appended to it; the caller sums several gradients it needs
    at once with ONE sum_wgrad_jobs(hold) (and must do so before reading any of them)."""
    _need_gpu(small, big)
    b, hs, ws, a = small.shape
    bc = big.shape[-1]
    assert tuple(big.shape[:3]) == (b, hs * stride, ws * stride), (small.shape, big.shape, stride)
    assert small.is_contiguous() and big.is_contiguous()
    n = lib().mdf_conv2d_wgrad_workspace(b, hs, ws, a, bc, ksize)
    work = torch.empty(n, device=small.device, dtype=torch.float32)
    dw = torch.empty((a, bc, ksize, ksize), device=small.device, dtype=torch.float32)
    if BATCH_WGRAD and hold is None and tuple(out_shape) == tuple(dw.shape) and _can_defer(param):
        ent = _pending_entry(dw.device.index, True)
        cur = torch.cuda.current_stream(dw.device)
        if cur not in ent[0]:
            ent[0].append(cur)
        dwp = dw.data_ptr()         # (the closure must not capture dw itself)
        ent[2].append(_DeferredWgrad("mdf_conv2d_wgrad_partial", lambda ns, st: (small.data_ptr(), big.data_ptr(), dwp, work.data_ptr(), b, hs, ws, a, bc,
                                                                                 ksize, stride, ctypes.byref(ns), st),
                                     small, big, dw, work, 2.0 * ksize * ksize * a * bc * b * hs * ws))
        return dw
    nslab = ctypes.c_int(0)
    _abi("mdf_conv2d_wgrad_partial", (small.data_ptr(), big.data_ptr(), dw.data_ptr(), work.data_ptr(), b, hs, ws, a, bc, ksize, stride,
                                      ctypes.byref(nslab), _stream(dw)), tag=f"wgrad2d {a}x{bc} k{ksize}s{stride} {hs}x{ws}x{b}",
         work={"flops": 2.0 * ksize * ksize * a * bc * b * hs * ws, "bytes": 4.0 * (small.numel() + big.numel()), "bound": "mfma"})
    if hold is not None:
        assert tuple(out_shape) == tuple(dw.shape)
        hold.append((work, dw.data_ptr(), nslab.value, dw.numel()))
        return dw
    _sum_later(work, dw, nslab.value, dw.numel(), param if tuple(out_shape) == tuple(dw.shape) else None)
    if tuple(out_shape) != tuple(dw.shape):
        pass     # (the image layer's weight gradient is computed over 4 padded input channels: summed at once, sliced below)
    return dw if tuple(out_shape) == tuple(dw.shape) else dw[:, :out_shape[1]].contiguous()


_K5_TAP = ((4, 2, 0), (-1, 3, 1))      # output parity p, 3x3 tap t -> 5x5 kernel index (-1: structurally zero)
_K5_IDX = {}


def _k5s2_dgrad_weight(w):
    """Conv2d(k5,s2,p2) weight [Cout,Cin,5,5] -> weight [4*Cin, Cout, 3, 3] of the stride-1 3x3 conv over dy whose output
    channel (py*2+px)*Cin + ci is dx[ci] at the pixels of parity (py,px): dx[2j+p] = sum_t dy[j+t-1] * w[k(p,t)]."""
    cout, cin = w.shape[:2]
    wt = torch.nn.functional.pad(w.detach().float().permute(1, 0, 2, 3), (0, 1, 0, 1))          # [Cin,Cout,6,6], index 5 = zero
    idx = _K5_IDX.get(w.device)                                                                  # (p,t) -> kernel index
    if idx is None:     # built once per device: torch.tensor(list, device=...) is a blocking copy (0.9 ms behind a busy GPU)
        idx = _K5_IDX[w.device] = torch.tensor([k if k >= 0 else 5 for p_ in _K5_TAP for k in p_], device=w.device)
    g = wt.index_select(2, idx).index_select(3, idx).reshape(cin, cout, 2, 3, 2, 3)              # [ci,co,py,ty,px,tx]
    return g.permute(2, 4, 0, 1, 3, 5).reshape(4 * cin, cout, 3, 3).contiguous()


def _dgrad2d_pack(conv):
    from .layers import cache_of_key
    k, stride, cin = conv.kernel_size[0], conv.stride[0], conv.in_channels
    w = conv.weight
    if k == 3 and stride == 1:
        return cache_of_key(conv, "dgrad").get((w,), lambda: ops.pack_conv2d_weight(w.detach().flip(2, 3).transpose(0, 1).contiguous()))
    if k == 5 and stride == 2:
        parts = _k5s2_parts(cin)

        def build():
            w4 = _k5s2_dgrad_weight(w)
            return [(ops.pack_conv2d_weight(w4[a:b].contiguous()), b - a) for a, b in parts]
        return cache_of_key(conv, "dgrad").get((w,), build)
    raise NotImplementedError(f"conv2d input gradient for k={k} stride={stride}")


def conv2d_dgrad(conv, dy, stat=None, groups=1):
    """Input gradient of a Conv2d(k3,s1,p1) or Conv2d(k5,s2,p2) layer, NHWC, on the forward conv kernels.  stat = (y_p, aux_p, red)
    (k3 only): the BatchNorm-backward sums of the producing layer ride in the launch's epilogue, as in conv3d_dgrad."""
    k, cin, cout = conv.kernel_size[0], conv.in_channels, conv.out_channels
    packs = _dgrad2d_pack(conv)
    if k == 3:
        if stat is not None:
            return ops.conv2d_train(dy, packs, cout, cin, 3, 1, False, 2, stat[2], groups, stat[0], stat[1])
        return ops.conv2d_nhwc(dy, packs, cout, cin, 3, 1)
    if len(packs) == 1 and cin in (8, 16):
        # the four parity classes are the rows of ONE launch: the conv's store interleaves them (pixel (2h + py, 2w + px), the
        # PixelShuffle(2) store of the refinement net's 8 -> 32 conv) -- no [.,h,w,4*cin] tensor and no copy of the full-resolution map
        return ops.conv2d_nhwc(dy, packs[0][0], cout, 4 * cin, 3, 1, pixel_shuffle2=True)
    zs = [ops.conv2d_nhwc(dy, wp, cout, n, 3, 1) for wp, n in packs]
    z = zs[0] if len(zs) == 1 else torch.cat(zs, dim=-1)
    b_, ho, wo, _ = z.shape
    return z.view(b_, ho, wo, 2, 2, cin).permute(0, 1, 3, 2, 4, 5).reshape(b_, 2 * ho, 2 * wo, cin)      # the 4 parity classes interleaved


class Tape2D:
    """Forward record of the feature pyramid's Conv2d + BatchNorm2d(batch statistics) + ReLU chain (backbone.py:17-45),
    `groups` independent module calls batched along the image axis (one group per view)."""

    def __init__(self, groups):
        self.groups, self.layers, self.pool = groups, [], None
        self.producer = {}

    def layer(self, conv, bn, x, planar_in=False, x_for_wgrad=None):
        k, stride = conv.kernel_size[0], conv.stride[0]
        from .layers import cache_of_key
        if self.pool is None:
            self.pool = step_pool(x.device)
        wp = cache_of_key(conv, "fwd").get((conv.weight,), lambda: ops.pack_conv2d_weight(conv.weight))
        c = conv.out_channels
        if FUSE_BN_SUMS:
            sums = self.pool.take(ops.STAT_SLICES * self.groups * 2 * c)
            y = ops.conv2d_train(x, wp, conv.in_channels, c, k, stride, planar_in, 1, sums, self.groups)                 # raw conv + statistics
        else:
            y = ops.conv2d_nhwc(x, wp, conv.in_channels, c, k, stride, planar_in=planar_in)                            # raw conv
            sums = bn_stats(y, y.numel() // c // self.groups, c, self.groups, pool=self.pool)
        n = y.numel() // c // self.groups
        z, aux = bn_finalize_apply(sums, bn, y, None, n, c, self.groups)
        self.producer[id(z)] = len(self.layers)
        self.layers.append((conv, bn, x if x_for_wgrad is None else x_for_wgrad, y, aux, z, x_for_wgrad is not None))
        return z

    def backward(self, grads):
        pg = {}
        pool = step_pool(self.layers[0][3].device)
        red_of = {}
        for li in reversed(range(len(self.layers))):
            conv, bn, x, y, aux, z, is_input = self.layers[li]
            dz = grads.pop(id(z))
            c = conv.out_channels
            n = y.numel() // c // self.groups
            dy, pg[bn.weight], pg[bn.bias] = bn_relu_backward(dz.contiguous(), y, aux, bn.weight, n, c, self.groups, pool=pool,
                                                               red=red_of.pop(li, None))
            pg[conv.weight] = conv2d_wgrad(dy, x, conv.kernel_size[0], conv.stride[0], tuple(conv.weight.shape), conv.weight)
            if not is_input:
                prod = self.producer.get(id(x))
                stat = None
                if FUSE_BN_SUMS and prod is not None and id(x) not in grads and conv.kernel_size[0] == 3 and conv.stride[0] == 1:
                    # x feeds this layer only (the pyramid outputs t2..t4 also receive a gradient from the heads and are consumed by
                    # k5-s2 layers): the launch's output is the complete dz of layer `prod`
                    pl = self.layers[prod]
                    stat = (pl[3], pl[4], pool.take(ops.STAT_SLICES * self.groups * 2 * pl[0].out_channels))
                    red_of[prod] = stat[2]
                dx = conv2d_dgrad(conv, dy, stat=stat, groups=self.groups)
                grads[id(x)] = dx if id(x) not in grads else grads[id(x)] + dx
        return pg


class TrunkTrainFn(torch.autograd.Function):
    """(images [G*B,3,H,W], G groups) -> (t2 [.,16,H/2,W/2], t3 [.,32,H/4,W/4], t4 [.,64,H/8,W/8]) of FPN_4Scales in training mode."""

    @staticmethod
    def forward(ctx, module, groups, imgs, *params):
        tape = Tape2D(groups)
        x = imgs.detach().float().contiguous()                              # planar, as the loader hands it over
        x4 = torch.zeros((x.shape[0], x.shape[2], x.shape[3], 4), device=x.device, dtype=torch.float32)
        x4[..., :3] = x.permute(0, 2, 3, 1)                                 # NHWC (padded to 4) for the first layer's weight gradient
        first = module.conv01[0]
        t = tape.layer(first.conv, first.bn, x, planar_in=True, x_for_wgrad=x4)
        outs = []
        for seq in (module.conv01[1:], module.conv12, module.conv23, module.conv34):
            for blk in seq:
                t = tape.layer(blk.conv, blk.bn, t)
            outs.append(t)
        ctx.tape, ctx.params, ctx.outs = tape, params, outs[1:]
        ctx.set_materialize_grads(False)
        return tuple(o.permute(0, 3, 1, 2) for o in outs[1:])

    @staticmethod
    def backward(ctx, *douts):
        grads = {}
        for o, g in zip(ctx.outs, douts):
            if g is not None:
                grads[id(o)] = g.permute(0, 2, 3, 1).contiguous()
        for o in ctx.outs:
            if id(o) not in grads:
                grads[id(o)] = torch.zeros_like(o)
        pg = ctx.tape.backward(grads)
        ctx.tape = None
        return (None, None, None) + tuple(pg.get(p) for p in ctx.params)


def trunk_train(module, imgs, groups):
    params = tuple(p for seq in (module.conv01, module.conv12, module.conv23, module.conv34) for p in seq.parameters())
    return TrunkTrainFn.apply(module, groups, imgs, *params)


# --------------------------------------------------------------------------- weight packing: one launch per step
_SRC_DIRECT, _SRC_SWAPFLIP, _SRC_K5S2, _SRC_PROB, _SRC_SHUFFLE2, _SRC_SWAP = range(6)


class PackPlan:
    """Every packed weight set the training kernels read (forward convs and their input-gradient convs: ~100 sets), re-packed
    after each optimizer update by ONE launch of mdf_pack_batch instead of ~250 flip / transpose / pad / pack launches.  The
    job table is built once per model (parameter storage does not move during training: the optimizer and load_state_dict
    write in place); `run` re-packs when a parameter's version moved and hands the buffers to the per-layer caches
    (layers._Folded) the layer code looks them up in."""

    def __init__(self, model):
        import torch.nn as nn
        from .layers import cache_of_key
        L = lib()
        self.jobs, self.entries = [], []          # jobs: ctypes arguments; entries: (cache, parameter, value)
        self.fpn = None                           # composed matrices of the feature pyramid's heads (see below)
        self.device = next(model.parameters()).device

        def buf(is3d, cin, cout, ntaps):
            n = L.mdf_conv3d_packed_size(cin, cout) if is3d else L.mdf_conv_packed_size(cin, cout, ntaps)
            return torch.empty(n, device=self.device, dtype=torch.float32)

        def job(param, is3d, tr, mode, cin, cout, ntaps, a0=0, a1=0):
            dst = buf(is3d, cin, cout, ntaps)
            self.jobs.append((param, dst, int(is3d), int(tr), mode, cin, cout, ntaps, a0, a1))
            return dst

        def entry(mod, name, param, value):
            self.entries.append((cache_of_key(mod, name), param, value))

        for reg in model.Regular:
            for m in reg.modules():
                w = getattr(m, "weight", None)
                if isinstance(m, nn.ConvTranspose3d):
                    entry(m, "fwd", w, job(w, 1, 1, _SRC_DIRECT, m.in_channels, m.out_channels, 27))
                    entry(m, "dgrad", w, job(w, 1, 0, _SRC_DIRECT, m.out_channels, m.in_channels, 27))     # stride-2 conv [out=Cin][in=Cout]
                elif isinstance(m, nn.Conv3d) and tuple(m.kernel_size) == (3, 3, 3):
                    if m.out_channels == 1:
                        entry(m, "probpack", w, job(w, 0, 0, _SRC_PROB, m.in_channels, 4, 9))
                        continue
                    entry(m, "fwd", w, job(w, 1, 0, _SRC_DIRECT, m.in_channels, m.out_channels, 27))
                    if m.stride[0] == 2:
                        entry(m, "dgrad", w, job(w, 1, 1, _SRC_DIRECT, m.out_channels, m.in_channels, 27))  # transposed conv [in=Cout][out=Cin]
                    else:
                        entry(m, "dgrad", w, job(w, 1, 0, _SRC_SWAPFLIP, m.out_channels, m.in_channels, 27))
        bb = model.Backbone
        if hasattr(bb, "conv34"):
            first = True
            for seq in (bb.conv01, bb.conv12, bb.conv23, bb.conv34):
                for blk in seq:
                    c = blk.conv
                    k, w = c.kernel_size[0], c.weight
                    entry(c, "fwd", w, job(w, 0, 0, _SRC_DIRECT, c.in_channels, c.out_channels, k * k))
                    if not first:
                        if k == 3 and c.stride[0] == 1:
                            entry(c, "dgrad", w, job(w, 0, 0, _SRC_SWAPFLIP, c.out_channels, c.in_channels, 9))
                        else:
                            entry(c, "dgrad", w, [(job(w, 0, 0, _SRC_K5S2, c.out_channels, hi - lo, 9, c.in_channels, lo), hi - lo)
                                                  for lo, hi in _k5s2_parts(c.in_channels)])
                    first = False
            for c in (bb.lat2, bb.lat3, bb.out2, bb.out3, bb.out4):
                entry(c, "fwd", c.weight, job(c.weight, 0, 0, _SRC_DIRECT, c.in_channels, c.out_channels, 1))
                entry(c, "dgrad", c.weight, job(c.weight, 0, 0, _SRC_SWAP, c.out_channels, c.in_channels, 1))
            c2, c3, cm = bb.out2.out_channels, bb.out3.out_channels, bb.out2.in_channels
            if (COMPOSE_FPN_HEADS and (bb.lat2.in_channels, bb.lat3.in_channels) == (c2, c3) and bb.lat2.bias is not None and bb.lat3.bias is not None
                    and bb.out2.bias is None and bb.out3.bias is None
                    and {bb.out3.in_channels, bb.out4.in_channels, bb.lat2.out_channels, bb.lat3.out_channels} == {cm}):
                # the heads through their algebra (FPNHeadsComposedFn): A2 = O2 L2, B3 = O2 L3, A3 = O3 L3 and three bias vectors are
                # formed by ONE launch ahead of the batched pack (mdf_fpn_compose_fwd), which then packs them like parameters
                comp = torch.empty(c2 * c2 + c2 * c3 + c3 * c3 + 2 * c2 + c3, device=self.device, dtype=torch.float32)
                o = [0]

                def piece(*shape):
                    n = 1
                    for d in shape:
                        n *= d
                    v = comp[o[0]:o[0] + n].view(shape)
                    o[0] += n
                    return v
                A2, B3, A3 = piece(c2, c2, 1, 1), piece(c2, c3, 1, 1), piece(c3, c3, 1, 1)
                self.fpn = {"bb": bb, "dims": (c2, c3, cm), "comp": comp, "e2": piece(c2), "e3": piece(c2), "f3": piece(c3),
                            "A2f": job(A2, 0, 0, _SRC_DIRECT, c2, c2, 1), "A2t": job(A2, 0, 0, _SRC_SWAP, c2, c2, 1),
                            "B3f": job(B3, 0, 0, _SRC_DIRECT, c3, c2, 1), "B3t": job(B3, 0, 0, _SRC_SWAP, c2, c3, 1),
                            "A3f": job(A3, 0, 0, _SRC_DIRECT, c3, c3, 1), "A3t": job(A3, 0, 0, _SRC_SWAP, c3, c3, 1),
                            "params": (bb.out2.weight, bb.out3.weight, bb.lat2.weight, bb.lat2.bias, bb.lat3.weight, bb.lat3.bias)}
        rf = getattr(model, "Refine", None)
        if rf is not None and hasattr(rf, "ress"):
            shuffled = rf.conv2[0]
            convs = [rf.conv0] + [blk.conv[i] for blk in rf.ress for i in (0, 2)] + [rf.conv1, rf.conv2[0], rf.conv2[2]]
            for c in convs:
                entry(c, "fwd", c.weight, job(c.weight, 0, 0, _SRC_SHUFFLE2 if c is shuffled else _SRC_DIRECT, c.in_channels, c.out_channels, 9))
                if c is not rf.conv0:
                    entry(c, "dgrad", c.weight, job(c.weight, 0, 0, _SRC_SWAPFLIP, c.out_channels, c.in_channels, 9))
        # job table (host) -> device, with the per-block job index
        nb = int(L.mdf_pack_job_bytes())
        table = (ctypes.c_char * (nb * len(self.jobs)))()
        block_job, first = [], 0
        for i, (param, dst, is3d, tr, mode, cin, cout, ntaps, a0, a1) in enumerate(self.jobs):
            nblk = int(L.mdf_pack_job_fill(ctypes.addressof(table), i, param.data_ptr(), dst.data_ptr(), is3d, tr, mode, cin, cout, ntaps,
                                           a0, a1, first))
            if nblk < 0:
                from . import MdfHipError
                raise MdfHipError(f"mdf_pack_job_fill: {L.mdf_last_error().decode()}")
            block_job += [i] * nblk
            first += nblk
        self.nblocks = first
        self.table = torch.frombuffer(bytearray(table), dtype=torch.uint8).to(self.device)
        self.block_job = torch.tensor(block_job, dtype=torch.int32).to(self.device)
        self.params = []
        seen = set()
        for _, p_, _ in self.entries:
            if id(p_) not in seen:
                seen.add(id(p_))
                self.params.append(p_)
        for p_ in (self.fpn["params"] if self.fpn is not None else ()):      # (the heads' biases have no packed set of their own)
            if id(p_) not in seen:
                seen.add(id(p_))
                self.params.append(p_)
        self.ptrs = tuple(p_.data_ptr() for p_ in self.params)
        self.versions = None

    def valid(self):
        return tuple(p_.data_ptr() for p_ in self.params) == self.ptrs

    def run(self):
        versions = tuple(p_._version for p_ in self.params)
        if versions != self.versions:
            if self.fpn is not None:
                f = self.fpn
                O2, O3, L2, b2, L3, b3 = f["params"]
                c2, c3, cm = f["dims"]
                _abi("mdf_fpn_compose_fwd", (O2.data_ptr(), O3.data_ptr(), L2.data_ptr(), b2.data_ptr(), L3.data_ptr(), b3.data_ptr(), c2, c3, cm,
                                             f["comp"].data_ptr(), _stream(f["comp"])), tag="composed FPN head matrices")
            _abi("mdf_pack_batch", (self.table.data_ptr(), self.block_job.data_ptr(), self.nblocks, _stream(self.table)),
                 tag=f"{len(self.jobs)} weight sets")
            self.versions = versions
        if self.fpn is not None:      # the heads find this step's composed packs on their module (valid while the parameters are these)
            self.fpn["key"] = tuple((p_.data_ptr(), p_._version) for p_ in self.fpn["params"])
            self.fpn["bb"].__dict__["_mdf_fpn"] = self.fpn
        idx = self.device.index
        for cache, p_, value in self.entries:
            cache.key, cache.val, cache.ev, cache.seen = ((p_.data_ptr(), p_._version, idx),), value, None, ()


def _k5s2_parts(cin):
    nout = 4 * cin
    return [(0, nout)] if nout <= 64 else [(0, nout // 2), (nout // 2, nout)]


def prepack(model):
    """Pack the forward and input-gradient weights of every conv layer the training kernels use in this step (the optimizer
    has just changed them).  Called at the top of CoreNet.forward in training mode, on the step's own stream."""
    plan = model.__dict__.get("_mdf_pack_plan")
    if plan is None or not plan.valid():
        plan = model.__dict__["_mdf_pack_plan"] = PackPlan(model)
    plan.run()
    step_pool(plan.device).reset()
    drop_stale_wgrad_sums(plan.device.index)


# --------------------------------------------------------------------------- FPN heads (1x1 convs + top-down adds) in training mode
def upsample2_backward(dfine, dcoarse=None):
    """up^T: [B,2h,2w,C] -> [B,h,w,C]; added into `dcoarse` when given (adjoint of F.interpolate x2 bilinear, backbone.py:60,62)."""
    b, h2, w2, c = dfine.shape
    out = torch.empty((b, h2 // 2, w2 // 2, c), device=dfine.device, dtype=torch.float32) if dcoarse is None else dcoarse
    _abi("mdf_upsample2_bilinear_bwd", (dfine.data_ptr(), out.data_ptr(), b, h2 // 2, w2 // 2, c, int(dcoarse is not None), _stream(out)),
         tag=f"upT {h2}x{w2}x{c}", work={"bytes": 4.0 * (dfine.numel() + out.numel()), "bound": "hbm"})
    return out


def _pack1x1(conv, transposed):
    from .layers import cache_of_key
    w = conv.weight
    if transposed:
        return cache_of_key(conv, "dgrad").get((w,), lambda: ops.pack_conv2d_weight(w.detach().transpose(0, 1).contiguous()))
    return cache_of_key(conv, "fwd").get((w,), lambda: ops.pack_conv2d_weight(w))


def _conv1x1(conv, x, transposed=False, res_up=None):
    cin, cout = (conv.out_channels, conv.in_channels) if transposed else (conv.in_channels, conv.out_channels)
    bias = None if (transposed or conv.bias is None) else conv.bias.detach()
    return ops.conv2d_nhwc(x, _pack1x1(conv, transposed), cin, cout, 1, 1, None, bias, False, None, 1.0, res_up)


class FPNHeadsTrainFn(torch.autograd.Function):
    """(t2, t3, t4) -> (out4(t4), out3(up(t4) + lat3(t3)), out2(up(up3) + lat2(t2)))   (backbone.py:59-63), NHWC kernels:
    the lateral 1x1 conv, its bias and the bilinear top-down add are one launch; backward = 1x1 convs with transposed weights,
    the adjoint upsampling kernel, 1x1 weight gradients on the MFMA wgrad kernel, bias gradients on the BN-statistics kernel."""

    @staticmethod
    def forward(ctx, m, t2, t3, t4, *params):
        t2n, t3n, t4n = ops.to_nhwc(t2.detach()), ops.to_nhwc(t3.detach()), ops.to_nhwc(t4.detach())
        up3 = _conv1x1(m.lat3, t3n, res_up=t4n)
        up2 = _conv1x1(m.lat2, t2n, res_up=up3)
        y4, y3, y2 = _conv1x1(m.out4, t4n), _conv1x1(m.out3, up3), _conv1x1(m.out2, up2)
        ctx.m, ctx.saved, ctx.params = m, (t2n, t3n, t4n, up3, up2), params
        ctx.set_materialize_grads(False)
        return ops.from_nhwc(y4), ops.from_nhwc(y3), ops.from_nhwc(y2)

    @staticmethod
    def backward(ctx, dy4, dy3, dy2):
        m = ctx.m
        t2n, t3n, t4n, up3, up2 = ctx.saved

        def nh(g, like):
            return torch.zeros_like(like) if g is None else ops.to_nhwc(g)
        dy4, dy3, dy2 = nh(dy4, t4n), ops.to_nhwc(dy3) if dy3 is not None else None, ops.to_nhwc(dy2) if dy2 is not None else None
        pg = {}
        pool = step_pool(t2n.device)

        def bias_grad(g):
            c = g.shape[-1]
            return bn_stats(g, g.numel() // c, c, pool=pool)[:c].float()
        d_up3 = None
        if dy2 is not None:
            d_up2 = _conv1x1(m.out2, dy2, transposed=True)
            pg[m.out2.weight] = conv2d_wgrad(dy2, up2, 1, 1, tuple(m.out2.weight.shape), m.out2.weight)
            pg[m.lat2.weight] = conv2d_wgrad(d_up2, t2n, 1, 1, tuple(m.lat2.weight.shape), m.lat2.weight)
            pg[m.lat2.bias] = bias_grad(d_up2)
            dt2 = _conv1x1(m.lat2, d_up2, transposed=True)
            d_up3 = upsample2_backward(d_up2)
        else:
            dt2 = torch.zeros_like(t2n)
        if dy3 is not None:
            g3 = _conv1x1(m.out3, dy3, transposed=True)
            pg[m.out3.weight] = conv2d_wgrad(dy3, up3, 1, 1, tuple(m.out3.weight.shape), m.out3.weight)
            d_up3 = g3 if d_up3 is None else d_up3.add_(g3)
        dt4 = _conv1x1(m.out4, dy4, transposed=True)
        pg[m.out4.weight] = conv2d_wgrad(dy4, t4n, 1, 1, tuple(m.out4.weight.shape), m.out4.weight)
        if d_up3 is not None:
            pg[m.lat3.weight] = conv2d_wgrad(d_up3, t3n, 1, 1, tuple(m.lat3.weight.shape), m.lat3.weight)
            pg[m.lat3.bias] = bias_grad(d_up3)
            dt3 = _conv1x1(m.lat3, d_up3, transposed=True)
            upsample2_backward(d_up3, dt4)
        else:
            dt3 = torch.zeros_like(t3n)
        ctx.saved = None
        return (None, ops.from_nhwc(dt2), ops.from_nhwc(dt3), ops.from_nhwc(dt4)) + tuple(pg.get(p) for p in ctx.params)


COMPOSE_FPN_HEADS = bool(int(_os.environ.get("MDF_FPN_COMPOSED", "1")))      # dev A/B: 0 = the 64-channel 1/2- and 1/4-resolution tensors are formed


class FPNHeadsComposedFn(torch.autograd.Function):
    """The same heads through their ALGEBRA (backbone.py:59-63; the eval path's `_composed_heads`, now with a backward): 1x1 convs and
    bilinear upsampling are linear and commute, so
        y4 = O4 t4,   y3 = up(O3 t4) + (O3 L3) t3 + O3 b3,   y2 = up(up(O2 t4) + (O2 L3) t3 + O2 b3) + (O2 L2) t2 + O2 b2
    and the 64-channel 1/4- and 1/2-resolution tensors (141 MB at cfg3, written and re-read five times per step forward and
    backward) are never formed: every large-resolution operand has 16 or 32 channels.  Backward: the gradients of the composed
    matrices (A2 = O2 L2, B3 = O2 L3, A3 = O3 L3) and bias vectors are 1x1 weight gradients / column sums over small-channel
    maps, mapped back onto the seven parameters by a dozen tiny matrix products."""

    @staticmethod
    def _plan(m):
        """This step's composed matrices, packed by the step's weight pack (PackPlan, mdf_fpn_compose_fwd) -- or None when the heads are
        called outside a prepared training step (then the same algebra runs as torch ops, below)."""
        f = m.__dict__.get("_mdf_fpn")
        if f is not None and f["key"] == tuple((p_.data_ptr(), p_._version) for p_ in f["params"]):
            return f
        return None

    @staticmethod
    def forward(ctx, m, t2, t3, t4, *params):
        t2n, t3n, t4n = ops.to_nhwc(t2.detach()), ops.to_nhwc(t3.detach()), ops.to_nhwc(t4.detach())
        f = FPNHeadsComposedFn._plan(m)
        if f is not None:
            c2, c3, cm = f["dims"]

            def cw(x, wp, cin, cout, bias=None, res_up=None):
                return ops.conv2d_nhwc(x, wp, cin, cout, 1, 1, None, bias, False, None, 1.0, res_up)
            with torch.no_grad():
                y4 = cw(t4n, _pack1x1(m.out4, False), cm, m.out4.out_channels)
                y3 = cw(t3n, f["A3f"], c3, c3, f["f3"], res_up=cw(t4n, _pack1x1(m.out3, False), cm, c3))
                u3 = cw(t3n, f["B3f"], c3, c2, f["e3"], res_up=cw(t4n, _pack1x1(m.out2, False), cm, c2))
                y2 = cw(t2n, f["A2f"], c2, c2, f["e2"], res_up=u3)
            ctx.m, ctx.saved, ctx.params, ctx.fast = m, (t2n, t3n, t4n), params, f
            ctx.set_materialize_grads(False)
            return ops.from_nhwc(y4), ops.from_nhwc(y3), ops.from_nhwc(y2)
        ctx.fast = None
        with torch.no_grad():
            O2, O3, O4 = (c.weight.detach().reshape(c.out_channels, c.in_channels) for c in (m.out2, m.out3, m.out4))
            L2, L3 = (c.weight.detach().reshape(c.out_channels, c.in_channels) for c in (m.lat2, m.lat3))
            b2, b3 = m.lat2.bias.detach(), m.lat3.bias.detach()
            A2, B3, A3 = O2 @ L2, O2 @ L3, O3 @ L3                    # [16,16], [16,32], [32,32]
            e2, e3, f3 = O2 @ b2, O2 @ b3, O3 @ b3

            def pk(w):
                return ops.pack_conv2d_weight(w.reshape(w.shape[0], w.shape[1], 1, 1).contiguous())

            def c1(x, w, bias=None, res_up=None, res=None):
                return ops.conv2d_nhwc(x, pk(w), w.shape[1], w.shape[0], 1, 1, None, bias, False, res, 1.0, res_up)
            y4 = c1(t4n, O4)
            y3 = c1(t3n, A3, f3.contiguous(), res_up=c1(t4n, O3))
            c3 = c1(t3n, B3, e3.contiguous(), res_up=c1(t4n, O2))
            y2 = c1(t2n, A2, e2.contiguous(), res_up=c3)
        ctx.m, ctx.saved, ctx.params = m, (t2n, t3n, t4n, O2, O3, O4, L2, L3, b2, b3, A2, B3, A3), params
        ctx.set_materialize_grads(False)
        return ops.from_nhwc(y4), ops.from_nhwc(y3), ops.from_nhwc(y2)

    @staticmethod
    def backward(ctx, g4, g3, g2):
        m = ctx.m
        if ctx.fast is not None:
            return FPNHeadsComposedFn._backward_fast(ctx, g4, g3, g2)
        t2n, t3n, t4n, O2, O3, O4, L2, L3, b2, b3, A2, B3, A3 = ctx.saved
        pool = step_pool(t2n.device)

        def pk(w):
            return ops.pack_conv2d_weight(w.reshape(w.shape[0], w.shape[1], 1, 1).contiguous())

        def c1t(g, w, res=None):       # w^T g: [.., out] -> [.., in]
            wt = w.t()
            return ops.conv2d_nhwc(g, pk(wt), wt.shape[1], wt.shape[0], 1, 1, None, None, False, res, 1.0, None)

        def colsum(g):
            c = g.shape[-1]
            return bn_stats(g, g.numel() // c, c, pool=pool)[:c].float()

        def wg(g, t):                  # sum_pixels g (x) t -> [g channels, t channels]
            return conv2d_wgrad(g, t, 1, 1, (g.shape[-1], t.shape[-1], 1, 1)).reshape(g.shape[-1], t.shape[-1])
        dO2 = torch.zeros_like(O2); dO3 = torch.zeros_like(O3)
        dL2 = torch.zeros_like(L2); dL3 = torch.zeros_like(L3)
        db2 = torch.zeros_like(b2); db3 = torch.zeros_like(b3)
        dt4 = None
        if g2 is not None:
            g2n = ops.to_nhwc(g2)
            dA2, s2 = wg(g2n, t2n), colsum(g2n)
            dt2 = c1t(g2n, A2)
            gc3 = upsample2_backward(g2n)                          # [.,h/4,w/4,16]
            dB3, sc3 = wg(gc3, t3n), colsum(gc3)
            dt3 = c1t(gc3, B3)
            gc4 = upsample2_backward(gc3)                          # [.,h/8,w/8,16]
            dO2 += wg(gc4, t4n) + dA2 @ L2.t() + torch.outer(s2, b2) + dB3 @ L3.t() + torch.outer(sc3, b3)
            dL2 += O2.t() @ dA2
            dL3 += O2.t() @ dB3
            db2 += O2.t() @ s2
            db3 += O2.t() @ sc3
            dt4 = c1t(gc4, O2)
        else:
            dt2, dt3 = torch.zeros_like(t2n), None
        if g3 is not None:
            g3n = ops.to_nhwc(g3)
            dA3, s3 = wg(g3n, t3n), colsum(g3n)
            dt3 = c1t(g3n, A3, res=dt3)
            ga4 = upsample2_backward(g3n)                          # [.,h/8,w/8,32]
            dO3 += wg(ga4, t4n) + dA3 @ L3.t() + torch.outer(s3, b3)
            dL3 += O3.t() @ dA3
            db3 += O3.t() @ s3
            dt4 = c1t(ga4, O3, res=dt4)
        if dt3 is None:
            dt3 = torch.zeros_like(t3n)
        pg = {m.out2.weight: dO2.reshape(m.out2.weight.shape), m.out3.weight: dO3.reshape(m.out3.weight.shape),
              m.lat2.weight: dL2.reshape(m.lat2.weight.shape), m.lat3.weight: dL3.reshape(m.lat3.weight.shape),
              m.lat2.bias: db2, m.lat3.bias: db3}
        if g4 is not None:
            g4n = ops.to_nhwc(g4)
            pg[m.out4.weight] = wg(g4n, t4n).reshape(m.out4.weight.shape)
            dt4 = c1t(g4n, O4, res=dt4)
        if dt4 is None:
            dt4 = torch.zeros_like(t4n)
        ctx.saved = None
        return (None, ops.from_nhwc(dt2), ops.from_nhwc(dt3), ops.from_nhwc(dt4)) + tuple(pg.get(p) for p in ctx.params)


def _fpn_backward_fast(ctx, g4, g3, g2):
    """The backward of FPNHeadsComposedFn on the step's packed composed matrices: the five large-map weight gradients are summed
    by ONE launch, the map back onto the seven parameters is ONE launch (mdf_fpn_compose_bwd)."""
    m, f = ctx.m, ctx.fast
    t2n, t3n, t4n = ctx.saved
    c2, c3, cm = f["dims"]
    dev = t2n.device
    pool = step_pool(dev)
    if g2 is None or g3 is None or g4 is None:      # (a head without a gradient: not a training step of this network)
        g4 = torch.zeros_like(ops.from_nhwc(t4n)[:, :m.out4.out_channels]) if g4 is None else g4
        g3 = torch.zeros((t3n.shape[0], c3, t3n.shape[1], t3n.shape[2]), device=dev) if g3 is None else g3
        g2 = torch.zeros((t2n.shape[0], c2, t2n.shape[1], t2n.shape[2]), device=dev) if g2 is None else g2
    hold = []

    def cwt(g, wp, cin, cout, res=None):            # W^T g on the transposed pack
        return ops.conv2d_nhwc(g, wp, cin, cout, 1, 1, None, None, False, res, 1.0, None)

    def wg(g, t):
        return conv2d_wgrad(g, t, 1, 1, (g.shape[-1], t.shape[-1], 1, 1), hold=hold)

    def colsum(g):
        c = g.shape[-1]
        return bn_stats(g, g.numel() // c, c, pool=pool)
    g2n, g3n, g4n = ops.to_nhwc(g2), ops.to_nhwc(g3), ops.to_nhwc(g4)
    dA2, s2 = wg(g2n, t2n), colsum(g2n)
    dt2 = cwt(g2n, f["A2t"], c2, c2)
    gc3 = upsample2_backward(g2n)                                  # [.,h/4,w/4,c2]
    dB3, sc3 = wg(gc3, t3n), colsum(gc3)
    dt3 = cwt(gc3, f["B3t"], c2, c3)
    gc4 = upsample2_backward(gc3)                                  # [.,h/8,w/8,c2]
    W2 = wg(gc4, t4n)
    dt4 = cwt(gc4, _pack1x1(m.out2, True), c2, cm)
    dA3, s3 = wg(g3n, t3n), colsum(g3n)
    dt3 = cwt(g3n, f["A3t"], c3, c3, res=dt3)
    ga4 = upsample2_backward(g3n)                                  # [.,h/8,w/8,c3]
    W3 = wg(ga4, t4n)
    dt4 = cwt(ga4, _pack1x1(m.out3, True), c3, cm, res=dt4)
    dO4 = conv2d_wgrad(g4n, t4n, 1, 1, tuple(m.out4.weight.shape), m.out4.weight)      # (summed with the rest of the pass)
    dt4 = cwt(g4n, _pack1x1(m.out4, True), m.out4.out_channels, cm, res=dt4)
    sum_wgrad_jobs(hold)
    O2, O3, L2, b2, L3, b3 = f["params"]
    out = {p_: torch.empty_like(p_) for p_ in (m.out2.weight, m.out3.weight, m.lat2.weight, m.lat3.weight, m.lat2.bias, m.lat3.bias)}
    _abi("mdf_fpn_compose_bwd", (O2.data_ptr(), O3.data_ptr(), L2.data_ptr(), b2.data_ptr(), L3.data_ptr(), b3.data_ptr(),
                                 dA2.data_ptr(), dB3.data_ptr(), dA3.data_ptr(), W2.data_ptr(), W3.data_ptr(),
                                 s2.data_ptr(), sc3.data_ptr(), s3.data_ptr(), c2, c3, cm,
                                 out[m.out2.weight].data_ptr(), out[m.out3.weight].data_ptr(), out[m.lat2.weight].data_ptr(),
                                 out[m.lat3.weight].data_ptr(), out[m.lat2.bias].data_ptr(), out[m.lat3.bias].data_ptr(), _stream(t2n)),
         tag="FPN head parameter gradients")
    out[m.out4.weight] = dO4
    ctx.saved = None
    return (None, ops.from_nhwc(dt2), ops.from_nhwc(dt3), ops.from_nhwc(dt4)) + tuple(out.get(p_) for p_ in ctx.params)


FPNHeadsComposedFn._backward_fast = staticmethod(_fpn_backward_fast)


def fpn_heads_train(module, t2, t3, t4):
    params = tuple(p for mod in (module.lat2, module.lat3, module.out2, module.out3, module.out4) for p in mod.parameters())
    fn = FPNHeadsComposedFn if (COMPOSE_FPN_HEADS and module.lat2.bias is not None and module.out2.bias is None) else FPNHeadsTrainFn
    return fn.apply(module, t2, t3, t4, *params)


# --------------------------------------------------------------------------- refinement net in training mode
def _conv3x3(conv, x, relu=False, res=None, res_scale=1.0, shuffle=False, transposed=False):
    from .layers import cache_of_key
    w = conv.weight
    if transposed:     # input gradient: flipped taps, swapped channels
        wp = cache_of_key(conv, "dgrad").get((w,), lambda: ops.pack_conv2d_weight(w.detach().flip(2, 3).transpose(0, 1).contiguous()))
        return ops.conv2d_nhwc(x, wp, conv.out_channels, conv.in_channels, 3, 1, None, None, relu, res, res_scale)
    if shuffle:
        wp = cache_of_key(conv, "fwd").get((w,), lambda: ops.pack_conv2d_weight(ops.shuffle2_rows(w)))
    else:
        wp = cache_of_key(conv, "fwd").get((w,), lambda: ops.pack_conv2d_weight(w))
    return ops.conv2d_nhwc(x, wp, conv.in_channels, conv.out_channels, 3, 1, None, None, relu, res, res_scale, pixel_shuffle2=shuffle)


class RefineTrainFn(torch.autograd.Function):
    """RefineNet2.forward (refine.py:25-46) in training mode on the 2-D conv kernels; the depth input is detached in the
    reference (refine.py:29), so the backward produces weight gradients only."""

    @staticmethod
    def forward(ctx, m, depth, lo, span, *params):
        bsz = depth.shape[0]
        # (depth - lo) / span and lo + o * span: one launch each with torch's roundings (mdf_range_affine_fwd, as in eval)
        x = ops.range_affine(depth.detach(), lo.reshape(bsz), span.reshape(bsz), 0).unsqueeze(-1)   # [B,h,w,1]
        x0 = _conv3x3(m.conv0, x)
        y, chain = x0, []
        for blk in m.ress:                                                                          # y + 0.1*conv(relu(conv(y)))
            t = _conv3x3(blk.conv[0], y, relu=True)
            y_new = _conv3x3(blk.conv[2], t, res=y, res_scale=0.1)
            chain.append((blk, y, t))
            y = y_new
        z = _conv3x3(m.conv1, y, res=x0)
        s_ = _conv3x3(m.conv2[0], z, shuffle=True)                                                  # [B,2h,2w,8]
        o = _conv3x3(m.conv2[2], s_)                                                                # [B,2h,2w,1]
        ctx.m, ctx.saved, ctx.params, ctx.span = m, (x, x0, chain, y, z, s_), params, span
        return ops.range_affine(o.squeeze(-1), lo.reshape(bsz), span.reshape(bsz), 1)

    @staticmethod
    def backward(ctx, dout):
        m = ctx.m
        x, x0, chain, y3, z, s_ = ctx.saved
        b, h, w, _ = z.shape
        pg = {}

        def wg(conv, small, big):
            pg[conv.weight] = conv2d_wgrad(small, big, 3, 1, tuple(conv.weight.shape), conv.weight)
        do = (dout.unsqueeze(1) * ctx.span).permute(0, 2, 3, 1).contiguous()                       # [B,2h,2w,1]
        wg(m.conv2[2], do, s_)
        ds = _conv3x3(m.conv2[2], do, transposed=True)                                              # [B,2h,2w,8]
        dq = ds.view(b, h, 2, w, 2, 8).permute(0, 1, 3, 5, 2, 4).reshape(b, h, w, 32)                # un-shuffle: channel c*4 + dy*2 + dx
        wg(m.conv2[0], dq, z)
        dz = _conv3x3(m.conv2[0], dq, transposed=True)                                              # d (x0 + conv1(y3))
        wg(m.conv1, dz, y3)
        dy = _conv3x3(m.conv1, dz, transposed=True)
        for blk, y_in, t in reversed(chain):
            g = dy * 0.1
            wg(blk.conv[2], g, t)
            dt = _conv3x3(blk.conv[2], g, transposed=True) * (t > 0)
            wg(blk.conv[0], dt, y_in)
            dy = _conv3x3(blk.conv[0], dt, transposed=True, res=dy)                                 # dy + d conv_a
        wg(m.conv0, dz + dy, x)
        ctx.saved = None
        return (None, None, None, None) + tuple(pg.get(p) for p in ctx.params)


def refine_train(module, depth, lo, span):
    return RefineTrainFn.apply(module, depth, lo, span, *tuple(module.parameters()))


# --------------------------------------------------------------------------- training loss
class LossTrainFn(torch.autograd.Function):
    """net/loss.py:10-27 on the device: (depth_min [B], est_0, gt_0, est_1, gt_1, ...) -> scalar loss."""

    @staticmethod
    def forward(ctx, floor, *pairs):
        ests = [_f32c(t.detach()) for t in pairs[0::2]]
        gts = [_f32c(t.detach()) for t in pairs[1::2]]
        ns = len(ests)
        dev = ests[0].device
        b = ests[0].shape[0]
        fl = floor.detach()
        if fl.dtype not in (torch.float32, torch.float64):
            fl = fl.float()
        f64, fstride = int(fl.dtype == torch.float64), fl.stride(0) if fl.dim() > 0 else 0
        acc = step_pool(dev).take(2 * ns)
        st = _stream(ests[0])
        for e, g in zip(ests, gts):
            assert e.shape == g.shape and e.shape[0] == b, (e.shape, g.shape)
        if ns <= 8:     # all scales in one launch (blockIdx.z = scale)
            ep = (ctypes.c_void_p * ns)(*[e.data_ptr() for e in ests])
            gp = (ctypes.c_void_p * ns)(*[g.data_ptr() for g in gts])
            pb = (ctypes.c_int64 * ns)(*[e.numel() // b for e in ests])
            _abi("mdf_masked_smooth_l1_reduce_multi", (ep, gp, pb, ns, fl.data_ptr(), f64, fstride, b, acc.data_ptr(), st), tag=f"loss {ns} scales",
                 work={"bytes": 8.0 * sum(e.numel() for e in ests), "bound": "hbm"})
        else:
            for s_, (e, g) in enumerate(zip(ests, gts)):
                _abi("mdf_masked_smooth_l1_reduce", (e.data_ptr(), g.data_ptr(), fl.data_ptr(), f64, fstride, b, e.numel() // b,
                                                     acc[2 * s_:].data_ptr(), st), tag=f"loss {tuple(e.shape)}",
                     work={"bytes": 8.0 * e.numel(), "bound": "hbm"})
        loss = torch.empty((), device=dev, dtype=torch.float32)
        inv = torch.empty(ns, device=dev, dtype=torch.float32)
        _abi("mdf_masked_smooth_l1_finalize", (acc.data_ptr(), ns, loss.data_ptr(), inv.data_ptr(), st))
        ctx.saved = (ests, gts, fl, f64, fstride, inv)
        ctx.needs = [pairs[2 * i].requires_grad for i in range(ns)]
        return loss

    @staticmethod
    def backward(ctx, dloss):
        ests, gts, fl, f64, fstride, inv = ctx.saved
        dl = _f32c(dloss)
        out = [None]
        ns = len(ests)
        if ns <= 8:
            b = ests[0].shape[0]
            des = [torch.empty_like(e) if need else None for e, need in zip(ests, ctx.needs)]
            if any(d is not None for d in des):
                ep = (ctypes.c_void_p * ns)(*[e.data_ptr() for e in ests])
                gp = (ctypes.c_void_p * ns)(*[g.data_ptr() for g in gts])
                dp = (ctypes.c_void_p * ns)(*[None if d is None else d.data_ptr() for d in des])
                pb = (ctypes.c_int64 * ns)(*[e.numel() // b for e in ests])
                _abi("mdf_masked_smooth_l1_bwd_multi", (ep, gp, pb, ns, fl.data_ptr(), f64, fstride, b, dl.data_ptr(), inv.data_ptr(), dp, _stream(ests[0])),
                     tag=f"loss bwd {ns} scales", work={"bytes": 12.0 * sum(e.numel() for e, d in zip(ests, des) if d is not None), "bound": "hbm"})
            for d in des:
                out += [d, None]
            ctx.saved = None
            return tuple(out)
        for s_, (e, g) in enumerate(zip(ests, gts)):
            de = None
            if ctx.needs[s_]:
                de = torch.empty_like(e)
                b = e.shape[0]
                _abi("mdf_masked_smooth_l1_bwd", (e.data_ptr(), g.data_ptr(), fl.data_ptr(), f64, fstride, b, e.numel() // b, dl.data_ptr(),
                                                  inv[s_:].data_ptr(), de.data_ptr(), _stream(de)), tag=f"loss bwd {tuple(e.shape)}",
                     work={"bytes": 12.0 * e.numel(), "bound": "hbm"})
            out += [de, None]
        ctx.saved = None
        return tuple(out)


def loss_train(floor, pairs):
    flat = [t for pr in pairs for t in pr]
    return LossTrainFn.apply(floor, *flat)
