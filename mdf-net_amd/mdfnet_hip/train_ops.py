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

import torch

from . import lib
from . import ops
from .ops import _abi, _f32c, _need_gpu, _stream, _src_array, _hypos_arg

BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------- BatchNorm3d(batch stats) + ReLU
def bn_stats(y, n, c):
    sums = torch.zeros(2 * c, device=y.device, dtype=torch.float64)
    _abi("mdf_bn_stats_fwd", (y.data_ptr(), n, c, sums.data_ptr(), _stream(y)), tag=f"stats C{c} N{n}",
         work={"bytes": 4.0 * n * c, "bound": "hbm"})
    return sums


def bn_finalize(sums, bn, n, c):
    """-> aux [4C] = (a, b, mean, invstd); updates the module's running statistics like nn.BatchNorm3d.train()."""
    aux = torch.empty(4 * c, device=sums.device, dtype=torch.float32)
    track = bn.track_running_stats and bn.running_mean is not None
    mom = BN_MOMENTUM if bn.momentum is None else bn.momentum
    _abi("mdf_bn_finalize_fwd", (sums.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), ctypes.c_float(bn.eps), ctypes.c_float(mom),
                                 n, c, aux.data_ptr(), bn.running_mean.data_ptr() if track else None,
                                 bn.running_var.data_ptr() if track else None,
                                 bn.num_batches_tracked.data_ptr() if track else None, _stream(aux)))
    return aux


def bn_relu_apply(y, aux, res, n, c):
    z = torch.empty_like(y)
    _abi("mdf_bn_relu_apply_fwd", (y.data_ptr(), aux.data_ptr(), None if res is None else res.data_ptr(), z.data_ptr(), n, c, _stream(z)),
         tag=f"apply C{c} N{n}", work={"bytes": 4.0 * n * c * (3 if res is not None else 2), "bound": "hbm"})
    return z


def bn_relu_backward(dz, y, aux, gamma, n, c):
    """-> (dy, dgamma, dbeta)."""
    red = torch.zeros(2 * c, device=y.device, dtype=torch.float64)
    _abi("mdf_bn_relu_bwd_reduce", (dz.data_ptr(), y.data_ptr(), aux.data_ptr(), n, c, red.data_ptr(), _stream(y)),
         tag=f"bwd-reduce C{c} N{n}", work={"bytes": 8.0 * n * c, "bound": "hbm"})
    dy = torch.empty_like(y)
    dgamma = torch.empty(c, device=y.device, dtype=torch.float32)
    dbeta = torch.empty(c, device=y.device, dtype=torch.float32)
    _abi("mdf_bn_relu_bwd", (dz.data_ptr(), y.data_ptr(), aux.data_ptr(), red.data_ptr(), gamma.data_ptr(), n, c, dy.data_ptr(),
                             dgamma.data_ptr(), dbeta.data_ptr(), _stream(y)),
         tag=f"bwd C{c} N{n}", work={"bytes": 12.0 * n * c, "bound": "hbm"})
    return dy, dgamma, dbeta


# --------------------------------------------------------------------------- conv weight / input gradients
def conv3d_wgrad(small, big, stride, out_shape):
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
    _abi("mdf_conv3d_wgrad", (small.data_ptr(), big.data_ptr(), dw.data_ptr(), work.data_ptr(), b, ds, hs, ws, a, bc, stride, 0,
                              _stream(dw)), tag=f"wgrad {a}x{bc} s{stride} {ds}x{hs}x{ws}",
         work={"flops": 2.0 * 27 * a * bc * b * ds * hs * ws, "bytes": 4.0 * (small.numel() + big.numel()), "bound": "mfma"})
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


def conv3d_dgrad(conv, transposed, dy, add_to=None):
    """dx = [add_to +] (input gradient of the layer) as one launch of the forward conv kernel family."""
    wp = dgrad_pack(conv, transposed)
    if transposed:        # backward of ConvTranspose3d(Cin->Cout): stride-2 conv Cout -> Cin
        return ops.conv3d_ndhwc(dy, wp, conv.out_channels, conv.in_channels, 2, False, None, None, False, add_to)
    if conv.stride[0] == 2:
        return ops.conv3d_ndhwc(dy, wp, conv.out_channels, conv.in_channels, 2, True, None, None, False, add_to)
    return ops.conv3d_ndhwc(dy, wp, conv.out_channels, conv.in_channels, 1, False, None, None, False, add_to)


# --------------------------------------------------------------------------- prob head backward
def prob_head_backward(prob, hypos, ddepth, dprob, x_feat, weight):
    """-> (dx_feat [B,D,h,w,C], dweight [1,C,3,3,3])."""
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
    _abi("mdf_prob_conv_dgrad", (dlogit.data_ptr(), _f32c(weight.detach()).data_ptr(), dx.data_ptr(), b, d, h, w, c, _stream(dx)),
         tag=f"1->{c} dgrad {d}x{h}x{w}", work={"bytes": 4.0 * (dlogit.numel() + dx.numel()), "bound": "hbm"})
    dw = conv3d_wgrad(dlogit.view(b, d, h, w, 1), x_feat, 1, tuple(weight.shape))
    return dx, dw


# --------------------------------------------------------------------------- regulariser: layer tape
class Tape:
    """Forward record of the regulariser's layer program (net/unit/regular.py: `features`), replayed backwards."""

    def __init__(self):
        self.layers = []

    def layer(self, conv, bn, x, res):
        tr = isinstance(conv, torch.nn.ConvTranspose3d)
        stride = conv.stride[0]
        wp = ops_pack_fwd(conv, tr)
        y = ops.conv3d_ndhwc(x, wp, conv.in_channels, conv.out_channels, stride, tr, None, None, False, None)   # raw conv
        c = conv.out_channels
        n = y.numel() // c
        aux = bn_finalize(bn_stats(y, n, c), bn, n, c)
        z = bn_relu_apply(y, aux, res, n, c)
        self.layers.append((conv, bn, tr, stride, x, y, aux, res, z))
        return z

    def backward(self, grads):
        """grads: {id(tensor): gradient} holding the gradient of the last layer's output; returns parameter grads
        {param: grad} and leaves the input gradients in `grads`."""
        pg = {}
        for conv, bn, tr, stride, x, y, aux, res, z in reversed(self.layers):
            dz = grads.pop(id(z))
            if res is not None:
                grads[id(res)] = dz if id(res) not in grads else grads[id(res)] + dz
            c = conv.out_channels
            n = y.numel() // c
            dy, pg[bn.weight], pg[bn.bias] = bn_relu_backward(dz, y, aux, bn.weight, n, c)
            if tr:      # ConvTranspose3d: small = x (input), big = dy (twice the size)
                pg[conv.weight] = conv3d_wgrad(x, dy, 2, tuple(conv.weight.shape))
            else:
                pg[conv.weight] = conv3d_wgrad(dy, x, stride, tuple(conv.weight.shape))
            grads[id(x)] = conv3d_dgrad(conv, tr, dy, add_to=grads.get(id(x)))
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
        wpack = ops.pack_prob_weight(module.prob.weight)
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
        dfeat, dwp = prob_head_backward(prob, ctx.hypos, ddepth, dprob, ctx.feat, module.prob.weight)
        grads = {id(ctx.feat): dfeat}
        pg = tape.backward(grads)
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


def _agg_call(pass_, ref, srcs, proj, hyp, pp, par, red_in, dcost, cost, wsum, red_out, dref, dsrcs, dcw, b, c, g, d, h, w):
    def ptr(t):
        return None if t is None else t.data_ptr()
    algo = 4.0 * b * ((len(srcs) + 1) * c * h * w + g * d * h * w)
    _abi("mdf_warp_aggregate_vec_train", (pass_, ref.data_ptr(), _src_array(srcs), proj.data_ptr(), hyp.data_ptr(), pp, par.data_ptr(),
                                          ptr(red_in), ptr(dcost), ptr(cost), ptr(wsum), ptr(red_out), ptr(dref),
                                          None if dsrcs is None else _src_array(dsrcs), ptr(dcw), b, c, g, d, h, w, len(srcs),
                                          _stream(ref)), tag=f"train pass{pass_} C{c}G{g}D{d} {w}x{h} V{len(srcs) + 1}",
         work={"bytes": algo * (2 if pass_ == _PASS_BWD else 1), "bound": "hbm"})


class AggregateTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, proj, hypos, cw, gamma, beta, w2, b2, *features):
        bn = module.depth_weight[0].bn
        feas = [ops.nhwc(f.detach()).permute(0, 2, 3, 1).contiguous() for f in features]      # [B,h,w,C] memory
        b, h, w, c = feas[0].shape
        g = module.ngroups
        d = hypos.shape[1]
        hyp, pp = _hypos_arg(hypos.detach(), h, w)
        nsrc = len(feas) - 1
        dev = feas[0].device
        n = b * d * h * w
        head = torch.cat([cw.detach().reshape(-1).float(), w2.detach().reshape(1).float(), b2.detach().reshape(1).float(),
                          gamma.detach().reshape(1).float(), torch.full((1,), 1.0 / n, device=dev)])
        par0 = torch.cat([head, torch.zeros(4 * nsrc, device=dev)])
        red = torch.zeros(2 * nsrc, device=dev, dtype=torch.float64)
        _agg_call(_PASS_STATS, feas[0], feas[1:], proj, hyp, pp, par0, None, None, None, None, red, None, None, None, b, c, g, d, h, w)
        mean = red[0::2] / n
        var = (red[1::2] / n - mean * mean).clamp_(min=0.0)
        invstd = torch.rsqrt(var + bn.eps)
        alpha = gamma.detach().double() * invstd
        shift = beta.detach().double() - mean * alpha
        par = torch.cat([head, torch.stack([alpha, shift, mean, invstd], dim=1).reshape(-1).float()])
        if bn.track_running_stats and bn.running_mean is not None:
            # the module is called once per source view (homoaggregate.py:35-40): n_src sequential momentum updates
            mom = BN_MOMENTUM if bn.momentum is None else bn.momentum
            coef = torch.tensor([mom * (1.0 - mom) ** (nsrc - 1 - v) for v in range(nsrc)], dtype=torch.float64).to(dev, non_blocking=True)
            keep = (1.0 - mom) ** nsrc
            unb = var * (n / max(n - 1, 1))
            bn.running_mean.mul_(keep).add_((coef * mean).sum().float())
            bn.running_var.mul_(keep).add_((coef * unb).sum().float())
            bn.num_batches_tracked.add_(nsrc)
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
        red = torch.zeros(2 * nsrc + 2, device=dev, dtype=torch.float64)
        _agg_call(_PASS_BWD_REDUCE, feas[0], feas[1:], ctx.proj, ctx.hyp, ctx.pp, ctx.par, None, dc, ctx.cost, ctx.wsum, red, None, None,
                  None, b, c, g, d, h, w)
        dref = torch.empty_like(feas[0])
        dhalf = [torch.zeros((b, h, w, g), device=dev, dtype=torch.float32) for _ in feas[1:]]   # even channel of every pair
        dcw = torch.zeros(g, device=dev, dtype=torch.float32)
        _agg_call(_PASS_BWD, feas[0], feas[1:], ctx.proj, ctx.hyp, ctx.pp, ctx.par, red, dc, ctx.cost, ctx.wsum, None, dref, dhalf, dcw,
                  b, c, g, d, h, w)
        dsrcs = [torch.stack((t, -t), dim=-1).reshape(b, h, w, c) for t in dhalf]      # softmax pair: d v1 = -d v0
        s_cw, s_gamma, s_beta, s_w2, s_b2 = ctx.wshapes
        dgamma = red[1:2 * nsrc:2].sum().float().reshape(s_gamma)
        dbeta = red[0:2 * nsrc:2].sum().float().reshape(s_beta)
        dw2 = red[2 * nsrc].float().reshape(s_w2)
        db2 = red[2 * nsrc + 1].float().reshape(s_b2)
        dfeas = [t.permute(0, 3, 1, 2) for t in [dref] + dsrcs]
        ctx.cost = ctx.wsum = None
        return (None, None, None, dcw.reshape(s_cw), dgamma, dbeta, dw2, db2) + tuple(dfeas)


def aggregate_train(module, features, proj, hypos):
    head = module.depth_weight
    return AggregateTrainFn.apply(module, proj, hypos, head[0].conv.weight, head[0].bn.weight, head[0].bn.bias, head[1].weight, head[1].bias,
                                  *features)
