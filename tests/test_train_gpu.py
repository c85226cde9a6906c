"""GPU: the TRAINING path on the hand-written kernels (mdfnet_hip/train_ops.py; BASELINE config 3, train.py:36-45).

Checkers: the reference's training golden directly (loss, depths, gradients; host-side values pinned); torch autograd over the
ORACLE's training forward for all 158 parameter gradients of the whole model; and, per slot (aggregation, regulariser, feature
pyramid, refinement net), autograd over the oracle's functional restatement in fp32 and in float64 (oracle/train_check.py) -- the
product package's own CPU route (mdfnet_hip/stockops.py) is not a checker here.  Single conv / BatchNorm layers are compared with
torch's nn modules (third-party L0 arithmetic).  The yardstick throughout is the float64 result: an fp32 implementation is judged by
its distance from it relative to the fp32 oracle's own (tests/golden/train_tiny_f64.npz for the reference's golden step)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from mdfnet_hip import ddp, ops, synth, train_ops
from modelutil import build_model

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def _l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


# ----------------------------------------------------------------------------------------------- kernels one by one
@pytest.mark.parametrize("c", [8, 16, 32, 64])
def test_bn_relu_train_forward_backward(c):
    torch.manual_seed(c)
    shape = (2, 6, 10, 12, c)
    n = 2 * 6 * 10 * 12
    y = torch.randn(shape) * 2 + 0.3
    res = torch.randn(shape)
    bn = nn.BatchNorm3d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    ref_bn = nn.BatchNorm3d(c)
    ref_bn.load_state_dict(bn.state_dict())
    ref_bn.train()
    yr = y.permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
    zr = res.permute(0, 4, 1, 2, 3) + F.relu(ref_bn(yr))
    dz = torch.randn(shape)
    zr.backward(dz.permute(0, 4, 1, 2, 3))
    bn = bn.to(DEV)
    yd = y.to(DEV)
    aux = train_ops.bn_finalize(train_ops.bn_stats(yd, n, c), bn, n, c)
    z = train_ops.bn_relu_apply(yd, aux, res.to(DEV), n, c)
    assert _rel(z, zr.permute(0, 2, 3, 4, 1)) < 1e-5
    assert _rel(bn.running_mean, ref_bn.running_mean) < 1e-5 and _rel(bn.running_var, ref_bn.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1
    dy, dgamma, dbeta = train_ops.bn_relu_backward(dz.to(DEV), yd, aux, bn.weight, n, c)
    assert _rel(dy, yr.grad.permute(0, 2, 3, 4, 1)) < 2e-5
    assert _rel(dgamma, ref_bn.weight.grad) < 2e-5 and _rel(dbeta, ref_bn.bias.grad) < 2e-5


@pytest.mark.parametrize("cin,cout,stride,tr", [
    (32, 16, 1, False), (16, 16, 1, False), (32, 32, 1, False), (64, 64, 1, False), (16, 8, 1, False), (8, 8, 1, False),
    (16, 32, 2, False), (32, 64, 2, False), (8, 16, 2, False), (64, 32, 2, True), (32, 16, 2, True), (16, 8, 2, True)])
def test_conv3d_input_and_weight_gradients(cin, cout, stride, tr):
    """dgrad (the forward kernel family on re-packed weights) and wgrad (csrc/wgrad.hip) of every layer kind of the two
    regularisers vs torch autograd on the CPU; odd w (masked chunk tails) and a non-multiple-of-16 row length."""
    torch.manual_seed(cin * 100 + cout + stride)
    d, h, w = (4, 6, 22) if not tr else (2, 3, 11)
    if stride == 2 and not tr:
        d, h, w = 4, 6, 20
    conv = (nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False) if tr
            else nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False))
    x = torch.randn(2, cin, d, h, w, requires_grad=True)
    y = conv(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    convd = conv.to(DEV)
    xd = ops.to_ndhwc(x.detach().to(DEV))
    dyd = ops.to_ndhwc(dy.to(DEV))
    dx = train_ops.conv3d_dgrad(convd, tr, dyd)
    assert _rel(ops.from_ndhwc(dx), x.grad) < 2e-5
    extra = torch.randn_like(dx)
    dx2 = train_ops.conv3d_dgrad(convd, tr, dyd, add_to=extra)          # fused accumulation into a skip gradient
    assert _rel(dx2, dx + extra) < 1e-5
    dw = train_ops.conv3d_wgrad(xd, dyd, 2, tuple(conv.weight.shape)) if tr else train_ops.conv3d_wgrad(dyd, xd, stride, tuple(conv.weight.shape))
    assert _rel(dw, conv.weight.grad.cpu()) < 3e-5


def _bn_sums_reference(out, stat_y=None, aux=None, groups=1):
    """float64 reference of the epilogue sums over a channels-last tensor [B,...,C] split into `groups` along B."""
    c = out.shape[-1]
    o = out.detach().cpu().double().reshape(groups, -1, c)
    if stat_y is None:
        return torch.cat([o.sum(1), (o * o).sum(1)], dim=1).reshape(-1)
    yv = stat_y.detach().cpu().double().reshape(groups, -1, c)
    a, b, mu, inv = (t.unsqueeze(1) for t in aux.detach().cpu().double().reshape(groups, 4, c).unbind(1))
    mask = (yv * a + b) > 0          # the sign of the kernels' fp32 fma(y, a, b): the double product is exact, rounding keeps the sign
    dr = torch.where(mask, o, torch.zeros_like(o))
    return torch.cat([dr.sum(1), (dr * (yv - mu) * inv).sum(1)], dim=1).reshape(-1)


def _sums_close(got, ref, terms):
    """sums of `terms` fp32 values: fp32 partials per wave tile, fp64 beyond -> relative to the sum of magnitudes, ~1e-6."""
    got, ref = got.cpu().double(), ref.double()
    scale = ref.abs().max().clamp(min=1e-30)
    return float((got - ref).abs().max() / scale)


@pytest.mark.parametrize("cin,cout,stride,tr,dhw", [
    # small volumes: conv3d_kernel (v1), every mode incl. split-K; large ones (>= 150k voxels): the LDS kernels (Winograd / w-phase / direct)
    (32, 16, 1, False, (4, 6, 22)), (64, 64, 1, False, (2, 5, 9)), (16, 32, 2, False, (4, 6, 20)), (8, 16, 2, False, (8, 36, 52)),
    (64, 32, 2, True, (2, 3, 11)), (16, 8, 2, True, (4, 18, 26)),
    (16, 16, 1, False, (8, 140, 150)), (32, 16, 1, False, (6, 130, 200)), (32, 32, 1, False, (6, 130, 200)), (16, 8, 1, False, (8, 140, 150)),
    (8, 8, 1, False, (8, 141, 151))])
def test_conv3d_epilogue_sums(cin, cout, stride, tr, dhw):
    """mdf_conv3d_train_fwd: the raw conv is the one mdf_conv3d_fwd computes (bit-identical), and the per-channel sums its epilogue
    accumulates equal float64 sums over that output -- mode 1 (sum y, sum y^2: the layer's batch statistics) and mode 2 (BatchNorm-
    backward sums of the output taken as dz against another layer's raw output and constants), with and without a skip operand."""
    torch.manual_seed(cin * 7 + cout + stride)
    d, h, w = dhw
    x = torch.randn(1, d, h, w, cin, device=DEV)
    wt = torch.randn((cin, cout, 3, 3, 3) if tr else (cout, cin, 3, 3, 3), device=DEV) * 0.1
    wp = ops.pack_conv3d_weight(wt, transposed=tr)
    y_plain = ops.conv3d_ndhwc(x, wp, cin, cout, stride, tr, None, None, False, None)
    S = ops.STAT_SLICES            # the launch spreads its blocks over S copies of the sums; their totals count
    sums = torch.zeros(S, 2 * cout, device=DEV, dtype=torch.float64)
    y = ops.conv3d_train(x, wp, cin, cout, stride, tr, None, 1, sums)
    assert torch.equal(y, y_plain)
    nvox = y.numel() // cout
    assert _sums_close(sums.sum(0), _bn_sums_reference(y), nvox) < 2e-6
    # mode 2 (+ skip operand)
    res = torch.randn_like(y)
    stat_y = torch.randn_like(y) * 2 + 0.3
    aux = torch.cat([torch.rand(cout) + 0.5, torch.randn(cout) * 0.3, torch.randn(cout) * 0.2, torch.rand(cout) + 0.5]).to(DEV)
    red = torch.zeros(1, 2 * cout, device=DEV, dtype=torch.float64)          # (one slice works too)
    dz = ops.conv3d_train(x, wp, cin, cout, stride, tr, res, 2, red, stat_y, aux)
    assert torch.equal(dz, ops.conv3d_ndhwc(x, wp, cin, cout, stride, tr, None, None, False, res))
    assert _sums_close(red.sum(0), _bn_sums_reference(dz, stat_y, aux), nvox) < 2e-6


@pytest.mark.parametrize("cin,cout,k,stride,hw", [(3, 8, 3, 1, (40, 72)), (8, 8, 3, 1, (40, 72)), (16, 16, 3, 1, (20, 36)), (32, 32, 3, 1, (22, 34)),
                                                   (64, 64, 3, 1, (10, 18)), (8, 16, 5, 2, (40, 72)), (16, 32, 5, 2, (20, 36)), (32, 64, 5, 2, (20, 36))])
def test_conv2d_epilogue_sums_per_group(cin, cout, k, stride, hw, monkeypatch):
    """mdf_conv2d_train_fwd with 3 BatchNorm groups of 2 images: a block's run of tiles crosses image and group boundaries."""
    monkeypatch.setenv("MDF_CONV_K5_WINOGRAD", "0")     # the training kernels are the direct form: compare with the eval launch in that form
    torch.manual_seed(cin + cout + k)
    h, w = hw
    groups, b = 3, 6
    planar = cin == 3
    x = torch.randn((b, cin, h, w) if planar else (b, h, w, cin), device=DEV)
    wt = torch.randn(cout, cin, k, k, device=DEV) * 0.1
    wp = ops.pack_conv2d_weight(wt)
    y_plain = ops.conv2d_nhwc(x, wp, cin, cout, k, stride, planar_in=planar)
    S = ops.STAT_SLICES
    sums = torch.zeros(S, groups * 2 * cout, device=DEV, dtype=torch.float64)
    y = ops.conv2d_train(x, wp, cin, cout, k, stride, planar, 1, sums, groups)
    assert torch.equal(y, y_plain)
    assert _sums_close(sums.sum(0), _bn_sums_reference(y, groups=groups), y.numel() // cout // groups) < 2e-6
    if k == 3 and not planar:
        stat_y = torch.randn_like(y) * 2 + 0.3
        aux = torch.stack([torch.cat([torch.rand(cout) + 0.5, torch.randn(cout) * 0.3, torch.randn(cout) * 0.2, torch.rand(cout) + 0.5])
                           for _ in range(groups)]).to(DEV)
        red = torch.zeros(S, groups * 2 * cout, device=DEV, dtype=torch.float64)
        dz = ops.conv2d_train(x, wp, cin, cout, k, stride, False, 2, red, groups, stat_y, aux.reshape(-1))
        assert torch.equal(dz, y_plain)
        assert _sums_close(red.sum(0), _bn_sums_reference(dz, stat_y, aux, groups=groups), y.numel() // cout // groups) < 2e-6


@pytest.mark.parametrize("stage", [0, 1])
def test_fused_bn_sums_equal_the_separate_passes(stage, seeded_sd, monkeypatch):
    """The regulariser's training forward + backward with the BatchNorm sums in the conv epilogues (default) against the same
    with mdf_bn_stats_fwd / mdf_bn_relu_bwd_reduce as passes of their own: same launches otherwise, so the results agree to the
    rounding of the sums (fp32 partials in a different order)."""
    torch.manual_seed(21 + stage)
    m = build_model()          # torch's default initialisation: a well-conditioned chain (the peaked golden recipe amplifies any
    reg = m.Regular[stage].train().to(DEV)      # rounding ~1e3x through its softmax, see test_regulariser_training_forward_backward)
    # (seeded instances checked to be well-conditioned -- both variants ~1e-5 from float64, scripts/diag_bn_fuse.py; some random
    # instances are not: batch statistics over the ~100 voxels of the deepest level can put BOTH variants 3e-3 from float64 and
    # 2e-4..2e-3 from each other)
    g, d, h, w = ((32, 48, 12, 20), (16, 24, 48, 80))[stage]
    torch.manual_seed(stage)
    cost = torch.rand(2, g, d, h, w, device=DEV)
    hyp = ((425 + 510 * torch.rand(2, 1, h, w)) + torch.linspace(-20, 20, d).reshape(1, d, 1, 1)).to(DEV)
    dd = torch.randn(2, h, w, device=DEV)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(train_ops, "FUSE_BN_SUMS", fused)
        ops.count_begin()
        c = cost.clone().requires_grad_(True)
        prob, depth = reg(c, hyp)
        depth.backward(dd)
        calls = ops.count_end()
        res[fused] = [prob.detach(), depth.detach(), c.grad] + [p.grad.clone() for p in reg.parameters()]
        reg.zero_grad()
        if fused:
            assert calls.get("mdf_bn_stats_fwd", 0) == 0 and calls.get("mdf_conv3d_train_fwd", 0) >= 10
            assert calls.get("mdf_bn_relu_bwd_reduce", 0) == 0 and calls.get("mdf_prob_conv_dgrad_stat", 0) == 1, calls   # the last layer's sums ride in the prob head's input-gradient launch
        else:
            assert calls.get("mdf_conv3d_train_fwd", 0) == 0 and calls.get("mdf_bn_relu_bwd_reduce", 0) >= 10
    for i, (a, b_) in enumerate(zip(res[True], res[False])):
        assert _l2(a, b_) < 1e-4, (i, _l2(a, b_))


@pytest.mark.parametrize("stage", [0, 2])
def test_batched_weight_gradient_launches_equal_the_separate_ones(stage, monkeypatch):
    """the weight-gradient launches of one backward pass, recorded and issued as one job-table launch per kernel
    form (mdf_wgrad_batch_begin / mdf_wgrad_batch_flush), run the SAME block bodies on the same operands as the separate launches:
    every gradient agrees to the rounding of the final sums, and the pass makes fewer launches."""
    torch.manual_seed(5 + stage)
    reg = build_model().Regular[stage].train().to(DEV)
    g, d, h, w = ((32, 48, 12, 20), (16, 24, 48, 80), (8, 8, 96, 160))[stage]
    cost = torch.rand(2, g, d, h, w, device=DEV)
    hyp = ((425 + 510 * torch.rand(2, 1, h, w)) + torch.linspace(-20, 20, d).reshape(1, d, 1, 1)).to(DEV)
    dd = torch.randn(2, h, w, device=DEV)
    res, calls = {}, {}
    for batched in (True, False):
        monkeypatch.setattr(train_ops, "BATCH_WGRAD", batched)
        ops.count_begin()
        c = cost.clone().requires_grad_(True)
        prob, depth = reg(c, hyp)
        depth.backward(dd)
        calls[batched] = ops.count_end()
        res[batched] = [c.grad] + [p.grad.clone() for p in reg.parameters()]
        reg.zero_grad()
    assert calls[True].get("mdf_wgrad_batch_flush", 0) == 1 and calls[True].get("mdf_conv3d_wgrad_partial", 0) == 0, calls[True]
    assert calls[False].get("mdf_wgrad_batch_flush", 0) == 0 and calls[False].get("mdf_conv3d_wgrad_partial", 0) >= 10, calls[False]
    for i, (a, b_) in enumerate(zip(res[True], res[False])):      # (the partial tiles are bit-identical -- next test; their final sums meet
        assert _l2(a, b_) < 1e-6, (i, _l2(a, b_))                 #  through fp32 atomics in arrival order: ~2e-7 between ANY two runs)


def test_job_table_launch_writes_the_same_partial_tiles_as_separate_launches():
    """The boundary itself: mdf_conv3d_wgrad_partial / mdf_conv2d_wgrad_partial between mdf_wgrad_batch_begin and mdf_wgrad_batch_flush
    only RECORD their launch; the flush issues one job-table launch per kernel form.  The partial tiles in the workspaces (the
    final sums meet through fp32 atomics in arrival order, so the tiles are the deterministic quantity) are bit-identical to the
    ones separate launches write -- mixed shapes, strides, 2-D and 3-D, in one batch."""
    import ctypes
    from mdfnet_hip import lib, check
    L = lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(3)
    jobs = []
    for (a, bc, s, b, ds, hs, ws) in [(8, 8, 1, 2, 8, 24, 40), (16, 8, 2, 2, 4, 12, 20), (16, 16, 1, 1, 6, 12, 20), (32, 16, 2, 1, 3, 6, 10),
                                      (8, 16, 1, 1, 5, 7, 9)]:
        small = torch.randn(b, ds, hs, ws, a, device=DEV)
        big = torch.randn(b, ds * s, hs * s, ws * s, bc, device=DEV)
        jobs.append(("mdf_conv3d_wgrad_partial", small, big, a * bc * 27, L.mdf_conv3d_wgrad_workspace(b, ds, hs, ws, a, bc),
                     lambda sm, bg, dw, wk, ns, a=a, bc=bc, s=s, b=b, ds=ds, hs=hs, ws=ws: (sm, bg, dw, wk, b, ds, hs, ws, a, bc, s, ns, st)))
    for (a, bc, k, s, b, hs, ws) in [(16, 8, 3, 1, 2, 48, 64), (32, 16, 5, 2, 1, 24, 32), (64, 32, 3, 1, 1, 12, 16)]:
        small = torch.randn(b, hs, ws, a, device=DEV)
        big = torch.randn(b, hs * s, ws * s, bc, device=DEV)
        jobs.append(("mdf_conv2d_wgrad_partial", small, big, a * bc * k * k, L.mdf_conv2d_wgrad_workspace(b, hs, ws, a, bc, k),
                     lambda sm, bg, dw, wk, ns, a=a, bc=bc, k=k, s=s, b=b, hs=hs, ws=ws: (sm, bg, dw, wk, b, hs, ws, a, bc, k, s, ns, st)))

    def run(batched):
        outs = []
        if batched:
            check(L.mdf_wgrad_batch_begin(), "mdf_wgrad_batch_begin")
        for entry, small, big, n, nwork, args in jobs:
            work = torch.full((nwork,), float("nan"), device=DEV)
            dw = torch.full((n,), float("nan"), device=DEV)
            ns = ctypes.c_int(0)
            check(getattr(L, entry)(*args(small.data_ptr(), big.data_ptr(), dw.data_ptr(), work.data_ptr(), ctypes.byref(ns))), entry)
            outs.append((work, dw, ns.value))
        if batched:
            torch.cuda.synchronize()
            assert sum(bool(torch.isnan(w).all()) for w, _, _ in outs) >= 5     # the LDS-staged forms have not launched yet
            check(L.mdf_wgrad_batch_flush(st), "mdf_wgrad_batch_flush")
        torch.cuda.synchronize()
        return outs
    sep, bat = run(False), run(True)
    for (entry, _, _, n, _, _), (w0, d0, n0), (w1, d1, n1) in zip(jobs, sep, bat):
        assert n0 == n1 and n0 >= 1
        assert torch.equal(w0[:n0 * n], w1[:n0 * n]) and not bool(torch.isnan(w0[:n0 * n]).any()), entry
        assert torch.equal(d0.isnan(), d1.isnan()) and torch.equal(d0.nan_to_num(), d1.nan_to_num())   # (zeroed, or left for the sum)
    # a flush with nothing recorded is a no-op; calls outside begin/flush launch at once again
    check(L.mdf_wgrad_batch_flush(st), "mdf_wgrad_batch_flush")
    again = run(False)
    assert all(torch.equal(a[0][:a[2] * j[3]], b_[0][:b_[2] * j[3]]) for a, b_, j in zip(again, sep, jobs))


@pytest.mark.parametrize("c,d", [(8, 8), (16, 24)])
def test_prob_head_backward(c, d):
    torch.manual_seed(c)
    b, h, w = 2, 12, 20
    conv = nn.Conv3d(c, 1, 3, padding=1, bias=False)
    x = torch.randn(b, c, d, h, w, requires_grad=True)
    hyp = 425 + 500 * torch.rand(b, d, h, w)
    prob = F.softmax(conv(x).squeeze(1), dim=1)
    depth = torch.sum(prob * hyp, 1)
    dd, dp = torch.randn_like(depth), 0.1 * torch.randn_like(prob)
    (depth * dd).sum().backward(retain_graph=True)
    gx1, gw1 = x.grad.clone(), conv.weight.grad.clone()
    x.grad = None; conv.weight.grad = None
    ((depth * dd).sum() + (prob * dp).sum()).backward()
    xd = ops.to_ndhwc(x.detach().to(DEV))
    w_dev = conv.weight.detach().to(DEV)
    probd, depthd = ops.prob_head(xd, w_dev, hyp.to(DEV))
    assert _rel(probd, prob) < 1e-4 and _rel(depthd, depth) < 1e-5
    dx, dw = train_ops.prob_head_backward(probd, hyp.to(DEV), dd.to(DEV), None, xd, w_dev)
    assert _rel(ops.from_ndhwc(dx), gx1) < 1e-4 and _rel(dw, gw1) < 1e-4
    dx, dw = train_ops.prob_head_backward(probd, hyp.to(DEV), dd.to(DEV), dp.to(DEV), xd, w_dev)
    assert _rel(ops.from_ndhwc(dx), x.grad) < 1e-4 and _rel(dw, conv.weight.grad) < 1e-4


@pytest.mark.parametrize("stage,b,h,w,nviews,per_pixel", [(0, 2, 12, 20, 3, False), (1, 1, 24, 36, 4, True), (2, 1, 32, 48, 5, True)])
def test_vector_aggregate_training_forward_backward(stage, b, h, w, nviews, per_pixel):
    """Homoaggre[s] in training mode (batch-statistics BatchNorm3d(1) per source-view call, running stats, gradient to
    reference and source features and to the five head parameters) vs autograd over the ORACLE's vector_aggregate (oracle/train_check.py)."""
    from net.unit.homoaggregate import VectorAggregate
    from net.unit.scale import scale_cam
    from oracle import train_check as TC
    c, g, d = ((64, 32, 48), (32, 16, 24), (16, 8, 8))[stage]
    torch.manual_seed(stage)
    mod = VectorAggregate(g)
    with torch.no_grad():
        for p in mod.parameters():
            p.add_(0.3 * torch.randn_like(p))
    sd = {k: v.clone() for k, v in mod.state_dict().items()}
    intr, extr, dr = synth.make_cameras(w * 2 ** (3 - stage), h * 2 ** (3 - stage), nviews, batch=b, rot_deg=2.0, seed=stage + 7)
    rp, sps = scale_cam(intr, extr, stage)
    feats = [torch.randn(b, c, h, w) for _ in range(nviews)]
    if per_pixel:
        hyp = (425 + 510 * torch.rand(b, 1, h, w)) + torch.linspace(-20, 20, d).reshape(1, d, 1, 1)
    else:
        hyp = torch.linspace(425, 935, d).reshape(1, d, 1, 1).repeat(b, 1, 1, 1)
    dcost = torch.randn(b, g, d, h, w)
    r32 = TC.aggregate(sd, g, feats, rp, sps, hyp, dcost, torch.float32)       # the oracle's VectorAggregate, autograd, fp32
    r64 = TC.aggregate(sd, g, feats, rp, sps, hyp, dcost, torch.float64)       # ... and in float64 (same sample positions up to
    mod.train().to(DEV)                                                         #     fp64-vs-fp32 rounding of the grid)
    fd = [f.to(DEV).requires_grad_(True) for f in feats]
    cost = mod(fd, rp.to(DEV), tuple(s.to(DEV) for s in sps), hyp.to(DEV))
    assert cost.shape == r32["cost"].shape and _rel(cost, r32["cost"]) < 2e-5
    cost.backward(dcost.to(DEV))
    bn = mod.depth_weight[0].bn
    want = r32["buffers"]
    assert _rel(bn.running_mean, want["depth_weight.0.bn.running_mean"]) < 1e-5 and _rel(bn.running_var, want["depth_weight.0.bn.running_var"]) < 1e-4
    assert int(bn.num_batches_tracked) == int(want["depth_weight.0.bn.num_batches_tracked"]) == nviews - 1
    print(f"\nstage {stage}: cost L2 error vs float64: HIP {_l2(cost, r64['cost']):.1e} | fp32 oracle {_l2(r32['cost'], r64['cost']):.1e}; "
          f"d ref-feature: HIP {_l2(fd[0].grad, r64['dfeats'][0]):.1e} | oracle {_l2(r32['dfeats'][0], r64['dfeats'][0]):.1e}; "
          f"d src-feature: HIP {_l2(fd[1].grad, r64['dfeats'][1]):.1e} | oracle {_l2(r32['dfeats'][1], r64['dfeats'][1]):.1e}")
    for i, (a_, r_) in enumerate(zip(fd, r32["dfeats"])):
        assert _rel(a_.grad, r_) < 2e-4, f"feature {i}"
    for k, pa in mod.named_parameters():
        # the scalar head parameters are sums of ~1e5 signed terms that cancel to ~1e-5 of sum|terms|: fp32 noise of EITHER
        # side (the fp32 oracle included) is ~1e-3 of the result
        assert _rel(pa.grad, r32["grads"][k]) < (5e-4 if pa.numel() > 1 else 1e-2), k


def _aggregate_case(stage, width, height, nviews, seed, hyp_spread=20.0):
    """Homoaggre[stage] in training mode at the stage's share of a width x height item: (module, device inputs)."""
    from net.unit.homoaggregate import VectorAggregate
    from net.unit.scale import scale_cam
    c, g, d = ((64, 32, 48), (32, 16, 24), (16, 8, 8))[stage]
    h, w = height >> (3 - stage), width >> (3 - stage)
    torch.manual_seed(seed)
    mod = VectorAggregate(g)
    with torch.no_grad():
        for p in mod.parameters():
            p.add_(0.3 * torch.randn_like(p))
    intr, extr, dr = synth.make_cameras(width, height, nviews, batch=1, rot_deg=2.0, seed=seed + 7)
    rp, sps = scale_cam(intr, extr, stage)
    feats = [torch.randn(1, c, h, w) for _ in range(nviews)]
    if stage == 0:
        hyp = torch.linspace(425, 935, d).reshape(1, d, 1, 1)
    else:
        hyp = (425 + 510 * torch.rand(1, 1, h, w)) + torch.linspace(-hyp_spread, hyp_spread, d).reshape(1, d, 1, 1)
    return mod.train().to(DEV), feats, rp, sps, hyp


@pytest.mark.parametrize("stage", [0, 1, 2])
def test_aggregate_scatter_claim_vs_all_atomics_full_size(stage, monkeypatch):
    """the claim-based NON-atomic LDS window updates of warp_bwd_kernel (csrc/warp_aggregate_train.hip) at the
    full cfg3 size (768x576, 5 views: 96x72x64ch, 192x144x32ch, 384x288x16ch), where a lost add would pass every finite /
    loss-decreasing check: d src and d ref against the SAME pass with every window update an LDS atomic (MDF_WARP_BWD_ATOMIC=1,
    read per call).  The two differ by fp32 summation order only.  (homoaggregate.py:16-20,35-46 backward)"""
    mod, feats, rp, sps, hyp = _aggregate_case(stage, 768, 576, 5, seed=40 + stage)
    torch.manual_seed(stage)
    grads = {}
    for mode in ("claim", "atomic"):
        if mode == "atomic":
            monkeypatch.setenv("MDF_WARP_BWD_ATOMIC", "1")
        else:
            monkeypatch.delenv("MDF_WARP_BWD_ATOMIC", raising=False)
        fd = [f.to(DEV).requires_grad_(True) for f in feats]
        cost = mod(fd, rp.to(DEV), tuple(s.to(DEV) for s in sps), hyp.to(DEV))
        if mode == "claim":
            dcost = torch.randn_like(cost)
        cost.backward(dcost)
        grads[mode] = [f.grad.clone() for f in fd] + [p.grad.clone() for p in mod.parameters()]
        mod.zero_grad()
    monkeypatch.delenv("MDF_WARP_BWD_ATOMIC", raising=False)
    worst = 0.0
    for i, (a, b) in enumerate(zip(grads["claim"], grads["atomic"])):
        assert torch.isfinite(a).all() and float(b.abs().max()) > 0
        e_max, e_l2 = _rel(a, b), _l2(a, b)
        worst = max(worst, e_max)
        # a lost add drops one tap contribution: ~1e-2..1 of that texel's value; reordering fp32 sums of <= a few hundred terms: ~1e-6.
        # (the head-parameter gradients behind the features are cancelling sums over every voxel through float atomics whose
        # order differs from run to run: looser)
        if i < len(feats):
            assert e_max < 2e-5 and e_l2 < 4e-6, (i, e_max, e_l2)
        else:
            assert e_max < 1e-3, (i, e_max, e_l2)
    print(f"\nstage {stage}: claim-based vs all-atomic scatter, worst max-rel difference {worst:.2e}")


@pytest.mark.parametrize("stage", [0, 2])
def test_aggregate_backward_nonfinite_gradient_does_not_fault(stage):
    """a non-finite upstream gradient (a diverged step, or the NaN statistics a z == 0 plane gives) must propagate as
    NaN like torch's grid_sample backward -- and must not index the scatter's LDS window out of range: an out-of-bounds tap has
    weight 0, lies outside the window's bounding box, and 0 * NaN = NaN used to pass the kernel's `value != 0` liveness test."""
    mod, feats, rp, sps, hyp = _aggregate_case(stage, 256, 192, 4, seed=60 + stage, hyp_spread=200.0)
    # strong parallax + a wide depth range: many samples leave the source maps (out-of-bounds taps next to live ones)
    sps = tuple(s.clone() for s in sps)
    for i, s_ in enumerate(sps):
        s_[:, 0, 3] += (-1) ** i * 0.4 * (256 >> (3 - stage)) * 650.0      # ~0.4 map widths of shift at mid range
    fd = [f.to(DEV).requires_grad_(True) for f in feats]
    cost = mod(fd, rp.to(DEV), tuple(s.to(DEV) for s in sps), hyp.to(DEV))
    dcost = torch.randn_like(cost)
    dcost[:, :, ::2] = float("nan")
    dcost[:, :, 1, ::3] = float("inf")
    cost.backward(dcost)
    torch.cuda.synchronize()                       # a fault would surface here
    assert torch.isnan(fd[1].grad).any() and torch.isnan(fd[0].grad).any()
    # z == 0 for every sample of source view 0 (row 2 of its projection zeroed, as the H4 golden does): NaN positions, NaN statistics
    sps0 = tuple(s.clone() for s in sps)
    sps0[0][:, 2, :] = 0.0
    mod.zero_grad()
    fd = [f.to(DEV).requires_grad_(True) for f in feats]
    cost = mod(fd, rp.to(DEV), tuple(s.to(DEV) for s in sps0), hyp.to(DEV))
    cost.backward(torch.randn_like(cost))
    torch.cuda.synchronize()
    assert torch.isnan(cost).any()
    # and the device still computes: the finite case right after
    mod.zero_grad()
    fd = [f.to(DEV).requires_grad_(True) for f in feats]
    cost = mod(fd, rp.to(DEV), tuple(s.to(DEV) for s in sps), hyp.to(DEV))
    cost.backward(torch.randn_like(cost))
    assert torch.isfinite(cost).all() and all(torch.isfinite(f.grad).all() for f in fd)


@pytest.mark.parametrize("weights", ["default_init", "seeded_peaked"])
@pytest.mark.parametrize("stage", [0, 1, 2])
def test_regulariser_training_forward_backward(stage, weights, seeded_sd):
    """Regular[s] + soft-argmin in training mode: outputs, input gradient, every parameter gradient and the BatchNorm running
    statistics against the ORACLE in float64 (oracle/train_check.py: autograd over oracle.mvs_oracle.regular under precision(f64)),
    judged by what fp32 itself can do: the fp32 oracle on the same input AND on three copies of the input moved by one fp32 ulp, each
    compared with the float64 result of the unperturbed input.  The HIP path may be at most 1.5 x as far from float64 as the farthest of those
    fp32 runs -- no floors, no per-configuration factors.
    default_init: torch's default initialisation, a well-conditioned chain.  seeded_peaked: the golden recipe (prob conv scaled up so
    that volumes are peaked): the softmax amplifies rounding anywhere in the 11-layer chain ~1e3x."""
    from oracle import train_check as TC
    torch.manual_seed(11 + stage)
    m = build_model()
    if weights == "seeded_peaked":
        m.load_state_dict(seeded_sd)
    import copy
    sd = m.Regular[stage].state_dict()
    reg = copy.deepcopy(m.Regular[stage]).to(DEV).train()
    g, d, h, w = ((32, 48, 12, 20), (16, 24, 24, 40), (8, 8, 48, 56))[stage]
    torch.manual_seed(stage + 3)
    cost = torch.rand(2, g, d, h, w)
    hyp = (425 + 510 * torch.rand(2, 1, h, w)) + torch.linspace(-20, 20, d).reshape(1, d, 1, 1)
    dd = torch.randn(2, h, w)
    want = TC.regulariser(sd, cost, hyp, dd, torch.float64)            # the oracle in float64: the yardstick

    def errors(r):
        e = {k: _l2(r[k], want[k]) for k in ("prob", "depth", "dcost")}
        e.update({k: _l2(g_, want["grads"][k]) for k, g_ in r["grads"].items()})
        return e

    # the fp32 oracle: unperturbed (its buffers are the running-statistics reference) + eight draws with the input AND the parameters
    # moved by one fp32 ulp (gen_golden.one_ulp): the error of a gradient is quantised by which near-zero ReLU units flip, and one run
    # samples one combination (scripts: the stage-0 golden-weight case takes 2.3e-5 / 5.2e-4 / 2.3e-3 on dcost over such draws)
    from oracle.gen_golden import one_ulp
    spread, first, cpu = {}, None, None
    for t in range(-1, 8):
        c, sdt = cost, sd
        if t >= 0:
            torch.manual_seed(200 + t)
            sdt = {k: (v * one_ulp(v.shape) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
            c = cost * one_ulp(cost.shape)
        r = TC.regulariser(sdt, c, hyp, dd, torch.float32)
        e = errors(r)
        spread = {k: max(spread.get(k, 0.0), v) for k, v in e.items()}
        if first is None:
            first, cpu = e, r
    cd = cost.to(DEV).requires_grad_(True)
    prob, depth = reg(cd, hyp.to(DEV))
    depth.backward(dd.to(DEV))
    assert all(p.grad is not None for p in reg.parameters())
    got = errors({"prob": prob, "depth": depth, "dcost": cd.grad, "grads": {k: p.grad for k, p in reg.named_parameters()}})
    rows = sorted(((got[k] / max(spread[k], 1e-30), got[k], first[k], spread[k], k) for k in got), reverse=True)
    print(f"\nstage {stage} {weights}: L2 error vs float64, HIP | fp32 oracle unperturbed | fp32 oracle worst of 9: "
          + ", ".join(f"{k} {got[k]:.1e} | {first[k]:.1e} | {spread[k]:.1e}" for k in ("prob", "depth", "dcost")))
    print(f"   worst HIP / fp32-worst ratios: {[(f'{q:.2f}', f'{a:.1e}', f'{c_:.1e}', k) for q, a, _, c_, k in rows[:4]]}; "
          f"median HIP / fp32-unperturbed {float(np.median([got[k] / max(first[k], 1e-30) for k in got])):.2f}")
    # Bars: (1) typical accuracy -- the median tensor is no farther from float64 than 1.5 x the unperturbed fp32 oracle (measured
    # 0.06-0.26: the centred softmax backward of prob_bwd.hip makes the HIP gradients 4-16 x MORE accurate than fp32 autograd);
    # (2) EVERY tensor within 1.5 x the fp32 spread (measured <= 1.02; no relaxation clause: what moves a gradient between the
    # quantised error levels is one ReLU decision, shown directly by test_relu_decisions_explain_the_gradient_spread).
    bad = [(k, a, c_) for q, a, _, c_, k in rows if q > 1.5]
    print(f"   tensors beyond 1.5 x the fp32 spread: {len(bad)} of {len(rows)}")
    assert not bad, bad
    assert float(np.median([got[k] / max(first[k], 1e-30) for k in got])) <= 1.5
    assert _l2(cd.grad, cpu["dcost"]) < 1e-2
    for k, pa in reg.named_parameters():
        assert _l2(pa.grad, cpu["grads"][k]) < 2e-2, (k, _l2(pa.grad, cpu["grads"][k]))
    for k, ba in reg.named_buffers():                       # running statistics: nn.BatchNorm's update from the oracle's batch statistics
        assert _rel(ba.float(), cpu["buffers"][k].float()) < 1e-4, k


@pytest.mark.parametrize("stage", [0, 2])
def test_relu_decisions_explain_the_gradient_spread(stage, seeded_sd):
    """the error of the regulariser's gradients against float64 is quantised (dcost of Regular[0] with the golden
    weights: 2.3e-5, 5.2e-4 or 2.3e-3 over one-ulp perturbations), explained so far only by perturbation statistics.  Direct proof:
    (1) every ReLU decision of the HIP forward is compared with the float64 oracle's (the tape holds each layer's raw conv output and
        its folded BatchNorm (a, b): the decision is y a + b > 0); the decisions differ on a handful of units, and every one of them has
        a float64 pre-activation within MARGIN = 2e-5 layer-rms of zero, i.e. within fp32 rounding after up to 11 layers;
    (2) the float64 oracle is run again FOLLOWING the HIP decisions (relu(x) := x * [HIP decision]): against that run the HIP
        gradients are at the fp32 rounding level -- at least 5x closer than to the free float64 run whenever decisions differ, and
        within FLOOR = 5e-5 of it in any case (measured <= 8e-6).  So the whole excess of the free comparison is the flipped units, not the kernels."""
    from oracle import train_check as TC
    import copy
    MARGIN, FLOOR = 2e-5, 5e-5
    torch.manual_seed(11 + stage)
    m = build_model()
    m.load_state_dict(seeded_sd)
    sd = m.Regular[stage].state_dict()
    reg = copy.deepcopy(m.Regular[stage]).to(DEV).train()
    g, d, h, w = ((32, 48, 12, 20), (16, 24, 24, 40), (8, 8, 48, 56))[stage]
    torch.manual_seed(stage + 3)
    cost = torch.rand(2, g, d, h, w)
    hyp = (425 + 510 * torch.rand(2, 1, h, w)) + torch.linspace(-20, 20, d).reshape(1, d, 1, 1)
    dd = torch.randn(2, h, w)
    pre64 = []
    free = TC.regulariser(sd, cost, hyp, dd, torch.float64, relu=lambda x: (pre64.append(x.detach()), torch.relu(x))[1])
    cd = cost.to(DEV).requires_grad_(True)
    prob, depth = reg(cd, hyp.to(DEV))
    depth.backward(dd.to(DEV))
    # the decisions were taken in the forward pass: re-run it for the tape (same kernels, same inputs: deterministic)
    cd2 = cost.to(DEV).requires_grad_(True)
    _, depth2 = reg(cd2, hyp.to(DEV))
    layers_ = depth2.grad_fn.tape.layers
    assert len(layers_) == len(pre64)
    hip_dec, flips, worst = [], 0, 0.0
    for (conv, bn, tr, stride, x, y, aux, res, z), p64 in zip(layers_, pre64):
        c = conv.out_channels
        pre = y * aux[:c] + aux[c:2 * c]                                  # NDHWC, the kernel's own folded scale / shift
        dec = ops.from_ndhwc((pre > 0).float()).cpu().bool()
        assert dec.shape == p64.shape
        dif = dec != (p64 > 0)
        flips += int(dif.sum())
        if dif.any():
            worst = max(worst, float((p64[dif].abs() / p64.pow(2).mean().sqrt()).max()))
        hip_dec.append(dec)
    it = iter(hip_dec)
    forced = TC.regulariser(sd, cost, hyp, dd, torch.float64, relu=lambda x: x * next(it).to(x.dtype))
    got = {"dcost": cd.grad, **{k: p.grad for k, p in reg.named_parameters()}}
    rows = []
    for k, v in got.items():
        e_free = _l2(v, free["dcost"] if k == "dcost" else free["grads"][k])
        e_forced = _l2(v, forced["dcost"] if k == "dcost" else forced["grads"][k])
        rows.append((e_free / max(e_forced, 1e-30), e_free, e_forced, k))
    rows.sort(reverse=True)
    units = sum(int(p.numel()) for p in pre64)
    print(f"\nstage {stage}: {flips} of {units} ReLU decisions differ from float64; farthest flipped pre-activation {worst:.1e} layer-rms from zero; "
          f"dcost error vs float64 free {_l2(cd.grad, free['dcost']):.1e}, following the HIP decisions {_l2(cd.grad, forced['dcost']):.1e}; "
          f"largest free / forced ratios {[(f'{q:.0f}x', f'{a:.1e}', f'{b_:.1e}', k) for q, a, b_, k in rows[:3]]}")
    assert worst <= MARGIN, (flips, worst)
    assert flips <= units // 100000 + 8, flips
    for q, e_free, e_forced, k in rows:
        assert e_forced <= FLOOR, (k, e_free, e_forced)
        if flips and e_free > 5 * FLOOR:
            assert e_forced * 5 <= e_free, (k, e_free, e_forced)


def test_frozen_or_hooked_weight_does_not_take_the_deferred_sum_route(seeded_sd):
    """the deferred weight-gradient sum keeps only the ADDRESS of dw.  For a frozen conv weight autograd drops the returned
    tensor (its block is re-used within the same backward pass), for a hooked parameter AccumulateGrad clones it: both must be summed
    at once.  One layer frozen + one hooked: every other gradient equals the all-trainable run's, the hooked one is complete, and a
    backward pass that raised leaves no pending job behind for the next step."""
    import copy
    from mdfnet_hip import train_ops
    torch.manual_seed(5)
    m = build_model()
    m.load_state_dict(seeded_sd)
    base = copy.deepcopy(m.Regular[1]).to(DEV).train()
    test = copy.deepcopy(m.Regular[1]).to(DEV).train()
    g, d, h, w = 16, 24, 24, 40
    cost = torch.rand(1, g, d, h, w, device=DEV)
    hyp = (425 + 510 * torch.rand(1, 1, h, w, device=DEV)) + torch.linspace(-20, 20, d, device=DEV).reshape(1, d, 1, 1)
    dd = torch.randn(1, h, w, device=DEV)
    names = [k for k, p in test.named_parameters() if p.dim() == 5]
    frozen, hooked = names[2], names[4]
    params = dict(test.named_parameters())
    params[frozen].requires_grad_(False)
    seen = {}
    params[hooked].register_hook(lambda gr: seen.__setitem__("hook", gr.clone()))
    for reg in (base, test):
        c = cost.clone().requires_grad_(True)
        _, depth = reg(c, hyp)
        depth.backward(dd)
        reg.dcost = c.grad
    torch.cuda.synchronize()
    assert not train_ops._PENDING_SUMS
    assert params[frozen].grad is None
    for (k, pb), (_, pt) in zip(base.named_parameters(), test.named_parameters()):
        if k == frozen:
            continue
        assert pt.grad is not None and torch.isfinite(pt.grad).all(), k
        assert _l2(pt.grad, pb.grad) < 2e-5, (k, _l2(pt.grad, pb.grad))
    assert _l2(seen["hook"], dict(base.named_parameters())[hooked].grad) < 2e-5        # complete when the hook saw it
    assert _l2(test.dcost, base.dcost) < 2e-5
    # stale jobs of a pass that never reached the engine's callback are dropped, not flushed
    train_ops._PENDING_SUMS[0] = [[], [("stale",)]]
    train_ops.drop_stale_wgrad_sums(0)
    assert not train_ops._PENDING_SUMS


# ----------------------------------------------------------------------------------------------- whole model
def _step(m, dev, g):
    from net.loss import Loss
    imgs, extr, intr, dr = synth.make_scene(96, 64, 3, batch=2, rot_deg=3.0, seed=31)
    out = m(imgs.to(dev), extr.to(dev), intr.to(dev), dr.to(dev))
    gt = {k: T(g["gt" + k]).to(dev) for k in ("3", "2", "1", "0")}
    loss = Loss()(out, gt, dr.to(dev))
    return out, loss


def test_training_step_on_gpu_vs_reference_golden(golden, seeded_sd):
    """One training step (forward, loss, backward, flat-bucket gradient, Adam) with Homoaggre / Regular / Depth_regress /
    Depth_hypos on the hand-written training kernels, against the REFERENCE's golden loss, depths and gradients."""
    g = golden("train_tiny.npz")
    m = build_model()
    m.load_state_dict(seeded_sd)
    m.train().to(DEV)
    bucket = ddp.FlatBucket(m)                      # single rank: gradients still live in ONE flat buffer
    assert bucket.flat.numel() == 1206380 and bucket.flat.is_cuda
    used = []
    orig = train_ops._abi
    train_ops._abi = lambda name, *a, **k: (used.append(name), orig(name, *a, **k))[1]
    # the host-side control-plane values (projection / camera products, gauss-fit row, log thresholds) are the BUILD host's,
    # recorded while the reference produced this golden (oracle/gen_golden.py:gen_train): everything the GPU computes is then
    # compared with the reference's own training step at the eval leg's bars, on any host
    rec = ops.recorded_host_values(projs=[g[f"host_proj{st}"] for st in range(3)], cams=[g[f"host_cam{st}"] for st in range(3)],
                                   fit_row=g["host_fit_row"],
                                   log_thresh={1: float(g["host_log_thresh1"]), 2: float(g["host_log_thresh2"])})
    try:
        with rec:
            out, loss = _step(m, DEV, g)
        bucket.zero_grad()
        loss.backward()
    finally:
        train_ops._abi = orig
    wgrad = "mdf_wgrad_batch_flush" if train_ops.BATCH_WGRAD else "mdf_conv3d_wgrad_partial"   # (one job-table launch per kernel form)
    assert {"mdf_warp_aggregate_vec_train", wgrad, "mdf_wgrad_sum_batch", "mdf_bn_relu_bwd", "mdf_prob_conv_dgrad_stat"} <= set(used)
    bucket.allreduce_gradients()
    # Yardstick: the same step through the REFERENCE in float64 (tests/golden/train_tiny_f64.npz, oracle/gen_golden.py:gen_train_f64 runs
    # /root/reference's own model under reference_in_float64(); the oracle's float64 run agrees with it to <= 2e-15).  The reference's
    # own fp32 result is e_ref away from it; the HIP path may be at most 1.5 x that far, AND at most 2 x e_ref from
    # the reference's fp32 golden itself (a binding bar against the reference's output; two fp32 results that
    # are each ~e_ref from the exact one are ~sqrt(2) e_ref apart if their errors are independent: measured 0.4-0.75 e_ref).
    g64 = golden("train_tiny_f64.npz")
    fails = []
    for i, d in enumerate(out["depth"]):
        mine = d.detach().cpu().numpy().astype(np.float64)
        e_hip, e_ref = np.abs(mine - g64[f"depth{i}"]).mean(), np.abs(g[f"depth{i}"] - g64[f"depth{i}"]).mean()
        err = np.abs(mine - g[f"depth{i}"])
        print(f"\ndepth{i}: mean |d| vs float64: HIP {e_hip:.3e}, reference fp32 {e_ref:.3e} (ratio {e_hip / e_ref:.2f}); "
              f"HIP vs reference fp32 golden: mean {err.mean():.3e} max {err.max():.3e}")
        fails += [("depth", i, e_hip, e_ref)] if e_hip > 1.5 * e_ref else []
        fails += [("depth vs the reference fp32 golden", i, err.mean(), e_ref)] if err.mean() > 2.0 * e_ref else []
        assert err.max() < 0.5, (i, err.max())
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=2e-5)
    params = dict(m.named_parameters())
    assert all(p.grad is not None for p in params.values())
    got = {k[5:]: params[k[5:]].grad.detach().cpu().numpy().copy() for k in g if k.startswith("grad:")}
    # Gradients, same yardstick: max-relative distance from the float64 gradient, HIP vs the reference's fp32 (the golden).
    ratios = []
    for k, mine in got.items():
        r64 = g64["grad:" + k]
        scale = np.abs(r64).max()
        e_hip, e_ref = np.abs(mine - r64).max() / scale, np.abs(g["grad:" + k] - r64).max() / scale
        rel = np.abs(mine - g["grad:" + k]).max() / np.abs(g["grad:" + k]).max()
        print(f"grad:{k}: max rel err vs float64: HIP {e_hip:.2e}, reference fp32 {e_ref:.2e} (ratio {e_hip / e_ref:.2f}); HIP vs reference fp32 {rel:.2e}")
        # the reference's run is ONE sample of a noisy quantity (peaked softmaxes, ReLU and mask decisions): `spread` is the farthest
        # the fp32 oracle lands from float64 over the unperturbed inputs and six one-ulp draws of images and parameters (gen_golden.fp32_spread)
        spread = max(e_ref, float(g64["spread:grad:" + k]))
        ratios.append(e_hip / e_ref)
        fails += [(k, e_hip, e_ref, spread)] if e_hip > 1.5 * spread else []
    print(f"median HIP / reference distance from float64 over the sampled gradients: {float(np.median(ratios)):.2f}")
    assert not fails, fails
    assert float(np.median(ratios)) <= 1.5, ratios
    bucket.zero_grad()
    for k, mine in got.items():
        params[k].grad = torch.from_numpy(mine).to(DEV)
    bucket.allreduce_gradients()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    opt.step()                                      # the optimizer consumes the bucket's views
    assert torch.isfinite(torch.cat([p.detach().reshape(-1) for p in m.parameters()])).all()


def test_all_parameter_gradients_vs_the_oracle(golden, seeded_sd):
    """Every one of the 158 parameter tensors: the HIP training path vs torch autograd over the ORACLE's training forward
    (oracle.core_forward(training=True) + mvs_loss, pinned to the reference's training golden by tests/test_oracle_golden.py) on
    this host's CPU -- the checker lives under oracle/, not in the product package."""
    from oracle import mvs_oracle as O
    g = golden("train_tiny.npz")
    imgs, extr, intr, dr = synth.make_scene(96, 64, 3, batch=2, rot_deg=3.0, seed=31)
    sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.clone()) for k, v in seeded_sd.items()}
    out_ref = O.core_forward(sd, imgs, extr, intr, dr, training=True)
    gt = {k: T(g["gt" + k]) for k in ("3", "2", "1", "0")}
    loss_ref = O.mvs_loss(out_ref["depth"], gt, dr)
    loss_ref.backward()
    m = build_model()
    m.load_state_dict(seeded_sd)
    m.train().to(DEV)
    ops.count_begin()
    out, loss = _step(m, DEV, g)
    loss.backward()
    calls = ops.count_end()
    # inside a prepared step the pyramid's heads run on the step's packed composed matrices: one launch forms them, one maps their
    # gradients back onto the seven parameters (csrc/fpn_compose.hip), one slab sum serves the five large-map products
    assert calls.get("mdf_fpn_compose_fwd") == 1 and calls.get("mdf_fpn_compose_bwd") == 1, calls
    for i, (a, b) in enumerate(zip(out["depth"], out_ref["depth"])):
        err = (a.detach().cpu() - b.detach()).abs()
        print(f"\ndepth{i}: HIP training path vs the oracle (same host): mean |d| {float(err.mean()):.3e} max {float(err.max()):.3e}")
    assert abs(float(loss) - float(loss_ref)) <= 2e-5 * abs(float(loss_ref))
    # float64 yardstick (the oracle in float64 on this host, oracle/gen_golden.py:train_f64): per tensor, the HIP gradient's L2 distance
    # from it against the fp32 oracle's own
    from oracle import gen_golden
    _, d64, g64 = gen_golden.train_f64(seeded_sd, {k: g["gt" + k] for k in ("3", "2", "1", "0")})
    for i, (a, b) in enumerate(zip(out["depth"], out_ref["depth"])):
        e_hip = np.abs(a.detach().cpu().numpy() - d64[i]).mean()
        e_ref = np.abs(b.detach().numpy() - d64[i]).mean()
        print(f"depth{i}: mean |d| vs float64: HIP {e_hip:.3e}, fp32 oracle {e_ref:.3e} (ratio {e_hip / e_ref:.2f})")
        assert e_hip <= 1.5 * e_ref, (i, e_hip, e_ref)
    # per tensor: (a) never farther from float64 than 1.5 x the farthest fp32-oracle run (unperturbed + six one-ulp draws of the
    # images and parameters, this host: gen_golden.fp32_spread) -- several gradients are discontinuous at rounding level, one fp32 run is one sample;
    # (b) typically as close as the fp32 oracle: median of e_hip / e_oracle <= 1.5 over the 158 tensors
    _, spread = gen_golden.fp32_spread(seeded_sd, {k: g["gt" + k] for k in ("3", "2", "1", "0")}, d64, g64, draws=6, metric="l2")
    rows = []
    for k, pa in m.named_parameters():
        assert sd[k].grad is not None, k
        r = torch.from_numpy(g64[k])
        rows.append((_l2(pa.grad, r), _l2(sd[k].grad, r), spread[k], k))
    by_spread = sorted(((eh / max(sp, 1e-30), eh, er, sp, k) for eh, er, sp, k in rows), reverse=True)
    med = float(np.median([eh / max(er, 1e-30) for eh, er, _, _ in rows]))
    print("\nparameter gradients, L2 rel err vs float64 -- worst HIP / fp32-spread:",
          [(f"{q:.2f}", f"HIP {eh:.1e}", f"oracle {er:.1e}", f"spread {sp:.1e}", k) for q, eh, er, sp, k in by_spread[:6]],
          f"; median HIP / fp32 oracle {med:.2f}; worst HIP error {max(eh for eh, *_ in rows):.1e}")
    assert by_spread[0][0] <= 1.5, by_spread[:5]
    assert med <= 1.5, med


def test_full_size_cfg3_training_step():
    """BASELINE config 3 at its full size: 768x576, 5 views, one sample per GPU, forward + loss + backward + Adam."""
    from net.loss import Loss
    m = build_model()
    m.load_state_dict(synth.seeded_state_dict(m.state_dict(), seed=1))
    m.train().to(DEV)
    bucket = ddp.FlatBucket(m)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    imgs, extr, intr, dr = synth.make_scene(768, 576, 5, batch=1, rot_deg=3.0, seed=77)
    rng = np.random.RandomState(5)
    gt = {k: T((425 + 510 * rng.rand(1, 576 // s, 768 // s)).astype(np.float32)).to(DEV) for k, s in (("3", 8), ("2", 4), ("1", 2), ("0", 1))}
    losses = []
    for it in range(2):
        out = m(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))
        loss = Loss()(out, gt, dr.to(DEV))
        bucket.zero_grad()
        loss.backward()
        bucket.allreduce_gradients()
        opt.step()
        losses.append(float(loss))
    assert out["depth"][3].shape == (1, 576, 768) and all(np.isfinite(losses))
    assert torch.isfinite(bucket.flat).all() and float(bucket.flat.abs().max()) > 0
    print("\ncfg3 full-size losses:", losses, "peak memory GiB:", torch.cuda.max_memory_allocated() / 2 ** 30)


# ----------------------------------------------------------------------------------------------- feature pyramid (2-D)
@pytest.mark.parametrize("cin,cout,k,stride", [(8, 8, 3, 1), (16, 16, 3, 1), (32, 32, 3, 1), (64, 64, 3, 1), (8, 16, 5, 2), (16, 32, 5, 2),
                                                (32, 64, 5, 2), (3, 8, 3, 1)])
def test_conv2d_input_and_weight_gradients(cin, cout, k, stride):
    torch.manual_seed(cin + cout + k)
    h, w = (12, 36) if stride == 1 else (16, 40)
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, bias=False)
    x = torch.randn(3, cin, h, w, requires_grad=True)
    y = conv(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    convd = conv.to(DEV)
    dyd = ops.to_nhwc(dy.to(DEV))
    if cin == 3:
        x4 = torch.zeros(3, h, w, 4, device=DEV)
        x4[..., :3] = x.detach().to(DEV).permute(0, 2, 3, 1)
        dw = train_ops.conv2d_wgrad(dyd, x4, k, stride, tuple(conv.weight.shape))
    else:
        xd = ops.to_nhwc(x.detach().to(DEV))
        dw = train_ops.conv2d_wgrad(dyd, xd, k, stride, tuple(conv.weight.shape))
        dx = train_ops.conv2d_dgrad(convd, dyd)
        assert _rel(ops.from_nhwc(dx), x.grad) < 2e-5
    assert _rel(dw, conv.weight.grad.cpu()) < 3e-5


def test_feature_pyramid_training_forward_backward():
    """FPN_4Scales in training mode, 3 views x batch 2 in one pass with per-view BatchNorm statistics: outputs, every
    parameter gradient and the running statistics vs V separate calls of the ORACLE's fpn_4scales in fp32 and in float64
    (oracle/train_check.py); the HIP path may be at most 1.5 x as far from float64 as the fp32 oracle."""
    from net.unit.backbone import FPN_4Scales
    from oracle import train_check as TC
    torch.manual_seed(5)
    mod = FPN_4Scales().train()
    sd = {k: v.clone() for k, v in mod.state_dict().items()}
    mod = mod.to(DEV)
    b, v, h, w = 2, 3, 64, 96
    imgs = torch.rand(b, v, 3, h, w)
    gouts = [[torch.randn(b, cc, h >> k, w >> k) for cc, k in ((64, 3), (32, 2), (16, 1))] for _ in range(v)]
    r32 = TC.pyramid(sd, imgs, gouts, torch.float32)
    r64 = TC.pyramid(sd, imgs, gouts, torch.float64)
    outs = mod.forward_views(imgs.to(DEV))
    sum((t * g.to(DEV)).sum() for o, go in zip(outs, gouts) for t, g in zip(o, go)).backward()
    for i in range(v):
        for a_, r_, r64_ in zip(outs[i], r32["outs"][i], r64["outs"][i]):
            assert a_.shape == r_.shape and _l2(a_, r64_) <= 1.5 * _l2(r_, r64_), (i, _l2(a_, r64_), _l2(r_, r64_))
    rows = []
    for k, pa in mod.named_parameters():
        assert pa.grad is not None, k
        rows.append((_l2(pa.grad, r64["grads"][k]), _l2(r32["grads"][k], r64["grads"][k]), k))
    ratios = sorted(((eh / max(ec, 1e-30), eh, ec, k) for eh, ec, k in rows), reverse=True)
    print(f"\nfeature pyramid: parameter gradients, L2 error vs float64: worst HIP / fp32 oracle {[(f'{q:.2f}', f'{eh:.1e}', f'{ec:.1e}', k) for q, eh, ec, k in ratios[:4]]}; "
          f"median ratio {float(np.median([q for q, *_ in ratios])):.2f}")
    assert ratios[0][0] <= 1.5, ratios[:4]
    for k, ba in mod.named_buffers():
        assert _rel(ba.float(), r32["buffers"][k].float()) < 1e-4, k


def test_refine_net_training_forward_backward():
    """RefineNet2 in training mode (no BatchNorm; detached input): output and every weight gradient vs the ORACLE's refine_net2 in
    fp32 and in float64 (oracle/train_check.py); at most 1.5 x as far from float64 as the fp32 oracle."""
    from net.unit.refine import RefineNet2
    from oracle import train_check as TC
    torch.manual_seed(9)
    mod = RefineNet2().train()
    sd = {k: v.clone() for k, v in mod.state_dict().items()}
    mod = mod.to(DEV)
    b, h, w = 2, 48, 72
    depth = 425 + 510 * torch.rand(b, h, w)
    dr = torch.tensor([[425.0, 935.0], [425.0, 935.0]], dtype=torch.float64)
    gout = torch.randn(b, 2 * h, 2 * w)
    r32 = TC.refine(sd, depth, dr, gout, torch.float32)
    r64 = TC.refine(sd, depth, dr, gout, torch.float64)
    out = mod(depth.to(DEV), dr.to(DEV))
    (out * gout.to(DEV)).sum().backward()
    assert out.shape == r32["out"].shape and _l2(out, r64["out"]) <= 1.5 * _l2(r32["out"], r64["out"])
    worst = 0.0
    for k, pa in mod.named_parameters():
        assert pa.grad is not None, k
        e_hip, e_cpu = _l2(pa.grad, r64["grads"][k]), _l2(r32["grads"][k], r64["grads"][k])
        worst = max(worst, e_hip / max(e_cpu, 1e-30))
        assert e_hip <= 1.5 * e_cpu, (k, e_hip, e_cpu)
    print(f"\nrefine net: worst HIP / fp32 oracle distance from float64 over the parameter gradients: {worst:.2f}")


def test_batched_weight_packing_equals_the_per_layer_packs(seeded_sd):
    """train_ops.PackPlan (one mdf_pack_batch launch for every weight set of a training step) writes bit-for-bit what the
    per-layer route builds with torch re-indexing (flip / transpose / pad / index_select) + mdf_conv*_pack_weights."""
    model = build_model()
    model.load_state_dict(seeded_sd)
    model = model.to(DEV).train()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    plan = train_ops.PackPlan(model)
    for job in plan.jobs:
        job[1].fill_(float("nan"))        # (the 3-D buffers are sized for the larger of the plain and transposed layouts)
    plan.run()
    torch.cuda.synchronize()
    assert len(plan.jobs) > 90 and plan.nblocks > 0
    S = train_ops
    for param, dst, is3d, tr, mode, cin, cout, ntaps, a0, a1 in plan.jobs:
        w = param.detach()
        if is3d:
            if tr:
                ref = ops.pack_conv3d_weight(w, transposed=True)
            elif mode == S._SRC_SWAPFLIP:
                ref = ops.pack_conv3d_weight(w.flip(2, 3, 4).transpose(0, 1).contiguous())
            else:                                 # direct: Conv3d weight, or a ConvTranspose3d weight read as [out=Cin][in=Cout]
                ref = ops.pack_conv3d_weight(w)
        elif mode == S._SRC_PROB:
            ref = ops.pack_prob_weight(w)
        elif mode == S._SRC_K5S2:
            ref = ops.pack_conv2d_weight(S._k5s2_dgrad_weight(w)[a1:a1 + cout].contiguous())
        elif mode == S._SRC_SHUFFLE2:
            ref = ops.pack_conv2d_weight(ops.shuffle2_rows(w))
        elif mode == S._SRC_SWAPFLIP:
            ref = ops.pack_conv2d_weight(w.flip(2, 3).transpose(0, 1).contiguous())
        elif mode == S._SRC_SWAP:
            ref = ops.pack_conv2d_weight(w.transpose(0, 1).contiguous())
        else:
            ref = ops.pack_conv2d_weight(w)
        assert ref.numel() == dst.numel()
        written = ~torch.isnan(dst)
        assert int(written.sum()) >= param.numel() and bool(written[:param.numel()].all())
        assert torch.equal(dst[written], ref[written]), (tuple(param.shape), is3d, tr, mode, cin, cout, ntaps)
    # and the caches hand the buffers out without re-packing
    from mdfnet_hip.layers import cache_of_key
    conv = next(m for m in model.Regular[0].modules() if isinstance(m, nn.Conv3d))
    got = cache_of_key(conv, "fwd").get((conv.weight,), lambda: pytest.fail("cache missed after PackPlan.run"))
    assert got.data_ptr() in {j[1].data_ptr() for j in plan.jobs}


@pytest.mark.parametrize("range_dtype", [torch.float64, torch.float32])
def test_fused_loss_value_and_gradient(range_dtype):
    """net/loss.py on the GPU (csrc/loss.hip: one reduction per scale, one finalize, one backward launch per scale) against
    the reference formulation (boolean-indexed smooth-L1 means, loss.py:19-25) on the CPU."""
    from net import loss as loss_mod
    torch.manual_seed(7)
    b = 2
    ests, gts = [], {}
    for k, (h, w) in zip(("3", "2", "1", "0"), ((9, 12), (18, 24), (36, 48), (72, 96))):
        gt = torch.rand(b, h, w) * 500 + 300
        gt[:, : h // 3] = 0.0                      # invalid region (gt <= depth_min)
        est = gt + torch.randn(b, h, w) * 1.5      # both branches of smooth-L1
        ests.append(est)
        gts[k] = gt
    dr = torch.tensor([[425.0, 935.0], [300.0 + 1e-7, 900.0]], dtype=range_dtype)
    gts["3"][1, -1, -1] = float(dr[1, 0].float())  # a gt that sits exactly at depth_min rounded to fp32
    crit = loss_mod.Loss()
    e_cpu = [e.clone().requires_grad_(True) for e in ests]
    ref = crit({"depth": e_cpu}, gts, dr)
    ref.backward()
    e_gpu = [e.to(DEV).requires_grad_(True) for e in ests]
    got = crit({"depth": e_gpu}, {k: v.to(DEV) for k, v in gts.items()}, dr.to(DEV))
    assert got.shape == () and got.dtype == torch.float32
    got.backward()
    assert abs(float(got) - float(ref)) <= 2e-6 * abs(float(ref))
    for a, r in zip(e_gpu, e_cpu):
        assert torch.equal(a.grad.cpu() != 0, r.grad != 0)                      # the same valid mask
        assert _rel(a.grad, r.grad) <= 2e-6


def test_flat_adam_matches_torch_adam(seeded_sd):
    """mdfnet_hip.optim.FlatAdam (one launch over all parameters) against torch.optim.Adam, 4 steps with the same random gradients:
    same update rule operation by operation; also the version counters move (the packed-weight caches key on them)."""
    from mdfnet_hip import optim as moptim
    torch.manual_seed(3)
    m_ref, m_hip = build_model(), build_model()
    m_ref.load_state_dict(seeded_sd); m_hip.load_state_dict(seeded_sd)
    m_ref, m_hip = m_ref.to(DEV), m_hip.to(DEV)
    opt_ref = torch.optim.Adam(m_ref.parameters(), lr=1e-3)
    bucket = ddp.FlatBucket(m_hip)
    opt_hip = moptim.FlatAdam(bucket, lr=1e-3)
    v0 = [p._version for p in bucket.params]
    for step in range(4):
        lr = 1e-3 * (1 - step / 8) ** 0.9                       # the poly schedule of train.py:34 moves lr between steps
        opt_ref.param_groups[0]["lr"] = lr
        opt_hip.param_groups[0]["lr"] = lr
        bucket.zero_grad()
        for pr, ph in zip(m_ref.parameters(), m_hip.parameters()):
            g = torch.randn_like(pr) * (10.0 ** float(torch.randint(-4, 1, (1,))))
            pr.grad = g.clone()
            ph.grad = g.clone()
        bucket.allreduce_gradients()
        opt_ref.step()
        opt_hip.step()
    assert all(p._version > v for p, v in zip(bucket.params, v0))
    worst = 0.0
    for (k, pr), ph in zip(m_ref.named_parameters(), m_hip.parameters()):
        err = float((pr - ph).abs().max() / pr.abs().max().clamp(min=1e-12))
        worst = max(worst, err)
        assert err <= 2e-6, (k, err)
    print(f"FlatAdam vs torch.optim.Adam after 4 steps: max relative parameter difference {worst:.2e}")


@pytest.mark.parametrize("a,bc", [(8, 8), (8, 16), (16, 8), (8, 4)])
@pytest.mark.parametrize("w", [16, 17, 63, 64, 65, 256, 257, 384, 511, 512, 513])
def test_wgrad_tap_packing_row_lengths(a, bc, w):
    """Few-channel weight gradients pack taps into the MFMA tile (shifts of dy in the rows, of x in the columns: wgrad_lds.hip, R > 0);
    the shifted row blocks sum over v - 1, so the tiles must reach one voxel past the row -- exercised at row lengths around every
    tile-width boundary, 2-D and 3-D, against torch's weight gradient on the CPU."""
    torch.manual_seed(a * 1000 + bc * 10 + w)
    for three_d in (False, True):
        if three_d:
            conv = nn.Conv3d(bc, a, 3, padding=1, bias=False)
            x = torch.randn(1, bc, 3, 4, w, requires_grad=True)
        else:
            conv = nn.Conv2d(bc, a, 3, padding=1, bias=False)
            x = torch.randn(2, bc, 5, w, requires_grad=True)
        y = conv(x)
        dy = torch.randn_like(y)
        y.backward(dy)
        if three_d:
            dw = train_ops.conv3d_wgrad(ops.to_ndhwc(dy.to(DEV)), ops.to_ndhwc(x.detach().to(DEV)), 1, tuple(conv.weight.shape))
        else:
            dw = train_ops.conv2d_wgrad(ops.to_nhwc(dy.to(DEV)), ops.to_nhwc(x.detach().to(DEV)), 3, 1, tuple(conv.weight.shape))
        assert _rel(dw, conv.weight.grad) < 3e-5, (three_d, _rel(dw, conv.weight.grad))


def test_fpn_head_algebra_kernels_vs_float64():
    """csrc/fpn_compose.hip on its own: the composed matrices and the map of their gradients back onto the seven parameters, against
    the same products in float64 (backbone.py:59-63 written out; train_ops.py:FPNHeadsComposedFn holds the torch form)."""
    import ctypes
    from mdfnet_hip import lib, check
    rng = np.random.RandomState(9)
    c2, c3, cm = 16, 32, 64
    f = lambda *s: T(rng.randn(*s).astype(np.float32))
    O2, O3, L2, b2, L3, b3 = f(c2, cm), f(c3, cm), f(cm, c2), f(cm), f(cm, c3), f(cm)
    dev = [t.to(DEV) for t in (O2, O3, L2, b2, L3, b3)]
    comp = torch.empty(c2 * c2 + c2 * c3 + c3 * c3 + 2 * c2 + c3, device=DEV)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib().mdf_fpn_compose_fwd(*[t.data_ptr() for t in dev], c2, c3, cm, comp.data_ptr(), st), "mdf_fpn_compose_fwd")
    d = [t.double() for t in (O2, O3, L2, b2, L3, b3)]
    exp = torch.cat([(d[0] @ d[2]).reshape(-1), (d[0] @ d[4]).reshape(-1), (d[1] @ d[4]).reshape(-1), d[0] @ d[3], d[0] @ d[5], d[1] @ d[5]])
    assert _rel(comp, exp) < 2e-6
    dA2, dB3, dA3, W2, W3 = f(c2, c2), f(c2, c3), f(c3, c3), f(c2, cm), f(c3, cm)
    s2, sc3, s3 = (T(rng.randn(n)) for n in (c2, c2, c3))          # float64, as mdf_bn_stats_fwd leaves the sums
    g = [t.to(DEV) for t in (dA2, dB3, dA3, W2, W3, s2, sc3, s3)]
    outs = [torch.empty(s, device=DEV) for s in ((c2, cm), (c3, cm), (cm, c2), (cm, c3), (cm,), (cm,))]
    check(lib().mdf_fpn_compose_bwd(*[t.data_ptr() for t in dev], *[t.data_ptr() for t in g], c2, c3, cm, *[t.data_ptr() for t in outs], st),
          "mdf_fpn_compose_bwd")
    dA2, dB3, dA3, W2, W3 = (t.double() for t in (dA2, dB3, dA3, W2, W3))
    s2f, sc3f, s3f = (t.float().double() for t in (s2, sc3, s3))
    exp = [W2 + dA2 @ d[2].t() + torch.outer(s2f, d[3]) + dB3 @ d[4].t() + torch.outer(sc3f, d[5]),
           W3 + dA3 @ d[4].t() + torch.outer(s3f, d[5]),
           d[0].t() @ dA2, d[0].t() @ dB3 + d[1].t() @ dA3, d[0].t() @ s2f, d[0].t() @ sc3f + d[1].t() @ s3f]
    for got, e in zip(outs, exp):
        assert _rel(got, e) < 3e-6
    # error convention: null pointer / bad shape -> MDF_EARG with a message, nothing launched
    assert lib().mdf_fpn_compose_fwd(None, *[t.data_ptr() for t in dev[1:]], c2, c3, cm, comp.data_ptr(), st) != 0
    assert b"null pointer" in lib().mdf_last_error()
    assert lib().mdf_fpn_compose_fwd(*[t.data_ptr() for t in dev], 0, c3, cm, comp.data_ptr(), st) != 0


@pytest.mark.parametrize("c,shape", [(8, (2, 5, 13, 37)), (16, (1, 9, 31, 70)), (8, (1, 1, 3, 5)), (16, (3, 2, 2, 2))])
def test_prob_conv_weight_gradient_on_the_vector_alus(c, shape):
    """dW of the `prob` conv (one output channel: wgrad.hip:wgrad_a1_valu_kernel, a thread per voxel and channel quad) against float64
    autograd of F.conv3d: odd sizes, volumes smaller than a block's stride, every border tap."""
    b, d, h, w = shape
    torch.manual_seed(c + d + w)
    x = torch.randn(b, c, d, h, w, dtype=torch.float64)
    wt = torch.randn(1, c, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(b, 1, d, h, w, dtype=torch.float64)
    F.conv3d(x, wt, padding=1).backward(dy)
    small = dy.float().permute(0, 2, 3, 4, 1).contiguous().to(DEV)           # [B,D,H,W,1]
    big = x.float().permute(0, 2, 3, 4, 1).contiguous().to(DEV)              # [B,D,H,W,C]
    dw = train_ops.conv3d_wgrad(small, big, 1, (1, c, 3, 3, 3))
    assert _rel(dw, wt.grad) < 2e-5
