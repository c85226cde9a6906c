"""GPU parity: MFMA 3-D conv layers and the full regularisers vs torch-CPU / reference goldens."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mdfnet_hip import ops, synth
from oracle import mvs_oracle as O
from modelutil import build_model

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"

COMBOS = [(32, 16, "s1"), (16, 16, "s1"), (32, 32, "s1"), (64, 64, "s1"), (16, 8, "s1"), (8, 8, "s1"),
          (16, 32, "s2"), (32, 64, "s2"), (8, 16, "s2"), (64, 32, "tr"), (32, 16, "tr"), (16, 8, "tr")]


@pytest.mark.parametrize("cin,cout,mode", COMBOS)
@pytest.mark.parametrize("shape", [(1, 4, 6, 8), (2, 5, 7, 9), (1, 8, 20, 36),
                                   (1, 1, 9, 37), (2, 2, 7, 33), (2, 3, 5, 21)])     # 1-3 planes: depth taps no voxel of a wave has are skipped (r05)
def test_conv3d_layer(cin, cout, mode, shape):
    b, d, h, w = shape
    rng = np.random.RandomState(cin * 131 + cout + d)
    x = T(rng.randn(b, cin, d, h, w).astype(np.float32))
    tr = mode == "tr"
    wt = T((rng.randn(*((cin, cout) if tr else (cout, cin)), 3, 3, 3) / np.sqrt(27 * cin)).astype(np.float32))
    alpha = T(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    beta = T(rng.uniform(-0.2, 0.2, cout).astype(np.float32))
    if tr:
        ref = F.conv_transpose3d(x, wt, None, 2, 1, 1)
    else:
        ref = F.conv3d(x, wt, None, 2 if mode == "s2" else 1, 1)
    res = T(rng.randn(*ref.shape).astype(np.float32))
    exp = F.relu(ref * alpha.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1)) + res
    wp = ops.pack_conv3d_weight(wt.to(DEV), tr)
    y = ops.conv3d_ndhwc(ops.to_ndhwc(x.to(DEV)), wp, cin, cout, 2 if mode != "s1" else 1, tr, alpha.to(DEV), beta.to(DEV),
                         True, ops.to_ndhwc(res.to(DEV)))
    got = ops.from_ndhwc(y).cpu()
    assert got.shape == exp.shape
    # fp32 fma chain over K = 27*cin terms, different summation order than oneDNN
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=1e-4, atol=2e-5)
    # no BN / relu / residual variant (raw conv)
    y2 = ops.conv3d_ndhwc(ops.to_ndhwc(x.to(DEV)), wp, cin, cout, 2 if mode != "s1" else 1, tr)
    np.testing.assert_allclose(ops.from_ndhwc(y2).cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("cin,cout,stride,shape", [(32, 32, 1, (1, 7, 67, 151)), (32, 32, 1, (2, 3, 90, 131)), (16, 32, 2, (1, 18, 160, 211)),
                                                   (8, 16, 2, (1, 12, 301, 399)), (8, 16, 2, (2, 9, 210, 300))])
def test_conv3d_layer_lds_weights_kernel_more_tiles_than_waves(cin, cout, stride, shape, monkeypatch):
    """conv3d_wlds_kernel (the layer's packed weight set in LDS, one persistent 16-wave block per CU): volumes with more m-tiles than the
    4096 waves of the grid -- every wave walks several tiles, the XCD chunks have remainders, the last tile is ragged -- against torch,
    and against conv3d_kernel (MDF_CONV3D_WLDS=0 / MDF_CONV3D_WLDS8=0) at the re-association level."""
    b, d, h, w = shape
    g = torch.Generator().manual_seed(cin * 10 + cout + w)
    x = torch.randn(b, cin, d, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) / np.sqrt(27 * cin)
    alpha, beta = torch.rand(cout, generator=g) + 0.5, torch.rand(cout, generator=g) * 0.4 - 0.2
    ref = F.conv3d(x, wt, None, stride, 1)
    assert ref.shape[2] * ref.shape[3] * ref.shape[4] * b > 4096 * 16 * (2 if cin == 8 else 1) and (stride == 2 or b * d * h * w < 150000)
    res = torch.randn(ref.shape, generator=g)
    exp = F.relu(ref * alpha.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1)) + res
    wp = ops.pack_conv3d_weight(wt.to(DEV), False)
    xd, rd = ops.to_ndhwc(x.to(DEV)), ops.to_ndhwc(res.to(DEV))
    y = ops.conv3d_ndhwc(xd, wp, cin, cout, stride, False, alpha.to(DEV), beta.to(DEV), True, rd)
    np.testing.assert_allclose(ops.from_ndhwc(y).cpu().numpy(), exp.numpy(), rtol=1e-4, atol=2e-5)
    monkeypatch.setenv("MDF_CONV3D_WLDS", "0")
    monkeypatch.setenv("MDF_CONV3D_WLDS8", "0")
    y0 = ops.conv3d_ndhwc(xd, wp, cin, cout, stride, False, alpha.to(DEV), beta.to(DEV), True, rd)
    assert (y - y0).abs().max().item() < 2e-5      # (same MFMAs; the split-K form sums its four partial chains last)


@pytest.mark.parametrize("cin,cout", [(32, 16), (16, 16), (32, 32), (16, 8), (8, 8), (16, 32), (8, 16)])
@pytest.mark.parametrize("shape", [(1, 6, 130, 201), (2, 5, 125, 131), (1, 3, 260, 197), (1, 90, 5, 401), (1, 2, 3, 25001), (1, 1, 400, 400)])
def test_conv3d_layer_lds_kernels(cin, cout, shape):
    """Volumes of >= 150 000 voxels take the LDS-staged kernels (conv_lds.hip; Cout = 8 in the w-phase form): ragged tiles
    in h and w (odd widths exercise the half-filled last w-phase pair / Winograd tile), D not a multiple of the depth chunk,
    batch 2, fewer rows than one tile (5, 3), a single plane."""
    b, d, h, w = shape
    assert b * d * h * w >= 150000
    g = torch.Generator().manual_seed(cin * 100 + cout + w)
    x = torch.randn(b, cin, d, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) / np.sqrt(27 * cin)
    alpha = torch.rand(cout, generator=g) + 0.5
    beta = torch.rand(cout, generator=g) * 0.4 - 0.2
    ref = F.conv3d(x, wt, None, 1, 1)
    res = torch.randn(ref.shape, generator=g)
    exp = F.relu(ref * alpha.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1)) + res
    wp = ops.pack_conv3d_weight(wt.to(DEV), False)
    xd = ops.to_ndhwc(x.to(DEV))
    y = ops.conv3d_ndhwc(xd, wp, cin, cout, 1, False, alpha.to(DEV), beta.to(DEV), True, ops.to_ndhwc(res.to(DEV)))
    np.testing.assert_allclose(ops.from_ndhwc(y).cpu().numpy(), exp.numpy(), rtol=1e-4, atol=2e-5)
    y2 = ops.conv3d_ndhwc(xd, wp, cin, cout, 1, False)
    np.testing.assert_allclose(ops.from_ndhwc(y2).cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("cin,cout", [(32, 16), (16, 16)])
@pytest.mark.parametrize("shape", [(1, 6, 130, 201), (2, 5, 125, 131), (1, 1, 400, 400), (1, 2, 3, 25001), (1, 90, 5, 401), (1, 7, 9, 33),
                                   (3, 4, 17, 40), (1, 13, 8, 32)])
def test_input_stationary_winograd_kernel_equals_the_output_stationary_one(cin, cout, shape, monkeypatch):
    """wino3d.hip (each input plane transformed once and multiplied into three rotating accumulator sets, stream-K over tile x depth)
    accumulates every output in the order of conv_lds.hip's Winograd form (kd, cin chunk, k): the results are bit-identical, with and
    without the epilogue (BN, ReLU, residual).  Shapes: ragged tiles, batch 2 and 3 (tile changes inside a block's range), a single
    plane, two planes, more planes than a block's share, a volume smaller than one tile's 32 columns."""
    b, d, h, w = shape
    g = torch.Generator().manual_seed(cin * 7 + cout + d * h + w)
    x = ops.to_ndhwc(torch.randn(b, cin, d, h, w, generator=g).to(DEV))
    wp = ops.pack_conv3d_weight((torch.randn(cout, cin, 3, 3, 3, generator=g) / np.sqrt(27 * cin)).to(DEV), False)
    alpha, beta = (torch.rand(cout, generator=g) + 0.5).to(DEV), (torch.rand(cout, generator=g) * 0.4 - 0.2).to(DEV)
    res = torch.randn(b, d, h, w, cout, generator=g).to(DEV)
    monkeypatch.setenv("MDF_CONV_LDS_MIN_VOXELS", "0")     # small volumes take the LDS kernels too
    out = {}
    for v in ("0", "1"):
        monkeypatch.setenv("MDF_CONV_WINO3D", v)
        out[v] = (ops.conv3d_ndhwc(x, wp, cin, cout, 1, False, alpha, beta, True, res), ops.conv3d_ndhwc(x, wp, cin, cout, 1, False))
    assert torch.equal(out["0"][0], out["1"][0]) and torch.equal(out["0"][1], out["1"][1])
    assert not torch.isnan(out["1"][0]).any()


@pytest.mark.parametrize("cin,cout", [(8, 16), (16, 32)])
@pytest.mark.parametrize("shape", [(1, 25, 203, 261), (2, 11, 190, 301), (1, 6, 401, 799), (1, 2, 700, 900), (1, 3, 640, 1000)])
def test_conv3d_stride2_layer_large_volumes(cin, cout, shape):
    """Stride-2 down-sampling layers at the sizes the regularisers run them (>= 150 000 output voxels): odd and even extents in
    d, h and w (the last output plane / row / column reads the zero halo or not), batch 2, a single output plane, two."""
    b, d, h, w = shape
    do, ho, wo = (d - 1) // 2 + 1, (h - 1) // 2 + 1, (w - 1) // 2 + 1
    assert b * do * ho * wo >= 150000
    g = torch.Generator().manual_seed(cin * 100 + cout + w)
    x = torch.randn(b, cin, d, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) / np.sqrt(27 * cin)
    alpha = torch.rand(cout, generator=g) + 0.5
    beta = torch.rand(cout, generator=g) * 0.4 - 0.2
    ref = F.conv3d(x, wt, None, 2, 1)
    assert tuple(ref.shape) == (b, cout, do, ho, wo)
    res = torch.randn(ref.shape, generator=g)
    exp = F.relu(ref * alpha.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1)) + res
    wp = ops.pack_conv3d_weight(wt.to(DEV), False)
    xd = ops.to_ndhwc(x.to(DEV))
    y = ops.conv3d_ndhwc(xd, wp, cin, cout, 2, False, alpha.to(DEV), beta.to(DEV), True, ops.to_ndhwc(res.to(DEV)))
    np.testing.assert_allclose(ops.from_ndhwc(y).cpu().numpy(), exp.numpy(), rtol=1e-4, atol=2e-5)
    y2 = ops.conv3d_ndhwc(xd, wp, cin, cout, 2, False)
    np.testing.assert_allclose(ops.from_ndhwc(y2).cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("cin,cout", [(32, 16), (16, 8), (64, 32)])
@pytest.mark.parametrize("shape", [(1, 6, 130, 201), (2, 4, 101, 187)])
def test_conv_transpose3d_layer_large_volumes(cin, cout, shape):
    """ConvTranspose3d(k3,s2,p1,op1)+BN+ReLU+skip on >= 150 000 input voxels (big-tile launch of conv3d_kernel, prefetched
    residual); ragged tiles, odd widths, batch 2."""
    b, d, h, w = shape
    assert b * d * h * w >= 150000
    g = torch.Generator().manual_seed(cin + cout + w)
    x = torch.randn(b, cin, d, h, w, generator=g)
    wt = torch.randn(cin, cout, 3, 3, 3, generator=g) / np.sqrt(27 * cin / 8)
    alpha = torch.rand(cout, generator=g) + 0.5
    beta = torch.rand(cout, generator=g) * 0.4 - 0.2
    ref = F.conv_transpose3d(x, wt, None, 2, 1, 1)
    res = torch.randn(ref.shape, generator=g)
    exp = F.relu(ref * alpha.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1)) + res
    wp = ops.pack_conv3d_weight(wt.to(DEV), True)
    xd = ops.to_ndhwc(x.to(DEV))
    y = ops.conv3d_ndhwc(xd, wp, cin, cout, 2, True, alpha.to(DEV), beta.to(DEV), True, ops.to_ndhwc(res.to(DEV)))
    np.testing.assert_allclose(ops.from_ndhwc(y).cpu().numpy(), exp.numpy(), rtol=1e-4, atol=2e-5)
    y2 = ops.conv3d_ndhwc(xd, wp, cin, cout, 2, True)
    np.testing.assert_allclose(ops.from_ndhwc(y2).cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("stage", [0, 1, 2])
def test_regulariser_vs_reference_golden(golden, seeded_sd, stage):
    g = golden("ops.npz")
    model = build_model()
    model.load_state_dict(seeded_sd)
    model.eval().to(DEV)
    cost = T(g[f"agg{stage}_cost"]).to(DEV)
    hyp = T(g[f"agg{stage}_hyp"]).to(DEV)
    prob, depth = model.Regular[stage](cost, hyp)
    np.testing.assert_allclose(prob.cpu().numpy(), g[f"reg{stage}_prob"], rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(depth.cpu().numpy(), g[f"reg{stage}_depth"], rtol=0, atol=5e-3)
    prob2 = model.Regular[stage](cost)
    assert torch.equal(prob2, prob)
    # the fused soft-argmin (what CoreNet uses when both slots are built-in) == the standalone Regress slot on the same prob
    assert torch.equal(model.Depth_regress(prob, hyp), depth)
    # standalone regress slot on the golden prob
    d2 = model.Depth_regress(T(g[f"reg{stage}_prob"]).to(DEV), hyp)
    np.testing.assert_allclose(d2.cpu().numpy(), g[f"reg{stage}_depth"], rtol=0, atol=3e-4)


def test_train_mode_runs_the_hip_training_kernels_with_autograd(seeded_sd):
    """model.train() on a GPU puts the regulariser's forward + backward on the hand-written training kernels
    (mdfnet_hip/train_ops.py:RegulariserTrainFn; batch-statistics BN) behind autograd -- an explicit mode, not a fallback."""
    from mdfnet_hip import ops as _ops
    _ops.count_begin()
    model = build_model()
    model.load_state_dict(seeded_sd)
    model.train().to(DEV)
    x = torch.randn(2, 16, 16, 16, 16, device=DEV, requires_grad=True)
    prob = model.Regular[1](x)
    assert prob.shape == (2, 16, 16, 16) and prob.requires_grad
    prob.sum().backward()
    calls = _ops.count_end()
    assert x.grad is not None and model.Regular[1].prob.weight.grad is not None
    assert calls.get("mdf_conv3d_wgrad_partial", 0) + calls.get("mdf_wgrad_batch_flush", 0) > 0 and calls.get("mdf_wgrad_sum_batch", 0) == 1 and calls.get("mdf_bn_relu_bwd", 0) > 0, calls


@pytest.mark.parametrize("cin,D,h,w", [(8, 8, 37, 53), (16, 48, 20, 70), (8, 24, 16, 16), (16, 3, 9, 65), (8, 1, 5, 7), (8, 60, 6, 6)])
def test_prob_head_routes_agree_with_conv3d_softmax(cin, D, h, w):
    """`prob` conv + softmax(D) + soft-argmin (regular.py:66-69 + regress.py:5-7): the partial-sum MFMA route and the direct
    kernel against torch's conv3d/softmax on the CPU, ragged tiles and D = 1 / 3 / >48 included."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(cin * 1000 + D)
    x = torch.randn(2, cin, D, h, w, generator=g)
    wt = torch.randn(1, cin, 3, 3, 3, generator=g) * 0.3
    hyp = torch.sort(torch.rand(2, D, h, w, generator=g) * 500 + 400, dim=1)[0]
    ref_p = F.softmax(F.conv3d(x, wt, padding=1).squeeze(1), dim=1)
    ref_d = (ref_p * hyp).sum(1)
    xd = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV)
    for direct in (False, True):
        p, d = ops.prob_head(xd, wt.to(DEV), hyp.to(DEV), direct=direct)
        np.testing.assert_allclose(p.cpu().numpy(), ref_p.numpy(), rtol=2e-4, atol=1e-7, err_msg=f"direct={direct}")
        np.testing.assert_allclose(d.cpu().numpy(), ref_d.numpy(), rtol=1e-5, err_msg=f"direct={direct}")
        assert torch.allclose(p.sum(1), torch.ones_like(p[:, 0]), atol=1e-5)


@pytest.mark.parametrize("cin,D,h,w", [(8, 8, 37, 53), (16, 24, 21, 130), (8, 1, 5, 7), (16, 3, 9, 65), (8, 60, 6, 70)])
def test_prob_head_in_one_launch_equals_the_two_launch_route(cin, D, h, w, monkeypatch):
    """mdf_prob_fused_fwd (partials kept in registers while the block walks the depth axis) against mdf_conv2d_fwd +
    mdf_prob_from_partials_fwd: same MFMA step, same combine / softmax / soft-argmin order, so bit for bit -- ragged tiles,
    D = 1 / 3 / > 48, per-pixel and per-plane hypotheses, with and without the depth output."""
    g = torch.Generator().manual_seed(cin * 77 + D)
    x = torch.randn(2, D, h, w, cin, generator=g).to(DEV)
    wt = (torch.randn(1, cin, 3, 3, 3, generator=g) * 0.3).to(DEV)
    hyps = [torch.sort(torch.rand(2, D, h, w, generator=g) * 500 + 400, dim=1)[0].to(DEV), (torch.rand(2, D, 1, 1, generator=g) * 500 + 400).to(DEV), None]
    for hyp in hyps:
        monkeypatch.setattr(ops, "PROB_FUSED_MIN_TILES", 0)
        ops.count_begin()
        got = ops.prob_head(x, wt, hyp)
        assert ops.count_end().get("mdf_prob_fused_fwd", 0) == 1
        monkeypatch.setattr(ops, "PROB_FUSED_MIN_TILES", 1 << 30)
        exp = ops.prob_head(x, wt, hyp)
        if hyp is None:
            assert torch.equal(got, exp)
        else:
            assert torch.equal(got[0], exp[0]) and torch.equal(got[1], exp[1])


@pytest.mark.parametrize("cin,cout", [(16, 8), (32, 16), (64, 32)])
@pytest.mark.parametrize("shape", [(1, 5, 37, 53), (2, 3, 20, 70), (1, 1, 9, 17), (1, 6, 101, 119)])     # the last: more m-tiles than the persistent form has waves
def test_transposed_conv_all_classes_kernel_equals_the_per_class_kernel(cin, cout, shape, monkeypatch):
    """convtr_all_kernel (every input fragment fetched once, fed to all four (pd, ph) parity classes) against
    conv3d_kernel<kTr> (one block per class): the taps of a class are accumulated in the same order, so bit for bit --
    ragged tiles, single-plane volume, with the skip connection and the folded BatchNorm + ReLU."""
    b, d, h, w = shape
    g = torch.Generator().manual_seed(cin + d * h)
    x = torch.randn(b, d, h, w, cin, generator=g).to(DEV)
    wt = (torch.randn(cin, cout, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).to(DEV)
    wp = ops.pack_conv3d_weight(wt, True)
    al, be = (torch.rand(cout, generator=g) + 0.5).to(DEV), (torch.randn(cout, generator=g) * 0.1).to(DEV)
    res = torch.randn(b, 2 * d, 2 * h, 2 * w, cout, generator=g).to(DEV)
    for kw in (dict(), dict(alpha=al, beta=be, relu=True, res=res)):
        monkeypatch.setenv("MDF_CONVTR_ALL_MIN_VOXELS", "-1")
        exp = ops.conv3d_ndhwc(x, wp, cin, cout, 2, True, **kw)
        monkeypatch.setenv("MDF_CONVTR_ALL_MIN_VOXELS", "0")
        # MDF_CONVTR_NS (64 -> 32 only): 1 = every n-tile of an m-tile in one wave, 2 / 4 = dealt out to two / four waves (small volumes)
        monkeypatch.setenv("MDF_CONVTR_CLS", "0")
        for ns in (("1", "2", "4") if cin == 64 else ("1",)):
            monkeypatch.setenv("MDF_CONVTR_NS", ns)
            got = ops.conv3d_ndhwc(x, wp, cin, cout, 2, True, **kw)
            assert torch.equal(got, exp), ns
        monkeypatch.delenv("MDF_CONVTR_NS")
        # MDF_CONVTR_WLDS (16 -> 8, 32 -> 16): the weight set in LDS, one persistent block per CU walking the m-tiles
        for wl in (("1", "2") if cin < 64 else ()):
            monkeypatch.setenv("MDF_CONVTR_WLDS", wl)
            got = ops.conv3d_ndhwc(x, wp, cin, cout, 2, True, **kw)
            assert torch.equal(got, exp), ("wlds", wl)
        monkeypatch.delenv("MDF_CONVTR_WLDS", raising=False)
        # MDF_CONVTR_CLS (64 -> 32): one parity class per persistent block, the class's tap slots in LDS
        for cl in (("1", "2") if cin == 64 else ()):
            monkeypatch.setenv("MDF_CONVTR_CLS", cl)
            got = ops.conv3d_ndhwc(x, wp, cin, cout, 2, True, **kw)
            assert torch.equal(got, exp), ("cls", cl)
        monkeypatch.delenv("MDF_CONVTR_CLS")
