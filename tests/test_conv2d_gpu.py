"""GPU parity: fused 2-D conv layers (feature pyramid / refinement) vs torch-CPU and the reference goldens."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mdfnet_hip import ops, synth
from modelutil import build_model

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"

# (cin, cout, k, stride) = every 2-D layer shape of FPN_4Scales and RefineNet2
COMBOS = [(3, 8, 3, 1), (8, 8, 3, 1), (8, 16, 5, 2), (16, 16, 3, 1), (16, 32, 5, 2), (32, 32, 3, 1), (32, 64, 5, 2),
          (64, 64, 3, 1), (64, 64, 1, 1), (32, 64, 1, 1), (64, 32, 1, 1), (16, 64, 1, 1), (64, 16, 1, 1),
          (1, 8, 3, 1), (8, 32, 3, 1), (8, 1, 3, 1),
          (16, 4, 3, 1), (8, 4, 3, 1)]       # prob-head partial sums: 4 outputs along w per MFMA column (w-phase form)


@pytest.mark.parametrize("cin,cout,k,stride", COMBOS)
@pytest.mark.parametrize("shape", [(1, 8, 16), (2, 13, 37), (1, 40, 136), (3, 9, 131)])
def test_conv2d_layer(cin, cout, k, stride, shape):
    b, h, w = shape
    rng = np.random.RandomState(cin * 7 + cout * 3 + k + h)
    x = T(rng.randn(b, cin, h, w).astype(np.float32))
    wt = T((rng.randn(cout, cin, k, k) / np.sqrt(k * k * cin)).astype(np.float32))
    alpha = T(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    beta = T(rng.uniform(-0.2, 0.2, cout).astype(np.float32))
    ref = F.conv2d(x, wt, None, stride, (k - 1) // 2)
    res = T(rng.randn(*ref.shape).astype(np.float32))
    exp = res + 0.1 * F.relu(ref * alpha.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1))
    wp = ops.pack_conv2d_weight(wt.to(DEV))
    y = ops.conv2d_nhwc(ops.to_nhwc(x.to(DEV)), wp, cin, cout, k, stride, alpha.to(DEV), beta.to(DEV), True,
                        ops.to_nhwc(res.to(DEV)), 0.1)
    got = ops.from_nhwc(y).cpu()
    assert got.shape == exp.shape
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=1e-4, atol=2e-5)
    y2 = ops.conv2d_nhwc(ops.to_nhwc(x.to(DEV)), wp, cin, cout, k, stride)      # raw conv, no epilogue
    np.testing.assert_allclose(ops.from_nhwc(y2).cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


def test_lateral_conv_with_fused_upsample_add():
    rng = np.random.RandomState(5)
    x = T(rng.randn(2, 32, 12, 20).astype(np.float32))
    top = T(rng.randn(2, 64, 6, 10).astype(np.float32))
    wt = T((rng.randn(64, 32, 1, 1) / 6).astype(np.float32))
    bias = T(rng.randn(64).astype(np.float32))
    exp = F.interpolate(top, scale_factor=2.0, mode="bilinear", align_corners=False) + F.conv2d(x, wt, bias)
    y = ops.conv2d_nhwc(ops.to_nhwc(x.to(DEV)), ops.pack_conv2d_weight(wt.to(DEV)), 32, 64, 1, 1, None, bias.to(DEV),
                        res_up=ops.to_nhwc(top.to(DEV)))
    np.testing.assert_allclose(ops.from_nhwc(y).cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 16), (32, 32), (32, 64), (16, 32)])
def test_streaming_1x1_kernel_equals_the_lds_kernel(cin, cout, monkeypatch):
    """conv1x1.hip against the LDS-tiled kernel for the same layer (MDF_CONV1X1=0): same MFMA order and epilogue, so
    bit for bit; ragged pixel count (the last run of tiles is partly empty), upsample-add, scale/shift/ReLU, residual."""
    rng = np.random.RandomState(cin + cout)
    b, h, w = 3, 26, 42                                              # 3276 pixels: not a multiple of a run (128 / 64 / 32 px)
    x = ops.to_nhwc(T(rng.randn(b, cin, h, w).astype(np.float32)).to(DEV))
    top = ops.to_nhwc(T(rng.randn(b, cout, h // 2, w // 2).astype(np.float32)).to(DEV))
    res = ops.to_nhwc(T(rng.randn(b, cout, h, w).astype(np.float32)).to(DEV))
    wp = ops.pack_conv2d_weight(T((rng.randn(cout, cin, 1, 1) / np.sqrt(cin)).astype(np.float32)).to(DEV))
    al, be = T(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(DEV), T(rng.randn(cout).astype(np.float32)).to(DEV)
    cases = [dict(alpha=None, beta=be, res_up=top), dict(alpha=al, beta=be, relu=True, res=res, res_scale=0.1), dict()]
    for kw in cases:
        monkeypatch.setenv("MDF_CONV1X1", "1")
        got = ops.conv2d_nhwc(x, wp, cin, cout, 1, 1, **kw)
        monkeypatch.setenv("MDF_CONV1X1", "0")
        exp = ops.conv2d_nhwc(x, wp, cin, cout, 1, 1, **kw)
        assert torch.equal(got, exp), kw.keys()


@pytest.mark.parametrize("shape", [(2, 21, 47), (1, 8, 30), (1, 5, 3), (3, 16, 61), (1, 64, 90)])
def test_refine_tail_in_one_launch(shape):
    """Conv2d(8,32) -> PixelShuffle(2) -> Conv2d(8,1) -> lo + y*span (refine.py:18-20,42-44) as mdf_refine_tail_fwd against torch on
    the CPU and against the three-launch route; tiles of 8 x 30 low-resolution pixels: exact, ragged and smaller-than-a-tile maps."""
    b, h, w = shape
    rng = np.random.RandomState(h * 100 + w)
    x = T(rng.randn(b, 8, h, w).astype(np.float32))
    w1 = T((rng.randn(32, 8, 3, 3) / np.sqrt(72)).astype(np.float32))
    w2 = T((rng.randn(1, 8, 3, 3) / np.sqrt(72)).astype(np.float32))
    lo = T(rng.uniform(400, 500, b).astype(np.float32))
    span = T(rng.uniform(300, 600, b).astype(np.float32))
    raw = F.conv2d(F.pixel_shuffle(F.conv2d(x, w1, None, 1, 1), 2), w2, None, 1, 1).squeeze(1)
    exp = lo.view(b, 1, 1) + raw * span.view(b, 1, 1)
    xd = ops.to_nhwc(x.to(DEV))
    wp = ops.pack_conv2d_weight(ops.shuffle2_rows(w1.to(DEV)))
    got = ops.refine_tail(xd, wp, w2.to(DEV), lo.to(DEV), span.to(DEV))
    assert got.shape == exp.shape
    np.testing.assert_allclose(got.cpu().numpy(), exp.numpy(), rtol=0, atol=span.max().item() * 4e-6)
    got_raw = ops.refine_tail(xd, wp, w2.to(DEV))
    np.testing.assert_allclose(got_raw.cpu().numpy(), raw.numpy(), rtol=0, atol=4e-6)
    mid = ops.conv2d_nhwc(xd, wp, 8, 32, 3, 1, pixel_shuffle2=True)                       # the three-launch route
    two = ops.conv2d_nhwc(mid, ops.pack_conv2d_weight(w2.to(DEV)), 8, 1, 3, 1).squeeze(-1)
    np.testing.assert_allclose(got_raw.cpu().numpy(), two.cpu().numpy(), rtol=0, atol=4e-6)


@pytest.mark.parametrize("chans", [(16, 32), (8, 16)])
@pytest.mark.parametrize("shape", [(2, 26, 70), (1, 8, 32), (1, 64, 132), (3, 2, 2), (2, 36, 196)])
def test_k5s2_as_winograd_over_the_parity_images(shape, chans, monkeypatch):
    """Conv2d(16, 32, k5, s2, p2) / Conv2d(8, 16, k5, s2, p2) run as a Winograd 3x3 conv over the four parity images of their input
    (wino2d.hip S2D; conv_lds.hip LdsConvParams::s2d for 16 -> 32 with MDF_CONV_WINO2D=0) against the direct 25-tap kernel and torch: tiles
    that hang over the right / bottom edge, a map smaller than a tile, several tiles per block."""
    b, h, w = shape
    cin, cout = chans
    rng = np.random.RandomState(h * 7 + w)
    x = T(rng.randn(b, cin, h, w).astype(np.float32))
    wt = T((rng.randn(cout, cin, 5, 5) / 20).astype(np.float32))
    al, be = T(rng.uniform(0.5, 1.5, cout).astype(np.float32)), T(rng.randn(cout).astype(np.float32) * 0.1)
    exp = F.relu(F.conv2d(x, wt, None, 2, 2) * al.view(1, -1, 1, 1) + be.view(1, -1, 1, 1))
    xd, wp = ops.to_nhwc(x.to(DEV)), ops.pack_conv2d_weight(wt.to(DEV))
    run = lambda: ops.from_nhwc(ops.conv2d_nhwc(xd, wp, cin, cout, 5, 2, al.to(DEV), be.to(DEV), True)).cpu()
    from mdfnet_hip import lib
    monkeypatch.setenv("MDF_CONV_K5_WINOGRAD", "1")
    got = run()
    assert lib().mdf_last_launch().decode().startswith("wino2d_kernel")
    monkeypatch.setenv("MDF_CONV_WINO2D", "0")
    old = run()                                             # conv_lds.hip's form (16 -> 32) / the direct kernel (8 -> 16)
    monkeypatch.setenv("MDF_CONV_K5_WINOGRAD", "0")
    direct = run()
    assert lib().mdf_last_launch().decode().startswith("conv_lds_kernel")
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(got.numpy(), direct.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(got.numpy(), old.numpy(), rtol=1e-4, atol=2e-5)
    if chans == (16, 32):
        assert torch.equal(got, old)        # same fragments, same MFMA order per accumulator: the two Winograd kernels agree bit for bit


def test_backbone_and_refine_vs_reference_golden(golden, seeded_sd):
    g = golden("ops.npz")
    m = build_model()
    m.load_state_dict(seeded_sd)
    m.eval().to(DEV)
    imgs, _, _, dr = synth.make_scene(96, 64, 3, batch=2, rot_deg=4.0, seed=5)
    with torch.no_grad():
        f8, f4, f2 = m.Backbone(imgs[:, 0].to(DEV))
        r = m.Refine(T(g["reg2_depth"]).to(DEV), dr.to(DEV))
    for a, k in ((f8, "fpn_f8"), (f4, "fpn_f4"), (f2, "fpn_f2")):
        assert a.shape == g[k].shape
        np.testing.assert_allclose(a.cpu().numpy(), g[k], rtol=1e-4, atol=2e-5)
        assert a.permute(0, 2, 3, 1).is_contiguous()        # NHWC memory for the aggregation kernel
    np.testing.assert_allclose(r.cpu().numpy(), g["refine_out"], rtol=0, atol=2e-3)   # mm; fp32 ulp at 600 = 6e-5, gain ~4


@pytest.mark.parametrize("shape", [(1, 9, 21), (2, 16, 70)])
def test_conv_with_fused_pixel_shuffle(shape):
    """refine.py:18-19: Conv2d(8 -> 32) followed by nn.PixelShuffle(2), written directly as [B,2H,2W,8] by the conv epilogue."""
    b, h, w = shape
    g = torch.Generator().manual_seed(h * w)
    x = torch.randn(b, 8, h, w, generator=g)
    wt = torch.randn(32, 8, 3, 3, generator=g) * 0.2
    exp = F.pixel_shuffle(F.conv2d(x, wt, None, 1, 1), 2)                      # [B,8,2h,2w]
    wp = ops.pack_conv2d_weight(ops.shuffle2_rows(wt.to(DEV)))
    y = ops.conv2d_nhwc(ops.to_nhwc(x.to(DEV)), wp, 8, 32, 3, 1, pixel_shuffle2=True)
    assert y.shape == (b, 2 * h, 2 * w, 8)
    np.testing.assert_allclose(y.permute(0, 3, 1, 2).cpu().numpy(), exp.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("shape", [(1, 8, 16), (2, 13, 37), (1, 40, 136), (3, 9, 131), (2, 64, 62), (1, 65, 63), (1, 200, 190), (5, 296, 400), (1, 3, 1), (2, 1, 70)])
def test_conv2d_pair_equals_the_two_layers(shape, monkeypatch):
    """mdf_conv2d_pair_fwd (conv_pair.hip: FPN_4Scales.conv01, both full-resolution ConvBNReLU layers in one launch) against the two
    single-layer launches it replaces and against torch on the CPU -- strips narrower / wider than 62 pixels, row counts around the
    segment boundaries, maps of one row / one column.  The MFMA form (MDF_CONV_PAIR_VALU=0: rolling LDS windows, same packed weights,
    tap order and epilogue arithmetic as the single-layer kernels) is BIT-IDENTICAL to them; the default vector-ALU form (every
    multiply-add a useful one, sums in tap order per lane) agrees to fp32 rounding."""
    n, h, w = shape
    rng = np.random.RandomState(n * 100 + h + w)
    x = T(rng.rand(n, 3, h, w).astype(np.float32))
    w1 = T((rng.randn(8, 3, 3, 3) / np.sqrt(27)).astype(np.float32))
    w2 = T((rng.randn(8, 8, 3, 3) / np.sqrt(72)).astype(np.float32))
    a1, b1 = T(rng.uniform(0.5, 1.5, 8).astype(np.float32)), T(rng.uniform(-0.3, 0.3, 8).astype(np.float32))
    a2, b2 = T(rng.uniform(0.5, 1.5, 8).astype(np.float32)), T(rng.uniform(-0.3, 0.3, 8).astype(np.float32))
    xd = x.to(DEV)
    wp1, wp2 = ops.pack_conv2d_weight(w1.to(DEV)), ops.pack_conv2d_weight(w2.to(DEV))
    t1 = ops.conv2d_nhwc(xd, wp1, 3, 8, 3, 1, a1.to(DEV), b1.to(DEV), True, planar_in=True)
    two = ops.conv2d_nhwc(t1, wp2, 8, 8, 3, 1, a2.to(DEV), b2.to(DEV), True)
    from mdfnet_hip import lib
    one = ops.conv2d_pair_planar(xd, wp1, a1.to(DEV), b1.to(DEV), wp2, a2.to(DEV), b2.to(DEV))
    assert lib().mdf_last_launch().decode().startswith("conv_pair_valu_kernel")
    monkeypatch.setenv("MDF_CONV_PAIR_VALU", "0")
    mfma = ops.conv2d_pair_planar(xd, wp1, a1.to(DEV), b1.to(DEV), wp2, a2.to(DEV), b2.to(DEV))
    assert lib().mdf_last_launch().decode().startswith("conv_pair_kernel")
    assert one.shape == two.shape == (n, h, w, 8) and not bool(torch.isnan(one).any())
    assert torch.equal(mfma, two), float((mfma - two).abs().max())
    np.testing.assert_allclose(one.cpu().numpy(), two.cpu().numpy(), rtol=2e-5, atol=2e-6)
    m1 = F.relu(F.conv2d(x, w1, None, 1, 1) * a1.view(1, -1, 1, 1) + b1.view(1, -1, 1, 1))
    exp = F.relu(F.conv2d(m1, w2, None, 1, 1) * a2.view(1, -1, 1, 1) + b2.view(1, -1, 1, 1))
    np.testing.assert_allclose(ops.from_nhwc(one).cpu().numpy(), exp.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("cin", [16, 32, 64])
@pytest.mark.parametrize("shape", [(1, 130, 201), (2, 125, 131), (1, 8, 32), (1, 3, 5), (3, 17, 40), (1, 61, 700)])
def test_wino2d_kernel_equals_the_conv_lds_winograd_form(cin, shape, monkeypatch):
    """wino2d.hip (pinned accumulators, lane-constant fill offsets, one transform cluster per chunk, tiles a whole step ahead) keeps
    conv_lds.hip's accumulation order per output: bit-identical with and without the epilogue (BN, ReLU, scaled residual), over ragged
    tiles, batch 2 / 3 (image changes inside a block's run of tiles), images smaller than one tile.  16 channels: MDF_WINO2D_16=1 (that
    layer stays on conv_lds.hip's form by default, it is memory-bound)."""
    b, h, w = shape
    g = torch.Generator().manual_seed(cin * 3 + h * w)
    x = ops.to_nhwc(torch.randn(b, cin, h, w, generator=g).to(DEV))
    wp = ops.pack_conv2d_weight((torch.randn(cin, cin, 3, 3, generator=g) / np.sqrt(9 * cin)).to(DEV))
    alpha, beta = (torch.rand(cin, generator=g) + 0.5).to(DEV), (torch.rand(cin, generator=g) * 0.4 - 0.2).to(DEV)
    res = torch.randn(b, h, w, cin, generator=g).to(DEV)
    monkeypatch.setenv("MDF_WINO2D_16", "1")
    out = {}
    for v in ("0", "1"):
        monkeypatch.setenv("MDF_CONV_WINO2D", v)
        out[v] = (ops.conv2d_nhwc(x, wp, cin, cin, 3, 1, alpha, beta, True, res, 0.1), ops.conv2d_nhwc(x, wp, cin, cin, 3, 1))
    assert torch.equal(out["0"][0], out["1"][0]) and torch.equal(out["0"][1], out["1"][1])
    assert not torch.isnan(out["1"][0]).any()


@pytest.mark.parametrize("shape", [(1, 8, 16), (2, 13, 37), (1, 40, 136), (3, 9, 131), (2, 64, 62), (1, 65, 63), (1, 200, 190), (1, 592, 800)])
def test_res_block_pair_equals_the_two_layers(shape):
    """mdf_conv2d_res_pair_fwd (res_pair.hip: a Res block of the refinement net, x + 0.1 * conv(relu(conv(x))), in one launch with
    rolling LDS windows) against the two single-layer launches it replaces -- same packed weights, tap order and epilogue arithmetic,
    so BIT-IDENTICAL; strips narrower / wider than 62 pixels, row counts around the 8-row steps and the segment boundaries -- and
    against torch on the CPU (net/unit/base.py:39-47)."""
    n, h, w = shape
    rng = np.random.RandomState(n * 100 + h + w + 7)
    x = T(rng.randn(n, 8, h, w).astype(np.float32))
    wa = T((rng.randn(8, 8, 3, 3) / np.sqrt(72)).astype(np.float32))
    wb = T((rng.randn(8, 8, 3, 3) / np.sqrt(72)).astype(np.float32))
    xd = ops.to_nhwc(x.to(DEV))
    pa, pb = ops.pack_conv2d_weight(wa.to(DEV)), ops.pack_conv2d_weight(wb.to(DEV))
    t = ops.conv2d_nhwc(xd, pa, 8, 8, 3, 1, None, None, True)
    two = ops.conv2d_nhwc(t, pb, 8, 8, 3, 1, None, None, False, xd, 0.1)
    one = ops.conv2d_res_pair(xd, pa, pb, 0.1)
    assert one.shape == two.shape == (n, h, w, 8)
    assert torch.equal(one, two), float((one - two).abs().max())
    exp = x + 0.1 * F.conv2d(F.relu(F.conv2d(x, wa, None, 1, 1)), wb, None, 1, 1)
    np.testing.assert_allclose(ops.from_nhwc(one).cpu().numpy(), exp.numpy(), rtol=1e-4, atol=2e-5)
    # x is not modified, and the call refuses to run in place
    assert torch.equal(ops.from_nhwc(xd).cpu(), x)


@pytest.mark.parametrize("shape", [(1, 8, 16), (2, 13, 37), (1, 65, 63), (2, 128, 160), (1, 592, 800)])
def test_refine_head_equals_range_mapping_plus_conv0(shape):
    """mdf_refine_head_fwd ((depth - lo) / span and RefineNet2.conv0 = Conv2d(1, 8, k3) in one launch on the vector ALUs, refine.py:29,36)
    against the two launches it replaces (mdf_range_affine_fwd + the MFMA conv): same roundings, same fma order -- BIT-IDENTICAL --
    and against torch on the CPU."""
    b, h, w = shape
    rng = np.random.RandomState(b + h + w)
    depth = T((425 + 510 * rng.rand(b, h, w)).astype(np.float32))
    lo = T(np.full(b, 425.0, np.float32) + np.arange(b, dtype=np.float32))
    span = T(np.full(b, 510.0, np.float32) - np.arange(b, dtype=np.float32))
    wt = T((rng.randn(8, 1, 3, 3) / 3.0).astype(np.float32))
    dd, lod, spd = depth.to(DEV), lo.to(DEV), span.to(DEV)
    x = ops.range_affine(dd, lod, spd, 0).unsqueeze(-1)
    two = ops.conv2d_nhwc(x, ops.pack_conv2d_weight(wt.to(DEV)), 1, 8, 3, 1, None, None, False)
    one = ops.refine_head(dd, lod, spd, wt.to(DEV))
    assert one.shape == two.shape == (b, h, w, 8)
    assert torch.equal(one, two), float((one - two).abs().max())
    exp = F.conv2d(((depth - lo.view(b, 1, 1)) / span.view(b, 1, 1)).unsqueeze(1), wt, None, 1, 1)
    np.testing.assert_allclose(ops.from_nhwc(one).cpu().numpy(), exp.numpy(), rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("b,h,w", [(1, 4, 6), (2, 10, 14), (5, 74, 100), (1, 148, 200)])
def test_fused_1x1_heads_equal_the_single_head_launches(b, h, w):
    """mdf_conv1x1_heads_fwd (conv1x1.hip: conv1x1_heads_kernel) -- the three composed heads of t4 (64 -> 64 / 32 / 16) and the two of t3
    (32 -> 32 / 16, each with its bilinear x2 upsample-add) as one launch per input -- against one mdf_conv2d_fwd launch per head:
    every output BIT-IDENTICAL (net/unit/backbone.py:58-66 in the composed form)."""
    rng = np.random.RandomState(b * 1000 + h + w)
    def head(cin, cout, bias):
        wt = T((rng.randn(cout, cin, 1, 1) / np.sqrt(cin)).astype(np.float32)).to(DEV)
        return (ops.pack_conv2d_weight(wt), (T(rng.randn(cout).astype(np.float32)).to(DEV) if bias else None), cin, cout)
    t4 = T(rng.randn(b, h, w, 64).astype(np.float32)).to(DEV)
    t3 = T(rng.randn(b, 2 * h, 2 * w, 32).astype(np.float32)).to(DEV)
    h4 = [head(64, 64, False), head(64, 32, False), head(64, 16, False)]
    h3 = [head(32, 32, True), head(32, 16, True)]
    single = lambda x, hd, up=None: ops.conv2d_nhwc(x, hd[0], hd[2], hd[3], 1, 1, None, hd[1], False, None, 1.0, up)
    y4, a4, c4 = ops.conv1x1_heads(t4, h4, [None, None, None])
    for got, hd in zip((y4, a4, c4), h4):
        assert torch.equal(got, single(t4, hd)), hd[3]
    y3, c3 = ops.conv1x1_heads(t3, h3, [a4, c4])
    assert torch.equal(y3, single(t3, h3[0], a4)) and torch.equal(c3, single(t3, h3[1], c4))
    assert y3.shape == (b, 2 * h, 2 * w, 32) and c3.shape == (b, 2 * h, 2 * w, 16)
