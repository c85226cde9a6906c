"""GPU parity: regression heads and the curve-fit hypothesis generator vs reference goldens."""
import numpy as np
import pytest
import torch

from mdfnet_hip import ops, synth
from oracle import mvs_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"


@pytest.mark.parametrize("stage", [0, 1, 2])
def test_depth_regression(golden, stage):
    g = golden("ops.npz")
    d = ops.depth_regress(T(g[f"reg{stage}_prob"]).to(DEV), T(g[f"agg{stage}_hyp"]).to(DEV))
    # bit-exact: the kernel mirrors ATen's cascade summation order (D = 48, 24, 8 all covered)
    assert np.array_equal(d.cpu().numpy(), g[f"reg{stage}_depth"])


def test_confidence_and_index(golden):
    g = golden("ops.npz")
    prob = T(g["reg2_prob"])
    c, idx = ops.confidence(prob.to(DEV), return_index=True)
    assert np.array_equal(c.cpu().numpy(), g["conf2"])
    exp_idx = O.confidence_index(prob)
    nbad = int((idx.cpu() != exp_idx).sum())
    assert idx.dtype == torch.int64 and nbad == 0, f"{nbad} confidence-index mismatches"


def test_hypos_gauss1_stage1(golden):
    g = golden("ops.npz")
    dr = synth.make_scene(96, 64, 3, batch=2, rot_deg=4.0, seed=5)[3]
    hyp0 = O.uniform_hypos(dr, 48)
    p0, d0 = T(g["reg0_prob"]).to(DEV), T(g["reg0_depth"]).to(DEV)
    row = ops.gauss1_fit_row(hyp0)
    assert torch.equal(row, O.gauss1_row0(hyp0.reshape(2, 48)))
    s = ops.hypos_fit(1, p0, d0, hyp0.to(DEV), row.to(DEV))
    # same row (host-computed with the reference's own calls), same sequential fp32 order as ATen's bmm; only
    # logf may differ by an ulp, which the cancelling sum amplifies to ~1e-5 relative in a few pixels.
    # Compared with the oracle run on THIS host: the row's bits depend on the host's BLAS/LAPACK (cond ~1e14).
    sn = s.cpu().numpy()
    live = O.gauss1_fit(T(g["reg0_prob"]), hyp0).numpy()
    np.testing.assert_allclose(sn, live, rtol=1e-4)
    print("gauss1 s: fraction of pixels not bit-identical to the same-host oracle:", float(np.mean(sn != live)))
    np.testing.assert_allclose(sn, g["hyp1_s"], rtol=5e-3)  # this host's row vs the build container's golden: H3 drift
    # GIVEN the row the real reference computed on the build host (ops.npz:hyp1_fit_row), the kernel reproduces the
    # reference's s: same sequential fp32 sum; the only ulp source left is logf (device libm vs the host's SLEEF), which
    # the cancelling 48-term sum (terms ~1e3 x larger than the result) amplifies to <= ~1e-4 relative in single pixels
    sg = ops.hypos_fit(1, p0, d0, hyp0.to(DEV), T(g["hyp1_fit_row"]).to(DEV)).cpu().numpy()
    rel = np.abs(sg - g["hyp1_s"]) / np.abs(g["hyp1_s"])
    print(f"gauss1 s given the reference's row: not bit-identical {float(np.mean(sg != g['hyp1_s'])):.3f}, rel max {rel.max():.2e} mean {rel.mean():.2e}")
    assert rel.max() <= 2e-4 and rel.mean() <= 1e-5
    lt = float(torch.log(torch.tensor(0.95)))
    out = ops.hypos_from_fit(1, T(g["hyp1_s"]).to(DEV), d0, dr.float().to(DEV), lt, 24, True)
    # same upsample/range arithmetic as ATen (bit-exact in all but ~0.1 % of entries: sqrt of a 1-ulp-different product)
    d = np.abs(out.cpu().numpy() - g["hyp1_out"])
    assert d.max() <= 1.3e-4 and np.mean(d > 0) < 5e-3


def test_hypos_laplace_stage2(golden):
    g = golden("ops.npz")
    dr = synth.make_scene(96, 64, 3, batch=2, rot_deg=4.0, seed=5)[3]
    p1, d1 = T(g["reg1_prob"]).to(DEV), T(g["reg1_depth"]).to(DEV)
    s = ops.hypos_fit(2, p1, d1, T(g["agg1_hyp"]).to(DEV))
    np.testing.assert_allclose(s.cpu().numpy(), g["hyp2_s"], rtol=2e-5)
    lt = float(torch.log(torch.tensor(1e-5)))
    out = ops.hypos_from_fit(2, s, d1, dr.float().to(DEV), lt, 8, True)
    np.testing.assert_allclose(out.cpu().numpy(), g["hyp2_out"], rtol=0, atol=5e-4)
    exact = ops.hypos_from_fit(2, T(g["hyp2_s"]).to(DEV), d1, dr.float().to(DEV), lt, 8, True)
    assert np.array_equal(exact.cpu().numpy(), g["hyp2_out"])  # given the reference's s: bit-exact
    same = ops.hypos_from_fit(2, s, d1, dr.float().to(DEV), lt, 8, False)
    assert same.shape == (2, 8) + tuple(d1.shape[1:])
