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
    # depth ~ 425..935 mm: 1 ulp = 6e-5; sum of <=48 products in a different order
    np.testing.assert_allclose(d.cpu().numpy(), g[f"reg{stage}_depth"], rtol=0, atol=3e-4)


def test_confidence_and_index(golden):
    g = golden("ops.npz")
    prob = T(g["reg2_prob"])
    c, idx = ops.confidence(prob.to(DEV), return_index=True)
    np.testing.assert_allclose(c.cpu().numpy(), g["conf2"], rtol=0, atol=2e-7)
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
    # the normal matrix is ill-conditioned (H3): b0 is a cancelling sum; compare the range it produces
    np.testing.assert_allclose(s.cpu().numpy(), g["hyp1_s"], rtol=2e-3)
    lt = float(torch.log(torch.tensor(0.95)))
    out = ops.hypos_from_fit(1, T(g["hyp1_s"]).to(DEV), d0, dr.float().to(DEV), lt, 24, True)
    np.testing.assert_allclose(out.cpu().numpy(), g["hyp1_out"], rtol=0, atol=5e-4)


def test_hypos_laplace_stage2(golden):
    g = golden("ops.npz")
    dr = synth.make_scene(96, 64, 3, batch=2, rot_deg=4.0, seed=5)[3]
    p1, d1 = T(g["reg1_prob"]).to(DEV), T(g["reg1_depth"]).to(DEV)
    s = ops.hypos_fit(2, p1, d1, T(g["agg1_hyp"]).to(DEV))
    np.testing.assert_allclose(s.cpu().numpy(), g["hyp2_s"], rtol=2e-5)
    lt = float(torch.log(torch.tensor(1e-5)))
    out = ops.hypos_from_fit(2, s, d1, dr.float().to(DEV), lt, 8, True)
    np.testing.assert_allclose(out.cpu().numpy(), g["hyp2_out"], rtol=0, atol=5e-4)
    same = ops.hypos_from_fit(2, s, d1, dr.float().to(DEV), lt, 8, False)
    assert same.shape == (2, 8) + tuple(d1.shape[1:])
