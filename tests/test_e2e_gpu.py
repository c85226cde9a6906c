"""GPU parity, end to end: CoreNet.forward (product, HIP kernels) vs goldens produced by the reference.
Metric = BASELINE.json's: mean |delta depth| <= 1e-3 (mm)."""
import numpy as np
import pytest
import torch

from mdfnet_hip import synth
from modelutil import build_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def model(seeded_sd):
    m = build_model()
    m.load_state_dict(seeded_sd)  # strict
    return m.eval().to(DEV)


@pytest.mark.parametrize("name", ["e2e_tiny.npz", "e2e_cfg1.npz", "e2e_5view.npz"])
def test_forward_vs_reference_golden(golden, model, name):
    g = golden(name)
    w, h, v, b, rot, seed = g["cfg"]
    imgs, extr, intr, dr = synth.make_scene(int(w), int(h), int(v), batch=int(b), rot_deg=float(rot), seed=int(seed))
    with torch.no_grad():
        out = model(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))
    depth, conf = out["depth"].cpu().numpy(), out["confidence"].cpu().numpy()
    assert depth.shape == g["depth"].shape and conf.shape == g["confidence"].shape
    err = np.abs(depth - g["depth"])
    print(f"{name}: mean|d depth| = {err.mean():.3e} mm, max = {err.max():.3e}, "
          f"conf max|d| = {np.abs(conf - g['confidence']).max():.3e}")
    assert err.mean() <= 1e-3, f"mean |delta depth| {err.mean()} > 1e-3"
    # confidence is a sum of 4 probabilities selected by an integer index; an index flip moves it by O(p)
    assert np.mean(np.abs(conf - g["confidence"]) > 1e-3) < 1e-3


def test_stagewise_vs_reference_trace(golden, model, seeded_sd):
    """Feed the product the same inputs and compare every per-stage tensor of the reference's trace."""
    g = golden("e2e_tiny.npz")
    w, h, v, b, rot, seed = g["cfg"]
    imgs, extr, intr, dr = synth.make_scene(int(w), int(h), int(v), batch=int(b), rot_deg=float(rot), seed=int(seed))
    tr = {}
    hooks = []
    for st in range(3):
        hooks.append(model.Homoaggre[st].register_forward_hook(lambda m, i, o, st=st: tr.__setitem__(f"cost{st}", o)))
        hooks.append(model.Regular[st].register_forward_hook(lambda m, i, o, st=st: tr.__setitem__(f"prob{st}", o)))
        hooks.append(model.Depth_hypos[st].register_forward_hook(lambda m, i, o, st=st: tr.__setitem__(f"hypos{st}", o)))
    with torch.no_grad():
        model(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))
    for hk in hooks:
        hk.remove()
    for st in range(3):
        for k, tol in (("hypos", 5e-3), ("cost", 2e-5), ("prob", 5e-4)):
            a, e = tr[f"{k}{st}"].cpu().numpy(), g[f"{k}{st}"]
            assert a.shape == e.shape
            d = np.abs(a - e)
            print(f"stage {st} {k}: max|d| {d.max():.3e} mean {d.mean():.3e}")
            assert d.max() <= tol, f"stage {st} {k}: {d.max()}"
