"""GPU parity, end to end: CoreNet.forward (product, HIP kernels) vs the reference.
Metric = BASELINE.json's: mean |delta depth| <= 1e-3 (mm).

The reference is not bit-stable across x86 hosts: its stage-1 gauss fit inverts a 3x3 fp32 matrix with
cond ~1e14 through LAPACK/BLAS, whose code paths differ between the build container's Xeon (where the goldens were
produced by the real reference) and the GPU box's EPYC (the reference algorithm run there differs from its own goldens by
1.2-1.4e-3 mm mean, scripts/diag_host_variance.py).  The host-side values the reference computed on the build host (the 4x4
projection products, the camera products, the fit row, log of the thresholds: a few hundred floats) are stored with the
goldens (oracle/gen_golden.py:HostValueRecorder), so both legs carry the metric's own bar:
  (A) same host: product vs the oracle executed live on this machine                         -> mean <= 1e-3
  (B) cross host: product, GIVEN the reference's own host values, vs the goldens of the real
      reference (everything the GPU computes is compared with the reference's output)        -> mean <= 1e-3
  (C) informational: product with this host's LAPACK values vs the goldens = the reference's own cross-host drift."""
import numpy as np
import pytest
import torch

from mdfnet_hip import synth
from oracle import mvs_oracle as O
from modelutil import build_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def model(seeded_sd):
    m = build_model()
    m.load_state_dict(seeded_sd)  # strict
    return m.eval().to(DEV)


def _recorded(g):
    from mdfnet_hip import ops
    return ops.recorded_host_values(projs=[g[f"host_proj{st}"] for st in range(3)], cams=[g[f"host_cam{st}"] for st in range(3)],
                                    fit_row=g["host_fit_row"],
                                    log_thresh={1: float(g["host_log_thresh1"]), 2: float(g["host_log_thresh2"])})


def _scene(g):
    w, h, v, b, rot, seed = g["cfg"]
    return synth.make_scene(int(w), int(h), int(v), batch=int(b), rot_deg=float(rot), seed=int(seed))


@pytest.mark.parametrize("name", ["e2e_tiny.npz", "e2e_cfg1.npz", "e2e_5view.npz"])
def test_forward_parity(golden, model, seeded_sd, name):
    g = golden(name)
    imgs, extr, intr, dr = _scene(g)
    with torch.no_grad():
        out = model(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))
        with _recorded(g):
            pinned = model(imgs.to(DEV), extr.clone().to(DEV), intr.clone().to(DEV), dr.clone().to(DEV))
    depth, conf = out["depth"].cpu().numpy(), out["confidence"].cpu().numpy()
    assert depth.shape == g["depth"].shape and conf.shape == g["confidence"].shape
    e_pin = np.abs(pinned["depth"].cpu().numpy() - g["depth"])
    c_pin = np.abs(pinned["confidence"].cpu().numpy() - g["confidence"])
    # live oracle on this host; warp through the explicit (host-independent, golden-pinned) arithmetic
    live = O.core_forward(seeded_sd, imgs, extr, intr, dr, warp=O.homo_warping_explicit)
    e_same = np.abs(depth - live["depth"].numpy())
    e_gold = np.abs(depth - g["depth"])
    e_ref_drift = np.abs(live["depth"].numpy() - g["depth"])
    c_same = np.abs(conf - live["confidence"].numpy())
    print(f"\n{name}: mean|d depth| (A) product-vs-oracle(same host) {e_same.mean():.3e} (max {e_same.max():.3e}) | "
          f"(B) product given the reference's host values vs golden {e_pin.mean():.3e} (max {e_pin.max():.3e}) | "
          f"(C) product with this host's LAPACK vs golden {e_gold.mean():.3e}, reference cross-host drift {e_ref_drift.mean():.3e} | "
          f"confidence max|d| same-host {c_same.max():.3e}, vs golden {c_pin.max():.3e}")
    assert e_same.mean() <= 1e-3, f"(A) same-host mean |delta depth| {e_same.mean()} > 1e-3"
    assert e_pin.mean() <= 1e-3, f"(B) vs the real reference's output, given its host values: {e_pin.mean()} > 1e-3"
    # confidence = sum of 4 probabilities picked by an integer index; an index flip moves it by O(p)
    assert np.mean(c_same > 1e-3) < 1e-3 and np.mean(c_pin > 1e-3) < 1e-3


def test_stagewise_vs_reference_trace(golden, model, seeded_sd):
    """Per-stage tensors of the product vs the reference's trace (goldens) and vs the live oracle."""
    g = golden("e2e_tiny.npz")
    imgs, extr, intr, dr = _scene(g)
    tr = {}
    hooks = []
    for st in range(3):
        hooks.append(model.Homoaggre[st].register_forward_hook(lambda m, i, o, st=st: tr.__setitem__(f"cost{st}", o)))
        hooks.append(model.Regular[st].register_forward_hook(lambda m, i, o, st=st: tr.__setitem__(f"prob{st}", o[0] if isinstance(o, tuple) else o)))   # (prob, depth) when fused
        hooks.append(model.Depth_hypos[st].register_forward_hook(lambda m, i, o, st=st: tr.__setitem__(f"hypos{st}", o)))
    _, live = O.core_forward(seeded_sd, imgs, extr, intr, dr, keep=True, warp=O.homo_warping_explicit)
    # Two fp32-equivalent forms of the backbone's 16 -> 32 k5-s2 layer (MDF_CONV_K5_WINOGRAD, read per call): the direct kernel keeps the
    # r02 bar of 6e-3 mm on the hypotheses against the golden; the Winograd form over the parity images (the default, -22 us per view)
    # moves the MAXIMUM over a stage's 9216 hypotheses to 5.5e-3 .. 6.3e-3 (mean 3.8e-4 in both, final depth error unchanged) and is
    # held to 8e-3.  Against the live oracle on this host both are held to 2e-3.  (per-form bars instead of one loosened bar.)
    import os
    try:
        for form, env, tol_hyp in (("direct k5-s2", "0", 6e-3), ("Winograd over parity images (default)", None, 8e-3)):
            if env is None:
                os.environ.pop("MDF_CONV_K5_WINOGRAD", None)
            else:
                os.environ["MDF_CONV_K5_WINOGRAD"] = env
            tr.clear()
            with torch.no_grad():
                model(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))
            print(f"\n[{form}]")
            for st in range(3):
                for k, tol_gold, tol_live in (("hypos", tol_hyp, 2e-3), ("cost", 2e-5, 2e-5), ("prob", 5e-4, 5e-4)):
                    a = tr[f"{k}{st}"].cpu().numpy()
                    dg, dl = np.abs(a - g[f"{k}{st}"]), np.abs(a - live[f"{k}{st}"].numpy())
                    print(f"stage {st} {k:5s}: vs golden max {dg.max():.3e} mean {dg.mean():.3e} | vs live oracle max {dl.max():.3e} "
                          f"mean {dl.mean():.3e}")
                    assert a.shape == g[f"{k}{st}"].shape
                    assert dg.max() <= tol_gold and dl.max() <= tol_live, f"{form}: stage {st} {k}: {dg.max():.3e} (bar {tol_gold}) / {dl.max():.3e} (bar {tol_live})"
    finally:
        os.environ.pop("MDF_CONV_K5_WINOGRAD", None)
        for hk in hooks:
            hk.remove()


@pytest.mark.parametrize("w,h,v,b,rng,base,seed", [
    (192, 128, 3, 2, (425.0, 935.0), 40.0, 41),       # batch of 2 (different cameras per sample)
    (384, 224, 7, 1, (0.5, 10.0), 0.25, 42),          # Tanks&Temples-like: 7 views, metric-scale depth range
    (256, 160, 11, 1, (425.0, 935.0), 25.0, 43),      # 11 views (EvalTanks.nviews, config.py:119): 10 source views
])
def test_other_configurations_same_host_parity(model, seeded_sd, w, h, v, b, rng, base, seed):
    imgs = synth.make_images(w, h, v, batch=b, seed=seed)
    intr, extr, dr = synth.make_cameras(w, h, v, batch=b, rot_deg=2.0, seed=seed + 1, depth_range=rng, baseline=base)
    with torch.no_grad():
        out = model(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))
    live = O.core_forward(seeded_sd, imgs, extr, intr, dr, warp=O.homo_warping_explicit)
    assert out["depth"].shape == (b, h, w) and out["confidence"].shape == (b, h, w)
    err = np.abs(out["depth"].cpu().numpy() - live["depth"].numpy())
    span = rng[1] - rng[0]
    print(f"\n{w}x{h}x{v} B={b} range={rng}: mean|d depth| {err.mean():.3e} max {err.max():.3e} (range span {span})")
    assert np.isfinite(out["depth"].cpu().numpy()).all()
    # BASELINE's 1e-3 mm is quoted at DTU scale (span 510 mm): scale the bar with the depth range
    assert err.mean() <= 1e-3 * span / 510.0


def test_full_size_cfg2_parity_vs_live_oracle(model, seeded_sd):
    """BASELINE config 2 at its full size (1600x1184, 5 views, hypotheses 48/24/8): the product against the oracle run live on
    this host's CPU (explicit-arithmetic warp, host-independent), the metric's own bar mean |d depth| <= 1e-3 mm; plus
    run-to-run determinism."""
    import time
    imgs, extr, intr, dr = synth.make_scene(1600, 1184, 5, rot_deg=3.0, seed=100)
    with torch.no_grad():
        out = model(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))
        out2 = model(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))
    assert torch.equal(out["depth"], out2["depth"]) and torch.equal(out["confidence"], out2["confidence"])
    t0 = time.time()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    live = O.core_forward(seeded_sd, imgs, extr, intr, dr, warp=O.homo_warping_explicit)
    err = np.abs(out["depth"].cpu().numpy() - live["depth"].numpy())
    cerr = np.abs(out["confidence"].cpu().numpy() - live["confidence"].numpy())
    print(f"\ncfg2 full size: mean|d depth| {err.mean():.3e} mm, max {err.max():.3e}, p99.9 {np.quantile(err, 0.999):.3e}; "
          f"confidence mean|d| {cerr.mean():.3e}; oracle took {time.time() - t0:.1f} s")
    assert out["depth"].shape == (1, 1184, 1600) and np.isfinite(err).all()
    assert err.mean() <= 1e-3
    assert cerr.mean() <= 1e-4


@pytest.mark.parametrize("views", [7, 11])
def test_full_size_cfg4_parity_vs_live_oracle(model, seeded_sd, views):
    """BASELINE config 4 at its full size: Tanks&Temples 1920x1056 (1080 cropped as load/tankseval.py:36), metric-scale depth
    range, 7 views (BASELINE.json) and 11 views (EvalTanks.nviews, config.py:119): the product against the oracle run live on
    this host's CPU over the whole image, the metric's bar scaled to the depth range; determinism; finiteness."""
    import time
    w, h, rng = 1920, 1056, (0.5, 10.0)
    imgs = synth.make_images(w, h, views, batch=1, seed=200 + views)
    # baseline 0.03 scene units: the sweep's parallax (f*B/z = 3470*0.03/z px: 208 px at z = 0.5, 10 px at z = 10) stays inside
    # the 1920-px frame, as for a camera orbiting a scene
    intr, extr, dr = synth.make_cameras(w, h, views, batch=1, rot_deg=2.0, seed=300 + views, depth_range=rng, baseline=0.03)
    with torch.no_grad():
        out = model(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))
        out2 = model(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))
    assert torch.equal(out["depth"], out2["depth"]) and torch.equal(out["confidence"], out2["confidence"])
    assert out["depth"].shape == (1, h, w) and bool(torch.isfinite(out["depth"]).all()) and bool(torch.isfinite(out["confidence"]).all())
    t0 = time.time()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    live = O.core_forward(seeded_sd, imgs, extr, intr, dr, warp=O.homo_warping_explicit)
    err = np.abs(out["depth"].cpu().numpy() - live["depth"].numpy())
    cerr = np.abs(out["confidence"].cpu().numpy() - live["confidence"].numpy())
    span = rng[1] - rng[0]
    print(f"\ncfg4 full size {w}x{h}x{views}: mean|d depth| {err.mean():.3e} (bar {1e-3 * span / 510.0:.2e}, range span {span}), max {err.max():.3e}; "
          f"confidence mean|d| {cerr.mean():.3e}; oracle took {time.time() - t0:.1f} s")
    assert err.mean() <= 1e-3 * span / 510.0          # BASELINE's 1e-3 mm is quoted at DTU scale (span 510 mm)
    assert cerr.mean() <= 1e-4
