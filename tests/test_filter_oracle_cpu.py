"""CPU: pin the consistency-filter oracle (oracle/filter_oracle.py) against goldens from the reference's own functions."""
import numpy as np
import torch

from oracle import filter_oracle as FO

T = torch.from_numpy


def _unpack(packed, h, w):
    return np.unpackbits(packed, axis=0)[:9].astype(bool)


def test_reproject_and_masks_match_reference(golden):
    g = golden("filter.npz")
    d, K, E = g["depths"], g["K"], g["E"]
    out = FO.reproject_with_depth(T(d[0]), T(K[0]), T(E[0]), T(d[1]), T(K[1]), T(E[1]))
    assert np.array_equal(torch.stack([o[0] for o in out]).numpy(), g["reproj1"], equal_nan=True)
    exp = FO.reproject_explicit(T(d[0]), T(K[0]), T(E[0]), T(d[1]), T(K[1]), T(E[1]))   # host-independent restatement
    assert np.array_equal(torch.stack([o[0] for o in exp]).numpy(), g["reproj1"], equal_nan=True)
    for v in range(1, d.shape[0]):
        masks, last, rep = FO.check_geometric_consistency(T(d[0]), T(K[0]), T(E[0]), T(d[v]), T(K[v]), T(E[v]))
        assert np.array_equal(torch.stack(masks)[:, 0].numpy(), _unpack(g[f"masks{v}"], *d.shape[1:]))
        assert np.array_equal(rep[0].numpy(), g[f"rep{v}"])
        m2, _, rep2 = FO.check_geometric_consistency(T(d[0]), T(K[0]), T(E[0]), T(d[v]), T(K[v]), T(E[v]), explicit=True)
        assert np.array_equal(torch.stack(m2)[:, 0].numpy(), _unpack(g[f"masks{v}"], *d.shape[1:])) and torch.equal(rep2, rep)
        counts = [int(m.sum()) for m in masks]
        assert counts == sorted(counts) and counts[0] < counts[-1]      # the nine thresholds actually differentiate


def test_fused_view_matches_reference(golden):
    g = golden("filter.npz")
    d, K, E = g["depths"], g["K"], g["E"]
    n = d.shape[0]
    r = FO.fuse_view(T(d[0]), T(g["conf"]), T(K[0]), T(E[0]), [T(d[v]) for v in range(1, n)], [T(K[v]) for v in range(1, n)],
                     [T(E[v]) for v in range(1, n)])
    assert np.array_equal(r["geo_mask"].numpy(), g["geo_mask"]) and np.array_equal(r["photo_mask"].numpy(), g["photo_mask"])
    assert np.array_equal(r["final_mask"].numpy(), g["final_mask"])
    assert np.array_equal(r["depth_avg"].numpy(), g["depth_avg"])
    assert 0 < int(g["final_mask"].sum()) < g["final_mask"].size
    pts = FO.backproject(r["depth_avg"], r["final_mask"], T(K[0]), T(E[0]))
    assert pts.shape == (int(g["final_mask"].sum()), 3) and np.isfinite(pts).all()
