"""CPU: eval data plane (PFM / pair / cam parsing, DTU-layout loader) against the reference's fixtures."""
import os

import numpy as np
import torch

from tools import data_io
from load import synthetic
from load.dtueval import LoadDataset


def test_pfm_bytes_identical_to_reference(golden, tmp_path):
    g = golden("io.npz")
    f = str(tmp_path / "a.pfm")
    data_io.save_pfm(f, g["img"])
    assert np.array_equal(np.frombuffer(open(f, "rb").read(), dtype=np.uint8), g["pfm_bytes"])
    assert open(f, "rb").read().startswith(b"Pf\n7 5\n-1.000000\n")
    back, scale = data_io.read_pfm(f)
    assert np.array_equal(back, g["pfm_back"]) and np.array_equal(back, g["img"]) and scale == float(g["pfm_scale"])


def test_pair_file_parse_matches_reference(golden, tmp_path):
    g = golden("io.npz")
    p = str(tmp_path / "pair.txt")
    open(p, "w").write("2\n0\n3 1 9.5 2 8.0 3 7.5\n1\n2 0 9.5 2 6.0\n")
    n, pairs = data_io.read_pairfile(p)
    assert n == int(g["pair_n"]) and [q[0] for q in pairs] == list(g["pair_ref"])
    assert pairs[0][1] == list(g["pair_src0"]) and pairs[1][1] == list(g["pair_src1"])


def test_synthetic_dtu_layout_loader(tmp_path):
    root = synthetic.write_dtu_eval_set(str(tmp_path / "dtu"), scans=(1, 4), nviews_total=5, width=160, height=128)
    ds = LoadDataset(root, os.path.join(root, "pair.txt"), [1, 4], nviews=3)
    assert len(ds) == 10
    item = ds[7]
    assert item["imgs"].shape == (3, 3, 128, 160) and item["imgs"].dtype == np.float32
    assert 0.0 <= item["imgs"].min() and item["imgs"].max() <= 1.0
    assert item["intrinsics"].shape == (3, 3, 3) and item["extrinsics"].shape == (3, 4, 4)
    assert item["depth_range"].dtype == np.float64 and list(item["depth_range"]) == [425.0, 935.0]
    assert item["filename"].format("depth_est", ".pfm") == "scan4/depth_est/00000002.pfm"
    k = item["intrinsics"][0]
    assert abs(k[0, 0] - 2892.33 * 160 / 1600) < 1e-3   # cam file round trip
