"""training data plane (load/dtutrain.py, load/blendedtrain.py counterparts) and one epoch of the training
driver on a synthetic training set.  OpenCV is absent, so the loaders' cv2.INTER_NEAREST is a restatement: pinned by its
index formula and by the properties the reference relies on (exact striding for the 1/2, 1/4, 1/8 maps)."""
import os
import random

import numpy as np
import pytest
import torch

from load import synthetic
from tools import data_io


def test_resize_nearest_is_opencv_style_not_centre_aligned():
    a = np.arange(7 * 10, dtype=np.float32).reshape(7, 10)
    # divisible sizes and power-of-two factors = plain striding (what the 4-scale ground truth uses)
    b = np.arange(16 * 24, dtype=np.float32).reshape(16, 24)
    for r in (2, 4, 8):
        assert np.array_equal(data_io.resize_nearest(b, (24 // r, 16 // r)), b[::r, ::r])
    # non-divisible: floor(dst * src/dst), clipped -- (10 -> 4): cols floor([0, 2.5, 5, 7.5]) = 0, 2, 5, 7; (7 -> 3): rows 0, 2, 4
    out = data_io.resize_nearest(a, (4, 3))
    assert np.array_equal(out, a[[0, 2, 4]][:, [0, 2, 5, 7]])
    # upscaling repeats, last index clipped
    up = data_io.resize_nearest(a[:2, :3], (6, 4))
    assert np.array_equal(up, a[[0, 0, 1, 1]][:, [0, 0, 1, 1, 2, 2]])
    assert out.dtype == a.dtype and out.flags["C_CONTIGUOUS"]


def test_dtu_train_loader_items(tmp_path):
    from load.dtutrain import LoadDataset
    root = synthetic.write_dtu_train_set(str(tmp_path / "dtu"), scenes=(2, 6), nviews_total=6, lightings=(0, 3), width=160, height=128)
    ds = LoadDataset(root, os.path.join(root, "Cameras", "pair.txt"), [2, 6], [0, 3], nviews=3, robust_train=False)
    assert len(ds) == 2 * 6 * 2
    assert ds.all_compose[0][:3] == [2, 0, 0] and ds.all_compose[1][:3] == [2, 3, 0] and ds.all_compose[2][:3] == [2, 0, 1]   # scene, lighting, ref
    it = ds[5]
    assert it["imgs"].shape == (3, 3, 128, 160) and it["imgs"].dtype == np.float32 and 0.0 <= it["imgs"].min() and it["imgs"].max() <= 1.0
    assert it["intrinsics"].shape == (3, 3, 3) and it["extrinsics"].shape == (3, 4, 4)
    assert list(it["ref_depths"].keys()) == ["3", "2", "1", "0"]
    assert [it["ref_depths"][k].shape for k in "3210"] == [(16, 20), (32, 40), (64, 80), (128, 160)]
    assert np.array_equal(it["ref_depths"]["2"], it["ref_depths"]["0"][::4, ::4])
    assert np.array_equal(it["depth_range"], np.array([425.0, 935.0])) and it["depth_range"].dtype == np.float64
    # the lighting picks the file: same geometry, different images
    a, b = ds[4], ds[5]            # same scene/ref, lighting 0 vs 3
    assert np.array_equal(a["extrinsics"], b["extrinsics"]) and not np.array_equal(a["imgs"], b["imgs"])
    # robust training: reference first, sources sampled without replacement from the ranked list minus its first entry
    ds_r = LoadDataset(root, os.path.join(root, "Cameras", "pair.txt"), [2], [0], nviews=4, robust_train=True)
    random.seed(3)
    first = ds_r[0]["extrinsics"]
    random.seed(3)
    assert np.array_equal(ds_r[0]["extrinsics"], first)                      # reproducible under the global `random` seed
    _, _, ref, srcs = ds_r.all_compose[0]
    cams = {v: data_io.read_cam_file(os.path.join(root, "Cameras", "{:0>8}_cam.txt".format(v)))[1] for v in range(6)}
    assert np.array_equal(first[0], cams[ref])
    allowed = [cams[s] for s in srcs[1:]]
    assert all(any(np.array_equal(e, c) for c in allowed) for e in first[1:])


def test_blendedmvs_loader_items(tmp_path):
    from load.blendedtrain import LoadDataset
    root = synthetic.write_blendedmvs_set(str(tmp_path / "bl"), scans=("scanA", "scanB"), nviews_total=5)
    ds = LoadDataset(root, nviews=5, robust_train=False)
    assert len(ds) == 2 * 4                                   # the reference view without sources is dropped
    scan, ref, srcs = ds.all_compose[3]                       # the short list (2 sources) is padded with its best source
    assert ref == 4 and srcs == [0, 1, 0, 0, 0]
    it = ds[0]
    assert it["imgs"].shape == (5, 3, 128, 160) and list(it["ref_depths"].keys()) == ["3", "2", "1", "0"]
    assert np.allclose(it["depth_range"], [420.0, 940.0]) and np.allclose(ds[4]["depth_range"], [421.0, 941.0])   # per scan, from the cam file
    it = ds[3]
    assert np.array_equal(it["extrinsics"][3], it["extrinsics"][1])         # padded views repeat source 0


def test_one_training_epoch_on_the_synthetic_dtu_train_set(tmp_path, monkeypatch, rehearsal_backend):
    """load -> DataLoader collate -> CoreNet.train() -> Loss -> FlatBucket -> Adam step -> checkpoint, as train.py wires them."""
    import importlib.util
    import torch.optim as optim
    from torch.utils.data import DataLoader
    from load.dtutrain import LoadDataset
    from mdfnet_hip import ddp
    from net import loss as loss_mod
    from modelutil import build_model
    root = synthetic.write_dtu_train_set(str(tmp_path / "dtu"), scenes=(2,), nviews_total=4, lightings=(0,), width=96, height=64)
    ds = LoadDataset(root, os.path.join(root, "Cameras", "pair.txt"), [2], [0], nviews=3, robust_train=True)
    spec = importlib.util.spec_from_file_location("mdf_train", os.path.join(os.path.dirname(data_io.__file__), "..", "train.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    torch.manual_seed(0)
    random.seed(0)
    model = build_model()
    bucket = ddp.FlatBucket(model)
    opt = optim.Adam([{"params": model.parameters(), "initial_lr": 1e-3}], lr=1e-3)
    before = [p.detach().clone() for p in model.parameters()]
    batches = DataLoader(ds, batch_size=2, shuffle=False, drop_last=True)
    mean_loss = tr.train_one_epoch(model, bucket, opt, loss_mod.Loss(), batches, torch.device("cpu"), log=lambda *a, **k: None, epoch=1)
    assert np.isfinite(mean_loss) and mean_loss > 0
    assert sum(int(not torch.equal(a, b)) for a, b in zip(before, model.parameters())) > 100     # the step moved the weights
    ck = str(tmp_path / "dtu_1.pth")
    torch.save({"epoch": 1, "model": model.state_dict()}, ck)
    m2 = build_model()
    m2.load_state_dict(torch.load(ck, map_location="cpu")["model"])          # train.py:60-68 checkpoint format, strict


REF = os.environ.get("MDF_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "load", "dtutrain.py")), reason="reference tree not present (GPU box)")
def test_loaders_match_the_reference_loaders_item_for_item(tmp_path):
    """Build container only: the reference's own load/dtutrain.py and load/blendedtrain.py, run in a subprocess on the same
    synthetic sets, return the same items (composition order, robust sampling under the same `random` seed, padding, ranges).
    OpenCV is absent here, so the subprocess gets a stub `cv2` whose `resize` is data_io.resize_nearest -- the nearest-resize
    itself is therefore NOT part of this comparison (it is pinned by its formula in the first test)."""
    import subprocess
    import sys
    droot = synthetic.write_dtu_train_set(str(tmp_path / "dtu"), scenes=(2, 6), nviews_total=6, lightings=(0, 3), width=96, height=64)
    broot = synthetic.write_blendedmvs_set(str(tmp_path / "bl"), scans=("scanA", "scanB"), nviews_total=5, width=96, height=64)
    stub = tmp_path / "stub"
    stub.mkdir()
    ours = os.path.join(os.path.dirname(data_io.__file__), "data_io.py")
    (stub / "cv2.py").write_text(
        "import importlib.util\n"
        "_s = importlib.util.spec_from_file_location('_mdf_data_io', %r); _m = importlib.util.module_from_spec(_s); _s.loader.exec_module(_m)\n"
        "INTER_NEAREST = 0\n"
        "def resize(img, size, interpolation=None):\n"
        "    assert interpolation == INTER_NEAREST\n"
        "    return _m.resize_nearest(img, size)\n" % ours)
    out = str(tmp_path / "ref_items.npz")
    code = (
        "import sys, random, numpy as np; sys.path[:0] = [%r, %r]\n"
        "import os; os.chdir('/tmp')\n"
        "from load.dtutrain import LoadDataset as D\n"
        "from load.blendedtrain import LoadDataset as B\n"
        "res = {}\n"
        "def put(tag, it):\n"
        "    for k in ('imgs', 'intrinsics', 'extrinsics', 'depth_range'): res[tag + k] = np.asarray(it[k])\n"
        "    for k, v in it['ref_depths'].items(): res[tag + 'gt' + k] = v\n"
        "for robust in (0, 1):\n"
        "    d = D(%r, %r, [2, 6], [0, 3], 3, bool(robust)); b = B(%r, 4, bool(robust))\n"
        "    res['len_d%%d' %% robust] = np.array(len(d)); res['len_b%%d' %% robust] = np.array(len(b))\n"
        "    random.seed(11)\n"
        "    for i in (0, 5, 13, len(d) - 1): put('d%%d_%%d_' %% (robust, i), d[i])\n"
        "    for i in (0, 3, len(b) - 1): put('b%%d_%%d_' %% (robust, i), b[i])\n"
        "np.savez(%r, **res)\n" % (REF, str(stub), droot, os.path.join(droot, "Cameras", "pair.txt"), broot, out))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = np.load(out)
    from load.blendedtrain import LoadDataset as B
    from load.dtutrain import LoadDataset as D
    n = 0
    for robust in (0, 1):
        d = D(droot, os.path.join(droot, "Cameras", "pair.txt"), [2, 6], [0, 3], 3, bool(robust))
        b = B(broot, 4, bool(robust))
        assert len(d) == int(ref["len_d%d" % robust]) and len(b) == int(ref["len_b%d" % robust])
        random.seed(11)
        items = [("d%d_%d_" % (robust, i), d[i]) for i in (0, 5, 13, len(d) - 1)] + [("b%d_%d_" % (robust, i), b[i]) for i in (0, 3, len(b) - 1)]
        for tag, it in items:
            for k in ("imgs", "intrinsics", "extrinsics", "depth_range"):
                a = np.asarray(it[k])
                assert a.dtype == ref[tag + k].dtype and np.array_equal(a, ref[tag + k]), tag + k
            for k, v in it["ref_depths"].items():
                assert np.array_equal(v, ref[tag + "gt" + k]), tag + k
            n += 1
    assert n == 14


def test_train_driver_two_ranks_gloo_on_synthetic_blendedmvs(tmp_path):
    """`torchrun --nproc-per-node 2 train.py -d blendedmvs` (gloo on CPU): DistributedSampler split, per-rank batch = global
    batch / world, flat-bucket gradient all-reduce, rank-0 checkpoint + epoch_loss.txt, both ranks end with equal weights."""
    import subprocess
    import sys
    root = tmp_path / "data"
    synthetic.write_blendedmvs_set(str(root / "blendedmvs768x576"), scans=("scanA", "scanB", "scanC"), nviews_total=7, width=96, height=64,
                                   short_pairs=False)
    pkg = os.path.dirname(os.path.dirname(data_io.__file__))
    env = dict(os.environ, MDF_DATA_ROOT=str(root), MDF_PTH_PATH=str(tmp_path / "pth"), MDF_MAX_EPOCH="1", OMP_NUM_THREADS="2",
               PYTHONPATH=pkg)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", os.path.join(pkg, "train.py"), "-d", "blendedmvs"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    ck = torch.load(str(tmp_path / "pth" / "blendedmvs_1.pth"), map_location="cpu")
    assert ck["epoch"] == 1 and len(ck["model"]) == 290 and not any(k.startswith("module.") for k in ck["model"])
    loss = float(open(str(tmp_path / "pth" / "epoch_loss.txt")).read().split()[0])
    assert np.isfinite(loss) and loss > 0
