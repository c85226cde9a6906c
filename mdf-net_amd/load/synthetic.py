"""Write a synthetic DTU-layout evaluation set to disk (no dataset exists offline): <root>/scan<N>/images/*.jpg,
<root>/scan<N>/cams/*_cam.txt, <root>/pair.txt — the layout load/dtueval.py reads."""
import os

import numpy as np

from mdfnet_hip import synth


def write_dtu_eval_set(root, scans=(1,), nviews_total=5, width=160, height=128, seed=0):
    from PIL import Image
    os.makedirs(root, exist_ok=True)
    imgs, extr, intr, _ = synth.make_scene(width, height, nviews_total, batch=1, rot_deg=2.0, seed=seed)
    for scan in scans:
        base = os.path.join(root, "scan{}".format(scan))
        os.makedirs(os.path.join(base, "images"), exist_ok=True)
        os.makedirs(os.path.join(base, "cams"), exist_ok=True)
        for v in range(nviews_total):
            arr = (imgs[0, v].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
            Image.fromarray(arr).save(os.path.join(base, "images", "{:0>8}.jpg".format(v)), quality=95)
            with open(os.path.join(base, "cams", "{:0>8}_cam.txt".format(v)), "w") as f:
                f.write("extrinsic\n")
                for r in extr[0, v].numpy():
                    f.write(" ".join("%.8f" % x for x in r) + "\n")
                f.write("\nintrinsic\n")
                for r in intr[0, v].numpy():
                    f.write(" ".join("%.8f" % x for x in r) + "\n")
                f.write("\n425.0 2.5\n")
    with open(os.path.join(root, "pair.txt"), "w") as f:
        f.write("%d\n" % nviews_total)
        for ref in range(nviews_total):
            srcs = [v for v in range(nviews_total) if v != ref]
            f.write("%d\n%d " % (ref, len(srcs)) + " ".join("%d %.2f" % (s, 100.0 - abs(s - ref)) for s in srcs) + "\n")
    return root


def _write_cam(path, extr, intr, last_line):
    with open(path, "w") as f:
        f.write("extrinsic\n")
        for r in extr:
            f.write(" ".join("%.8f" % x for x in r) + "\n")
        f.write("\nintrinsic\n")
        for r in intr:
            f.write(" ".join("%.8f" % x for x in r) + "\n")
        f.write("\n" + last_line + "\n")


def _plane_depth(height, width, seed, lo=500.0, hi=850.0):
    """A smooth synthetic ground-truth depth map with a hole (zeros = invalid, masked out by the loss)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    d = lo + (hi - lo) * (0.5 + 0.3 * np.sin(xx / width * 3.0 + rng.rand()) * np.cos(yy / height * 2.0 + rng.rand()))
    d[: height // 8, : width // 8] = 0.0
    return d.astype(np.float32)


def write_dtu_train_set(root, scenes=(2,), nviews_total=6, lightings=(0, 1), width=160, height=128, seed=0):
    """<root>/Rectified/scan<N>_train/rect_<view+1:03>_<light>_r5000.png, <root>/Cameras/<view:08>_cam.txt + pair.txt,
    <root>/Depths/scan<N>_train/depth_map_<view:04>.pfm -- the layout load/dtutrain.py reads."""
    from PIL import Image
    from tools import data_io
    imgs, extr, intr, _ = synth.make_scene(width, height, nviews_total, batch=1, rot_deg=2.0, seed=seed)
    os.makedirs(os.path.join(root, "Cameras"), exist_ok=True)
    for v in range(nviews_total):
        _write_cam(os.path.join(root, "Cameras", "{:0>8}_cam.txt".format(v)), extr[0, v].numpy(), intr[0, v].numpy(), "425.0 2.5")
    with open(os.path.join(root, "Cameras", "pair.txt"), "w") as f:
        f.write("%d\n" % nviews_total)
        for ref in range(nviews_total):
            srcs = [v for v in range(nviews_total) if v != ref]
            f.write("%d\n%d " % (ref, len(srcs)) + " ".join("%d %.2f" % (s, 100.0 - abs(s - ref)) for s in srcs) + "\n")
    for scene in scenes:
        folder = "scan{}_train".format(scene)
        os.makedirs(os.path.join(root, "Rectified", folder), exist_ok=True)
        os.makedirs(os.path.join(root, "Depths", folder), exist_ok=True)
        for v in range(nviews_total):
            arr = imgs[0, v].permute(1, 2, 0).numpy()
            for li in lightings:
                lit = np.clip(arr * (0.8 + 0.05 * li), 0, 1)
                Image.fromarray((lit * 255).astype(np.uint8)).save(
                    os.path.join(root, "Rectified", folder, "rect_{:0>3}_{}_r5000.png".format(v + 1, li)))
            data_io.save_pfm(os.path.join(root, "Depths", folder, "depth_map_{:0>4}.pfm".format(v)), _plane_depth(height, width, seed + v))
    return root


def write_blendedmvs_set(root, scans=("scanA", "scanB"), nviews_total=5, width=160, height=128, seed=0, short_pairs=True):
    """<root>/training_list.txt, <root>/<scan>/{blended_images/<v:08>.jpg, cams/<v:08>_cam.txt, cams/pair.txt,
    rendered_depth_maps/<v:08>.pfm} -- the layout load/blendedtrain.py reads.  With short_pairs the last reference view lists
    only 2 source views (the loader pads it) and one lists none (the loader drops it)."""
    from PIL import Image
    from tools import data_io
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "training_list.txt"), "w") as f:
        f.write("\n".join(scans) + "\n")
    for si, scan in enumerate(scans):
        imgs, extr, intr, _ = synth.make_scene(width, height, nviews_total, batch=1, rot_deg=2.0, seed=seed + 10 * si)
        for sub in ("blended_images", "cams", "rendered_depth_maps"):
            os.makedirs(os.path.join(root, scan, sub), exist_ok=True)
        for v in range(nviews_total):
            arr = (imgs[0, v].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
            Image.fromarray(arr).save(os.path.join(root, scan, "blended_images", "{:0>8}.jpg".format(v)), quality=95)
            _write_cam(os.path.join(root, scan, "cams", "{:0>8}_cam.txt".format(v)), extr[0, v].numpy(), intr[0, v].numpy(),
                       "%.4f 2.5 128 %.4f" % (420.0 + si, 940.0 + si))
            data_io.save_pfm(os.path.join(root, scan, "rendered_depth_maps", "{:0>8}.pfm".format(v)), _plane_depth(height, width, seed + v + si))
        with open(os.path.join(root, scan, "cams", "pair.txt"), "w") as f:
            f.write("%d\n" % nviews_total)
            for ref in range(nviews_total):
                srcs = [v for v in range(nviews_total) if v != ref]
                if short_pairs and ref == nviews_total - 1:
                    srcs = srcs[:2]
                if short_pairs and ref == nviews_total - 2:
                    srcs = []
                f.write("%d\n%d " % (ref, len(srcs)) + " ".join("%d %.2f" % (s, 100.0 - abs(s - ref)) for s in srcs) + "\n")
    return root
