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
