"""Depth-map filtering + fusion driver (counterpart of the reference's tools/filter/dynamic_filter_gpu.py:12-301).

Same entry points (`filter`, `check_geometric_consistency`), CLI flags (-f/-d/-s/-e/-r/-o) and outputs
(<eval>/<scan>/<filter>/{%08d_photo,_geo,_final}.png, <ref>_depth_est.pfm, <scan>.ply), but every reference view is ONE
fused HIP kernel over all its source views (mdf_consistency_fuse_fwd) instead of ~40 tensor passes per source view, and
scans shard over ranks (one process per GPU, no collective).  The .ply is written directly (binary little-endian, the
format plyfile emits) so `plyfile` is not needed."""
import argparse
import os
import time

import numpy as np
import torch

from mdfnet_hip import ops, shard
from tools.data_io import read_pfm, save_pfm, read_pairfile, read_img, read_cam_file


def save_mask(filename, mask):
    from PIL import Image
    Image.fromarray(mask.astype(np.uint8) * 255).save(filename)


def write_ply(path, xyz, rgb):
    """Vertices with float x,y,z + uchar red,green,blue; binary_little_endian 1.0 (what PlyData([el]).write produces)."""
    xyz = np.asarray(xyz, dtype="<f4").reshape(-1, 3)
    rgb = np.asarray(rgb, dtype=np.uint8).reshape(-1, 3)
    rec = np.empty(len(xyz), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    rec["red"], rec["green"], rec["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % len(rec))
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        rec.tofile(f)


def read_ply(path):
    with open(path, "rb") as f:
        n = None
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            if line == "end_header":
                break
        rec = np.fromfile(f, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")], count=n)
    return np.stack([rec["x"], rec["y"], rec["z"]], 1), np.stack([rec["red"], rec["green"], rec["blue"]], 1)


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src,
                                thre1=4, thre2=1300.):
    """dynamic_filter_gpu.py:166-191 -> (list of 9 masks [1,h,w], last mask, depth_reprojected [1,h,w])."""
    r = ops.consistency_fuse(depth_ref, torch.ones_like(depth_ref), intrinsics_ref, extrinsics_ref, [depth_src],
                             [intrinsics_src], [extrinsics_src], 0.0, 1, float(thre1), float(thre2), per_view=True)
    masks = [r["view_masks"][0, i].unsqueeze(0) for i in range(9)]
    return masks, masks[-1], r["rep"]


def backproject(depth_avg, mask, intrinsics, extrinsics):
    """filter():130-143, numpy semantics of the reference (int64 pixel grid * float32 depth -> float64)."""
    h, w = depth_avg.shape
    x, y = np.meshgrid(np.arange(0, w), np.arange(0, h))
    x, y, d = x[mask], y[mask], depth_avg[mask]
    cam = np.matmul(np.linalg.inv(intrinsics), np.vstack((x, y, np.ones_like(x))) * d)
    return np.matmul(np.linalg.inv(extrinsics), np.vstack((cam, np.ones_like(x))))[:3].transpose((1, 0))


def filter(dataset_root, scan, img_folder, cam_folder, eval_folder, filter_folder, outply_folder,
           photo_threshold=0.8, nconditions=5, thre1=4, thre2=1300., device=None, log=print):
    device = device or torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    scan_location = os.path.join(dataset_root, scan)
    eval_location = os.path.join(eval_folder, scan)
    workspace = os.path.join(eval_location, filter_folder)
    os.makedirs(workspace, exist_ok=True)
    _, pairs = read_pairfile(os.path.join(scan_location, "pair.txt"))
    cams, depths = {}, {}

    def cam(v):
        if v not in cams:
            k, e = read_cam_file(os.path.join(scan_location, cam_folder, "{:0>8}_cam.txt".format(v)))
            cams[v] = (torch.from_numpy(k.copy()), torch.from_numpy(e.copy()))
        return cams[v]

    def depth(v):   # every depth map is uploaded once per scan and stays resident (a DTU scan is 49 x 7.7 MB)
        if v not in depths:
            depths[v] = torch.from_numpy(read_pfm(os.path.join(eval_location, "depth_est", "{:0>8}.pfm".format(v)))[0].copy()).float().to(device)
        return depths[v]

    vertexs, vertex_colors = [], []
    for ref_view, src_views in pairs:
        if len(src_views) == 0:
            continue
        t0 = time.time()
        ref_img = read_img(os.path.join(scan_location, img_folder, "{:0>8}.jpg".format(ref_view)))
        conf = torch.from_numpy(read_pfm(os.path.join(eval_location, "confidence", "{:0>8}.pfm".format(ref_view)))[0].copy()).float().to(device)
        k_ref, e_ref = cam(ref_view)
        r = ops.consistency_fuse(depth(ref_view), conf, k_ref, e_ref, [depth(v) for v in src_views], [cam(v)[0] for v in src_views],
                                 [cam(v)[1] for v in src_views], photo_threshold, nconditions, float(thre1), float(thre2))
        depth_avg = r["depth_avg"].cpu().numpy()
        photo, geo, final = (r[k].cpu().numpy() for k in ("photo_mask", "geo_mask", "final_mask"))
        ref_depth = depth(ref_view).cpu().numpy()
        log("processing {}, ref-view{:0>2}, photo/geo/final-mask:{}/{}/{}".format(scan_location, ref_view, photo.sum(), geo.sum(),
                                                                                 final.sum()), " time:", time.time() - t0)
        save_mask(os.path.join(workspace, "{:0>8}_photo.png".format(ref_view)), photo)
        save_mask(os.path.join(workspace, "{:0>8}_geo.png".format(ref_view)), geo)
        save_mask(os.path.join(workspace, "{:0>8}_final.png".format(ref_view)), final)
        save_pfm(os.path.join(workspace, "{}".format(ref_view) + "_" + "depth_est.pfm"), ref_depth * final.astype(np.float32))
        h, w = depth_avg.shape
        vertexs.append(backproject(depth_avg, final, k_ref.numpy(), e_ref.numpy()))
        vertex_colors.append((ref_img[:h, :w, :][final] * 255).astype(np.uint8))
    if not vertexs:
        return None
    out_dir = eval_location if outply_folder is None else outply_folder
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, scan + ".ply")
    write_ply(path, np.concatenate(vertexs, 0), np.concatenate(vertex_colors, 0))
    log("saving the final model to", path)
    return path


def main():
    parser = argparse.ArgumentParser(description="filter to mask cloudpoints...")
    parser.add_argument("-f", "--filter_folder", default="filter", type=str)
    parser.add_argument("-d", "--dataset", default="tanks", type=str, help="dtu or tanks")
    parser.add_argument("-s", "--set", default="intermediate", type=str)
    parser.add_argument("-e", "--eval_folder", default=os.environ.get("MDF_OUTPUT_ROOT", "/hy-tmp/outputs"), type=str)
    parser.add_argument("-r", "--root_folder", default=os.environ.get("MDF_DATA_ROOT", "/hy-nas"), type=str)
    parser.add_argument("-o", "--outply_folder", default=None, type=str)
    args = parser.parse_args()
    rank, world, _ = shard.init()
    if args.dataset == "dtu":
        root = os.path.join(args.root_folder, "dtu1600x1200")
        scans = ["scan" + s for s in os.environ.get("MDF_DTU_SCANS", "11").split(",")]
        img_folder, cam_folder, ncond = "images", "cams", 5
    elif args.dataset == "tanks":
        root = os.path.join(args.root_folder, "TankandTemples", args.set)
        scans = {"intermediate": ["Family", "Francis", "Horse", "Lighthouse", "M60", "Panther", "Playground", "Train"],
                 "advanced": ["Auditorium", "Ballroom", "Courtroom", "Museum", "Palace", "Temple"]}[args.set]
        img_folder, cam_folder, ncond = "images", "cams_1", (5 if args.set == "intermediate" else 1)
    else:
        raise SystemExit("please use dtu or tanks dataset")
    for i in shard.shard_items(len(scans), rank, world):     # scans are independent: shard them, no collective
        t0 = time.time()
        filter(root, scans[i], img_folder, cam_folder, args.eval_folder, args.filter_folder, args.outply_folder, 0.8, ncond, 4, 1300)
        print("scan:", scans[i], "all time:", time.time() - t0)
    shard.barrier()


if __name__ == "__main__":
    main()
