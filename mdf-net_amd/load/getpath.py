"""Where the datasets keep their files (the layouts the reference's load/getpath.py:4-45 encodes), as one table.

    kind   mode         relative path (fields: scan folder, view id, lighting)
"""
import os

_LAYOUT = {
    # DTU training set (640x512): lighting-specific rectified images, shared cameras, per-scan depth maps
    ("img", "train"): lambda s, v, l: ("Rectified", s, "rect_%03d_%s_r5000.png" % (v + 1, l)),
    ("cam", "train"): lambda s, v, l: ("Cameras", "%08d_cam.txt" % v),
    ("depth", "train"): lambda s, v, l: ("Depths", s, "depth_map_%04d.pfm" % v),
    # DTU evaluation set (1600x1200) and Tanks&Temples: per-scan images/ and cams/ (cams_1/ for T&T)
    ("img", "eval"): lambda s, v, l: (s, "images", "%08d.jpg" % v),
    ("cam", "eval"): lambda s, v, l: (s, "cams", "%08d_cam.txt" % v),
    ("img", "tanks"): lambda s, v, l: (s, "images", "%08d.jpg" % v),
    ("cam", "tanks"): lambda s, v, l: (s, "cams_1", "%08d_cam.txt" % v),
    # BlendedMVS (768x576)
    ("img", "blendedmvs"): lambda s, v, l: (s, "blended_images", "%08d.jpg" % v),
    ("cam", "blendedmvs"): lambda s, v, l: (s, "cams", "%08d_cam.txt" % v),
    ("depth", "blendedmvs"): lambda s, v, l: (s, "rendered_depth_maps", "%08d.pfm" % v),
}


def _path(kind, root, scan, view, lighting, mode):
    entry = _LAYOUT.get((kind, mode))
    return None if entry is None else os.path.join(root, *entry(scan, view, lighting))


def get_img_path(dataset_path, scan_folder, view_id, lighting=None, mode=""):
    return _path("img", dataset_path, scan_folder, view_id, lighting, mode)


def get_cam_path(dataset_path, scan_folder, view_id, mode):
    return _path("cam", dataset_path, scan_folder, view_id, None, mode)


def get_depth_path(dataset_path, scan_folder, view_id, mode):
    return _path("depth", dataset_path, scan_folder, view_id, None, mode)
