"""Dataset path templates (counterpart of load/getpath.py:4-32)."""
import os


def get_img_path(dataset_path, scan_folder, view_id, lighting=None, mode=""):
    if mode == "train":
        return os.path.join(dataset_path, "Rectified", scan_folder, "rect_{:0>3}_{}_r5000.png".format(view_id + 1, lighting))
    if mode in ("eval", "tanks"):
        return os.path.join(dataset_path, scan_folder, "images", "{:0>8}.jpg".format(view_id))
    if mode == "blendedmvs":
        return os.path.join(dataset_path, "{}/blended_images/{:0>8}.jpg".format(scan_folder, view_id))
    return None


def get_cam_path(dataset_path, scan_folder, view_id, mode):
    if mode == "train":
        return os.path.join(dataset_path, "Cameras", "{:0>8}_cam.txt".format(view_id))
    if mode == "eval":
        return os.path.join(dataset_path, scan_folder, "cams", "{:0>8}_cam.txt".format(view_id))
    if mode == "tanks":
        return os.path.join(dataset_path, scan_folder, "cams_1", "{:0>8}_cam.txt".format(view_id))
    if mode == "blendedmvs":
        return os.path.join(dataset_path, "{}/cams/{:0>8}_cam.txt".format(scan_folder, view_id))
    return None


def get_depth_path(dataset_path, scan_folder, view_id, mode):
    """Ground-truth depth maps (load/getpath.py:34-45)."""
    if mode == "train":
        return os.path.join(dataset_path, "Depths", scan_folder, "depth_map_{:0>4}.pfm".format(view_id))
    if mode == "blendedmvs":
        return os.path.join(dataset_path, "{}/rendered_depth_maps/{:0>8}.pfm".format(scan_folder, view_id))
    return None
