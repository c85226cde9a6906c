"""DTU training set (counterpart of the reference's load/dtutrain.py:10-89): items = scene x reference view x lighting;
each yields V images [V,3,H,W] in [0,1], cameras, the reference view's ground-truth depth at 4 scales (keys "3","2","1","0"
= 1/8, 1/4, 1/2, 1/1, nearest-neighbour reduced) and the fixed DTU depth range."""
import random

import numpy as np
import torch

from load.getpath import get_cam_path, get_depth_path, get_img_path
from tools import data_io


def ground_truth_pyramid(ref_depth):
    """dtutrain.py:52-58: {"3": 1/8, "2": 1/4, "1": 1/2, "0": full}, cv2.INTER_NEAREST semantics."""
    h, w = ref_depth.shape
    return {"3": data_io.resize_nearest(ref_depth, (w // 8, h // 8)), "2": data_io.resize_nearest(ref_depth, (w // 4, h // 4)),
            "1": data_io.resize_nearest(ref_depth, (w // 2, h // 2)), "0": ref_depth}


class LoadDataset(torch.utils.data.Dataset):
    def __init__(self, datasetpath, pairpath, scencelist, lighting_label, nviews, robust_train=False):
        super().__init__()
        self.datasetpath, self.scenelist, self.lighting_label = datasetpath, scencelist, lighting_label
        self.nviews, self.robust_train = nviews, robust_train
        self.num_viewpoint, self.pairs = data_io.read_pairfile(pairpath)
        self.all_compose = [[scene, lighting, r, s] for scene in self.scenelist for r, s in self.pairs
                            for lighting in self.lighting_label]                       # dtutrain.py:79-87 (same order)

    def __len__(self):
        return len(self.scenelist) * len(self.pairs) * len(self.lighting_label)

    def __getitem__(self, item):
        scene, lighting, ref_view, src_views = self.all_compose[item]
        rs_views = [ref_view] + src_views[:self.nviews - 1]
        if self.robust_train:                                                          # dtutrain.py:33-35
            index = random.sample(range(1, len(src_views), 1), self.nviews - 1)
            rs_views = [ref_view] + [src_views[i] for i in index]
        scan_folder = "scan{}_train".format(scene)
        imgs, extrinsics, intrinsics, ref_depths = [], [], [], {}
        for i, vid in enumerate(rs_views):
            imgs.append(data_io.read_img(get_img_path(self.datasetpath, scan_folder, vid, lighting, mode="train")))
            intrinsic, extrinsic = data_io.read_cam_file(get_cam_path(self.datasetpath, scan_folder, vid, mode="train"))
            extrinsics.append(extrinsic)
            intrinsics.append(intrinsic)
            if i == 0:
                depth = np.array(data_io.read_pfm(get_depth_path(self.datasetpath, scan_folder, vid, mode="train"))[0], dtype=np.float32)
                ref_depths = ground_truth_pyramid(depth)
        return {"imgs": np.stack(imgs).transpose([0, 3, 1, 2]), "intrinsics": np.stack(intrinsics), "extrinsics": np.stack(extrinsics),
                "ref_depths": ref_depths, "depth_range": np.array([425.0, 935.0])}
