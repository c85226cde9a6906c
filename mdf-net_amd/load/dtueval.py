"""DTU evaluation items (counterpart of load/dtueval.py:9-61): one item per (scan, reference view)."""
from typing import List

import numpy as np
import torch

from tools import data_io
from load.getpath import get_img_path, get_cam_path


class LoadDataset(torch.utils.data.Dataset):
    CROP_ROWS = 1184   # dtueval.py:34 — the stage-0 regulariser needs H/8 divisible by 4

    def __init__(self, datasetpath: str, pairpath: str, scencelist: List, nviews: int) -> None:
        super().__init__()
        self.datasetpath, self.scenelist, self.nviews = datasetpath, scencelist, nviews
        self.num_viewpoint, self.pairs = data_io.read_pairfile(pairpath)
        self.all_compose = [[scene, ref, srcs] for scene in scencelist for ref, srcs in self.pairs]

    def __len__(self):
        return len(self.all_compose)

    def __getitem__(self, item):
        scene, ref_view, src_views = self.all_compose[item]
        folder = "scan{}".format(scene)
        imgs, extrinsics, intrinsics = [], [], []
        for vid in [ref_view] + src_views[:self.nviews - 1]:
            imgs.append(data_io.read_img(get_img_path(self.datasetpath, folder, vid, mode="eval"))[:self.CROP_ROWS])
            k, e = data_io.read_cam_file(get_cam_path(self.datasetpath, folder, vid, mode="eval"))
            intrinsics.append(k)
            extrinsics.append(e)
        return {"imgs": np.stack(imgs).transpose([0, 3, 1, 2]), "intrinsics": np.stack(intrinsics),
                "extrinsics": np.stack(extrinsics), "depth_range": np.array([425.0, 935.0]),
                "view_ids": np.array([ref_view] + src_views[:self.nviews - 1], dtype=np.int64), "scan": folder,
                "filename": folder + "/{}/" + "{:0>8}".format(ref_view) + "{}"}
