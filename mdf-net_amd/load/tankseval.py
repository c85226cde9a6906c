"""Tanks&Temples evaluation items (counterpart of load/tankseval.py:9-66): per-scene pair file, depth range from
the reference view's cam file, rows cropped to 1056."""
import os
from typing import List

import numpy as np
import torch

from tools import data_io
from load.getpath import get_img_path, get_cam_path


class LoadDataset(torch.utils.data.Dataset):
    CROP_ROWS = 1056

    def __init__(self, datasetpath: str, scenelist: List, nviews: int) -> None:
        super().__init__()
        self.datasetpath, self.nviews = datasetpath, nviews
        self.all_compose = []
        for scan in scenelist:
            _, pairs = data_io.read_pairfile(os.path.join(datasetpath, scan, "pair.txt"))
            self.all_compose += [[scan, ref, srcs] for ref, srcs in pairs]

    def __len__(self):
        return len(self.all_compose)

    def __getitem__(self, item):
        scene, ref_view, src_views = self.all_compose[item]
        imgs, extrinsics, intrinsics, ranges = [], [], [], []
        for vid in [ref_view] + src_views[:self.nviews - 1]:
            imgs.append(data_io.read_img(get_img_path(self.datasetpath, scene, vid, mode="tanks"))[:self.CROP_ROWS])
            k, e, r = data_io.read_cam_file(get_cam_path(self.datasetpath, scene, vid, mode="tanks"), with_range=True)
            intrinsics.append(k)
            extrinsics.append(e)
            ranges.append(r)
        return {"imgs": np.stack(imgs).transpose([0, 3, 1, 2]), "intrinsics": np.stack(intrinsics),
                "extrinsics": np.stack(extrinsics), "depth_range": ranges[0],
                "view_ids": np.array([ref_view] + src_views[:self.nviews - 1], dtype=np.int64), "scan": scene,
                "filename": scene + "/{}/" + "{:0>8}".format(ref_view) + "{}"}
